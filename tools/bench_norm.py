"""Timings of the train-mode BatchNorm launches at the ResNet-50 / generator shapes (development aid):
    python tools/bench_norm.py [crops]          (RG_BN_REG=0 selects the loop kernels for an A/B)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "reid-gan_amd"))
import torch
from rg_hip import ops

dev = torch.device("cuda:0")


def timeit(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    print("RG_BN_REG=%s  crops %d" % (os.environ.get("RG_BN_REG", "1"), N))
    print("%-22s %6s | %9s %9s us | GB/s fwd  bwd (algorithmic: 2 / 4 tensor passes)" % ("shape", "fused", "fwd", "bwd"))
    for C, H, W in ((64, 64, 32), (256, 64, 32), (128, 32, 16), (512, 32, 16), (256, 16, 8), (1024, 16, 8), (512, 8, 4), (2048, 8, 4),
                    (2048, 16, 8), (128, 16, 8), (256, 8, 4)):
        x = torch.randn(N, C, H, W, device=dev)
        dy = torch.randn_like(x)
        g, b = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
        rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
        if not ops.bn_train_fused_ok(x):
            print("%-22s %6s" % ((N, C, H, W), "no"))
            continue
        y, mean, stat = ops.bn_train_fwd_fused(x, g, b, None, rm, rv, 1e-5, 0.1, ops.ACT_RELU, 0.0)
        tf = timeit(lambda: ops.bn_train_fwd_fused(x, g, b, None, rm, rv, 1e-5, 0.1, ops.ACT_RELU, 0.0))
        tb = timeit(lambda: ops.bn_train_bwd_fused(x, dy, y, mean, stat, g, ops.ACT_RELU, 0.0))
        nb = x.numel() * 4 / 1e3
        print("%-22s %6s | %9.2f %9.2f    | %7.0f %7.0f" % ((N, C, H, W), "yes", tf, tb, 2 * nb / tf, 4 * nb / tb))


if __name__ == "__main__":
    main()
