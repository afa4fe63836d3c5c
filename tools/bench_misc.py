"""Timings of the odd-shaped layers of the FD-GAN step that tools/bench_conv.py does not cover: the (8,4) bottleneck
convolutions of the generator, the 7x7 stem's data gradient to the image, the 64->3 output deconvolution and the
stem max-pool backward (development aid)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "reid-gan_amd"))
import torch
from rg_hip import ops

dev = torch.device("cuda:0")


def timeit(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    shapes = [("de_avg (as conv 512->2432 (8,4))", 512, 8, 4, 2432, 8, 4, 1, 0),
              ("en_avg conv 512->128 (8,4)", 512, 8, 4, 128, 8, 4, 1, 0),
              ("de_avg, 1x1 view (16384->2432)", 16384, 1, 1, 2432, 1, 1, 1, 0),
              ("en_avg, 1x1 view (16384->128)", 16384, 1, 1, 128, 1, 1, 1, 0),
              ("stem 3->64 7x7/2", 3, 256, 128, 64, 7, 7, 2, 3),
              ("de1 (as conv 3->64 4x4/2)", 3, 256, 128, 64, 4, 4, 2, 1),
              ("dp1 21->64 4x4/2", 21, 256, 128, 64, 4, 4, 2, 1),
              ("dp4 256->512 4x4/1", 256, 32, 16, 512, 4, 4, 1, 1),
              ("dp5 512->1 4x4/1", 512, 31, 15, 1, 4, 4, 1, 1)]
    print("%-36s %9s | %8s %8s %8s ms   (GB moved by the op's tensors)" % ("layer", "GFLOP", "fwd", "dgrad", "wgrad"))
    for name, C, H, W, K, kh, kw, s, p in shapes:
        x = torch.randn(N, C, H, W, device=dev)
        w = torch.randn(K, C, kh, kw, device=dev) * 0.05
        y = ops.conv2d_fwd(x, w, s, p)
        dy = torch.randn_like(y)
        fl = 2.0 * y.numel() * C * kh * kw
        tf = timeit(lambda: ops.conv2d_fwd(x, w, s, p))
        td = timeit(lambda: ops.conv2d_dgrad(dy, w, (H, W), s, p))
        tw = timeit(lambda: ops.conv2d_wgrad(x, dy, (K, C, kh, kw), s, p))
        gb = (x.numel() + w.numel() + y.numel()) * 4 / 1e9
        print("%-36s %9.2f | %8.3f %8.3f %8.3f      %.3f" % (name, fl / 1e9, tf, td, tw, gb))
    x = torch.randn(N, 64, 128, 64, device=dev)
    y, arg = ops.maxpool2d_fwd(x)
    dy = torch.randn_like(y)
    tf = timeit(lambda: ops.maxpool2d_fwd(x))
    tb = timeit(lambda: ops.maxpool2d_bwd(dy, arg, x.shape))
    print("maxpool 3x3/2 on [%d,64,128,64]: fwd %.3f ms  bwd %.3f ms  (x %.3f GB, y %.3f GB)" % (
        N, tf, tb, x.numel() * 4 / 1e9, y.numel() * 4 / 1e9))


if __name__ == "__main__":
    main()
