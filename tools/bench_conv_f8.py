"""Per-shape timing of the fp8 convolution GEMMs on the DPTN layer geometries at 128 samples (development aid):
python tools/bench_conv_f8.py [N]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "reid-gan_amd"))
import torch
from rg_hip import lowp, ops

dev = torch.device("cuda:0")
# name, count per step, C, H, W, K, k, stride, pad   (tools: oracle hook dump of ODPTNModel.step)
SHAPES = [("g.enc 64>64 3x3", 5, 64, 64, 32, 64, 3, 1, 1), ("g.enc 64>128 4x4/2", 3, 64, 64, 32, 128, 4, 2, 1),
          ("g 128>128 3x3", 3, 128, 32, 16, 128, 3, 1, 1), ("g 128>256 4x4/2", 3, 128, 32, 16, 256, 4, 2, 1),
          ("g 256>256 3x3", 15, 256, 16, 8, 256, 3, 1, 1), ("g 256>256 1x1", 6, 256, 16, 8, 256, 1, 1, 0),
          ("g 256>128 3x3", 2, 256, 16, 8, 128, 3, 1, 1), ("g 128>64 3x3", 2, 128, 32, 16, 64, 3, 1, 1),
          ("d 32>32 4x4/2", 3, 32, 128, 64, 32, 4, 2, 1), ("d 32>64 4x4/2", 3, 32, 64, 32, 64, 4, 2, 1),
          ("d 64>128 4x4/2", 3, 64, 32, 16, 128, 4, 2, 1)]


def timeit(fn, reps=20):
    for _ in range(3):
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
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    st = lowp.F8States(dev, capacity=4)
    st.policy = "jit"
    sx, sw, sdy = st.new(lowp.E4M3), st.new(lowp.E4M3), st.new(lowp.E5M2)
    st.finalize()
    tot = [0.0, 0.0, 0.0, 0.0]
    print("%-22s %3s %8s | %8s %8s %8s us | %7s %7s %7s TFLOP/s | out MB" % ("layer", "cnt", "GFLOP", "fwd", "dgrad", "wgrad", "fwd", "dgrad", "wgrad"))
    for name, cnt, C, H, W, K, k, s, p in SHAPES:
        x = torch.randn(N, C, H, W, device=dev)
        w = torch.randn(K, C, k, k, device=dev) * 0.05
        P, Q = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
        dy = torch.randn(N, K, P, Q, device=dev) * 1e-3
        sx.prepare(x); sw.prepare(w); sdy.prepare(dy)
        xq, xq_t = lowp.quantize_dual(x, sx)
        wq, wq_t = lowp.quantize_dual(w, sw)
        dyq, dyq_t = lowp.quantize_dual(dy, sdy)
        g = (N, C, H, W, K, k, k, s, s, p, p)
        fl = 2.0 * N * P * Q * K * C * k * k
        tf = timeit(lambda: lowp.conv_fwd(xq, wq, g))
        td = timeit(lambda: lowp.conv_dgrad(dyq, wq_t, g, (H, W)))
        tw = timeit(lambda: lowp.conv_wgrad(xq_t, dyq_t, g))
        for i, t in enumerate((tf, td, tw)):
            tot[i] += cnt * t
        tot[3] += cnt * fl
        print("%-22s %3d %8.2f | %8.1f %8.1f %8.1f    | %7.0f %7.0f %7.0f         | %6.1f" % (
            name, cnt, fl / 1e9, tf, td, tw, fl / tf / 1e6, fl / td / 1e6, fl / tw / 1e6, N * K * P * Q * 4 / 1e6))
    print("TOTAL per step: fwd %.2f ms  dgrad %.2f ms  wgrad %.2f ms;  %.1f / %.1f / %.1f TFLOP/s" % (
        tot[0] / 1e3, tot[1] / 1e3, tot[2] / 1e3, tot[3] / tot[0] / 1e6, tot[3] / tot[1] / 1e6, tot[3] / tot[2] / 1e6))


if __name__ == "__main__":
    main()
