"""Timing of the fp8 dual-layout quantiser (development aid): python tools/bench_quant.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "reid-gan_amd"))
import torch
from rg_hip import lowp

dev = torch.device("cuda:0")


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
    st = lowp.F8States(dev, capacity=4)
    st.policy = "delayed"
    s = st.new(lowp.E4M3)
    st.finalize()
    print("%-22s %10s %10s %10s %10s   GB/s(both)" % ("shape", "both us", "a only", "b only", "copy us"))
    for shape in ((128, 64, 128, 64), (128, 128, 64, 32), (128, 256, 32, 16), (128, 256, 16, 8), (64, 128, 64, 32), (128, 64, 64, 32)):
        x = torch.randn(shape, device=dev)
        s.prepare(x)
        y = torch.empty_like(x)
        t_both = timeit(lambda: lowp.quantize_dual(x, s, True, True))
        t_a = timeit(lambda: lowp.quantize_dual(x, s, True, False))
        t_b = timeit(lambda: lowp.quantize_dual(x, s, False, True))
        t_c = timeit(lambda: y.copy_(x))
        gb = x.numel() * 4 * 1.5 / 1e9
        print("%-22s %10.1f %10.1f %10.1f %10.1f   %8.0f" % (str(shape), t_both, t_a, t_b, t_c, gb / (t_both * 1e-6)))


if __name__ == "__main__":
    main()
