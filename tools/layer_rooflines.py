#!/usr/bin/env python3
"""Per-layer roofline of the conv family from the tables of tools/bench_conv.py:
    python tools/layer_rooflines.py profiles/r04_conv_shapes_resnet_n32.txt profiles/r04_conv_shapes_gan_n32.txt > profiles/r04_conv_layer_rooflines.txt
For each layer and direction: t_mfma = FLOPs / MFMA_RATE, t_hbm = algorithmic bytes / HBM_RATE, bound = max, eff = bound / measured."""
import re
import sys

HBM_RATE = 5.0e12    # attainable streaming rate (MI355X_MICROARCH.md: 5.5-6.2 TB/s measured for plain reads / stores; 8 TB/s spec)
MFMA_RATE = 220e12   # split-bf16 rate of a staging-free fragment-read + MFMA loop at the 1.45-1.55 GHz the chip holds under dense bf16
                     # MFMA issue (profiles/r04_micro_gemm_pl.txt: 209-222 TFLOP/s; 416.7 at the nominal 2.4 GHz)


def parse(path):
    rows = []
    for ln in open(path):
        m = re.match(r"(\S.*?)\s+(\d+)\s+(\d+),(\d+),(\d+),(\d+)->(\d+) k(\d+) s(\d+) p(\d+)\s+([\d.]+) \|\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)/([\d.]+)/([\d.]+)", ln)
        if m:
            g = m.groups()
            rows.append((g[0], int(g[1])) + tuple(int(v) for v in g[2:10]) + (float(g[10]),) + tuple(float(v) for v in g[14:17]))
    return rows


def main():
    print("Per-layer roofline of the conv family at 32 crops (times: tools/bench_conv.py, per-launch HIP events, the kernels the step runs:\n"
          "filter re-layout built once, kernel choice measured).  t_mfma = FLOPs / %.0f TFLOP/s (the split-bf16 rate a staging-free loop sustains\n"
          "at the clock the chip holds under dense bf16 MFMA issue; 416.7 at the nominal 2.4 GHz), t_hbm = algorithmic bytes (one read of each\n"
          "operand, one write of the result, fp32) / %.0f TB/s, bound = max of the two, eff = bound / measured.  'HBM' marks layers whose memory\n"
          "time exceeds their matrix time.  The full data gradients of stem7x7 / en1 / dp1 / de1 are NOT what the step runs (their inputs are\n"
          "images / pose maps: the step needs at most 3 channels of them, conv_dgrad_smallc_px_kernel); the tool runs them for completeness." % (MFMA_RATE / 1e12, HBM_RATE / 1e12))
    for path in sys.argv[1:]:
        rows = parse(path)
        print("\n%-26s %3s %-28s | %-30s | %-30s | %-30s" % (path.split("/")[-1][:26], "cnt", "N,C,H,W->K k s p", "fwd    us  mfma   hbm  eff", "dgrad  us  mfma   hbm  eff", "wgrad  us  mfma   hbm  eff"))
        tot, totb = [0.0] * 3, [0.0] * 3
        for name, cnt, N, C, H, W, K, k, s, p, gf, tf, td, tw in rows:
            P, Q = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
            by = 4.0 * (N * C * H * W + K * C * k * k + N * K * P * Q)
            cells = []
            for i, t in enumerate((tf, td, tw)):
                tm, th = gf * 1e9 / MFMA_RATE * 1e6, by / HBM_RATE * 1e6
                b = max(tm, th)
                tot[i] += cnt * t * 1e3
                totb[i] += cnt * b
                cells.append("%6.1f %5.1f %5.1f %4.2f %s" % (t * 1e3, tm, th, b / (t * 1e3) if t > 0 else 0.0, "HBM" if th > tm else "   "))
            print("%-26s %3d %-28s | %-30s | %-30s | %-30s" % (name[:26], cnt, "%d,%d,%d,%d->%d k%d s%d p%d" % (N, C, H, W, K, k, s, p), cells[0], cells[1], cells[2]))
        print("TOTAL measured us fwd / dgrad / wgrad %.0f / %.0f / %.0f; sum of bounds %.0f / %.0f / %.0f; eff %.2f / %.2f / %.2f" % (
            tot[0], tot[1], tot[2], totb[0], totb[1], totb[2], totb[0] / tot[0], totb[1] / tot[1], totb[2] / tot[2]))


if __name__ == "__main__":
    main()
