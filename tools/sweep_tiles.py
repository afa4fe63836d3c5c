"""For every conv geometry of the FD-GAN step: the planner's choice vs every forced (tile, split-K) choice, fwd and dgrad
(development aid for the cost model in conv_igemm.hip:plan_gemm).  usage: sweep_tiles.py N [resnet|gan]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "reid-gan_amd"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from rg_hip import ops
from rg_hip.lib import lib
from bench_conv import resnet_shapes, uniq
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
which = sys.argv[2] if len(sys.argv) > 2 else "resnet"


def t(fn, reps=6):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


if which == "resnet":
    shapes = uniq(resnet_shapes(256, 128))
else:
    shapes = [("en2", 1, 64, 128, 64, 128, 4, 2, 1), ("en3", 1, 128, 64, 32, 256, 4, 2, 1), ("en4", 1, 256, 32, 16, 512, 4, 2, 1),
              ("en5", 1, 512, 16, 8, 512, 4, 2, 1), ("dp4", 1, 256, 32, 16, 512, 4, 1, 1)]
tot = {"fwd": [0.0, 0.0], "dgrad": [0.0, 0.0]}
for name, cnt, C, H, W, K, k, s, p in shapes:
    if C < 16:
        continue
    x = torch.randn(N, C, H, W, device=dev); w = torch.randn(K, C, k, k, device=dev) * 0.05
    y = ops.conv2d_fwd(x, w, s, p); dy = torch.randn_like(y)
    wk = ops.weights_to_krsc(w) if k > 1 and C % 4 == 0 else None
    for kind, fn in (("fwd", lambda: ops.conv2d_fwd(x, w, s, p, w_krsc=wk)), ("dgrad", lambda: ops.conv2d_dgrad(dy, w, (H, W), s, p, w_krsc=wk))):
        lib.rg_conv_set_force(-1, -1)
        ops._ws_sizes.clear()
        base = t(fn)
        best, bcfg = base, "plan"
        for tile in (0, 1, 2):
            for sp in (1, 2, 3, 4, 6, 8):
                lib.rg_conv_set_force(tile, sp)
                ops._ws_sizes.clear()
                try:
                    us = t(fn, 4)
                except Exception:
                    continue
                if us < best:
                    best, bcfg = us, "t%d s%d" % (tile, sp)
        lib.rg_conv_set_force(-1, -1)
        ops._ws_sizes.clear()
        tot[kind][0] += base * cnt; tot[kind][1] += best * cnt
        flag = "  <-- %.0f%%" % (100 * (base - best) / base) if best < 0.93 * base else ""
        print("%-11s x%d %-5s plan %6.1f us  best %6.1f us (%s)%s" % (name, cnt, kind, base, best, bcfg, flag))
for kind in tot:
    print("TOTAL %s: plan %.2f ms  best %.2f ms" % (kind, tot[kind][0] / 1e3, tot[kind][1] / 1e3))
