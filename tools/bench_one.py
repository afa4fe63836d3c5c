"""Repeatedly launch one conv shape (for PMC runs): python tools/bench_one.py fwd|dgrad|wgrad N C H W K k s p [reps]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "reid-gan_amd"))
import torch
from rg_hip import ops
kind = sys.argv[1]
N, C, H, W, K, k, s, p = [int(v) for v in sys.argv[2:10]]
reps = int(sys.argv[10]) if len(sys.argv) > 10 else 10
dev = torch.device("cuda:0")
x = torch.randn(N, C, H, W, device=dev)
w = torch.randn(K, C, k, k, device=dev) * 0.05
y = ops.conv2d_fwd(x, w, s, p)
dy = torch.randn_like(y)
wk = ops.weights_to_krsc(w) if k > 1 and C % 4 == 0 else None
torch.cuda.synchronize()
for _ in range(reps):
    if kind == "fwd":
        ops.conv2d_fwd(x, w, s, p)
    elif kind == "dgrad":
        ops.conv2d_dgrad(dy, w, (H, W), s, p, w_krsc=wk)
    else:
        ops.conv2d_wgrad(x, dy, (K, C, k, k), s, p)
torch.cuda.synchronize()
