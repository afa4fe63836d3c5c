"""How many quantiser launches of one DPTN step (config 5 geometry, eager) are filters / activations / gradients, and their sizes."""
import collections
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
os.environ["RG_NET_GRAPHS"] = "0"
import __graft_entry__  # noqa: F401
import torch
import bench
from rg_hip import lowp

cnt = collections.Counter()
orig_dual, orig_grad = lowp.quantize_dual, lowp.quantize_grad_dual
phase = ["?"]


def dual(x, state, want_a=True, want_b=True):
    cnt[(phase[0], tuple(x.shape), want_a, want_b)] += 1
    return orig_dual(x, state, want_a, want_b)


def grad(dy, state, want_a=True, want_b=True, **kw):
    cnt[("grad", tuple(dy.shape), want_a, want_b)] += 1
    return orig_grad(dy, state, want_a, want_b, **kw)


lowp.quantize_dual, lowp.quantize_grad_dual = dual, grad
ow, oa = lowp.F8Layer.weights, lowp.F8Layer.quant_act_both


def weights(self, w, key=None):
    phase[0] = "filter"
    try:
        return ow(self, w, key)
    finally:
        phase[0] = "?"


def act(self, x, want):
    phase[0] = "act"
    try:
        return oa(self, x, want)
    finally:
        phase[0] = "?"


lowp.F8Layer.weights, lowp.F8Layer.quant_act_both = weights, act
w = bench.WORKLOADS["5"]()
w.build(torch.device("cuda:0"), 0)
step = w.step
for _ in range(3):
    step()
torch.cuda.synchronize()
cnt.clear()
step()
torch.cuda.synchronize()
tot = collections.Counter()
for (ph, shape, a, b), n in sorted(cnt.items(), key=lambda kv: (kv[0][0], -kv[1])):
    tot[ph] += n
    print("%-7s %-22s a=%d b=%d  x %d" % (ph, shape, a, b, n))
print(dict(tot))
