"""debug: per-parameter gradient / update differences of the DPTN step (HIP vs oracle) for step 0"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "reid-gan_amd")); sys.path.insert(0, REPO)
import torch
from tests.golden import cases_dptn as C
from tests.test_dptn_gpu import _build

dev = torch.device("cuda:0")
mode = sys.argv[1] if len(sys.argv) > 1 else "vanilla"
m, om = _build(dev, mode)
d = C.inputs(); dd = {k: v.to(dev) for k, v in d.items()}
om.set_input(d); om.forward()
m.set_input(dd); m.forward()
print("fwd fake_t", ((m.fake_image_t.cpu() - om.fake_image_t).abs().max() / om.fake_image_t.abs().max()).item())
om.optimizer_D.zero_grad(); om.backward_D()
m.optimizer_D.zero_grad(); m.backward_D()
def cmp(pg, og, what):
    rows = []
    for k, p in pg.items():
        r = og[k].grad
        if r is None or p.grad is None:
            rows.append((k, "none", None if p.grad is None else float(p.grad.abs().max()), None if r is None else float(r.abs().max())))
            continue
        e = (p.grad.cpu().double() - r.double()).norm().item() / max(r.double().norm().item(), 1e-30)
        rows.append((k, e, float(r.abs().max()), float(p.grad.abs().max())))
    bad = [r for r in rows if r[1] == "none" or r[1] > 2e-3]
    print(what, "tensors", len(rows), "bad", len(bad))
    for r in bad[:40]:
        print("   ", r)
cmp(dict(m.net_D.module.named_parameters()), dict(om.net_D.named_parameters()), "D grads")
om.optimizer_D.step(); m.optimizer_D.step()
om.optimizer_G.zero_grad(); om.backward_G()
m.optimizer_G.zero_grad(); m.backward_G()
cmp(dict(m.net_G.module.named_parameters()), dict(om.net_G.named_parameters()), "G grads")
before = {k: v.detach().clone() for k, v in om.net_G.named_parameters()}
om.optimizer_G.step(); m.optimizer_G.step()
pg, og = dict(m.net_G.module.named_parameters()), dict(om.net_G.named_parameters())
rows = []
for k in og:
    do = (og[k].detach() - before[k]).double()
    dh = (pg[k].detach().cpu() - before[k]).double()
    rows.append((k, (dh - do).abs().max().item(), do.abs().max().item(), float(og[k].grad.abs().max())))
rows.sort(key=lambda r: -r[1])
print("largest update differences (name, max|dh-do|, max|do|, max|grad|):")
for r in rows[:25]:
    print("   ", r)
