"""debug: per-tensor gradient cosine (fp8 HIP vs fp32 oracle) of DPTNGenerator with fixed cotangents, in tape order"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "reid-gan_amd")); sys.path.insert(0, REPO)
import torch
from tests.golden import cases_dptn as C
from tests.test_dptn_gpu import _build
dev = torch.device("cuda:0")
dtype = sys.argv[1] if len(sys.argv) > 1 else "fp8"
m, om = _build(dev, "hinge", conv_dtype=dtype)
if dtype == "fp8" and len(sys.argv) > 2:
    from rg_hip import lowp
    for st in m._f8_states: st.policy = sys.argv[2]
if "nokink" in sys.argv:
    from tests.test_dptn_gpu import _remove_kinks
    _remove_kinks([m.net_G, m.net_D], [om.net_G, om.net_D])
d = C.inputs(); dd = {k: v.to(dev) for k, v in d.items()}
om.set_input(d); om.forward(); m.set_input(dd); m.forward()
def cos(a, b):
    a, b = a.detach().double().cpu().flatten(), b.detach().double().flatten()
    return float((a * b).sum() / (a.norm() * b.norm() + 1e-300))
g = torch.Generator().manual_seed(11)
ct, cs_ = torch.randn(om.fake_image_t.shape, generator=g), torch.randn(om.fake_image_s.shape, generator=g)
((om.fake_image_t * ct).sum() + (om.fake_image_s * cs_).sum()).backward()
torch.autograd.backward([m.fake_image_t, m.fake_image_s], [ct.to(dev), cs_.to(dev)])
pg, og = dict(m.net_G.module.named_parameters()), dict(om.net_G.named_parameters())
for k in og:
    if og[k].grad is None: continue
    if og[k].dim() < 2: continue
    print("%-70s cos %.4f  |ref| %.3e" % (k, cos(pg[k].grad, og[k].grad), float(og[k].grad.norm())))
