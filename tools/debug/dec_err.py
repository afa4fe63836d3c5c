"""DECGenerator1 gradient error against an fp64 evaluation of the oracle: HIP vs fp64 and CPU-fp32 vs fp64 (is a tolerance miss
arithmetic noise on an ill-conditioned case, or a defect?)."""
import copy
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import __graft_entry__  # noqa: F401  (puts the package on sys.path)
import torch
from tests.golden import cases_dualgan as C
from dual_gan.models import networks as N

dev = torch.device("cuda:0")
on, feat = C.decgen1_case()
o64 = copy.deepcopy(on).double()
rg = N.DECGenerator1(64, 256, 3, 'instance', 'LeakyReLU', False, False, 3, 3)
rg.load_state_dict(on.state_dict())
rg = rg.to(dev).train()
g = torch.Generator().manual_seed(7)
f32, f64, fd = feat.clone().requires_grad_(True), feat.double().requires_grad_(True), feat.to(dev).requires_grad_(True)
y32, y64, yd = on(f32), o64(f64), rg(fd)
dy = torch.randn(y32.shape, generator=g)
y32.backward(dy); y64.backward(dy.double()); yd.backward(dy.to(dev))


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).norm() / b.norm()).item()


print("fwd   hip %.2e cpu32 %.2e" % (rel(yd, y64), rel(y32, y64)))
print("dfeat hip %.2e cpu32 %.2e  hip-vs-cpu32 %.2e" % (rel(fd.grad, f64.grad), rel(f32.grad, f64.grad), rel(fd.grad, f32.grad)))
p32, p64, pd = dict(on.named_parameters()), dict(o64.named_parameters()), dict(rg.named_parameters())
for k in p32:
    print("%-34s hip %.2e cpu32 %.2e" % (k, rel(pd[k].grad, p64[k].grad), rel(p32[k].grad, p64[k].grad)))
