import sys, os
sys.path.insert(0, os.path.join(os.getcwd(), "reid-gan_amd")); sys.path.insert(0, os.getcwd())
import torch
from oracle import ref_torch as O
import reid.models as RM
dev = torch.device("cuda:0")
torch.manual_seed(1)
o = O.OReidResNet(50, cut_at_pooling=True)
r = RM.create('resnet50', pretrained=False, cut_at_pooling=True)
r.load_state_dict(o.state_dict()); r.to(dev)
o.train(); r.train()
for n_img, hw in ((4, (128, 64)), (16, (128, 64))):
    x = O.synth_images(n_img, hw[0], hw[1], seed=2)
    xo = x.clone().requires_grad_(True); xr = x.clone().to(dev).requires_grad_(True)
    o.zero_grad(); r.zero_grad()
    fo, fr = o(xo), r(xr)
    def rel(a, b):
        a, b = a.detach().double().cpu(), b.detach().double().cpu()
        return ((a-b).abs().max() / b.abs().max().clamp_min(1e-12)).item()
    print("N", n_img, "fwd rel", rel(fr, fo))
    g = torch.randn(fo.shape, generator=torch.Generator().manual_seed(3))
    fo.backward(g); fr.backward(g.to(dev))
    print("dx rel", rel(xr.grad, xo.grad))
    og = dict(o.named_parameters())
    for n, p in r.named_parameters():
        if p.grad is None: continue
        e = rel(p.grad, og[n].grad)
        if e > 1e-3 or n.endswith("conv1.weight") and "layer" not in n or "layer4.2" in n or "layer1.0.conv1" in n:
            print("  %-40s %.3e  |g|max %.3e" % (n, e, og[n].grad.abs().max().item()))
# double-precision oracle to see which side is off
od = O.OReidResNet(50, cut_at_pooling=True).double(); od.load_state_dict({k: v.double() if v.is_floating_point() else v for k, v in o.state_dict().items()})
torch.manual_seed(1)
o2 = O.OReidResNet(50, cut_at_pooling=True); od = O.OReidResNet(50, cut_at_pooling=True)
od.load_state_dict(o2.state_dict()); od = od.double(); o2.train(); od.train()
r2 = RM.create('resnet50', pretrained=False, cut_at_pooling=True); r2.load_state_dict(o2.state_dict()); r2.to(dev).train()
x = O.synth_images(4, 128, 64, seed=2)
x32 = x.clone().requires_grad_(True); x64 = x.double().clone().requires_grad_(True); xg = x.clone().to(dev).requires_grad_(True)
f32, f64, fg = o2(x32), od(x64), r2(xg)
g = torch.randn(f32.shape, generator=torch.Generator().manual_seed(3))
f32.backward(g); f64.backward(g.double()); fg.backward(g.to(dev))
print("vs fp64: cpu32 dx", rel(x32.grad, x64.grad), " hip dx", rel(xg.grad, x64.grad), " cpu32 fwd", rel(f32, f64), " hip fwd", rel(fg, f64))
