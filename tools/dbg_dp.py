import sys, os
sys.path.insert(0, os.path.join(os.getcwd(), "reid-gan_amd")); sys.path.insert(0, os.getcwd())
import torch
from oracle import ref_torch as O
import fdgan.networks as N
from fdgan.losses import GANLoss
dev = torch.device("cuda:0")
def rel(a,b):
    a,b=a.detach().double().cpu(),b.detach().double().cpu(); return ((a-b).abs().max()/b.abs().max().clamp_min(1e-12)).item()
for norm, slope in (("instance", 1.0), ("batch", 1.0), ("batch", 0.2)):
    torch.manual_seed(10)
    o = O.OPatchDiscriminator(21, norm); o.apply(O.o_weights_init_normal)
    r = N.NLayerDiscriminator(21, norm_layer=N.get_norm_layer(norm)); r.load_state_dict(o.state_dict()); r.to(dev).train(); o.train()
    o = o.double()
    for m in o.modules():
        if isinstance(m, torch.nn.LeakyReLU): m.negative_slope = slope
    for m in r.modules():
        if hasattr(m, "SLOPE") and m.__class__.__name__ == "LeakyReLU": m.SLOPE = slope; m.negative_slope = slope
    x = torch.cat((O.synth_posemaps(3, seed=11), O.synth_images(3, seed=12)), 1)
    xo = x.double().clone().requires_grad_(True); xr = x.clone().to(dev).requires_grad_(True)
    # intermediate activations of the oracle
    acts = []
    h = xo
    for m in o.model:
        h = m(h); acts.append(h)
    yo = h
    yr = r(xr)
    print(norm, slope, "logits", rel(yr, yo))
    O.o_gan_loss(yo, True).backward(); GANLoss()(yr, True).backward()
    print("  dx", rel(xr.grad, xo.grad))
    og = dict(o.named_parameters())
    for n, p in r.named_parameters():
        print("  %-20s %.3e |ref|max %.3e" % (n, rel(p.grad, og[n].grad), og[n].grad.abs().max().item()))
