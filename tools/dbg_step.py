import sys, os, argparse
sys.path.insert(0, os.path.join(os.getcwd(), "reid-gan_amd")); sys.path.insert(0, os.getcwd())
import torch
from oracle import ref_torch as O
from fdgan.model import FDGANModel
sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from test_modules_gpu import _opt
torch.manual_seed(17)
b = 2
oE = O.OSiameseNet(O.OReidResNet(50, cut_at_pooling=True), O.OEltwiseSubEmbed(True, True, 2048, 2))
oDi = O.OSiameseNet(O.OReidResNet(50, cut_at_pooling=True), O.OEltwiseSubEmbed(True, True, 2048, 1))
for net in (oE, oDi):
    net.embed_model.classifier.weight.data.normal_(0, 0.05)
    for m in net.modules():
        if isinstance(m, (torch.nn.BatchNorm2d, torch.nn.BatchNorm1d)):
            m.running_mean.normal_(0, 0.05); m.running_var.uniform_(0.8, 1.2); m.weight.data.uniform_(0.4, 0.6)
oG = O.OPoseGenerator(128, 2048, 256, dropout=0.0); oG.apply(O.o_weights_init_normal)
oDp = O.OPatchDiscriminator(21); oDp.apply(O.o_weights_init_normal)
model = FDGANModel(_opt())
model.net_E.module.load_state_dict(oE.state_dict()); model.net_G.module.load_state_dict(oG.state_dict())
model.net_Di.module.load_state_dict(oDi.state_dict()); model.net_Dp.module.load_state_dict(oDp.state_dict())
model.reset_model_status()
ostep = O.OFDGANStep(oE, oG, oDi, oDp, lr=0.001, stage=2, lambda_recon=100.0, lambda_veri=10.0, lambda_sp=10.0)
origin, target, pose, labels, noise = O.synth_fdgan_batch(b, seed=100)
ref_losses, ref_fake = ostep.step(origin, target, pose, labels, noise)
pid1 = torch.arange(b); pid2 = torch.where(labels == 1, pid1, pid1 + 1000)
in1 = dict(pid=pid1, origin=origin[:b], target=target[:b], posemap=pose[:b], noise=noise[:b])
in2 = dict(pid=pid2, origin=origin[b:], target=target[b:], posemap=pose[b:])
model.set_input((in1, in2)); model.optimize_parameters()
print(ref_losses); print(model.get_current_errors())
for name, rn, on in (("E", model.net_E.module, oE), ("G", model.net_G.module, oG), ("Di", model.net_Di.module, oDi), ("Dp", model.net_Dp.module, oDp)):
    og = dict(on.named_parameters()); num = den = 0.0; worst = (0, None); nsign = ntot = 0
    for n, p in rn.named_parameters():
        if og[n].grad is None: continue
        a, r = p.grad.detach().double().cpu(), og[n].grad.double()
        d = a - r; num += d.pow(2).sum().item(); den += r.pow(2).sum().item()
        l2 = d.norm().item() / max(r.norm().item(), 1e-30)
        if l2 > worst[0]: worst = (l2, n)
        nsign += ((a * r) < 0).sum().item(); ntot += r.numel()
        z1 = ((r == 0) & (a != 0)); z2 = ((r != 0) & (a == 0))
        if z1.sum().item() + z2.sum().item() > 0 and name in ("E", "G"):
            print("   %-45s ref0&hip!=0: %d (max |hip| %.2e)  ref!=0&hip0: %d (max |ref| %.2e)  |ref|max %.2e" % (n, z1.sum().item(), a[z1].abs().max().item() if z1.any() else 0, z2.sum().item(), r[z2].abs().max().item() if z2.any() else 0, r.abs().max().item()))
    print(name, "global L2 grad err %.3e worst tensor %.3e %s; sign mismatches %d / %d; |g| rms %.3e" % ((num/den)**.5, worst[0], worst[1], nsign, ntot, (den/ntot)**.5))
    # magnitude distribution of oracle grads
    allg = torch.cat([p.grad.flatten().abs() for p in on.parameters() if p.grad is not None])
    q = torch.quantile(allg[torch.randperm(allg.numel())[:2000000]], torch.tensor([0.01, 0.1, 0.5, 0.9, 0.99]))
    print("   |g| quantiles 1/10/50/90/99%:", ["%.2e" % v for v in q.tolist()], "zeros:", (allg == 0).sum().item())
