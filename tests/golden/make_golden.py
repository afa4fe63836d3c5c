"""Generates tests/golden/*.npz by importing the REFERENCE's own modules (build container only:
/root/reference is mounted here and does not exist on the GPU box).

For every fixture the oracle restatement (oracle/ref_torch.py) is built with seeded weights, the SAME
state_dict is loaded into the reference module, both run on the SAME seeded inputs on the CPU, the script
asserts they agree (<= 1e-5 relative) and stores the REFERENCE outputs (sub-sampled) as the fixture.
tests/test_oracle_golden.py later re-checks the oracle against these files without the reference.

Reference files are loaded by path; nothing from them is copied into the repository.  `FD/fdgan/networks.py`
has an unused top-level `import torchvision` (:11) — torchvision is not installed in this image, so an empty
placeholder module object is registered for that one import (recipe recorded in SURVEY.md §8c).  The
torchvision ResNet itself is NOT available: the trunk is pinned against the reference's in-tree
`resnet_ibn_a.Bottleneck(ibn=False)` assembled [3,4,6,3] by this script following `_make_layer` (:141-159).

Usage:  python tests/golden/make_golden.py
"""
from __future__ import absolute_import, print_function

import importlib.util
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
from oracle import ref_torch as O  # noqa: E402
from tests.golden import cases as C  # noqa: E402
from tests.golden.cases import sub  # noqa: E402

FD = "/root/reference/FD-GAN-master"
CC = "/root/reference/cluster-contrast-reid-main"


def load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def check(a, b, what, tol=1e-5):
    a, b = a.detach().double(), b.detach().double()
    err = (a - b).abs().max().item()
    scale = max(b.abs().max().item(), 1e-12)
    print("  %-34s max|oracle-ref| = %.3e (scale %.3e)" % (what, err, scale))
    assert err <= tol * scale + 1e-9, what


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    sys.modules.setdefault("torchvision", types.ModuleType("torchvision"))
    ref_net = load(os.path.join(FD, "fdgan/networks.py"), "ref_fd_networks")
    ref_loss = load(os.path.join(FD, "fdgan/losses.py"), "ref_fd_losses")
    ref_embed = load(os.path.join(FD, "reid/models/embedding.py"), "ref_embedding")
    ref_multi = load(os.path.join(FD, "reid/models/multi_branch.py"), "ref_multi_branch")
    ref_cm = load(os.path.join(CC, "clustercontrast/models/cm.py"), "ref_cm")
    ref_pool = load(os.path.join(CC, "clustercontrast/models/pooling.py"), "ref_pooling")
    ref_ibn = load(os.path.join(CC, "clustercontrast/models/resnet_ibn_a.py"), "ref_resnet_ibn_a")
    out = {}

    # ---- generator (train-mode BN, drop = 0) ---------------------------------------------------
    for cl in (0, 2):
        og, (pose, feat, z) = C.generator_case(cl)
        rg = ref_net.CustomPoseGenerator(128, 2048, 256, dropout=0.0,
                                         norm_layer=ref_net.get_norm_layer('batch'), connect_layers=cl)
        rg.load_state_dict(og.state_dict())
        rg.train()
        yo, yr = og(pose.clone(), feat.clone(), z.clone()), rg(pose.clone(), feat.clone(), z.clone())
        check(yo, yr, "G fwd connect_layers=%d" % cl)
        out["g_fwd_cl%d" % cl], out["g_fwd_cl%d_stats" % cl] = sub(yr)

    # ---- PatchGAN discriminator ----------------------------------------------------------------
    od, x = C.discriminator_case('batch')
    rd = ref_net.NLayerDiscriminator(21, norm_layer=ref_net.get_norm_layer('batch'))
    rd.load_state_dict(od.state_dict())
    yo, yr = od(x.clone()), rd(x.clone())
    check(yo, yr, "D_pd fwd")
    out["dp_fwd"], out["dp_fwd_stats"] = sub(yr)
    # instance-norm variant (bias on, affine off)
    od, x = C.discriminator_case('instance')
    rd = ref_net.NLayerDiscriminator(21, norm_layer=ref_net.get_norm_layer('instance'))
    rd.load_state_dict(od.state_dict())
    yo, yr = od(x.clone()), rd(x.clone())
    check(yo, yr, "D_pd fwd (instance norm)")
    out["dp_in_fwd"], out["dp_in_fwd_stats"] = sub(yr)

    # ---- GANLoss -------------------------------------------------------------------------------
    pred = C.ganloss_case()
    crit = ref_loss.GANLoss(smooth=False)
    vals = []
    for real in (True, False):
        lo, lr = O.o_gan_loss(pred, real), crit(pred, real)
        check(lo, lr, "GANLoss real=%s" % real)
        vals.append(lr.item())
    out["ganloss"] = np.array(vals)

    # ---- EltwiseSubEmbed / SiameseNet ----------------------------------------------------------
    oe, (f1, f2) = C.embed_case()
    re_ = ref_embed.EltwiseSubEmbed(use_batch_norm=True, use_classifier=True, num_features=2048, num_classes=2)
    re_.load_state_dict(oe.state_dict())
    for mode in ("train", "eval"):
        getattr(oe, mode)()
        getattr(re_, mode)()
        yo, yr = oe(f1, f2), re_(f1, f2)
        check(yo, yr, "EltwiseSubEmbed %s" % mode)
        out["embed_" + mode] = yr.detach().numpy().astype(np.float64)
        re_.load_state_dict(oe.state_dict())        # keep running stats aligned

    # ---- Bottleneck and the [3,4,6,3] trunk assembled from the reference's Bottleneck ------------
    oreid, trunk_sd, imgs = C.trunk_case()

    class RefTrunk(nn.Module):                # assembly only; every block is the reference's class
        def __init__(self):
            super(RefTrunk, self).__init__()
            self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
            self.bn1 = nn.BatchNorm2d(64)
            self.relu = nn.ReLU(inplace=True)
            self.maxpool = nn.MaxPool2d(3, 2, 1)
            cin = 64
            for li, (w, n, s) in enumerate(((64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2)), 1):
                blocks = []
                for bi in range(n):
                    st = s if bi == 0 else 1
                    ds = None
                    if st != 1 or cin != w * 4:
                        ds = nn.Sequential(nn.Conv2d(cin, w * 4, 1, st, bias=False), nn.BatchNorm2d(w * 4))
                    blocks.append(ref_ibn.Bottleneck(cin, w, False, st, ds))
                    cin = w * 4
                setattr(self, "layer%d" % li, nn.Sequential(*blocks))

        def forward(self, x):
            x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
            return self.layer4(self.layer3(self.layer2(self.layer1(x))))

    rt = RefTrunk()
    sd = {k: v for k, v in trunk_sd.items() if not k.startswith("fc.")}
    rt.load_state_dict(sd)
    for mode in ("eval", "train"):
        getattr(rt, mode)()
        getattr(oreid, mode)()
        fr = rt(imgs.clone())
        fr = F.avg_pool2d(fr, fr.shape[2:]).flatten(1)       # wrapper logic of FD/reid/models/resnet.py:71-72
        fo = oreid(imgs.clone())
        check(fo, fr, "ResNet-50 trunk + avgpool (%s)" % mode, tol=2e-5)
        out["resnet50_trunk_%s" % mode], out["resnet50_trunk_%s_stats" % mode] = sub(fr)
        rt.load_state_dict(sd)
        oreid.base.load_state_dict(trunk_sd)

    # ---- GeM pooling ---------------------------------------------------------------------------
    og_, xg, wsum = C.gem_case()
    rg_ = ref_pool.GeneralizedMeanPoolingP()
    xo, xr = xg.clone().requires_grad_(True), xg.clone().requires_grad_(True)
    yo, yr = og_(xo), rg_(xr)
    check(yo, yr, "GeM fwd")
    (yo * wsum).sum().backward()
    (yr * wsum).sum().backward()
    check(xo.grad, xr.grad, "GeM dx")
    check(og_.p.grad, rg_.p.grad, "GeM dp")
    out["gem_fwd"] = yr.detach().flatten().numpy().astype(np.float64)
    out["gem_dp"] = rg_.p.grad.numpy().astype(np.float64)
    out["gem_dx"], _ = sub(xr.grad)

    # ---- CM / CM_Hard: logits, input gradient (pre-update bank) and the bank after the ordered update ----
    bank, feats, labels, gout = C.cm_case()
    for name, rfn, ofn in (("cm", ref_cm.cm, O.OCM), ("cm_hard", ref_cm.cm_hard, O.OCMHard)):
        b_r, b_o = bank.clone(), bank.clone()
        xr, xo = feats.clone().requires_grad_(True), feats.clone().requires_grad_(True)
        yr = rfn(xr, labels, b_r, 0.2)
        yo = ofn.apply(xo, labels, b_o, torch.Tensor([0.2]))
        yr.backward(gout)
        yo.backward(gout)
        check(yo, yr, name + " logits")
        check(xo.grad, xr.grad, name + " grad_inputs")
        check(b_o, b_r, name + " bank after update")
        out[name + "_logits"], _ = sub(yr)
        out[name + "_grad"], _ = sub(xr.grad)
        out[name + "_bank"], out[name + "_bank_stats"] = sub(b_r, 2048)

    np.savez_compressed(os.path.join(HERE, "reference_modules.npz"), **out)
    print("wrote", os.path.join(HERE, "reference_modules.npz"), "with", len(out), "arrays")


if __name__ == "__main__":
    main()
