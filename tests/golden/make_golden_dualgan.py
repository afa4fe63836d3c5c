"""Generates tests/golden/reference_dualgan.npz by importing the REFERENCE's own dual_gan modules (build container only).

Loaded by path from /root/reference/cluster-contrast-reid-main/dual_gan/models: base_function.py, PTM.py, networks.py,
external_function.py.  networks.py / PTM.py use package-relative imports, so empty package objects `dual_gan` and
`dual_gan.models` (with __path__ pointing INTO THE REFERENCE TREE) are registered; two imports of things that are not
installed and are never called on this path get empty placeholder module objects: `torchvision(.models)`
(external_function.py:3, used only by VGG19) and `clustercontrast.utils.data.diff_augs` (networks.py:8, used only by
Resize_ReID; the real file needs torchvision).  Nothing of the reference is copied into the repository.

For every fixture the oracle (oracle/ref_dualgan.py) is built with seeded weights, the SAME state_dict is loaded into the
reference module, both run on the same inputs, the script asserts agreement and stores the REFERENCE outputs.
`bicubic_normalize` has no importable reference (diff_augs needs torchvision): it is anchored on torch's own
F.interpolate, which is what torchvision's tensor resize calls.

Usage:  python tests/golden/make_golden_dualgan.py
"""
from __future__ import absolute_import, print_function

import importlib
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
from oracle import ref_dualgan as D  # noqa: E402
from tests.golden import cases_dualgan as C  # noqa: E402
from tests.golden.cases import sub  # noqa: E402

CC = "/root/reference/cluster-contrast-reid-main"


def check(a, b, what, tol=1e-5):
    a, b = a.detach().double(), b.detach().double()
    err = (a - b).abs().max().item()
    scale = max(b.abs().max().item(), 1e-12)
    print("  %-34s max|oracle-ref| = %.3e (scale %.3e)" % (what, err, scale))
    assert err <= tol * scale + 1e-9, what


def import_reference():
    for name in ("torchvision", "torchvision.models"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["torchvision"].models = sys.modules["torchvision.models"]
    for name in ("clustercontrast", "clustercontrast.utils", "clustercontrast.utils.data"):
        sys.modules.setdefault(name, types.ModuleType(name))
    shell = types.ModuleType("clustercontrast.utils.data.diff_augs")
    shell.my_transform = shell.my_normalize = None
    # Resize_ReID (networks.py:160) calls my_resize: torchvision's tensor resize is F.interpolate(bicubic) — the anchor the
    # bicubic fixture already uses; everything after it in that module is the reference's own code
    shell.my_resize = lambda X, size=(256, 128): torch.nn.functional.interpolate(X, size=size, mode='bicubic', align_corners=False)
    sys.modules["clustercontrast.utils.data.diff_augs"] = shell
    pkg = types.ModuleType("dual_gan")
    pkg.__path__ = [os.path.join(CC, "dual_gan")]
    sys.modules["dual_gan"] = pkg
    sub_pkg = types.ModuleType("dual_gan.models")
    sub_pkg.__path__ = [os.path.join(CC, "dual_gan", "models")]
    sys.modules["dual_gan.models"] = sub_pkg
    net = importlib.import_module("dual_gan.models.networks")
    ext = importlib.import_module("dual_gan.models.external_function")
    ptm = importlib.import_module("dual_gan.models.PTM")
    return net, ext, ptm


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    ref_net, ref_ext, ref_ptm = import_reference()
    out = {}

    print("PCTM")
    on, (q, v) = C.pctm_case()
    rn = ref_ptm.PCTM(d_model=64, nhead=2, num_CABs=2, num_TTBs=2, dim_feedforward=64, activation="LeakyReLU",
                      affine=True, norm='instance')
    rn.load_state_dict(on.state_dict())
    yo, yr = on(q, v), rn(q, v)
    check(yo, yr, "pctm_fwd")
    out["pctm_fwd"], out["pctm_fwd_stats"] = sub(yr)

    print("PoseGenerator1")
    on, (feat, pose) = C.posegen1_case()
    rn = ref_net.PoseGenerator1(64, 18, 256, 3, 'instance', 'LeakyReLU', False, False, 3, True, 2, 2, 2)
    rn.load_state_dict(on.state_dict())
    rn.train()
    feat_r = feat.clone().requires_grad_(True)
    feat_o = feat.clone().requires_grad_(True)
    yo, yr = on(feat_o, pose), rn(feat_r, pose)
    check(yo, yr, "posegen1_fwd")
    g = torch.Generator().manual_seed(5)
    dy = torch.randn(yr.shape, generator=g)
    yo.backward(dy)
    yr.backward(dy)
    check(feat_o.grad, feat_r.grad, "posegen1_dfeat", 1e-4)
    out["posegen1_fwd"], out["posegen1_fwd_stats"] = sub(yr)
    out["posegen1_dfeat"], out["posegen1_dfeat_stats"] = sub(feat_r.grad)
    gsel = ["block0.model.0.weight", "encoder1.model.5.weight", "feature_block.model.0.weight",
            "PCTM.encoder.layers.0.self_attn.in_proj_weight", "PCTM.decoder.layers.1.multihead_attn.out_proj.weight",
            "PCTM.decoder.layers.0.linear1.weight", "PCTM.decoder.norm.weight", "decoder0.model.0.weight",
            "decoder2.shortcut.0.weight", "outconv.conv1.bias"]
    pr, po = dict(rn.named_parameters()), dict(on.named_parameters())
    for k in gsel:
        check(po[k].grad, pr[k].grad, "posegen1 grad " + k, 2e-4)
        out["posegen1_g_" + k], _ = sub(pr[k].grad)

    print("AEGenerator")
    on, x = C.aegen_case()
    rn = ref_net.AEGenerator(3, 64, 256, 3, 'instance', 'LeakyReLU', False, False, 3, 3)
    rn.load_state_dict(on.state_dict())
    rn.train()
    yo, yr = on(x), rn(x)
    check(yo, yr, "aegen_fwd")
    check(on.forward_enc(x), rn.forward_enc(x), "aegen_enc")
    out["aegen_fwd"], out["aegen_fwd_stats"] = sub(yr)
    out["aegen_enc"], out["aegen_enc_stats"] = sub(rn.forward_enc(x))

    print("DECGenerator1 / DECGenerator")
    for tag, case, ctor, gsel in (
            ("decgen1", C.decgen1_case,
             lambda: ref_net.DECGenerator1(64, 256, 3, 'instance', 'LeakyReLU', False, False, 3, 3),
             ["feature_block.model.0.weight", "mblock0.conv1.weight", "mblock2.bypass.bias", "decoder0.model.2.weight",
              "decoder2.shortcut.0.weight", "outconv.conv1.weight"]),
            ("decgen", C.decgen_case,
             lambda: ref_net.DECGenerator(64, 2048, 3, 'instance', 'LeakyReLU', False, False, 3),
             ["resblock.conv1.weight", "resblock.bypass.weight", "decoder1.model.5.weight", "outconv.conv1.bias"])):
        on, feat = case()
        rn = ctor()
        rn.load_state_dict(on.state_dict())
        rn.train()
        feat_r, feat_o = feat.clone().requires_grad_(True), feat.clone().requires_grad_(True)
        yo, yr = on(feat_o), rn(feat_r)
        check(yo, yr, tag + "_fwd")
        g = torch.Generator().manual_seed(7)
        dy = torch.randn(yr.shape, generator=g)
        yo.backward(dy)
        yr.backward(dy)
        check(feat_o.grad, feat_r.grad, tag + "_dfeat", 1e-4)
        out[tag + "_fwd"], out[tag + "_fwd_stats"] = sub(yr)
        out[tag + "_dfeat"], out[tag + "_dfeat_stats"] = sub(feat_r.grad)
        pr, po = dict(rn.named_parameters()), dict(on.named_parameters())
        for k in gsel:
            check(po[k].grad, pr[k].grad, tag + " grad " + k, 2e-4)
            out[tag + "_g_" + k], _ = sub(pr[k].grad)

    print("Resize_ReID (--use_adp)")
    on, x = C.resize_reid_case()
    rn = ref_net.Resize_ReID(image_nc=3)
    rn.load_state_dict(on.state_dict())
    rn.train()
    xo, xr = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    yo, yr = on(xo), rn(xr)
    check(yo, yr, "resize_reid_fwd")
    g = torch.Generator().manual_seed(8)
    dy = torch.randn(yr.shape, generator=g)
    yo.backward(dy)
    yr.backward(dy)
    check(xo.grad, xr.grad, "resize_reid_dx", 1e-4)
    out["resize_reid_fwd"], out["resize_reid_fwd_stats"] = sub(yr)
    out["resize_reid_dx"], out["resize_reid_dx_stats"] = sub(xr.grad)
    pr, po = dict(rn.named_parameters()), dict(on.named_parameters())
    for k in ["resblock1.conv1.weight_orig", "resblock2.bypass.weight_orig", "resblock3.conv2.bias", "resblock2.model.0.weight"]:
        check(po[k].grad, pr[k].grad, "resize_reid grad " + k, 2e-4)
        out["resize_reid_g_" + k], _ = sub(pr[k].grad)

    print("FDGenerator (--model_gen FD)")
    on, (feat, noise) = C.fdgen_case()
    rn = ref_net.FDGenerator(256, 64, output_nc=3, noise_nc=512, fuse_mode='add')
    rn.load_state_dict(on.state_dict())
    rn.train()
    fo, fr = feat.clone().requires_grad_(True), feat.clone().requires_grad_(True)
    no, nr = noise.clone().requires_grad_(True), noise.clone().requires_grad_(True)
    yo, yr = on(fo, no), rn(fr, nr)
    check(yo, yr, "fdgen_fwd")
    g = torch.Generator().manual_seed(9)
    dy = torch.randn(yr.shape, generator=g)
    yo.backward(dy)
    yr.backward(dy)
    check(fo.grad, fr.grad, "fdgen_dfeat", 1e-4)
    check(no.grad, nr.grad, "fdgen_dnoise", 1e-4)
    out["fdgen_fwd"], out["fdgen_fwd_stats"] = sub(yr)
    out["fdgen_dfeat"], _ = sub(fr.grad)
    out["fdgen_dnoise"], _ = sub(nr.grad)
    pr, po = dict(rn.named_parameters()), dict(on.named_parameters())
    for k in ["W_reid.weight", "W_noise.weight", "de_avg.1.weight", "de_conv4.1.weight", "de_conv2.2.weight", "de_conv1.1.weight"]:
        check(po[k].grad, pr[k].grad, "fdgen grad " + k, 2e-4)
        out["fdgen_g_" + k], _ = sub(pr[k].grad)

    print("DPTNGenerator")
    on, (xs, ps, pt) = C.dptn_case()
    rn = ref_net.DPTNGenerator(3, 18, 64, 256, 3, 'instance', 'LeakyReLU', False, False, 3, 3, True, 2, 2, 2)
    rn.load_state_dict(on.state_dict())
    rn.train()
    (to, so), (tr, sr) = on(xs, ps, pt), rn(xs, ps, pt)
    check(to, tr, "dptn_fwd_t")
    check(so, sr, "dptn_fwd_s")
    g = torch.Generator().manual_seed(6)
    dt, ds = torch.randn(tr.shape, generator=g), torch.randn(sr.shape, generator=g)
    ((to * dt).sum() + (so * ds).sum()).backward()
    ((tr * dt).sum() + (sr * ds).sum()).backward()
    out["dptn_fwd_t"], out["dptn_fwd_t_stats"] = sub(tr)
    out["dptn_fwd_s"], out["dptn_fwd_s_stats"] = sub(sr)
    pr, po = dict(rn.named_parameters()), dict(on.named_parameters())
    for k in ["block0.model.0.weight", "mblock1.conv2.weight", "PTM.decoder.layers.0.multihead_attn.in_proj_weight",
              "source_encoder.encoder1.model.2.weight", "decoder1.model.2.weight", "outconv.conv1.weight"]:
        check(po[k].grad, pr[k].grad, "dptn grad " + k, 2e-4)
        out["dptn_g_" + k], _ = sub(pr[k].grad)
    check(on(xs, ps, pt, False)[0], rn(xs, ps, pt, False)[0], "dptn_fwd_t (is_train=False)")

    print("ResDiscriminator (spectral norm)")
    on, x = C.resdisc_case()
    rn = ref_net.ResDiscriminator(3, 32, 128, 3, 'none', 'LeakyReLU', True)
    rn.load_state_dict(on.state_dict())
    rn.train()
    for it in range(2):                       # two training forwards: u, v advance one power iteration each
        yo, yr = on(x), rn(x)
        check(yo, yr, "resdisc_fwd%d" % it)
        out["resdisc_fwd%d" % it] = yr.detach().flatten().numpy().astype(np.float64)
    (yo ** 2).mean().backward()
    (yr ** 2).mean().backward()
    pr, po = dict(rn.named_parameters()), dict(on.named_parameters())
    for k in ["block0.model.0.weight_orig", "encoder1.model.3.weight_orig", "encoder0.shortcut.1.weight_orig", "conv.weight_orig",
              "conv.bias"]:
        check(po[k].grad, pr[k].grad, "resdisc grad " + k, 1e-4)
        out["resdisc_g_" + k], _ = sub(pr[k].grad)
    out["resdisc_u_block0"] = dict(rn.named_buffers())["block0.model.0.weight_u"].numpy().astype(np.float64)

    print("GANLoss lsgan")
    pred = C.lsgan_case()
    crit = ref_ext.GANLoss('lsgan')
    vals = []
    for real in (True, False):
        check(D.o_lsgan(pred, real, True), crit(pred, real, True), "lsgan disc %s" % real)
        check(D.o_lsgan(pred, real, False), crit(pred, real, False), "lsgan gen %s" % real)
        vals += [crit(pred, real, True).item(), crit(pred, real, False).mean().item()]
    out["lsgan"] = np.array(vals)

    print("CM_gan (clustercontrast/models/cm.py, loaded by path)")
    import importlib.util
    from oracle import ref_torch as O
    from tests.golden import cases as C0
    spec = importlib.util.spec_from_file_location("ref_cc_cm", os.path.join(CC, "clustercontrast/models/cm.py"))
    ref_cm = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref_cm)
    bank, feats, labels, gout = C0.cm_case()
    gbank = torch.nn.functional.normalize(bank.flip(0) + 0.1, dim=1)
    gfeat = feats.flip(1) * 3.0 + 0.2
    res = []
    for fn in (O.OCMGan, ref_cm.CM_gan):
        b1, b2 = bank.clone(), gbank.clone()
        x = feats.clone().requires_grad_(True)
        y = fn.apply(x, gfeat, labels, b1, b2, torch.Tensor([0.2]))
        y.backward(gout)
        res.append((y.detach(), x.grad, b1, b2))
    for a, b, what in zip(res[0], res[1], ("logits", "grad", "bank", "gan_bank")):
        check(a, b, "cm_gan " + what)
    out["cm_gan_logits"], _ = sub(res[1][0])
    out["cm_gan_grad"], _ = sub(res[1][1])
    out["cm_gan_bank"], _ = sub(res[1][2])
    out["cm_gan_gbank"], _ = sub(res[1][3])

    print("bicubic + normalize (torch anchor)")
    x = C.bicubic_case()
    y = D.o_my_transform(x, (64, 32))
    out["bicubic_normalize"], out["bicubic_normalize_stats"] = sub(y)

    path = os.path.join(HERE, "reference_dualgan.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays")


if __name__ == "__main__":
    main()
