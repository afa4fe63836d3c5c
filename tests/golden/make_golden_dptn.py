"""Generates tests/golden/reference_dptn.npz by running the REFERENCE's own DPTNModel code on the CPU (build container only).

What is imported from /root/reference/cluster-contrast-reid-main/dual_gan/models (by package path, nothing is copied):
`DPTN_model.py` (DPTNModel: forward, backward_D_basic, backward_D, backward_G_basic, backward_G, optimize_parameters),
`external_function.py` (GANLoss in all four modes, cal_gradient_penalty, VGGLoss, VGG19), `networks.py`, `base_function.py`,
`PTM.py`, `base_model.py`.

Placeholders (empty module objects) stand in for imports that are absent here and are never called on the training step:
`cv2`, `dual_gan.gan_util` (the file does not exist in the reference tree), `clustercontrast.utils.data.pose_utils`
(needs skimage) and `clustercontrast.utils.data.diff_augs` (needs torchvision) — visualisation / adaptor helpers.
torchvision itself is absent (no wheel offline): `torchvision.models.vgg19` is provided by the oracle's restatement of the
published VGG-19 feature stack with SEEDED weights (oracle.ref_dualgan.o_tv_vgg19_features — third-party arithmetic,
"unpinned"); the reference's own `VGG19` slicing and `VGGLoss` run on top of it and are what this script pins.
`DPTNModel.set_input` is `.cuda()`-bound: the four input tensors are assigned directly.  The wgangp penalty draws its
interpolation weights with torch.rand (external_function.py:89): the script seeds the global generator right before the
reference call and hands the same draw to the oracle.

For every fixture the oracle (oracle/ref_dualgan.py) runs on the same weights and inputs; the script asserts agreement
and stores the REFERENCE values.      Usage:  python tests/golden/make_golden_dptn.py
"""
from __future__ import absolute_import, print_function

import argparse
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
from tests.golden import cases_dptn as C  # noqa: E402
from tests.golden.cases import sub  # noqa: E402

CC = "/root/reference/cluster-contrast-reid-main"


def check(a, b, what, tol=2e-5):
    a, b = torch.as_tensor(a).detach().double(), torch.as_tensor(b).detach().double()
    err = (a - b).abs().max().item()
    scale = max(b.abs().max().item(), 1e-12)
    print("  %-44s max|oracle-ref| = %.3e (scale %.3e)" % (what, err, scale))
    assert err <= tol * scale + 1e-9, what


def import_reference():
    tv, tvm = types.ModuleType("torchvision"), types.ModuleType("torchvision.models")
    tvm.vgg19 = lambda pretrained=True: types.SimpleNamespace(features=C.vgg_features())
    tv.models = tvm
    sys.modules["torchvision"], sys.modules["torchvision.models"] = tv, tvm
    for name in ("cv2", "clustercontrast", "clustercontrast.utils", "clustercontrast.utils.data",
                 "clustercontrast.utils.data.pose_utils"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["clustercontrast.utils.data"].pose_utils = sys.modules["clustercontrast.utils.data.pose_utils"]
    shell = types.ModuleType("clustercontrast.utils.data.diff_augs")
    shell.my_resize = shell.my_transform = shell.my_normalize = None
    sys.modules["clustercontrast.utils.data.diff_augs"] = shell
    pkg = types.ModuleType("dual_gan")
    pkg.__path__ = [os.path.join(CC, "dual_gan")]
    pkg.gan_util = types.ModuleType("dual_gan.gan_util")
    sys.modules["dual_gan"], sys.modules["dual_gan.gan_util"] = pkg, pkg.gan_util
    sub_pkg = types.ModuleType("dual_gan.models")
    sub_pkg.__path__ = [os.path.join(CC, "dual_gan", "models")]
    sys.modules["dual_gan.models"] = sub_pkg
    ext = importlib.import_module("dual_gan.models.external_function")
    dptn = importlib.import_module("dual_gan.models.DPTN_model")
    return ext, dptn


def ref_opt(gan_mode):
    return argparse.Namespace(
        gan_train=True, checkpoints_dir="/tmp/rg_golden", name="dptn", load_pretrain="", old_size=(128, 64), model_gen="DPTN",
        image_nc=3, pose_nc=18, norm="instance", use_spect_g=False, use_spect_d=True, use_coord=False, nhead=2, num_CABs=2,
        num_TTBs=2, use_adp=False, dis_layers=3, init_type="orthogonal", verbose=False, pool_size=0, gan_lr=2e-4,
        gan_mode=gan_mode, device="cpu", beta1=0.5, ratio_g2d=0.1, gan_lr_policy="lambda", iter_start=0, niter=100,
        niter_decay=100, continue_train=False, which_epoch="latest", lambda_rec=C.LAMBDAS["lambda_rec"],
        lambda_g=C.LAMBDAS["lambda_g"], lambda_style=C.LAMBDAS["lambda_style"], lambda_content=C.LAMBDAS["lambda_content"],
        t_s_ratio=C.LAMBDAS["t_s_ratio"], gpu_ids=[])


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    ref_ext, ref_dptn = import_reference()
    out = {}

    print("GANLoss, all modes (external_function.py:46-69)")
    pred = C.ganloss_case()
    for mode in C.GAN_MODES:
        crit = ref_ext.GANLoss(mode)
        vals = []
        for real in (True, False):
            for is_disc in (True, False):
                r, o = crit(pred, real, is_disc), D.o_ganloss(pred, real, is_disc, mode)
                check(o, r, "ganloss %s real=%s disc=%s" % (mode, real, is_disc))
                vals.append(r.mean().item())
        out["ganloss_" + mode] = np.array(vals)

    print("VGG19 slicing + VGGLoss (external_function.py:107-147,226-347) on the seeded VGG-19 stack")
    x, y = C.vgg_pair()
    ref_vgg = ref_ext.VGGLoss()
    ora_vgg = D.OVGGLoss(D.OVGG19(C.vgg_features()))
    xr, xo = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    (cr, sr), (co, so) = ref_vgg(xr, y), ora_vgg(xo, y)
    check(co, cr, "vgg content")
    check(so, sr, "vgg style")
    (cr + 500.0 * sr).backward()
    (co + 500.0 * so).backward()
    check(xo.grad, xr.grad, "vgg d/dx", 1e-4)
    out["vgg_losses"] = np.array([cr.item(), sr.item()])
    out["vgg_dx"], out["vgg_dx_stats"] = sub(xr.grad)
    feats = ref_vgg.vgg(x)
    out["vgg_relu_means"] = np.array([feats[k].mean().item() for k, _, _ in D._VGG_SLICES])

    print("cal_gradient_penalty (external_function.py:72-104)")
    _, net_D = C.nets()
    ref_D = ref_dptn.networks.ResDiscriminator(3, 32, 128, 3, 'none', 'LeakyReLU', True)
    ref_D.load_state_dict(net_D.state_dict())
    ref_D.train()
    d = C.inputs()
    real, fake = d['Xt'], d['Xs']
    torch.manual_seed(77)
    gp_r, gr = ref_ext.cal_gradient_penalty(ref_D, real, fake)
    torch.manual_seed(77)
    alpha = torch.rand(real.shape[0], 1)
    gp_o, go = D.o_cal_gradient_penalty(net_D, real, fake, alpha)
    check(gp_o, gp_r, "gradient penalty")
    check(go, gr, "gp input gradients", 1e-4)
    gp_r.backward()
    gp_o.backward()
    pr, po = dict(ref_D.named_parameters()), dict(net_D.named_parameters())
    for k in C.PROBES_D:
        check(po[k].grad, pr[k].grad, "gp grad " + k, 2e-4)
        out["gp_g_" + k], _ = sub(pr[k].grad)
    out["gp_value"] = np.array([gp_r.item()])
    out["gp_alpha"] = alpha.numpy().astype(np.float64)

    print("DPTNModel.optimize_parameters (DPTN_model.py:216-225), 2 steps per mode")
    for mode, with_vgg in (("hinge", False), ("hinge", True), ("vanilla", False), ("wgangp", False)):
        tag = mode + ("_vgg" if with_vgg else "")
        print(" mode", tag)
        om = C.model(mode, with_vgg)
        rm = ref_dptn.DPTNModel(ref_opt(mode))
        rm.net_G.load_state_dict(om.net_G.state_dict())
        rm.net_D.load_state_dict(om.net_D.state_dict())
        rm.net_G.train()
        rm.net_D.train()
        if not with_vgg:
            # BASELINE config 5 runs without the perceptual terms (no ImageNet weights): zero losses of the right type
            rm.Vggloss = lambda a, b: (torch.zeros(()), torch.zeros(()))
        d = C.inputs()
        for step in range(2):
            rm.source_image, rm.source_pose = d['Xs'], d['Ps']
            rm.target_image, rm.target_pose = d['Xt'], d['Pt']
            om.set_input(d)
            if mode == "wgangp":
                torch.manual_seed(300 + step)
                om.gp_alpha = C.gp_alpha(step)
                torch.manual_seed(300 + step)                  # global generator: same draw as Generator(300 + step)
                assert torch.equal(torch.rand(C.B, 1), om.gp_alpha)
                torch.manual_seed(300 + step)
            rm.optimize_parameters()
            om.optimize_parameters()
            re, oe = rm.get_current_errors(), om.get_current_errors()
            for k in re:
                check(oe[k], re[k], "%s step %d loss %s" % (tag, step, k), 1e-4)
            out["dptn_%s_losses_%d" % (tag, step)] = np.array([re[k] for k in re])
            check(om.fake_image_t, rm.fake_image_t, "%s step %d fake_t" % (tag, step), 1e-4)
            out["dptn_%s_fake_t_%d" % (tag, step)], _ = sub(rm.fake_image_t.detach())
            out["dptn_%s_fake_s_%d" % (tag, step)], _ = sub(rm.fake_image_s.detach())
        pr, po = dict(rm.net_G.named_parameters()), dict(om.net_G.named_parameters())
        for k in C.PROBES_G:
            check(po[k], pr[k], "%s param %s" % (tag, k), 1e-4)
            out["dptn_%s_p_%s" % (tag, k)], _ = sub(pr[k].detach())
        pr, po = dict(rm.net_D.named_parameters()), dict(om.net_D.named_parameters())
        for k in C.PROBES_D:
            check(po[k], pr[k], "%s D param %s" % (tag, k), 1e-4)
            out["dptn_%s_pd_%s" % (tag, k)], _ = sub(pr[k].detach())

    print("lsgan: the reference fails in backward_G (non-scalar loss_ad_gen_t, DPTN_model.py:203-213)")
    rm = ref_dptn.DPTNModel(ref_opt("lsgan"))
    rm.Vggloss = lambda a, b: (torch.zeros(()), torch.zeros(()))
    d = C.inputs()
    rm.source_image, rm.source_pose, rm.target_image, rm.target_pose = d['Xs'], d['Ps'], d['Xt'], d['Pt']
    try:
        rm.optimize_parameters()
        raise AssertionError("expected the reference to fail with lsgan")
    except RuntimeError as e:
        print("  reference raised RuntimeError:", str(e)[:90])
        out["lsgan_error"] = np.frombuffer(str(e).encode(), dtype=np.uint8).astype(np.float64)

    path = os.path.join(HERE, "reference_dptn.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays")


if __name__ == "__main__":
    main()
