"""BASELINE config 5 on the MI355X: the DPTNModel step (dual_gan/models/DPTN_model.py on the HIP tape programs) against
the oracle restatement (oracle/ref_dualgan.ODPTNModel, which make_golden_dptn.py found bit-identical to the reference's own
DPTNModel on the CPU) and directly against the committed reference fixtures (tests/golden/reference_dptn.npz).

fp32 path: losses, generated images and post-step parameters within 1e-3 (max norm) over two optimizer steps for the hinge,
vanilla and wgangp (gradient penalty: second-order weight gradients) objectives and with the VGG perceptual / style terms.
fp8 path: DECLARED tolerance — every loss within 5e-2 relative (+1e-3 absolute for losses near zero) of the fp32 oracle,
generated images within 6e-2 relative L2.
"""
import argparse
import os

import numpy as np
import pytest
import torch

from oracle import ref_dualgan as D
from tests.golden import cases_dptn as C
from tests.golden.cases import sub
from tests.test_modules_gpu import _check, _check_l2

pytestmark = pytest.mark.gpu

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_dptn.npz"))
NAMES = ['app_gen_s', 'content_gen_s', 'style_gen_s', 'app_gen_t', 'ad_gen_t', 'dis_img_gen_t', 'content_gen_t', 'style_gen_t']


def _opt(gan_mode, no_vgg=True, conv_dtype="fp32"):
    return argparse.Namespace(
        model="DPTN", gan_train=True, checkpoints_dir="/tmp/rg_ckpt", name="dptn_t", load_pretrain="", model_gen="DPTN",
        image_nc=3, pose_nc=18, norm="instance", use_spect_g=False, use_spect_d=True, use_coord=False, nhead=2, num_CABs=2,
        num_TTBs=2, use_adp=False, dis_layers=3, init_type="orthogonal", verbose=False, pool_size=0, gan_lr=2e-4,
        gan_mode=gan_mode, beta1=0.5, ratio_g2d=0.1, gan_lr_policy="lambda", iter_start=0, niter=100, niter_decay=100,
        continue_train=False, which_epoch="latest", no_vgg_loss=no_vgg, vgg_weights="", conv_dtype=conv_dtype,
        old_size=(128, 64), **C.LAMBDAS)


def _build(dev, gan_mode, with_vgg=False, conv_dtype="fp32"):
    from dual_gan.models.models import create_model
    om = C.model(gan_mode, with_vgg)
    m = create_model(_opt(gan_mode, no_vgg=not with_vgg, conv_dtype=conv_dtype))
    assert m.name() == "DPTNModel"
    m.net_G.module.load_state_dict(om.net_G.state_dict())
    m.net_D.module.load_state_dict(om.net_D.state_dict())
    if with_vgg:
        m.Vggloss.vgg.load_torchvision({"features." + k: v for k, v in C.vgg_features().state_dict().items()})
    m.net_G.train()
    m.net_D.train()
    return m, om


def _run_steps(m, om, dev, gan_mode, steps=2):
    d = C.inputs()
    dd = {k: v.to(dev) for k, v in d.items()}
    out = []
    for step in range(steps):
        if gan_mode == "wgangp":
            om.gp_alpha = C.gp_alpha(step)
            m.gp_alpha = C.gp_alpha(step)
        om.step(d)
        m.set_input(dd)
        m.optimize_parameters()
        out.append((m.get_current_errors(), om.get_current_errors(), m.fake_image_t.detach().cpu(), om.fake_image_t.detach()))
    return out


@pytest.mark.parametrize("gan_mode,with_vgg", [("hinge", False), ("vanilla", False), ("wgangp", False), ("hinge", True)])
def test_dptn_step_matches_oracle_and_reference_fixture(dev, gan_mode, with_vgg):
    m, om = _build(dev, gan_mode, with_vgg)
    tag = gan_mode + ("_vgg" if with_vgg else "")
    res = _run_steps(m, om, dev, gan_mode)
    for step, (got, ref, fake, ofake) in enumerate(res):
        gold = GOLD["dptn_%s_losses_%d" % (tag, step)]
        for i, k in enumerate(NAMES):
            scale = max(abs(ref[k]), 1e-3)
            assert abs(got[k] - ref[k]) <= 1e-3 * scale, "step %d %s: %.6f vs oracle %.6f" % (step, k, got[k], ref[k])
            assert abs(got[k] - gold[i]) <= 1e-3 * max(abs(gold[i]), 1e-3), "step %d %s vs reference fixture" % (step, k)
        _check(fake, ofake, 1e-3, "fake_t step %d" % step)
        s, _ = sub(fake)
        gref = GOLD["dptn_%s_fake_t_%d" % (tag, step)]
        assert np.abs(np.asarray(s, dtype=np.float64).reshape(gref.shape) - gref).max() <= 1e-3 * np.abs(gref).max()
    # parameters after two Adam steps: relative L2 (Adam's first steps are sign-like: an element whose gradient is at
    # rounding level can move by 2 lr either way)
    pg, og = dict(m.net_G.module.named_parameters()), dict(om.net_G.named_parameters())
    for k in C.PROBES_G:
        _check_l2(pg[k], og[k], 2e-3, "param " + k, tol_max=5e-2)
    pd, od = dict(m.net_D.module.named_parameters()), dict(om.net_D.named_parameters())
    for k in C.PROBES_D:
        _check_l2(pd[k], od[k], 2e-3, "D param " + k, tol_max=5e-2)


def test_dptn_lsgan_fails_where_the_reference_fails(dev):
    m, _ = _build(dev, "lsgan")
    d = {k: v.to(dev) for k, v in C.inputs().items()}
    m.set_input(d)
    with pytest.raises(RuntimeError, match="scalar outputs"):
        m.optimize_parameters()
    msg = bytes(GOLD["lsgan_error"].astype(np.uint8)).decode()
    assert "grad can be implicitly created only for scalar outputs" in msg


def test_gradient_penalty_against_reference_fixture(dev):
    """cal_gradient_penalty on the HIP discriminator: value, input gradients and the second-order weight gradients."""
    from dual_gan.models import external_function, networks
    _, net_D = C.nets()
    rg = networks.ResDiscriminator(3, 32, 128, 3, 'none', 'LeakyReLU', True)
    rg.load_state_dict(net_D.state_dict())
    rg.to(dev).train()
    d = C.inputs()
    real, fake = d['Xt'], d['Xs']
    alpha = torch.from_numpy(GOLD["gp_alpha"]).float()
    gp_o, go = D.o_cal_gradient_penalty(net_D, real, fake, alpha)
    gp_o.backward()
    gp, g = external_function.cal_gradient_penalty(rg, real.to(dev), fake.to(dev), alpha=alpha)
    gp.backward()
    assert abs(float(gp) - float(GOLD["gp_value"][0])) <= 1e-3 * abs(float(GOLD["gp_value"][0]))
    _check(g, go, 1e-3, "gp input gradients")
    po, pr = dict(net_D.named_parameters()), dict(rg.named_parameters())
    for k in C.PROBES_D:
        _check_l2(pr[k].grad, po[k].grad, 2e-3, "gp grad " + k, tol_max=5e-2)
        gold = GOLD["gp_g_" + k]
        s, _ = sub(pr[k].grad.cpu())
        assert np.abs(np.asarray(s, dtype=np.float64).reshape(gold.shape) - gold).max() <= 5e-3 * np.abs(gold).max(), k


def test_vgg_loss_against_reference_fixture(dev):
    from dual_gan.models import external_function
    vl = external_function.VGGLoss().to(dev)
    vl.vgg.load_torchvision({"features." + k: v for k, v in C.vgg_features().state_dict().items()})
    x, y = C.vgg_pair()
    xd = x.to(dev).requires_grad_(True)
    content, style = vl(xd, y.to(dev))
    (content + 500.0 * style).backward()
    ref = GOLD["vgg_losses"]
    assert abs(float(content) - ref[0]) <= 1e-3 * abs(ref[0]) and abs(float(style) - ref[1]) <= 1e-3 * abs(ref[1])
    s, _ = sub(xd.grad.cpu())
    gold = GOLD["vgg_dx"]
    assert np.abs(np.asarray(s, dtype=np.float64).reshape(gold.shape) - gold).max() <= 2e-3 * np.abs(gold).max()
    feats = vl.vgg(x.to(dev))
    means = np.array([float(feats[k].mean()) for k, _, _ in D._VGG_SLICES])
    assert np.abs(means - GOLD["vgg_relu_means"]).max() <= 1e-3 * np.abs(GOLD["vgg_relu_means"]).max()
    assert list(feats) == [k for k, _, _ in D._VGG_SLICES]


def test_dptn_step_fp8_declared_tolerance(dev):
    """conv_dtype='fp8': all convolutions of net_G / net_D on the fp8 MFMA family.  Declared tolerance against the fp32
    oracle step: losses 5e-2 relative (+1e-3), generated images 6e-2 relative L2."""
    m, om = _build(dev, "hinge", conv_dtype="fp8")
    from rg_hip import lowp
    assert lowp.states_of(m.net_G) is not None and lowp.states_of(m.net_D) is not None
    res = _run_steps(m, om, dev, "hinge")
    for step, (got, ref, fake, ofake) in enumerate(res):
        for k in NAMES:
            assert abs(got[k] - ref[k]) <= 5e-2 * abs(ref[k]) + 1e-3, "step %d %s: fp8 %.5f vs fp32 oracle %.5f" % (step, k, got[k], ref[k])
        l2 = (fake.double() - ofake.double()).norm().item() / ofake.double().norm().item()
        assert l2 <= 6e-2, "fake_t step %d: rel L2 %.3e" % (step, l2)
