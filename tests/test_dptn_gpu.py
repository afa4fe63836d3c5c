"""BASELINE config 5 on the MI355X: the DPTNModel step (dual_gan/models/DPTN_model.py on the HIP tape programs) against
the oracle restatement (oracle/ref_dualgan.ODPTNModel, which make_golden_dptn.py found bit-identical to the reference's own
DPTNModel on the CPU) and directly against the committed reference fixtures (tests/golden/reference_dptn.npz).

fp32 path: losses, generated images and post-step parameters within 1e-3 (max norm) over two optimizer steps for the hinge,
vanilla and wgangp (gradient penalty: second-order weight gradients) objectives and with the VGG perceptual / style terms.
fp8 path: DECLARED tolerance (test_dptn_step_fp8_declared_tolerance) — forward images within 1.2e-1 relative L2 (about
twenty fp8 layers deep at <= 6e-2 each, adding in quadrature, partly damped by the normalisation layers), losses within 5e-2,
gradient cosine >= 0.95; the layer-level tolerances are in tests/test_f8_gpu.py.
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


def _remove_kinks(hip_nets, oracle_nets):
    """LeakyReLU(0.1) -> slope 1 on both sides: the networks become smooth (norms, tanh, softmax, attention remain), so two
    exact implementations must agree to rounding instead of to 'a few elements took the other branch'."""
    from rg_hip import nn as rnn
    from rg_hip.ops import ACT_LEAKY
    for net in hip_nets:
        for mod in net.modules():
            if isinstance(mod, rnn.LeakyReLU):
                mod.negative_slope = mod.SLOPE = 1.0
            if getattr(mod, "_act", None) is not None and mod._act[0] == ACT_LEAKY:
                mod._act = (ACT_LEAKY, 1.0)
    for net in oracle_nets:
        for mod in net.modules():
            if isinstance(mod, torch.nn.LeakyReLU):
                mod.negative_slope = 1.0


@pytest.mark.parametrize("gan_mode", ["hinge", pytest.param("vanilla", marks=pytest.mark.slow)])
def test_dptn_gradients_exact_without_kinks(dev, gan_mode):
    """Every gradient of the DPTN step (D update, then G update through the updated-forward D) against an fp64 run of the
    oracle with the LeakyReLU kinks removed: max-norm per tensor <= 1e-4 of the tensor's largest entry (fp32 InstanceNorm over
    32-element maps is the least well conditioned piece; measured ~1e-5).  This is what makes the 1e-2-level L2 tolerances of
    the kinked comparisons below a statement about branch flips, not about the kernels."""
    m, om = _build(dev, gan_mode)
    _remove_kinks([m.net_G, m.net_D], [om.net_G, om.net_D])
    om.net_G.double()
    om.net_D.double()
    d = C.inputs()
    om.set_input({k: v.double() for k, v in d.items()})
    m.set_input({k: v.to(dev) for k, v in d.items()})
    om.forward()
    m.forward()
    _check(m.fake_image_t, om.fake_image_t, 2e-5, "fake_t")
    _check(m.fake_image_s, om.fake_image_s, 2e-5, "fake_s")
    om.optimizer_D.zero_grad()
    om.backward_D()
    m.optimizer_D.zero_grad()
    m.backward_D()
    worst = 0.0
    for k, p in m.net_D.module.named_parameters():
        worst = max(worst, _check(p.grad, dict(om.net_D.named_parameters())[k].grad, 1e-4, "D grad " + k))
    om.optimizer_G.zero_grad()
    om.backward_G()
    m.optimizer_G.zero_grad()
    m.backward_G()
    og = dict(om.net_G.named_parameters())
    gmax = max(float(p.grad.abs().max()) for p in og.values())
    for k, p in m.net_G.module.named_parameters():
        ref = og[k].grad
        if float(ref.abs().max()) < 1e-6 * gmax:          # biases in front of a norm: zero in exact arithmetic
            assert float(p.grad.abs().max()) <= 1e-4 * gmax, k
            continue
        worst = max(worst, _check(p.grad, ref, 1e-4, "G grad " + k))
    print("worst gradient max-norm error without kinks: %.2e" % worst)


# default run: the hinge step (BASELINE config 5's mode); the vanilla / wgangp modes and the VGG variant repeat the same programs with
# another loss head (-m "gpu and slow")
@pytest.mark.parametrize("gan_mode,with_vgg", [("hinge", False), pytest.param("vanilla", False, marks=pytest.mark.slow),
                                               pytest.param("wgangp", False, marks=pytest.mark.slow),      # (the penalty itself: test_gradient_penalty_against_reference_fixture)
                                               pytest.param("hinge", True, marks=pytest.mark.slow)])
def test_dptn_step_matches_oracle_and_reference_fixture(dev, gan_mode, with_vgg):
    """Step 0 (identical weights on both sides): every loss and both generated images at 1e-3, against the oracle AND against
    the values recorded from the reference's own DPTNModel.  Step 1 follows one Adam update of G and D: Adam's first step is
    lr * g / (|g| + 1e-8), i.e. lr * sign(g) — a LeakyReLU element that takes the other branch perturbs every gradient below it
    by ~1 / sqrt(#elements) (0.2-0.4 %, measured), which flips the sign of the smallest ~0.3 % of the gradient entries and
    moves those parameters by 2 lr (tools/debug/dptn_step_diff.py); the exactness of the kernels is
    test_dptn_gradients_exact_without_kinks.  Step-1 quantities are therefore held to 2e-2."""
    m, om = _build(dev, gan_mode, with_vgg)
    tag = gan_mode + ("_vgg" if with_vgg else "")
    res = _run_steps(m, om, dev, gan_mode)
    for step, (got, ref, fake, ofake) in enumerate(res):
        rtol, atol = (1e-3, 1e-6) if step == 0 else (2e-2, 2e-5)
        gold = GOLD["dptn_%s_losses_%d" % (tag, step)]
        for i, k in enumerate(NAMES):
            assert abs(got[k] - ref[k]) <= rtol * abs(ref[k]) + atol, "step %d %s: %.6f vs oracle %.6f" % (step, k, got[k], ref[k])
            assert abs(got[k] - gold[i]) <= rtol * abs(gold[i]) + atol, "step %d %s vs reference fixture" % (step, k)
        s, _ = sub(fake)
        gref = GOLD["dptn_%s_fake_t_%d" % (tag, step)]
        gerr = np.abs(np.asarray(s, dtype=np.float64).reshape(gref.shape) - gref).max() / np.abs(gref).max()
        if step == 0:
            _check(fake, ofake, 1e-3, "fake_t step 0")
            assert gerr <= 1e-3, gerr
        else:
            _check_l2(fake, ofake, 2e-2, "fake_t step 1", tol_max=1e-1)
            assert gerr <= 1e-1, gerr
    # parameters after two Adam steps: relative L2 (sign-like updates, see above)
    pg, og = dict(m.net_G.module.named_parameters()), dict(om.net_G.named_parameters())
    # (tests/test_oracle_golden_dptn.py::test_reference_step_is_sensitive_to_rounding measures the reference arithmetic's own
    # spread under a 3e-7 weight perturbation: up to 5e-3 relative L2 / 4e-2 max-norm on these tensors after two steps)
    for k in C.PROBES_G:
        _check_l2(pg[k], og[k], 2e-2, "param " + k, tol_max=1.5e-1)
    pd, od = dict(m.net_D.module.named_parameters()), dict(om.net_D.named_parameters())
    for k in C.PROBES_D:
        _check_l2(pd[k], od[k], 2e-2, "D param " + k, tol_max=1.5e-1)


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
    """conv_dtype='fp8': the convolutions of net_G / net_D on the fp8 MFMA family (first / last layers fp32).  Declared
    tolerance against the fp32 oracle:
      * forward with identical weights: generated images <= 1.2e-1 relative L2, losses <= 5e-2 relative (+1e-3; +2e-2 for
        ad_gen_t, a mean of signed discriminator outputs near zero);
      * gradients with identical weights, identical cotangents at the network outputs and the LeakyReLU kinks removed: cosine
        similarity with the fp32 oracle's gradient >= 0.92 for every filter tensor of G (all ~50 layers), the probed tensors
        of D and D's input gradient (e5m2 gradients carry 2 mantissa bits: ~7 % noise per layer, uncorrelated between
        elements; measured 0.94 .. 0.99);
      * after two optimizer steps: losses still within 5e-2.  Images / parameters are NOT compared after an update: Adam's
        first steps are lr * sign(g), so gradient noise of a few % flips the direction of the smallest entries on either side
        (the fp32 reference does the same under rounding-level noise, test_reference_step_is_sensitive_to_rounding)."""
    m, om = _build(dev, "hinge", conv_dtype="fp8")
    from rg_hip import lowp
    assert lowp.states_of(m.net_G) is not None and lowp.states_of(m.net_D) is not None
    d = C.inputs()
    dd = {k: v.to(dev) for k, v in d.items()}
    # ---- forward at identical weights ---------------------------------------------------------------------------
    om.set_input(d)
    om.forward()
    m.set_input(dd)
    m.forward()
    l2 = (m.fake_image_t.cpu().double() - om.fake_image_t.double()).norm().item() / om.fake_image_t.double().norm().item()
    print("fp8 forward: fake_t rel L2 vs fp32 oracle %.3e" % l2)
    assert l2 <= 1.2e-1

    # ---- gradient quality: identical weights, identical cotangents at the outputs, LeakyReLU kinks removed on both sides.
    # (With the kinks in, the fp8 forward's 9 % activation noise puts ~4 % of the LeakyReLU(0.1) elements on the other branch,
    # which changes the local gradient by 90 % there: the cosine then falls by ~0.95 per block — 0.36 at the first layer,
    # tools/debug/f8_grad_depth.py — a property of comparing gradients along two different forward trajectories, not of the
    # backward kernels; without kinks it stays >= 0.94 through all ~50 layers.)
    def cos(a, b):
        a, b = a.detach().double().cpu().flatten(), b.detach().double().flatten()
        return float((a * b).sum() / (a.norm() * b.norm()))
    mk, omk = _build(dev, "hinge", conv_dtype="fp8")
    _remove_kinks([mk.net_G, mk.net_D], [omk.net_G, omk.net_D])
    omk.set_input(d)
    omk.forward()
    mk.set_input(dd)
    mk.forward()
    g = torch.Generator().manual_seed(11)
    ct, cs_ = torch.randn(omk.fake_image_t.shape, generator=g), torch.randn(omk.fake_image_s.shape, generator=g)
    ((omk.fake_image_t * ct).sum() + (omk.fake_image_s * cs_).sum()).backward()
    torch.autograd.backward([mk.fake_image_t, mk.fake_image_s], [ct.to(dev), cs_.to(dev)])
    pg, og = dict(mk.net_G.module.named_parameters()), dict(omk.net_G.named_parameters())
    cs = {k: cos(pg[k].grad, og[k].grad) for k in og if og[k].dim() > 1 and og[k].grad is not None}
    xr = d['Xt'].clone().requires_grad_(True)
    xd = dd['Xt'].clone().requires_grad_(True)
    yo = omk.net_D(xr)
    cd = torch.randn(yo.shape, generator=g)
    yo.backward(cd)
    mk.net_D(xd).backward(cd.to(dev))
    pd, od = dict(mk.net_D.module.named_parameters()), dict(omk.net_D.named_parameters())
    cs.update({"D." + k: cos(pd[k].grad, od[k].grad) for k in C.PROBES_D})
    cs["D.input"] = cos(xd.grad, xr.grad)
    worst = min(cs, key=cs.get)
    print("fp8 gradient cosines vs fp32 oracle (no kinks): min %.4f at %s, median %.4f over %d tensors"
          % (cs[worst], worst, sorted(cs.values())[len(cs) // 2], len(cs)))
    assert cs[worst] >= 0.92, (worst, cs[worst])


# default run: the forward / gradient bounds above, graphed == eager bit for bit in fp8 (tests/test_netgraph_gpu.py) and the config-5
# bench line; the two CPU-oracle optimizer steps of this one take 15-30 s of host time
@pytest.mark.slow
def test_dptn_fp8_losses_track_the_fp32_oracle_over_two_steps(dev):
    """third clause of the declared fp8 tolerance (test_dptn_step_fp8_declared_tolerance): after two optimizer steps the losses
    are still within 5e-2 of the fp32 oracle's"""
    m2, om2 = _build(dev, "hinge", conv_dtype="fp8")
    for step, (got, ref, fake, ofake) in enumerate(_run_steps(m2, om2, dev, "hinge")):
        print("fp8 step %d losses %s" % (step, {k: round(got[k], 5) for k in NAMES}))
        for k in NAMES:
            # ad_gen_t = -lambda_g * mean(D(fake)) is a mean of signed values near zero: absolute floor 2e-2 (lambda_g = 5)
            atol = 2e-2 if k == "ad_gen_t" else 1e-3
            assert abs(got[k] - ref[k]) <= 5e-2 * abs(ref[k]) + atol, "step %d %s: fp8 %.5f vs fp32 oracle %.5f" % (step, k, got[k], ref[k])


# default run: tests/test_netgraph_gpu.py compares graphed with eager steps bit for bit in both arithmetics, tests/test_fullsize_gpu.py
# two config-2 steps between runs; these two repeat that for the eager DPTN step alone
@pytest.mark.slow
@pytest.mark.parametrize("conv_dtype", ["fp32", "fp8"])
def test_dptn_steps_are_bit_identical_between_runs(dev, conv_dtype):
    """three optimizer steps from the same state twice: identical losses, generated images and parameters, bit for bit — the
    split-K reductions, the integer-atomic amax collection of the fp8 scaling states, the side-stream weight gradients and the
    batched spectral norm are all order-independent"""
    def run():
        torch.manual_seed(5)
        m, _ = _build(dev, "hinge", conv_dtype=conv_dtype)
        dd = {k: v.to(dev) for k, v in C.inputs().items()}
        losses = []
        for _ in range(3):
            m.set_input(dd)
            m.optimize_parameters()
            losses.append([float(v) for v in m.get_current_errors().values()])
        torch.cuda.synchronize()
        params = torch.cat([p.detach().flatten() for p in list(m.net_G.parameters()) + list(m.net_D.parameters())])
        return losses, m.fake_image_t.detach().clone(), params.clone()

    l1, f1, p1 = run()
    l2, f2, p2 = run()
    assert l1 == l2
    assert torch.equal(f1, f2)
    assert torch.equal(p1, p2)


def test_use_adp_adaptor_and_adaptor_only_training(dev):
    """--use_adp (DPTN_model.py:56-59, 91-105, 142-154): Resize_ReID against the oracle (== reference, golden `resize_reid_*`), then a
    DPTNModel without --gan_train whose synthesize() / synthesize_pair() pass through the adaptor and whose only optimizer is the
    adaptor's."""
    import copy
    from dual_gan.models import networks as N
    from dual_gan.models.models import create_model
    from oracle import ref_dualgan as D
    from tests.golden import cases_dualgan as CD
    from tests.test_modules_gpu import _check_grads
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_dualgan.npz"))
    on, x = CD.resize_reid_case()
    rg = N.Resize_ReID(image_nc=3)
    rg.load_state_dict(on.state_dict())
    rg = rg.to(dev).train()
    o64 = copy.deepcopy(on).double()            # gradients judged against fp64 (train-mode BatchNorm over two samples, ReLU kinks)
    xo, xd = x.double().requires_grad_(True), x.to(dev).requires_grad_(True)
    yo, y = o64(xo), rg(xd)
    _check(y, yo, 1e-3, "Resize_ReID fwd")
    ref = gold["resize_reid_fwd"]
    got = np.asarray(sub(y.detach().cpu())[0], dtype=np.float64).reshape(ref.shape)
    assert np.abs(got - ref).max() <= 1e-3 * np.abs(ref).max()
    g = torch.Generator().manual_seed(8)
    dy = torch.randn(yo.shape, generator=g)
    yo.backward(dy.double())
    y.backward(dy.to(dev))
    _check_l2(xd.grad, xo.grad, 5e-3, "Resize_ReID dx")
    _check_grads(rg, o64, 5e-3, "Resize_ReID grads", tol_tensor=5e-2)
    # the model: no --gan_train, --use_adp
    opt = _opt("hinge")
    opt.gan_train, opt.use_adp = False, True
    m = create_model(opt)
    assert m.model_names == ['G', 'A'] and not hasattr(m, "optimizer_G") and hasattr(m, "optimizer_A")
    om = C.model("hinge", False)
    m.net_G.module.load_state_dict(om.net_G.state_dict())
    oa, _ = CD.resize_reid_case()
    m.net_A.load_state_dict(oa.state_dict())
    assert not m.net_G.training                 # only the adaptor trains
    og = om.net_G.eval()
    oa.train()
    d = C.inputs()
    m.set_input({k: v.to(dev) for k, v in d.items()})
    ft, fs = m.synthesize(True)
    with torch.no_grad():
        rt, rs = og(d['Xs'], d['Ps'], d['Pt'], True)
    ot, os_ = oa(rt), oa(rs)
    assert tuple(ft.shape) == (rt.shape[0], 3, 256, 128)
    _check(ft, ot, 1e-3, "synthesize(use_adp) target branch")
    _check(fs, os_, 1e-3, "synthesize(use_adp) source branch")
    fn = m.synthesize_pair()
    with torch.no_grad():
        rn, _ = og(torch.flip(d['Xs'], dims=[0]), torch.flip(d['Ps'], dims=[0]), d['Pt'], False)
    _check(fn, oa(rn), 1e-3, "synthesize_pair(use_adp)")          # third training forward of the adaptor on both sides
    # one adaptor-only update on a surrogate loss
    w = torch.randn(ot.shape, generator=g)
    opt_o = torch.optim.Adam(oa.parameters(), lr=2e-4, betas=(0.5, 0.999))
    opt_o.zero_grad()
    oa2_t = oa(rt)                              # forwards four and five, in the model's order (target branch, then source branch)
    oa(rs)
    (oa2_t * w).mean().backward()
    opt_o.step()
    m.optimizer_A.zero_grad()
    ft2, _ = m.synthesize(True)
    from rg_hip.tape import backward as rg_backward
    rg_backward(RF_mean(ft2, w.to(dev)))
    m.optimizer_A.step()
    po = dict(oa.named_parameters())
    moved = 0.0
    for n, p in m.net_A.named_parameters():
        assert (p.detach().cpu() - po[n].detach()).abs().max().item() <= 2.5 * 2e-4 + 1e-6, n
        moved = max(moved, (p.detach().cpu() - dict(CD.resize_reid_case()[0].named_parameters())[n]).abs().max().item())
    assert moved >= 1e-4                        # the adaptor really stepped


def RF_mean(x, w):
    """(x * w).mean() through torch ops (test-side loss)"""
    return (x * w).mean()
