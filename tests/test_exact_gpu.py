"""Exactness of the kernel compositions (parity items 6a / 6b).

(a) Kink-free comparisons.  Two exact fp32 implementations of a ReLU network agree only up to "a few elements took the other
branch" (pre-activations differ by ~1e-6, so ~1 element per layer does), which forces the kinked gradient tests onto L2
tolerances of 1e-2.  Here the kinks are taken out of play on BOTH sides, so every gradient must agree with an fp64 run of the
oracle in the MAX norm at rounding level (2e-5 of the tensor's largest entry):
  * ResNet-50 trunk, eval-mode and train-mode BatchNorm: every BatchNorm bias is set to +8 (weights ~1), which puts every
    pre-ReLU value > 0 — the SAME fused programs run (BatchNorm folded into the conv epilogues / train-mode statistics, ReLU
    masks in the dgrad epilogues, residual adds), their masks simply never bite.  Max-pool keeps its argmax kink: the stem's
    conv1 / bn1 gradients get 2e-3 (one window whose two largest values lie within rounding moves one of ~5e5 terms);
  * CustomPoseGenerator (train-mode BatchNorm): the same shift, positive pose maps / noise and |w| in the norm-less first
    layer keep its LeakyReLU(0.2) / ReLU inputs positive;
  * PoseGenerator1 and ResDiscriminator (module-level LeakyReLU): slope 1.
(b) HIP outputs against the committed reference fixtures directly (tests/golden/reference_modules.npz, same cases.py inputs).
"""
import copy
import os

import numpy as np
import pytest
import torch

from oracle import ref_torch as O
from tests.golden import cases as C
from tests.golden.cases import sub
from tests.test_modules_gpu import _check

pytestmark = pytest.mark.gpu

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_modules.npz"))


def _shift_bn(mod, shift=8.0, seed=0):
    g = torch.Generator().manual_seed(seed)
    for m in mod.modules():
        if isinstance(m, (torch.nn.BatchNorm2d, torch.nn.BatchNorm1d)):
            with torch.no_grad():
                m.bias.fill_(shift)
                m.weight.copy_(1.0 + 0.1 * torch.rand(m.weight.shape, generator=g))
                m.running_mean.copy_(0.1 * torch.randn(m.running_mean.shape, generator=g))
                m.running_var.copy_(0.8 + 0.4 * torch.rand(m.running_var.shape, generator=g))


def _round_to_fp32(mod):
    """fp64 oracle holding exactly the fp32 parameter values the HIP model gets"""
    mod.load_state_dict({k: (v.float().double() if v.is_floating_point() else v) for k, v in mod.state_dict().items()})


def _calibrate_bn(mod, forward, margin=1.0):
    """One forward pass of the oracle that lifts every BatchNorm bias, channel by channel, until the smallest value the layer
    emits on this input is `margin` (+ 1 % of the layer's largest): the residual stream of an untrained ResNet grows into the
    thousands, so no fixed shift keeps all 53 pre-ReLU maps positive.  Returns the smallest BatchNorm output of a second,
    unmodified pass (the proof that no ReLU bites)."""
    lows = []

    def lift(m, inp, out):
        red = [d for d in range(out.dim()) if d != 1]
        lo, hi = out.amin(red), out.abs().amax()
        delta = (margin + 0.01 * hi) - lo
        with torch.no_grad():
            m.bias.add_(delta.to(m.bias.dtype))
        return out + delta.view([1, -1] + [1] * (out.dim() - 2))

    def probe(m, inp, out):
        lows.append(float(out.min()))

    bns = [m for m in mod.modules() if isinstance(m, (torch.nn.BatchNorm2d, torch.nn.BatchNorm1d))]
    for hook in (lift, probe):
        hs = [m.register_forward_hook(hook) for m in bns]
        with torch.no_grad():
            forward()
        for h in hs:
            h.remove()
    return min(lows)


def _maxerr(got, ref64, floor=0.0):
    ref64 = ref64.detach().double().cpu()
    return (got.detach().double().cpu() - ref64).abs().max().item() / max(ref64.abs().max().item(), floor, 1e-300)


def _grads_exact(rg, o64, tol=2e-5, loose=(), loose_tol=2e-3, o32=None, factor=4.0):
    """max-norm error of every parameter gradient against the fp64 oracle, relative to the tensor's largest entry.  Bound: `tol`
    (rounding level).  Where the arithmetic itself is ill-conditioned in fp32 (train-mode BatchNorm over a handful of elements
    divides rounding noise by a tiny sigma), the oracle's own fp32 CPU run `o32` gives the scale: the bound becomes
    factor x the worst tensor of that run, never less than `tol`."""
    og = dict(o64.named_parameters())
    gmax = max(p.grad.abs().max().item() for p in og.values() if p.grad is not None)
    cpu_worst = 0.0
    if o32 is not None:
        for n, p in o32.named_parameters():
            if p.grad is not None:
                cpu_worst = max(cpu_worst, _maxerr(p.grad, og[n].grad, 1e-3 * gmax))
    bound = max(tol, factor * cpu_worst)
    worst = 0.0
    for n, p in rg.named_parameters():
        ref = og[n].grad
        if ref is None:
            assert p.grad is None, n
            continue
        assert p.grad is not None, n
        err = _maxerr(p.grad, ref, 1e-3 * gmax)
        is_loose = any(n.startswith(k) or ("." + k) in n for k in loose)
        t = max(loose_tol, bound) if is_loose else bound
        assert err <= t, "%s: max-norm err %.3e > %.1e (fp32 CPU oracle worst %.2e)" % (n, err, t, cpu_worst)
        if not is_loose:
            worst = max(worst, err)
    return worst, cpu_worst


def _fwd_exact(got, ref64, what, tol=2e-5, ref32=None, factor=4.0):
    e = _maxerr(got, ref64)
    e32 = _maxerr(ref32, ref64) if ref32 is not None else 0.0
    assert e <= max(tol, factor * e32), "%s: max-norm err %.3e (fp32 CPU oracle %.2e)" % (what, e, e32)
    return e


@pytest.mark.parametrize("mode", ["eval", "train"])
def test_resnet50_trunk_exact_without_kinks(dev, mode):
    import reid.models as RM
    torch.manual_seed(40)
    o = O.OReidResNet(50, cut_at_pooling=True)
    _shift_bn(o, 8.0, 1)
    x = O.synth_images(4, 64, 32, seed=3)
    o = o.double()
    getattr(o, mode)()
    assert _calibrate_bn(o, lambda: o(x.double())) >= 1.0
    _round_to_fp32(o)
    r = RM.create('resnet50', cut_at_pooling=True, pretrained=False)
    r.load_state_dict({k: v.float() if v.is_floating_point() else v for k, v in o.state_dict().items()})
    r.to(dev)
    getattr(r, mode)()
    # train mode: BatchNorm over 4 x (2 x 1) elements in layer4 divides rounding noise by small sigmas — the oracle's own fp32
    # run measures that conditioning (eval mode needs no anchor: plain rounding level)
    o32 = copy.deepcopy(o).float() if mode == "train" else None
    xo = x.double().clone().requires_grad_(True)
    xr = x.clone().to(dev).requires_grad_(True)
    yo, yr = o(xo), r(xr)
    assert float(yo.min()) > 0
    g = torch.Generator().manual_seed(5)
    cot = torch.randn(yo.shape, generator=g)
    y32 = None
    if o32 is not None:
        x32 = x.clone().requires_grad_(True)
        y32 = o32(x32)
        (y32 * cot).sum().backward()
    _fwd_exact(yr, yo, "features", ref32=y32)
    (yo * cot.double()).sum().backward()
    (yr * cot.to(dev)).sum().backward()
    # the stem sits behind the max-pool (argmax kink): its filter / BN gradients and dx get the looser bound
    worst, cpu = _grads_exact(r, o, loose=("base.conv1", "base.bn1"), o32=o32)
    err_dx = _maxerr(xr.grad, xo.grad)
    assert err_dx <= max(2e-3, 4.0 * (_maxerr(x32.grad, xo.grad) if o32 is not None else 0.0)), err_dx
    print("trunk (%s BN): worst gradient max-norm error %.2e (fp32 CPU oracle %.2e), dx %.2e" % (mode, worst, cpu, err_dx))


def test_pose_generator_exact_without_kinks(dev):
    import fdgan.networks as N
    torch.manual_seed(10)
    o = O.OPoseGenerator(128, 2048, 256, dropout=0.0, norm='batch', connect_layers=0)
    o.apply(O.o_weights_init_normal)
    _shift_bn(o, 8.0, 2)
    with torch.no_grad():
        o.en_conv1.weight.abs_()                       # no norm behind it: positive filters on positive pose maps
    r = N.CustomPoseGenerator(128, 2048, 256, dropout=0.0, norm_layer=N.get_norm_layer('batch'), fuse_mode='cat',
                              connect_layers=0)
    o = o.double().train()
    g = torch.Generator().manual_seed(4)
    pose = O.synth_posemaps(3, seed=3) + 0.5
    feat = torch.randn(3, 2048, 1, 1, generator=g).abs() + 0.1
    z = torch.randn(3, 256, 1, 1, generator=g).abs() + 0.1
    assert _calibrate_bn(o, lambda: o(pose.double(), feat.double(), z.double())) >= 1.0
    _round_to_fp32(o)
    r.load_state_dict({k: v.float() if v.is_floating_point() else v for k, v in o.state_dict().items()})
    r.to(dev).train()
    o32 = copy.deepcopy(o).float()          # BatchNorm over 3 x (1 x 1) elements at the bottleneck: anchored like the trunk
    fo = feat.double().clone().requires_grad_(True)
    fr = feat.clone().to(dev).requires_grad_(True)
    f32 = feat.clone().requires_grad_(True)
    yo = o(pose.double(), fo, z.double())
    yr = r(pose.to(dev), fr, z.to(dev))
    y32 = o32(pose, f32, z)
    _fwd_exact(yr, yo, "fake", ref32=y32)
    cot = torch.randn(yo.shape, generator=g)
    (yo * cot.double()).sum().backward()
    (yr * cot.to(dev)).sum().backward()
    (y32 * cot).sum().backward()
    _fwd_exact(fr.grad, fo.grad, "d reid feature", ref32=f32.grad)
    worst, cpu = _grads_exact(r, o, o32=o32)
    print("CustomPoseGenerator: worst gradient max-norm error %.2e (fp32 CPU oracle %.2e)" % (worst, cpu))


def test_dualgan_nets_exact_without_kinks(dev):
    from dual_gan.models import networks as N
    from tests.golden import cases_dualgan as CD
    from tests.test_dptn_gpu import _remove_kinks
    # PoseGenerator1
    on, (feat, pose) = CD.posegen1_case()
    rg = N.PoseGenerator1(64, 18, 256, 3, 'instance', 'LeakyReLU', False, False, 3, True, 2, 2, 2)
    rg.load_state_dict(on.state_dict())
    rg.to(dev).train()
    _remove_kinks([rg], [on])
    on = on.double()
    fo = feat.double().clone().requires_grad_(True)
    fr = feat.clone().to(dev).requires_grad_(True)
    yo, yr = on(fo, pose.double()), rg(fr, pose.to(dev))
    _check(yr, yo, 2e-5, "posegen1 fwd")
    g = torch.Generator().manual_seed(5)
    cot = torch.randn(yo.shape, generator=g)
    (yo * cot.double()).sum().backward()
    (yr * cot.to(dev)).sum().backward()
    _check(fr.grad, fo.grad, 5e-5, "posegen1 d feature")
    w1, _ = _grads_exact(rg, on, tol=1e-4)
    # ResDiscriminator (spectral norm: both sides run the same power iteration)
    od, x = CD.resdisc_case()
    rd = N.ResDiscriminator(3, 32, 128, 3, 'none', 'LeakyReLU', True)
    rd.load_state_dict(od.state_dict())
    rd.to(dev).train()
    _remove_kinks([rd], [od])
    od = od.double()
    xo = x.double().clone().requires_grad_(True)
    xr = x.clone().to(dev).requires_grad_(True)
    yo, yr = od(xo), rd(xr)
    _check(yr, yo, 2e-5, "resdisc fwd")
    (yo ** 2).mean().backward()
    (yr ** 2).mean().backward()
    _check(xr.grad, xo.grad, 5e-5, "resdisc dx")
    w2, _ = _grads_exact(rd, od, tol=1e-4)
    print("PoseGenerator1 / ResDiscriminator: worst gradient max-norm errors %.2e / %.2e" % (w1, w2))


# ---- (b) the committed reference fixtures, directly ---------------------------------------------------------------------
def _gold(got, key, tol=1e-3):
    ref = GOLD[key]
    got = np.asarray(got, dtype=np.float64).reshape(ref.shape)
    err = np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-12)
    assert err <= tol, "%s: %.3e" % (key, err)


@pytest.mark.parametrize("cl", [0, 2])
def test_generator_vs_reference_fixture(dev, cl):
    import fdgan.networks as N
    og, (pose, feat, z) = C.generator_case(cl)
    r = N.CustomPoseGenerator(128, 2048, 256, dropout=0.0, norm_layer=N.get_norm_layer('batch'), fuse_mode='cat',
                              connect_layers=cl)
    r.load_state_dict(og.state_dict())
    r.to(dev).train()
    s, st = sub(r(pose.to(dev), feat.to(dev), z.to(dev)).cpu())
    _gold(s, "g_fwd_cl%d" % cl)
    ref = GOLD["g_fwd_cl%d_stats" % cl]
    assert abs(st[1] - ref[1]) <= 1e-3 * abs(ref[1]) and st[2] == ref[2]


@pytest.mark.parametrize("norm,key", [("batch", "dp_fwd"), ("instance", "dp_in_fwd")])
def test_discriminator_vs_reference_fixture(dev, norm, key):
    import fdgan.networks as N
    od, x = C.discriminator_case(norm)
    r = N.NLayerDiscriminator(21, norm_layer=N.get_norm_layer(norm))
    r.load_state_dict(od.state_dict())
    r.to(dev)
    r.train(od.training)
    s, _ = sub(r(x.to(dev)).cpu())
    _gold(s, key)


def test_embed_ganloss_gem_vs_reference_fixture(dev):
    from fdgan.losses import GANLoss
    from reid.models.embedding import EltwiseSubEmbed
    from clustercontrast.models.pooling import GeneralizedMeanPoolingP
    oe, (f1, f2) = C.embed_case()
    e = EltwiseSubEmbed(use_batch_norm=True, use_classifier=True, num_features=2048, num_classes=2)
    e.load_state_dict(oe.state_dict())
    e.to(dev)
    for mode in ("train", "eval"):
        getattr(e, mode)()
        _gold(e(f1.to(dev), f2.to(dev)).detach().cpu().numpy(), "embed_" + mode)
    pred = C.ganloss_case().to(dev)
    _gold([float(GANLoss()(pred, True)), float(GANLoss()(pred, False))], "ganloss")
    _, x, dy = C.gem_case()
    gem = GeneralizedMeanPoolingP().to(dev)
    xd = x.to(dev).requires_grad_(True)
    y = gem(xd)
    y.backward(dy.to(dev).view_as(y))
    _gold(y.detach().cpu().numpy(), "gem_fwd")
    _gold(sub(xd.grad.cpu())[0], "gem_dx")
    _gold(gem.p.grad.cpu().numpy(), "gem_dp")


def test_trunk_and_cluster_memory_vs_reference_fixture(dev):
    import reid.models as RM
    from clustercontrast.models.cm import cm, cm_hard
    oreid, trunk_sd, imgs = C.trunk_case()
    r = RM.create('resnet50', cut_at_pooling=True, pretrained=False)
    r.load_state_dict(oreid.state_dict())
    r.to(dev)
    for mode in ("eval", "train"):
        getattr(r, mode)()
        s, _ = sub(r(imgs.to(dev)).detach().cpu())
        _gold(s, "resnet50_trunk_%s" % mode)
        r.base.load_state_dict(trunk_sd)
    bank, feats, labels, gout = C.cm_case()
    for fn, tag in ((cm, "cm"), (cm_hard, "cm_hard")):
        b = bank.clone().to(dev)
        x = feats.clone().to(dev).requires_grad_(True)
        y = fn(x, labels.to(dev), b, 0.2)
        y.backward(gout.to(dev))
        _gold(sub(y.detach().cpu())[0], tag + "_logits")
        _gold(sub(x.grad.cpu())[0], tag + "_grad")
        _gold(sub(b.cpu(), 2048)[0], tag + "_bank")
