"""Module- and step-level parity on the MI355X: the drop-in modules (reid / fdgan packages on the HIP tape
runtime) against the oracle restatement on identical seeded weights and inputs.

Tolerance: the north star asks for 1e-3 relative fp32.  Forward outputs and losses are checked in the max norm
at 1e-3 (measured: ~1e-5).  Gradients need a flip-robust metric: two fp32 implementations disagree by a few
1e-6 on pre-activations, so about one element per layer lands on the other side of a (Leaky)ReLU kink; a single
such flip changes one row of a weight gradient by ~1/sqrt(#terms) (2e-2 in the max norm for the 256->512 PatchGAN
layer, measured) although every kernel is exact — test_patch_discriminator_without_kinks shows all gradients agree
to 1e-5 once the kink is removed (slope 1).  Gradients are therefore compared by relative L2 error per tensor
(1e-2) and over all parameters (2e-3), with a loose max-norm guard against gross errors; quantities that are
noise-dominated even inside the reference (train-mode BatchNorm backward through 50 layers at tiny batch, where
torch's fp32 CPU result is several % from its own fp64 result) are judged against an fp64 run of the oracle.
"""
import argparse

import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(got, ref):
    got, ref = got.detach().double().cpu(), ref.detach().double().cpu()
    assert got.shape == ref.shape, (got.shape, ref.shape)
    return (got - ref).abs().max().item() / max(ref.abs().max().item(), 1e-12)


def _check(got, ref, tol, what):
    r = _rel(got, ref)
    assert r <= tol, "%s: rel err %.3e > %.1e" % (what, r, tol)
    return r


def _check_anchored(got, ref32, ref64, what, floor=1e-3, factor=4.0):
    """Ill-conditioned quantities (train-mode BatchNorm backward through 50 layers at tiny batch: torch's own
    fp32 CPU result is several % away from its fp64 result) are judged against an fp64 run of the oracle:
    the HIP result may be at most `factor` x as far from fp64 as the reference's fp32 CPU path is."""
    r64 = ref64.detach().double().cpu()
    e_hip = (got.detach().double().cpu() - r64).norm().item() / r64.norm().item()
    e_cpu = (ref32.detach().double().cpu() - r64).norm().item() / r64.norm().item()
    assert e_hip <= max(floor, factor * e_cpu), "%s: L2 hip-vs-fp64 %.3e, cpu32-vs-fp64 %.3e" % (what, e_hip, e_cpu)
    return e_hip, e_cpu


def _check_grads_anchored(rg_mod, o32, o64, what, floor=2e-3, factor=3.0, tensor_floor=5e-2, tight=5e-3, slack=4):
    """Noise-dominated gradients (see _check_anchored): per tensor the errors of two fp32 implementations are independent draws,
    so compare (a) the global relative L2 error over ALL parameters, (b) every tensor against the WORST tensor of the reference's
    own fp32 path, and (c) the NUMBER of output channels (first-dimension slices: filter rows, affine / bias entries) that are off.
    (`tensor_floor`: one activation that takes the other LeakyReLU / ReLU branch — the two implementations differ by a few 1e-7 in
    front of the kink — moves entries of the gradients of the channel it belongs to by up to a few per cent of that tensor's largest
    entry; measured 0.7-2.4e-2 across the rounding variants of the conv kernels, while the kink-free twins of these tests hold 2e-5.)
    (c) ties that allowance to a COUNT: a branch flip perturbs the channel(s) downstream of it — behind an
    InstanceNorm on a small map it shifts that instance's statistics, i.e. one whole filter row (measured: 393 entries of one
    4x4 filter tensor, all in two rows) — while an indexing bug moves most rows of a tensor.  Rows holding an entry further than
    `tight` (5e-3 of the tensor's largest entry) from the fp64 gradient are counted on both sides: the HIP path may have at most
    `factor` x as many as the reference's fp32 CPU path shows against the same fp64 run (its own flips and noise, counted) plus
    `slack`, over the whole module; per tensor the count is bounded by `slack`, `factor` x the reference's count for that tensor and
    `factor` x the fraction of rows its worst tensor has off (the noise-dominated train-mode BatchNorm trunks have a third of their
    rows off on BOTH sides: there the bound is the reference's own level, elsewhere it is `slack` rows)."""
    g32, g64 = dict(o32.named_parameters()), dict(o64.named_parameters())
    num_h = num_c = den = 0.0
    per_h, per_c, scale = {}, {}, {}
    err_h, err_c = {}, {}
    for n, p in rg_mod.named_parameters():
        if g64[n].grad is None:
            assert p.grad is None, n
            continue
        r64 = g64[n].grad
        h, c = p.grad.detach().double().cpu(), g32[n].grad.double()
        num_h += (h - r64).pow(2).sum().item()
        num_c += (c - r64).pow(2).sum().item()
        den += r64.pow(2).sum().item()
        err_h[n], err_c[n] = (h - r64).abs(), (c - r64).abs()
        per_h[n], per_c[n] = err_h[n].max().item(), err_c[n].max().item()
        scale[n] = r64.abs().max().item()
    gmax = max(scale.values())
    rows_h, rows_c, nrows = {}, {}, {}
    for n in per_h:          # structurally-zero gradients: measure against >= 1e-4 of the module's largest
        sc = max(scale[n], 1e-4 * gmax)
        per_h[n], per_c[n] = per_h[n] / sc, per_c[n] / sc
        eh, ec = err_h[n].reshape(err_h[n].shape[0], -1) if err_h[n].dim() > 0 else err_h[n].reshape(1, 1), None
        ec = err_c[n].reshape(eh.shape)
        rows_h[n] = int((eh.max(dim=1).values > tight * sc).sum().item())
        rows_c[n] = int((ec.max(dim=1).values > tight * sc).sum().item())
        nrows[n] = eh.shape[0]
    l2_h, l2_c = (num_h / den) ** 0.5, (num_c / den) ** 0.5
    assert l2_h <= max(floor, factor * l2_c), "%s: global L2 hip %.3e vs cpu32 %.3e" % (what, l2_h, l2_c)
    worst_c = max(per_c.values())
    bad = {n: e for n, e in per_h.items() if e > max(tensor_floor, factor * worst_c)}
    assert not bad, "%s: tensors beyond %.1fx the reference's worst (%.3e): %s" % (what, factor, worst_c, bad)
    tot_h, tot_c = sum(rows_h.values()), sum(rows_c.values())
    assert tot_h <= factor * tot_c + slack, "%s: %d rows hold entries beyond %.0e of their tensor's scale (reference fp32 path: %d)" % (
        what, tot_h, tight, tot_c)
    worst_frac_c = max(rows_c[n] / float(nrows[n]) for n in rows_c)
    many = {n: (k, nrows[n]) for n, k in rows_h.items() if k > max(slack, factor * rows_c[n], factor * worst_frac_c * nrows[n])}
    assert not many, ("%s: tensors with more off rows than %d, %.0fx the reference's for that tensor and %.0fx its worst tensor's fraction "
                      "(%.3f): %s" % (what, slack, factor, factor, worst_frac_c, many))
    print("%s: off rows (entry > %.0e) hip %d / cpu32 %d" % (what, tight, tot_h, tot_c))
    return l2_h, l2_c


def _check_grads(rg_mod, o_mod, tol, what, skip=(), tol_tensor=1e-2, tol_max=1e-1):
    """relative L2 error over all parameters <= tol, per tensor <= tol_tensor, max norm <= tol_max.  The max-norm bound is a
    statement about ReLU / LeakyReLU branch flips, not about kernels (measured <= 5e-2; the kink-free tests in test_exact_gpu.py
    and *_without_kinks hold 2e-5 / 1e-4 in the max norm)."""
    og = dict(o_mod.named_parameters())
    gmax = max(p.grad.abs().max().item() for p in og.values() if p.grad is not None)
    num = den = 0.0
    for n, p in rg_mod.named_parameters():
        if any(s_ in n for s_ in skip):
            continue
        ref = og[n].grad
        if ref is None:
            assert p.grad is None, "%s: %s has a gradient, the oracle has none" % (what, n)
            continue
        assert p.grad is not None, "%s: %s has no gradient" % (what, n)
        d = p.grad.detach().double().cpu() - ref.double()
        num += d.pow(2).sum().item()
        den += ref.double().pow(2).sum().item()
        # gradients that are zero in exact arithmetic (a bias in front of a train-mode norm) hold only rounding
        # noise on both sides: measure against at least 1e-4 of the module's largest gradient
        floor = 1e-4 * gmax
        l2 = d.norm().item() / max(ref.double().norm().item(), floor * ref.numel() ** 0.5)
        mx = d.abs().max().item() / max(ref.abs().max().item(), floor)
        assert l2 <= tol_tensor, "%s: %s rel L2 err %.3e" % (what, n, l2)
        assert mx <= tol_max, "%s: %s max-norm err %.3e" % (what, n, mx)
    g = (num / max(den, 1e-300)) ** 0.5
    assert g <= tol, "%s: global rel L2 grad err %.3e" % (what, g)
    return g


def _check_l2(got, ref, tol, what, tol_max=1e-1):
    got, ref = got.detach().double().cpu(), ref.detach().double().cpu()
    l2 = (got - ref).norm().item() / max(ref.norm().item(), 1e-300)
    assert l2 <= tol, "%s: rel L2 err %.3e" % (what, l2)
    assert _rel(got, ref) <= tol_max, "%s: max-norm err %.3e" % (what, _rel(got, ref))


@pytest.mark.parametrize("mode", ["eval", "train"])
def test_reid_resnet50(dev, mode):
    from oracle import ref_torch as O
    import reid.models as RM
    torch.manual_seed(1)
    o = O.OReidResNet(50, cut_at_pooling=True)
    for m in o.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.1)
            m.running_var.uniform_(0.8, 1.2)
            m.weight.data.uniform_(0.4, 0.6)
            m.bias.data.normal_(0, 0.1)
    r = RM.create('resnet50', pretrained=False, cut_at_pooling=True)
    r.load_state_dict(o.state_dict())
    r.to(dev)
    getattr(o, mode)()
    getattr(r, mode)()
    x = O.synth_images(4, 128, 64, seed=2)
    xo = x.clone().requires_grad_(True)
    xr = x.clone().to(dev).requires_grad_(True)
    if mode == "train":
        o64 = O.OReidResNet(50, cut_at_pooling=True)
        o64.load_state_dict(o.state_dict())
        o64 = o64.double().train()
        x64 = x.double().clone().requires_grad_(True)
        f64 = o64(x64)
    fo, fr = o(xo), r(xr)
    _check(fr, fo, 1e-3, "features")
    g = torch.randn(fo.shape, generator=torch.Generator().manual_seed(3))
    fo.backward(g)
    fr.backward(g.to(dev))
    if mode == "eval":
        _check_l2(xr.grad, xo.grad, 2e-3, "input grad")
        _check_grads(r, o, 2e-3, "resnet50 " + mode)
    else:
        f64.backward(g.double())
        _check_anchored(xr.grad, xo.grad, x64.grad, "input grad")
        _check_grads_anchored(r, o, o64, "resnet50 train")
    if mode == "train":
        sr, so = r.state_dict(), o.state_dict()
        for k in so:
            if "running" in k:
                _check(sr[k], so[k], 1e-3, k)
            if "num_batches_tracked" in k:
                assert int(sr[k]) == int(so[k])


def test_reid_resnet_head_variants(dev):
    from oracle import ref_torch as O
    import reid.models as RM
    torch.manual_seed(4)
    for kw in (dict(num_features=128, norm=True), dict(num_features=64, num_classes=10), dict(num_classes=7)):
        o = O.OReidResNet(18, **kw)
        r = RM.create('resnet18', pretrained=False, **kw)
        r.load_state_dict(o.state_dict())
        r.to(dev).train()
        o.train()
        o64 = O.OReidResNet(18, **kw)
        o64.load_state_dict(o.state_dict())
        o64 = o64.double().train()
        x = O.synth_images(6, 64, 32, seed=5)
        yo, yr, y64 = o(x), r(x.to(dev)), o64(x.double())
        _check(yr, yo, 1e-3, "head %s" % kw)
        g = torch.randn(yo.shape, generator=torch.Generator().manual_seed(6))
        yo.backward(g)
        yr.backward(g.to(dev))
        y64.backward(g.double())
        _check_grads_anchored(r, o, o64, "resnet18 head %s" % kw)


# (norm='instance' cannot run in the reference either: en_avg ends in a 1x1 map and torch's instance_norm refuses
# a single spatial element in training mode)
@pytest.mark.parametrize("norm,cl", [pytest.param("batch", 0, marks=pytest.mark.slow), ("batch", 3)])      # cl = 3 runs every layer of cl = 0 plus the skips
def test_generator(dev, norm, cl):
    from oracle import ref_torch as O
    import fdgan.networks as N
    torch.manual_seed(7)
    o = O.OPoseGenerator(128, 2048, 256, dropout=0.0, norm=norm, connect_layers=cl)
    o.apply(O.o_weights_init_normal)
    r = N.CustomPoseGenerator(128, 2048, 256, dropout=0.0, norm_layer=N.get_norm_layer(norm), connect_layers=cl)
    r.load_state_dict(o.state_dict())
    r.to(dev).train()
    o.train()
    pose = O.synth_posemaps(3, seed=8)
    gg = torch.Generator().manual_seed(9)
    feat = torch.randn(3, 2048, 1, 1, generator=gg).abs()
    z = torch.randn(3, 256, 1, 1, generator=gg)
    fo = feat.clone().requires_grad_(True)
    fr = feat.clone().to(dev).requires_grad_(True)
    o64 = O.OPoseGenerator(128, 2048, 256, dropout=0.0, norm=norm, connect_layers=cl)
    o64.load_state_dict(o.state_dict())
    o64 = o64.double().train()
    f64 = feat.double().clone().requires_grad_(True)
    y64 = o64(pose.double(), f64, z.double())
    yo = o(pose.clone(), fo, z.clone())
    yr = r(pose.to(dev), fr, z.to(dev))
    _check(yr, yo, 1e-3, "fake images")
    g = torch.randn(yo.shape, generator=gg)
    yo.backward(g)
    yr.backward(g.to(dev))
    y64.backward(g.double())
    _check_anchored(fr.grad, fo.grad, f64.grad, "d reid feature")
    _check_grads_anchored(r, o, o64, "generator %s cl=%d" % (norm, cl))


@pytest.mark.parametrize("norm", ["batch", "instance"])
def test_patch_discriminator(dev, norm):
    from oracle import ref_torch as O
    import fdgan.networks as N
    torch.manual_seed(10)
    o = O.OPatchDiscriminator(21, norm)
    o.apply(O.o_weights_init_normal)
    r = N.NLayerDiscriminator(21, norm_layer=N.get_norm_layer(norm))
    r.load_state_dict(o.state_dict())
    r.to(dev).train()
    o.train()
    x = torch.cat((O.synth_posemaps(3, seed=11), O.synth_images(3, seed=12)), 1)
    xo = x.clone().requires_grad_(True)
    xr = x.clone().to(dev).requires_grad_(True)
    o64 = O.OPatchDiscriminator(21, norm)
    o64.load_state_dict(o.state_dict())
    o64 = o64.double().train()
    x64 = x.double().clone().requires_grad_(True)
    O.o_gan_loss(o64(x64), True).backward()
    yo, yr = o(xo), r(xr)
    _check(yr, yo, 1e-3, "patch logits")
    from fdgan.losses import GANLoss
    lo, lr = O.o_gan_loss(yo, True), GANLoss()(yr, True)
    _check(lr, lo, 1e-4, "GANLoss")
    lo.backward()
    lr.backward()
    _check_anchored(xr.grad, xo.grad, x64.grad, "input grad", floor=2e-3)
    _check_grads_anchored(r, o, o64, "D_pd " + norm)


@pytest.mark.parametrize("norm", ["batch", "instance"])
def test_patch_discriminator_without_kinks(dev, norm):
    """LeakyReLU slope 1 removes the only discontinuity: every gradient must then agree to 1e-5 in the MAX norm
    with an fp64 oracle — the composition (tape order, fused epilogues, split-K, statistics) is exact."""
    from oracle import ref_torch as O
    import fdgan.networks as N
    from fdgan.losses import GANLoss
    torch.manual_seed(10)
    o = O.OPatchDiscriminator(21, norm)
    o.apply(O.o_weights_init_normal)
    r = N.NLayerDiscriminator(21, norm_layer=N.get_norm_layer(norm))
    r.load_state_dict(o.state_dict())
    r.to(dev).train()
    o = o.double().train()
    for m in o.modules():
        if isinstance(m, torch.nn.LeakyReLU):
            m.negative_slope = 1.0
    for m in r.modules():
        if m.__class__.__name__ == "LeakyReLU":
            m.SLOPE = m.negative_slope = 1.0
    x = torch.cat((O.synth_posemaps(3, seed=11), O.synth_images(3, seed=12)), 1)
    xo = x.double().clone().requires_grad_(True)
    xr = x.clone().to(dev).requires_grad_(True)
    yo, yr = o(xo), r(xr)
    _check(yr, yo, 2e-5, "logits")
    O.o_gan_loss(yo, True).backward()
    GANLoss()(yr, True).backward()
    _check(xr.grad, xo.grad, 2e-5, "input grad")
    og = dict(o.named_parameters())
    gmax = max(p.grad.abs().max().item() for p in og.values())
    for n, p in r.named_parameters():
        d = (p.grad.detach().double().cpu() - og[n].grad).abs().max().item()
        assert d <= 2e-5 * max(og[n].grad.abs().max().item(), 1e-2 * gmax), (n, d)


def _make_siamese(O, dev, num_classes):
    import reid.models as RM
    from reid.models.embedding import EltwiseSubEmbed
    from reid.models.multi_branch import SiameseNet
    o = O.OSiameseNet(O.OReidResNet(50, cut_at_pooling=True),
                      O.OEltwiseSubEmbed(use_batch_norm=True, use_classifier=True, num_features=2048,
                                         num_classes=num_classes))
    o.embed_model.classifier.weight.data.normal_(0, 0.05)
    for m in o.modules():
        if isinstance(m, (torch.nn.BatchNorm2d, torch.nn.BatchNorm1d)):
            m.running_mean.normal_(0, 0.05)
            m.running_var.uniform_(0.8, 1.2)
            m.weight.data.uniform_(0.4, 0.6)
    r = SiameseNet(RM.create('resnet50', pretrained=False, cut_at_pooling=True),
                   EltwiseSubEmbed(use_batch_norm=True, use_classifier=True, num_features=2048,
                                   num_classes=num_classes))
    r.load_state_dict(o.state_dict())
    return o, r.to(dev)


def test_siamese_eval_bn_and_shared_branch(dev):
    from oracle import ref_torch as O
    torch.manual_seed(13)
    o, r = _make_siamese(O, dev, 1)
    o.train()
    r.train()
    O.o_set_bn_eval(o)
    from fdgan.networks import set_bn_fix
    r.apply(set_bn_fix)
    a, b, c = (O.synth_images(2, 128, 64, seed=s) for s in (14, 15, 16))
    # reference pattern of backward_Di: two calls sharing the first branch
    _, _, p1 = o(a, b)
    _, _, p2 = o(a, c)
    lo = (O.o_gan_loss(p1, True) + O.o_gan_loss(p2, False)) * 0.5
    lo.backward()
    q1, q2 = r.forward_shared(a.to(dev), [b.to(dev), c.to(dev)])
    _check(q1, p1, 1e-3, "pred_real")
    _check(q2, p2, 1e-3, "pred_fake")
    from fdgan.losses import GANLoss
    crit = GANLoss()
    from fdgan.model import _weighted
    lr = _weighted([(crit(q1, True), 0.5), (crit(q2, False), 0.5)])
    _check(lr, lo, 1e-4, "loss_Di")
    lr.backward()
    _check_grads(r, o, 2e-3, "D_id shared-branch", skip=("fc.",))
    # plain two-branch call with gradients to the second input only (generator update pattern)
    o.zero_grad()
    xo = c.clone().requires_grad_(True)
    xr = c.clone().to(dev).requires_grad_(True)
    _, _, po = o(a, xo)
    from rg_hip.tape import no_param_grad
    with no_param_grad(r):
        _, _, pr = r(a.to(dev), xr)
    _check(pr, po, 1e-3, "pred (x2 only)")
    O.o_gan_loss(po, True).backward()
    crit(pr, True).backward()
    _check_l2(xr.grad, xo.grad, 2e-3, "d fake through D_id")


def _opt(**kw):
    d = dict(stage=2, checkpoints="/tmp/rg_ckpt", name="t", norm="batch", drop=0.0, connect_layers=0, fuse_mode="cat",
             pose_feature_size=128, noise_feature_size=256, arch="resnet50", lr=0.001, niter=50, niter_decay=50,
             lambda_recon=100.0, lambda_veri=10.0, lambda_sp=10.0, smooth_label=False, random_init=True, quiet=True,
             batch_size=2)
    d.update(kw)
    return argparse.Namespace(**d)


def test_fdgan_step_matches_oracle(dev):
    """Two consecutive optimize_parameters() calls: all seven losses, the generated images and probes of the
    updated parameters of all four networks."""
    from oracle import ref_torch as O
    from fdgan.model import FDGANModel
    torch.manual_seed(17)
    b = 2
    oE = O.OSiameseNet(O.OReidResNet(50, cut_at_pooling=True),
                       O.OEltwiseSubEmbed(True, True, 2048, 2))
    oDi = O.OSiameseNet(O.OReidResNet(50, cut_at_pooling=True),
                        O.OEltwiseSubEmbed(True, True, 2048, 1))
    for net in (oE, oDi):
        net.embed_model.classifier.weight.data.normal_(0, 0.05)
        for m in net.modules():
            if isinstance(m, (torch.nn.BatchNorm2d, torch.nn.BatchNorm1d)):
                m.running_mean.normal_(0, 0.05)
                m.running_var.uniform_(0.8, 1.2)
                m.weight.data.uniform_(0.4, 0.6)
    oG = O.OPoseGenerator(128, 2048, 256, dropout=0.0)
    oG.apply(O.o_weights_init_normal)
    oDp = O.OPatchDiscriminator(21)
    oDp.apply(O.o_weights_init_normal)

    model = FDGANModel(_opt())
    model.net_E.module.load_state_dict(oE.state_dict())
    model.net_G.module.load_state_dict(oG.state_dict())
    model.net_Di.module.load_state_dict(oDi.state_dict())
    model.net_Dp.module.load_state_dict(oDp.state_dict())
    model.reset_model_status()
    ostep = O.OFDGANStep(oE, oG, oDi, oDp, lr=0.001, stage=2, lambda_recon=100.0, lambda_veri=10.0, lambda_sp=10.0)
    init_state = {n: {k: v.clone() for k, v in net.state_dict().items()}
                  for n, net in (("E", oE), ("G", oG), ("Di", oDi), ("Dp", oDp))}

    for it in range(2):
        origin, target, pose, labels, noise = O.synth_fdgan_batch(b, seed=100 + it)
        ref_losses, ref_fake = ostep.step(origin, target, pose, labels, noise)
        pid1 = torch.arange(b)
        pid2 = torch.where(labels == 1, pid1, pid1 + 1000)
        in1 = dict(pid=pid1, origin=origin[:b], target=target[:b], posemap=pose[:b], noise=noise[:b])
        in2 = dict(pid=pid2, origin=origin[b:], target=target[b:], posemap=pose[b:])
        model.set_input((in1, in2))
        model.optimize_parameters()
        got = model.get_current_errors()
        if it == 0:
            _check(model.fake, ref_fake, 1e-3, "step 0 fake")
        else:
            # after an Adam step (|update| ~ lr whatever |g| is, so rounding-level gradient elements move by +-lr
            # with a sign that is noise in BOTH implementations) images agree in the L2 sense
            _check_l2(model.fake, ref_fake, 2e-3, "step %d fake" % it, tol_max=5e-2)
        for k, v in ref_losses.items():
            assert abs(got[k] - v) <= 1e-3 * max(abs(v), 1e-3), "step %d loss %s: %r vs %r" % (it, k, got[k], v)
    # ---- updated parameters ------------------------------------------------------------------------
    # SGD nets (D_id, D_pd): the update is linear in the gradient -> compare the accumulated update in L2.
    # Adam nets (E, G): the first steps move EVERY element by ~lr*sign(g) whatever |g| is, so elements whose
    # gradient sits at rounding level get a noise sign in both implementations (measured after ONE step: gradient
    # rel. L2 error 2e-3, 0.04 % sign mismatches, identical zero pattern; at the second step Adam's g/sqrt(v)
    # turns the >20 % relative noise of the many |g| < 1e-6 elements into update differences): bound every
    # element by the reach of the two steps, the mean difference by lr/10 and the far fraction by 10 %.
    lr_adam, n_steps = 0.001 * 0.1, 2
    for name, rn, on, kind in (("E", model.net_E.module, oE, "adam"), ("G", model.net_G.module, oG, "adam"),
                               ("Di", model.net_Di.module, oDi, "sgd"), ("Dp", model.net_Dp.module, oDp, "sgd")):
        so, s0 = on.state_dict(), init_state[name]
        num = den = 0.0
        far = tot = 0
        sabs = 0.0
        for k, v in rn.state_dict().items():
            if v.dtype != torch.float32 or "running" in k:
                continue
            a, r_ = v.detach().double().cpu(), so[k].double()
            if kind == "sgd":
                num += (a - r_).pow(2).sum().item()
                den += (r_ - s0[k].double()).pow(2).sum().item()
            else:
                d = (a - r_).abs()
                assert d.max().item() <= 3.0 * lr_adam * n_steps, (name, k, d.max().item())
                far += (d > 0.1 * lr_adam).sum().item()
                sabs += d.sum().item()
                tot += d.numel()
        if kind == "sgd":
            assert (num / max(den, 1e-300)) ** 0.5 <= 5e-3, "net_%s SGD update rel L2 err %.3e" % (name, (num / den) ** 0.5)
        else:
            assert far <= 0.10 * tot, "net_%s: %d of %d elements more than lr/10 apart" % (name, far, tot)
            assert sabs / tot <= 0.1 * lr_adam, "net_%s: mean |dp| %.3e" % (name, sabs / tot)


def test_joint_step_4b_fdgan_adaptor(dev):
    """BASELINE config 4b: the joint ReID + GAN trainer step with FDGANModel in the GAN role (fdgan.adaptor.FDGANAdaptor)
    against oracle.o_joint_step_fd.  Step-0 losses are forward quantities (1e-3); the SGD discriminators' updates are
    linear in the gradient (L2), the Adam nets are bounded by the step size."""
    import torch.nn.functional as F
    from oracle import ref_torch as O
    from fdgan.model import FDGANModel
    from fdgan.adaptor import FDGANAdaptor
    import clustercontrast.models as M
    from clustercontrast.models.cm import ClusterMemory
    from clustercontrast.trainers import ClusterContrastWithGANTrainer
    from rg_hip import optim as roptim
    torch.manual_seed(23)
    b = 2
    oE = O.OSiameseNet(O.OReidResNet(50, cut_at_pooling=True), O.OEltwiseSubEmbed(True, True, 2048, 2))
    oDi = O.OSiameseNet(O.OReidResNet(50, cut_at_pooling=True), O.OEltwiseSubEmbed(True, True, 2048, 1))
    for net in (oE, oDi):
        net.embed_model.classifier.weight.data.normal_(0, 0.05)
        for m in net.modules():
            if isinstance(m, (torch.nn.BatchNorm2d, torch.nn.BatchNorm1d)):
                m.running_mean.normal_(0, 0.05)
                m.running_var.uniform_(0.8, 1.2)
                m.weight.data.uniform_(0.4, 0.6)
    oG = O.OPoseGenerator(128, 2048, 256, dropout=0.0)
    oG.apply(O.o_weights_init_normal)
    oDp = O.OPatchDiscriminator(21)
    oDp.apply(O.o_weights_init_normal)
    model = FDGANModel(_opt())
    model.net_E.module.load_state_dict(oE.state_dict())
    model.net_G.module.load_state_dict(oG.state_dict())
    model.net_Di.module.load_state_dict(oDi.state_dict())
    model.net_Dp.module.load_state_dict(oDp.state_dict())
    model.reset_model_status()
    ofd = O.OFDGANStep(oE, oG, oDi, oDp, lr=0.001, stage=2, lambda_recon=100.0, lambda_veri=10.0, lambda_sp=10.0)
    init_D = {n: {k: v.clone() for k, v in net.state_dict().items()} for n, net in (("Di", oDi), ("Dp", oDp))}

    oenc = O.OCCResNet(50, pooling_type="gem")
    renc = M.create('resnet50', pretrained=False, pooling_type="gem")
    renc.load_state_dict(oenc.state_dict())
    renc.to(dev).train()
    oenc.train()
    K, Dm = 32, oenc.num_features
    g = torch.Generator().manual_seed(6)
    bank = F.normalize(torch.randn(K, Dm, generator=g), dim=1)
    om = O.OClusterMemory(Dm, K, temp=0.05, momentum=0.1)
    om.features = bank.clone()
    rm = ClusterMemory(Dm, K, temp=0.05, momentum=0.1).to(dev)
    rm.features = bank.clone().to(dev)
    oopt = torch.optim.Adam([{"params": [p]} for p in oenc.parameters() if p.requires_grad], lr=3.5e-4, weight_decay=5e-4)
    ropt = roptim.Adam([{"params": [p]} for p in renc.parameters() if p.requires_grad], lr=3.5e-4, weight_decay=5e-4)
    gan = FDGANAdaptor(model)
    trainer = ClusterContrastWithGANTrainer(renc, GAN=gan, memory=rm)

    imgs = O.synth_images(8, 128, 64, seed=10)
    labels = torch.randint(0, K, (2,), generator=g).repeat_interleave(4)
    batch = O.synth_fdgan_batch(b, seed=100)
    origin, target, pose, lab, noise = batch
    ref, ref_fake = O.o_joint_step_fd(ofd, oenc, om, oopt, imgs, labels, batch)
    pid1 = torch.arange(b)
    pid2 = torch.where(lab == 1, pid1, pid1 + 1000)
    in1 = dict(pid=pid1, origin=origin[:b], target=target[:b], posemap=pose[:b], noise=noise[:b])
    in2 = dict(pid=pid2, origin=origin[b:], target=target[b:], posemap=pose[b:])
    gan.set_input((in1, in2))
    loss = trainer.joint_step(imgs.to(dev), labels.to(dev), torch.arange(8).to(dev), ropt)
    errs = gan.get_current_errors()
    assert abs(loss.item() - ref['loss']) <= 1e-3 * abs(ref['loss']), (loss.item(), ref['loss'])
    for k in ('G', 'D_i', 'D_p'):
        assert abs(errs[k] - ref[k]) <= 1e-3 * max(abs(ref[k]), 1e-3), (k, errs[k], ref[k])
    _check(gan.fake_image, ref_fake, 1e-3, "fake")
    _check_l2(rm.features, om.features, 1e-3, "bank after the step")
    for name, rn, on in (("Di", model.net_Di.module, oDi), ("Dp", model.net_Dp.module, oDp)):
        num = den = 0.0
        so, s0 = on.state_dict(), init_D[name]
        for k, v in rn.state_dict().items():
            if v.dtype != torch.float32 or "running" in k:
                continue
            a, r_ = v.detach().double().cpu(), so[k].double()
            num += (a - r_).pow(2).sum().item()
            den += (r_ - s0[k].double()).pow(2).sum().item()
        assert den > 0 and (num / den) ** 0.5 <= 1e-2, (name, (num / max(den, 1e-300)) ** 0.5)
    lr_adam = 1e-4
    for rn, on in ((model.net_E.module, oE), (model.net_G.module, oG)):
        so = on.state_dict()
        for k, v in rn.state_dict().items():
            if v.dtype == torch.float32 and "running" not in k:
                assert (v.detach().double().cpu() - so[k].double()).abs().max().item() <= 3.0 * lr_adam, k


def test_fdgan_step_through_rccl_reducers(dev, monkeypatch):
    """The data-parallel path on one rank: torch.distributed over RCCL (backend "nccl"), the gradient arenas of all three
    optimizers all-reduced in place (async, overlapped with the next backward; side / aux streams active).  With world
    size 1 the collective is an identity, so two steps must reproduce the un-reduced run exactly."""
    import torch.distributed as dist
    from fdgan.model import FDGANModel
    from oracle import ref_torch as O

    def run(force):
        monkeypatch.setenv("RG_FORCE_REDUCE", "1" if force else "0")
        torch.manual_seed(5)
        model = FDGANModel(_opt())
        assert all(r.active() == force for r in model.reducers)
        model.reset_model_status()
        out = []
        for it in range(2):
            origin, target, pose, labels, noise = O.synth_fdgan_batch(2, seed=300 + it)
            pid1 = torch.arange(2)
            pid2 = torch.where(labels == 1, pid1, pid1 + 1000)
            model.set_input((dict(pid=pid1, origin=origin[:2], target=target[:2], posemap=pose[:2], noise=noise[:2]),
                             dict(pid=pid2, origin=origin[2:], target=target[2:], posemap=pose[2:])))
            model.optimize_parameters()
            out.append(model.get_current_errors())
        return out

    if not dist.is_initialized():
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29617", rank=0, world_size=1)
    try:
        ref = run(False)
        got = run(True)
    finally:
        dist.destroy_process_group()
    for a, b in zip(ref, got):
        for k in a:
            assert abs(a[k] - b[k]) <= 1e-6 * max(abs(a[k]), 1e-3), (k, a[k], b[k])
