"""dual_gan modules on the MI355X (HIP tape programs of reid-gan_amd/dual_gan) against the oracle restatement
(oracle/ref_dualgan.py, pinned on the reference by tests/golden/reference_dualgan.npz) on identical seeded weights and
inputs.  Forward 1e-3 max-norm (measured ~1e-5); gradients by the flip-robust L2 metrics of test_modules_gpu.py."""
import argparse
import copy
import os

import numpy as np
import pytest
import torch

from tests.test_modules_gpu import _check, _check_grads, _check_l2

pytestmark = pytest.mark.gpu

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_dualgan.npz"))


def _load(rg_mod, o_mod, dev):
    missing = rg_mod.load_state_dict(o_mod.state_dict(), strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    return rg_mod.to(dev)


def test_multihead_attention_public_api(dev):
    """rg_hip.attention.MultiheadAttention called like nn.MultiheadAttention ([L, B, E]) — self and cross attention."""
    from rg_hip.attention import MultiheadAttention
    torch.manual_seed(1)
    ref = torch.nn.MultiheadAttention(64, 2).double()
    with torch.no_grad():
        ref.in_proj_bias.normal_(0, 0.1)
        ref.out_proj.bias.normal_(0, 0.1)
    mha = MultiheadAttention(64, 2)
    mha.load_state_dict({k: v.float() for k, v in ref.state_dict().items()})
    mha.to(dev)
    g = torch.Generator().manual_seed(2)
    for (L, S, shared) in ((12, 12, True), (10, 7, False)):
        q = torch.randn(L, 3, 64, generator=g)
        kv = q if shared else torch.randn(S, 3, 64, generator=g)
        v2 = None if shared else torch.randn(S, 3, 64, generator=g)
        qr = q.double().requires_grad_(True)
        kr = qr if shared else kv.double().requires_grad_(True)
        vr = kr if shared else v2.double().requires_grad_(True)
        yr = ref(qr, kr, vr)[0]
        dy = torch.randn(yr.shape, generator=g)
        ref.zero_grad()
        yr.backward(dy.double())
        qd = q.to(dev).requires_grad_(True)
        kd = qd if shared else kv.to(dev).requires_grad_(True)
        vd = kd if shared else v2.to(dev).requires_grad_(True)
        mha.zero_grad()
        y = mha(qd, kd, vd)[0]
        y.backward(dy.to(dev))
        _check(y, yr, 2e-5, "mha out")
        _check(qd.grad, qr.grad, 5e-5, "mha dq")
        if not shared:
            _check(kd.grad, kr.grad, 5e-5, "mha dk")
            _check(vd.grad, vr.grad, 5e-5, "mha dv")
        for n, p in mha.named_parameters():
            _check(p.grad, dict(ref.named_parameters())[n].grad, 5e-5, "mha grad " + n)


def test_pctm(dev):
    from dual_gan.models.PTM import PCTM
    from tests.golden import cases_dualgan as C
    from tests.golden.cases import sub
    on, (q, v) = C.pctm_case()
    rg = _load(PCTM(d_model=64, nhead=2, num_CABs=2, num_TTBs=2, dim_feedforward=64, activation="LeakyReLU", affine=True,
                    norm='instance'), on, dev)
    qo, vo = q.clone().requires_grad_(True), v.clone().requires_grad_(True)
    qd, vd = q.to(dev).requires_grad_(True), v.to(dev).requires_grad_(True)
    yo, y = on(qo, vo), rg(qd, vd)
    _check(y, yo, 1e-3, "pctm fwd")
    s, _ = sub(y.detach().cpu())
    ref = GOLD["pctm_fwd"]
    assert np.abs(np.asarray(s, dtype=np.float64).reshape(ref.shape) - ref).max() <= 1e-3 * np.abs(ref).max()
    g = torch.Generator().manual_seed(3)
    dy = torch.randn(yo.shape, generator=g)
    yo.backward(dy)
    y.backward(dy.to(dev))
    _check_l2(qd.grad, qo.grad, 2e-3, "pctm dquery")
    _check_l2(vd.grad, vo.grad, 2e-3, "pctm dvalue")
    _check_grads(rg, on, 2e-3, "pctm grads")


def test_posegen1(dev):
    from dual_gan.models import networks as N
    from tests.golden import cases_dualgan as C
    from tests.golden.cases import sub
    on, (feat, pose) = C.posegen1_case()
    rg = _load(N.PoseGenerator1(64, 18, 256, 3, 'instance', 'LeakyReLU', False, False, 3, True, 2, 2, 2), on, dev)
    rg.train()
    fo = feat.clone().requires_grad_(True)
    fd = feat.to(dev).requires_grad_(True)
    yo, y = on(fo, pose), rg(fd, pose.to(dev))
    r = _check(y, yo, 1e-3, "posegen1 fwd")
    s, _ = sub(y.detach().cpu())
    ref = GOLD["posegen1_fwd"]
    assert np.abs(np.asarray(s, dtype=np.float64).reshape(ref.shape) - ref).max() <= 1e-3 * np.abs(ref).max()
    g = torch.Generator().manual_seed(5)
    dy = torch.randn(yo.shape, generator=g)
    yo.backward(dy)
    y.backward(dy.to(dev))
    _check_l2(fd.grad, fo.grad, 5e-3, "posegen1 dfeat")
    _check_grads(rg, on, 5e-3, "posegen1 grads", tol_tensor=2e-2)
    print("posegen1 fwd rel err %.2e" % r)


def test_resdiscriminator(dev):
    from dual_gan.models import networks as N
    from tests.golden import cases_dualgan as C
    on, x = C.resdisc_case()
    rg = _load(N.ResDiscriminator(3, 32, 128, 3, 'none', 'LeakyReLU', True), on, dev)
    rg.train()
    xd = x.to(dev).requires_grad_(True)
    xo = x.clone().requires_grad_(True)
    for it in range(2):                                   # u, v advance once per training forward on both sides
        yo, y = on(xo), rg(xd)
        _check(y, yo, 1e-3, "resdisc fwd %d" % it)
        ref = GOLD["resdisc_fwd%d" % it]
        got = y.detach().double().cpu().flatten().numpy()
        assert np.abs(got - ref).max() <= 1e-3 * np.abs(ref).max()
    (yo ** 2).mean().backward()
    (y ** 2).mean().backward()
    _check_l2(xd.grad, xo.grad, 5e-3, "resdisc dx")
    _check_grads(rg, on, 5e-3, "resdisc grads", tol_tensor=2e-2)
    ob = dict(on.named_buffers())
    for n, b in rg.named_buffers():
        _check(b, ob[n], 1e-4, "resdisc buffer " + n)
    rg.eval()
    on.eval()
    _check(rg(xd.detach()), on(x), 1e-3, "resdisc eval fwd")


def _gan_opt(**kw):
    opt = dict(gan_train=True, checkpoints_dir="/tmp/rg_ckpt", name="t", load_pretrain="", model_gen="Pose", num_feats=256,
               layers_g=3, image_nc=3, pose_nc=18, norm="instance", use_spect_g=False, use_spect_d=True, use_coord=False,
               num_blocks=3, nhead=2, num_CABs=2, num_TTBs=2, dis_layers=3, init_type="orthogonal", verbose=False,
               pool_size=0, gan_lr=2e-4, gan_mode="lsgan", no_vgg_loss=True, beta1=0.5, ratio_g2d=0.1, lambda_rec=2.0,
               lambda_g=5.0, gan_lr_policy="lambda", iter_start=0, niter=100, niter_decay=100, continue_train=False,
               which_epoch="latest", bipath_gan=False, use_adp=False, lambda_fus=0.8)
    opt.update(kw)
    return argparse.Namespace(**opt)


def _ae_pair(dev):
    """AEModel (HIP) and OAEModel (oracle) on identical weights."""
    from dual_gan.models.models import create_model
    from oracle import ref_dualgan as D
    from tests.golden import cases_dualgan as C
    opt = _gan_opt(model="AE")
    model = create_model(opt)
    og, _ = C.posegen1_case()
    od, _ = C.resdisc_case()
    model.net_G.module.load_state_dict(og.state_dict())
    model.net_D.module.load_state_dict(od.state_dict())
    return model, D.OAEModel(og, od), D


def _adam_close(rg_mod, o_mod, lr, steps, what):
    """After `steps` Adam steps every parameter moved by <= ~lr per step on both sides; first-step updates are
    sign-like, so compare against the step size, not the parameter scale."""
    op = dict(o_mod.named_parameters())
    far = tot = 0
    worst = 0.0
    for n, p in rg_mod.named_parameters():
        d = (p.detach().cpu().double() - op[n].detach().double()).abs()
        worst = max(worst, d.max().item())
        far += (d > 0.5 * lr * steps).sum().item()
        tot += d.numel()
    assert worst <= 3.0 * lr * steps, "%s: a parameter differs by %.3e (> 3 lr steps)" % (what, worst)
    assert far <= 0.02 * tot, "%s: %.2f%% of the parameters differ by more than half a step" % (what, 100.0 * far / tot)


def test_aemodel_gan_step(dev):
    """AEModel.synthesize_p + optimize_generated (D update, then G update) for two steps: losses within 1e-3 at step
    0; parameters Adam-close; step-1 losses (after sign-like first Adam updates) within 1e-2."""
    model, omodel, D = _ae_pair(dev)
    g = torch.Generator().manual_seed(11)
    for it in range(2):
        inp = D.synth_dualgan_inputs(4, 64, 32, seed=20 + it)
        feat = torch.nn.functional.normalize(torch.randn(4, 2048, 8, 4, generator=g).abs(), dim=1)
        omodel.set_input(inp)
        model.set_input(inp)
        fo = omodel.synthesize_p(feat)
        fr = model.synthesize_p(feat.to(dev))
        _check(fr, fo, 1e-3 if it == 0 else 2e-2, "fake image step %d" % it)
        omodel.optimize_generated()
        model.optimize_generated()
        errs = model.get_current_errors()
        tol = 1e-3 if it == 0 else 1e-2
        assert abs(errs["D"] - omodel.loss_D.item()) <= tol * abs(omodel.loss_D.item()), (it, errs, omodel.loss_D.item())
        assert abs(errs["G"] - omodel.loss_G.item()) <= tol * abs(omodel.loss_G.item()), (it, errs, omodel.loss_G.item())
        _adam_close(model.net_G.module, omodel.net_G, 2e-4, it + 1, "G params step %d" % it)
        _adam_close(model.net_D.module, omodel.net_D, 2e-5, it + 1, "D params step %d" % it)
    assert model.get_current_learning_rate() == (2e-4, 2e-5)
    model.update_learning_rate()


def test_aemodel_get_loss_G_with_cluster_reconstruction(dev):
    """AEModel.get_loss_G(need_cm=True, cluster_features=...) — the reference's default arguments (AE_model.py:355-372): the generator
    objective and the per-sample reconstruction error of the image synthesised from the cluster features, with gradients."""
    model, omodel, D = _ae_pair(dev)
    inp = D.synth_dualgan_inputs(2, 64, 32, seed=41)
    g = torch.Generator().manual_seed(42)
    feat = torch.nn.functional.normalize(torch.randn(2, 2048, 8, 4, generator=g).abs(), dim=1)
    cfeat = torch.nn.functional.normalize(torch.randn(2, 2048, 8, 4, generator=g).abs(), dim=1)
    omodel.set_input(inp)
    omodel.synthesize_p(feat)
    lo, ro = omodel.get_loss_G(need_cm=True, cluster_features=cfeat)
    model.set_input(inp)
    model.synthesize_p(feat.to(dev))
    lg, rg = model.get_loss_G(need_cm=True, cluster_features=cfeat.to(dev))
    assert abs(float(lg) - float(lo)) <= 1e-3 * abs(float(lo)), (float(lg), float(lo))
    assert tuple(rg.shape) == (2,)
    _check(rg, ro, 1e-3, "loss_rec per sample")
    w = torch.tensor([0.3, 1.7])
    omodel.optimizer_G.zero_grad()
    (lo + (ro * w).sum()).backward()
    model.optimizer_G.zero_grad()
    from rg_hip.tape import backward as rg_backward
    rg_backward(lg + (rg * w.to(dev)).sum())
    _check_grads(model.net_G.module, omodel.net_G, 5e-3, "get_loss_G(need_cm) grads", tol_tensor=5e-2)
    with pytest.raises(TypeError):
        model.get_loss_G(need_cm=True)
    # get_L1_loss (AE_model.py:378-390): per-sample reconstruction error, alone and with the per-sample discriminator term
    omodel.synthesize_p(feat)
    model.synthesize_p(feat.to(dev))
    _check(model.get_L1_loss(), omodel.get_L1_loss(), 1e-3, "get_L1_loss")
    lo, lg = omodel.get_L1_loss(with_dis=True), model.get_L1_loss(with_dis=True)
    _check(lg, lo, 1e-3, "get_L1_loss(with_dis)")
    omodel.optimizer_G.zero_grad()
    (lo * w).sum().backward()
    model.optimizer_G.zero_grad()
    rg_backward((lg * w.to(dev)).sum())
    _check_grads(model.net_G.module, omodel.net_G, 5e-3, "get_L1_loss(with_dis) grads", tol_tensor=5e-2)


def test_joint_step_4a(dev):
    """BASELINE config 4a — the joint ReID + GAN step as committed (trainers_b.py:617-774): ResNet-50 cluster-contrast
    encoder (GeM) -> PoseGenerator1 from the detached feature map -> lsgan / L1 generator loss + cluster-contrast loss ->
    D step -> one backward through G and the encoder -> both optimizers.  Step-0 losses are forward quantities (1e-3);
    the memory bank and the G / D parameters after the step are checked Adam-aware."""
    import torch.nn.functional as F
    from oracle import ref_torch as O
    import clustercontrast.models as M
    from clustercontrast.models.cm import ClusterMemory
    from clustercontrast.trainers import ClusterContrastWithGANTrainer
    from rg_hip import optim as roptim
    model, omodel, D = _ae_pair(dev)
    torch.manual_seed(3)
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
    trainer = ClusterContrastWithGANTrainer(renc, GAN=model, memory=rm)

    B = 8
    imgs = O.synth_images(B, 128, 64, seed=10)                 # encoder map 8x4 at 128x64
    gan_in = D.synth_dualgan_inputs(B, 64, 32, seed=12)        # pose tokens 8x4 at 64x32
    labels = torch.randint(0, K, (2,), generator=g).repeat_interleave(4)
    indexes = torch.arange(B)
    conf = torch.rand(64, generator=g) + 0.5

    lo, lo_cl, lo_G, lo_D = D.o_joint_step(oenc, om, omodel, oopt, imgs, labels, gan_in, conf_mask=conf[indexes])
    model.set_input(gan_in)
    lr = trainer.joint_step(imgs.to(dev), labels.to(dev), indexes.to(dev), ropt, conf_weight=conf.to(dev))
    errs = model.get_current_errors()
    assert abs(lr.item() - lo.item()) <= 1e-3 * abs(lo.item()), (lr.item(), lo.item())
    assert abs(errs["G"] - lo_G.item()) <= 1e-3 * abs(lo_G.item()), (errs, lo_G.item())
    assert abs(errs["D"] - lo_D.item()) <= 1e-3 * abs(lo_D.item()), (errs, lo_D.item())
    _check(model.fake_image, omodel.fake_image, 1e-3, "fake image")
    _check_l2(rm.features, om.features, 1e-3, "bank after the step")
    _adam_close(model.net_G.module, omodel.net_G, 2e-4, 1, "G params")
    _adam_close(model.net_D.module, omodel.net_D, 2e-5, 1, "D params")


def test_aegenerator_and_gan_step(dev):
    """model_gen='AE': AEGenerator forward / forward_enc / forward_dec against the oracle (== reference) and the
    stand-alone GAN step AEModel.optimize_parameters() that GANTrainer.train_gan drives (trainers.py:286-335)."""
    from dual_gan.models.models import create_model
    from oracle import ref_dualgan as D
    from tests.golden import cases_dualgan as C
    from tests.golden.cases import sub
    og, x = C.aegen_case()
    od, _ = C.resdisc_case()
    model = create_model(_gan_opt(model="AE", model_gen="AE"))
    model.net_G.module.load_state_dict(og.state_dict())
    model.net_D.module.load_state_dict(od.state_dict())
    G = model.net_G.module
    y = G(x.to(dev))
    _check(y, og(x), 1e-3, "aegen fwd")
    ref = GOLD["aegen_fwd"]
    got = np.asarray(sub(y.detach().cpu())[0], dtype=np.float64).reshape(ref.shape)
    assert np.abs(got - ref).max() <= 1e-3 * np.abs(ref).max()
    # enc / dec as separate autograd nodes, gradients to the parameters of both halves
    xo, xd = x.clone().requires_grad_(True), x.to(dev).requires_grad_(True)
    fo, fd = og.forward_enc(xo), G.forward_enc(xd)
    _check(fd, fo, 1e-3, "forward_enc")
    yo, yd = og.forward_dec(fo), G.forward_dec(fd)
    _check(yd, yo, 1e-3, "forward_dec")
    g = torch.Generator().manual_seed(4)
    dy = torch.randn(yo.shape, generator=g)
    yo.backward(dy)
    yd.backward(dy.to(dev))
    _check_l2(xd.grad, xo.grad, 5e-3, "d image")
    _check_grads(G, og, 5e-3, "aegen grads", tol_tensor=2e-2)
    G.zero_grad()
    og.zero_grad()
    # hard_mix: same selections and mixture
    f4o = og.forward_enc(D.synth_dualgan_inputs(4, 64, 32, seed=33)['Xs']).detach()
    reid = torch.randn(4, 32, generator=g)
    mo = D.o_hard_mix(f4o, reid, 2, 0.8)
    mr = model.hard_mix(f4o.to(dev), reid.to(dev), 2)
    _check(mr, mo, 1e-5, "hard_mix")
    # stand-alone GAN step (D update, then G update) on an AE forward
    omodel = D.OAEModel(og, od)
    inp = D.synth_dualgan_inputs(4, 64, 32, seed=31)
    omodel.set_input(inp)
    omodel.fake_image = og(inp['Xs'])
    omodel.optimize_generated()
    model.set_input(inp)
    model.optimize_parameters()
    errs = model.get_current_errors()
    assert abs(errs["D"] - omodel.loss_D.item()) <= 1e-3 * abs(omodel.loss_D.item()), (errs, omodel.loss_D.item())
    assert abs(errs["G"] - omodel.loss_G.item()) <= 1e-3 * abs(omodel.loss_G.item()), (errs, omodel.loss_G.item())
    _adam_close(model.net_G.module, omodel.net_G, 2e-4, 1, "G params")
    _adam_close(model.net_D.module, omodel.net_D, 2e-5, 1, "D params")


def test_dec_generators(dev):
    """`--model_gen DEC` (DECGenerator1, networks.py:401-444) and the older DECGenerator (networks.py:356-398): forward, input
    gradient and parameter gradients against the oracle (== reference, golden fixture), then define_G's 'DEC' branch."""
    from dual_gan.models import networks as N
    from tests.golden import cases_dualgan as C
    from tests.golden.cases import sub
    for tag, case, ctor in (
            ("decgen1", C.decgen1_case, lambda: N.DECGenerator1(64, 256, 3, 'instance', 'LeakyReLU', False, False, 3, 3)),
            ("decgen", C.decgen_case, lambda: N.DECGenerator(64, 2048, 3, 'instance', 'LeakyReLU', False, False, 3))):
        on, feat = case()
        rg = _load(ctor(), on, dev)
        rg.train()
        # gradients are judged against an fp64 evaluation of the oracle: on this case torch's own fp32 CPU backward is 1.2e-2
        # (relative L2) away from fp64 — a LeakyReLU gate decided the other way behind an instance norm — while the HIP path
        # is 2e-6 away (tools/debug/dec_err.py)
        on = copy.deepcopy(on).double()
        fo = feat.double().requires_grad_(True)
        fd = feat.to(dev).requires_grad_(True)
        yo, y = on(fo), rg(fd)
        _check(y, yo, 1e-3, tag + " fwd")
        ref = GOLD[tag + "_fwd"]
        got = np.asarray(sub(y.detach().cpu())[0], dtype=np.float64).reshape(ref.shape)
        assert np.abs(got - ref).max() <= 1e-3 * np.abs(ref).max()
        g = torch.Generator().manual_seed(7)
        dy = torch.randn(yo.shape, generator=g)
        yo.backward(dy.double())
        y.backward(dy.to(dev))
        _check_l2(fd.grad, fo.grad, 1e-3, tag + " dfeat")
        _check_grads(rg, on, 1e-3, tag + " grads", tol_tensor=2e-2)
    # define_G builds DECGenerator1 for 'DEC' with the reference's argument order (networks.py:21-22)
    opt = argparse.Namespace(model_gen='DEC', init_type='orthogonal', gpu_ids=[0])
    net = N.define_G(opt, image_nc=3, pose_nc=18, ngf=64, img_f=256, encoder_layer=3, norm='instance', activation='LeakyReLU',
                     use_spect=False, use_coord=False, output_nc=3, num_blocks=3)
    inner = net.module if hasattr(net, "module") else net
    assert type(inner).__name__ == "DECGenerator1"
    on, feat = C.decgen1_case()
    inner.load_state_dict(on.state_dict())
    _check(net(feat.to(dev)), on(feat), 1e-3, "define_G DEC fwd")
    with pytest.raises(NotImplementedError):
        N.define_G(argparse.Namespace(model_gen="PoseAE", init_type='orthogonal', gpu_ids=[0]), 3, 18)
    # AEModel with --model_gen DEC: synthesize(features) (AE_model.py:209-210), then the stand-alone D / G update on it
    from dual_gan.models.models import create_model
    from oracle import ref_dualgan as D
    od, _ = C.resdisc_case()
    model = create_model(_gan_opt(model="AE", model_gen="DEC"))
    model.net_G.module.load_state_dict(on.state_dict())
    model.net_D.module.load_state_dict(od.state_dict())
    omodel = D.OAEModel(on, od)
    inp = D.synth_dualgan_inputs(2, 64, 32, seed=34)
    omodel.set_input(inp)
    omodel.fake_image = on(feat)
    omodel.optimize_generated()
    model.set_input(inp)
    model.synthesize(feat.to(dev))
    _check(model.fake_image, omodel.fake_image, 1e-3, "AEModel DEC synthesize")
    model.optimize_generated()
    errs = model.get_current_errors()
    assert abs(errs["D"] - omodel.loss_D.item()) <= 1e-3 * abs(omodel.loss_D.item()), (errs, omodel.loss_D.item())
    assert abs(errs["G"] - omodel.loss_G.item()) <= 1e-3 * abs(omodel.loss_G.item()), (errs, omodel.loss_G.item())
    _adam_close(model.net_G.module, omodel.net_G, 2e-4, 1, "G params (DEC)")
    _adam_close(model.net_D.module, omodel.net_D, 2e-5, 1, "D params (DEC)")


def test_fd_generator(dev):
    """`--model_gen FD` (FDGenerator, networks.py:449-538, 'add' fusion of the ReID vector and a 512-d noise): forward, both input
    gradients and the parameter gradients against an fp64 run of the oracle (== reference, golden `fdgen_*`); define_G builds it, and
    calling it the way the reference's own AEModel does — without the noise — fails as it does there."""
    from dual_gan.models import networks as N
    from tests.golden import cases_dualgan as C
    from tests.golden.cases import sub
    on, (feat, noise) = C.fdgen_case()
    rg = _load(N.FDGenerator(256, 64, output_nc=3, noise_nc=512, fuse_mode='add'), on, dev)
    rg.train()
    o64 = copy.deepcopy(on).double()
    fo, no = feat.double().requires_grad_(True), noise.double().requires_grad_(True)
    fd, nd = feat.to(dev).requires_grad_(True), noise.to(dev).requires_grad_(True)
    yo, y = o64(fo, no), rg(fd, nd)
    _check(y, yo, 1e-3, "fdgen fwd")
    ref = GOLD["fdgen_fwd"]
    got = np.asarray(sub(y.detach().cpu())[0], dtype=np.float64).reshape(ref.shape)
    assert np.abs(got - ref).max() <= 1e-3 * np.abs(ref).max()
    g = torch.Generator().manual_seed(9)
    dy = torch.randn(yo.shape, generator=g)
    yo.backward(dy.double())
    y.backward(dy.to(dev))
    _check_l2(fd.grad, fo.grad, 5e-3, "fdgen dfeat")
    _check_l2(nd.grad, no.grad, 5e-3, "fdgen dnoise")
    _check_grads(rg, o64, 5e-3, "fdgen grads", tol_tensor=5e-2)
    opt = argparse.Namespace(model_gen='FD', init_type='orthogonal', gpu_ids=[0])
    net = N.define_G(opt, image_nc=3, pose_nc=18, ngf=64, img_f=256, encoder_layer=3, norm='instance', activation='LeakyReLU',
                     use_spect=False, use_coord=False, output_nc=3, num_blocks=3)
    inner = net.module if hasattr(net, "module") else net
    assert type(inner).__name__ == "FDGenerator" and inner.fuse_mode == 'add'
    inner.load_state_dict(on.state_dict())
    _check(net(feat.to(dev), noise.to(dev)), on(feat, noise), 1e-3, "define_G FD fwd")
    with pytest.raises(AttributeError):
        net(feat.to(dev))                       # AEModel.synthesize(features) in the reference: noise is None


def test_dptn_generator(dev):
    """DPTNGenerator (both branches batched through the shared blocks) against the oracle (== reference): both outputs,
    all parameter gradients (shared weights receive both branches' contributions), the image gradient, and the
    inference form (is_train=False)."""
    from dual_gan.models import networks as N
    from tests.golden import cases_dualgan as C
    from tests.golden.cases import sub
    on, (xs, ps, pt) = C.dptn_case()
    rg = _load(N.DPTNGenerator(3, 18, 64, 256, 3, 'instance', 'LeakyReLU', False, False, 3, 3, True, 2, 2, 2), on, dev)
    rg.train()
    xo, xd = xs.clone().requires_grad_(True), xs.to(dev).requires_grad_(True)
    (to, so), (tr, sr) = on(xo, ps, pt), rg(xd, ps.to(dev), pt.to(dev))
    _check(tr, to, 1e-3, "dptn out_t")
    _check(sr, so, 1e-3, "dptn out_s")
    ref = GOLD["dptn_fwd_t"]
    got = np.asarray(sub(tr.detach().cpu())[0], dtype=np.float64).reshape(ref.shape)
    assert np.abs(got - ref).max() <= 1e-3 * np.abs(ref).max()
    g = torch.Generator().manual_seed(6)
    dt, ds = torch.randn(to.shape, generator=g), torch.randn(so.shape, generator=g)
    ((to * dt).sum() + (so * ds).sum()).backward()
    torch.autograd.backward([tr, sr], [dt.to(dev), ds.to(dev)])
    _check_l2(xd.grad, xo.grad, 5e-3, "dptn d source")
    _check_grads(rg, on, 5e-3, "dptn grads", tol_tensor=2e-2)
    with torch.no_grad():
        ti, si = rg(xs.to(dev), ps.to(dev), pt.to(dev), False)
    assert si is None
    _check(ti, on(xs, ps, pt, False)[0], 1e-3, "dptn inference out_t")


def test_ganloss_modes(dev):
    """external_function.GANLoss: lsgan / vanilla / hinge / wgangp, discriminator and generator forms, value + gradient."""
    import torch.nn.functional as F
    from dual_gan.models.external_function import GANLoss
    g = torch.Generator().manual_seed(17)
    pred = torch.randn(3, 1, 16, 8, generator=g) * 2

    def ref(mode, x, real, disc):
        if mode == 'lsgan':
            l = (x - (1.0 if real else 0.0)) ** 2
            return l.mean() if disc else l
        if mode == 'vanilla':
            return F.binary_cross_entropy_with_logits(x, torch.full_like(x, 1.0 if real else 0.0))
        if disc:
            xx = -x if real else x
            return F.relu(1 + xx).mean() if mode == 'hinge' else xx.mean()
        return -x.mean()
    for mode in ('lsgan', 'vanilla', 'hinge', 'wgangp'):
        crit = GANLoss(mode)
        for real in (True, False):
            for disc in (True, False):
                xr = pred.double().requires_grad_(True)
                xd = pred.to(dev).requires_grad_(True)
                lr, ld = ref(mode, xr, real, disc), crit(xd, real, disc)
                _check(ld, lr, 2e-5, "%s real=%s disc=%s" % (mode, real, disc))
                lr.sum().backward()
                (ld.sum() if ld.dim() else ld).backward()
                _check(xd.grad, xr.grad, 2e-5, "%s grad" % mode)
    with pytest.raises(NotImplementedError):
        GANLoss('nonsense')
