"""CPU: the dual_gan oracle (oracle/ref_dualgan.py) against the golden vectors recorded from the reference's own
dual_gan modules by tests/golden/make_golden_dualgan.py.  No reference and no GPU needed."""
import os

import numpy as np
import torch

from oracle import ref_dualgan as D
from tests.golden import cases_dualgan as C
from tests.golden.cases import sub

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_dualgan.npz"))


def _cmp(got, key, tol=2e-5):
    ref = GOLD[key]
    got = np.asarray(got, dtype=np.float64).reshape(ref.shape)
    scale = max(np.abs(ref).max(), 1e-12)
    err = np.abs(got - ref).max()
    assert err <= tol * scale, "%s: %.3e vs scale %.3e" % (key, err, scale)


def test_pctm():
    net, (q, v) = C.pctm_case()
    s, st = sub(net(q, v))
    _cmp(s, "pctm_fwd")
    _cmp(st, "pctm_fwd_stats")


def test_posegen1_forward_backward():
    net, (feat, pose) = C.posegen1_case()
    feat = feat.clone().requires_grad_(True)
    y = net(feat, pose)
    s, st = sub(y)
    _cmp(s, "posegen1_fwd")
    _cmp(st, "posegen1_fwd_stats")
    g = torch.Generator().manual_seed(5)
    y.backward(torch.randn(y.shape, generator=g))
    s, st = sub(feat.grad)
    _cmp(s, "posegen1_dfeat", 1e-4)
    params = dict(net.named_parameters())
    for key in GOLD.files:
        if key.startswith("posegen1_g_"):
            _cmp(sub(params[key[len("posegen1_g_"):]].grad)[0], key, 2e-4)


def test_resdiscriminator_spectral_norm():
    net, x = C.resdisc_case()
    for it in range(2):
        y = net(x)
        _cmp(y.detach().flatten().numpy(), "resdisc_fwd%d" % it)
    (y ** 2).mean().backward()
    params = dict(net.named_parameters())
    for key in GOLD.files:
        if key.startswith("resdisc_g_"):
            _cmp(sub(params[key[len("resdisc_g_"):]].grad)[0], key, 1e-4)
    _cmp(dict(net.named_buffers())["block0.model.0.weight_u"].numpy(), "resdisc_u_block0")


def test_lsgan_and_bicubic():
    pred = C.lsgan_case()
    vals = []
    for real in (True, False):
        vals += [D.o_lsgan(pred, real, True).item(), D.o_lsgan(pred, real, False).mean().item()]
    _cmp(vals, "lsgan")
    s, st = sub(D.o_my_transform(C.bicubic_case(), (64, 32)))
    _cmp(s, "bicubic_normalize")
    _cmp(st, "bicubic_normalize_stats")


def test_cm_gan():
    import torch.nn.functional as F
    from oracle import ref_torch as O
    from tests.golden import cases as C0
    bank, feats, labels, gout = C0.cm_case()
    gbank = F.normalize(bank.flip(0) + 0.1, dim=1)
    gfeat = feats.flip(1) * 3.0 + 0.2
    x = feats.clone().requires_grad_(True)
    y = O.OCMGan.apply(x, gfeat, labels, bank, gbank, torch.Tensor([0.2]))
    y.backward(gout)
    _cmp(sub(y.detach())[0], "cm_gan_logits")
    _cmp(sub(x.grad)[0], "cm_gan_grad")
    _cmp(sub(bank)[0], "cm_gan_bank")
    _cmp(sub(gbank)[0], "cm_gan_gbank")


def test_aegenerator():
    net, x = C.aegen_case()
    s, st = sub(net(x))
    _cmp(s, "aegen_fwd")
    _cmp(st, "aegen_fwd_stats")
    s, st = sub(net.forward_enc(x))
    _cmp(s, "aegen_enc")
    _cmp(st, "aegen_enc_stats")


def test_dec_generators():
    for tag, case in (("decgen1", C.decgen1_case), ("decgen", C.decgen_case)):
        net, feat = case()
        feat = feat.clone().requires_grad_(True)
        y = net(feat)
        s, st = sub(y)
        _cmp(s, tag + "_fwd")
        _cmp(st, tag + "_fwd_stats")
        g = torch.Generator().manual_seed(7)
        y.backward(torch.randn(y.shape, generator=g))
        _cmp(sub(feat.grad)[0], tag + "_dfeat", 1e-4)
        params = dict(net.named_parameters())
        pre = tag + "_g_"
        for key in GOLD.files:
            if key.startswith(pre):
                _cmp(sub(params[key[len(pre):]].grad)[0], key, 2e-4)


def test_resize_reid_adaptor():
    net, x = C.resize_reid_case()
    x = x.clone().requires_grad_(True)
    y = net(x)
    assert tuple(y.shape) == (2, 3, 256, 128)
    s, st = sub(y)
    _cmp(s, "resize_reid_fwd")
    _cmp(st, "resize_reid_fwd_stats")
    g = torch.Generator().manual_seed(8)
    y.backward(torch.randn(y.shape, generator=g))
    _cmp(sub(x.grad)[0], "resize_reid_dx", 1e-4)
    params = dict(net.named_parameters())
    for key in GOLD.files:
        if key.startswith("resize_reid_g_"):
            _cmp(sub(params[key[len("resize_reid_g_"):]].grad)[0], key, 2e-4)


def test_fd_generator():
    net, (feat, noise) = C.fdgen_case()
    feat, noise = feat.clone().requires_grad_(True), noise.clone().requires_grad_(True)
    y = net(feat, noise)
    assert tuple(y.shape) == (3, 3, 256, 128)
    s, st = sub(y)
    _cmp(s, "fdgen_fwd")
    _cmp(st, "fdgen_fwd_stats")
    g = torch.Generator().manual_seed(9)
    y.backward(torch.randn(y.shape, generator=g))
    _cmp(sub(feat.grad)[0], "fdgen_dfeat", 1e-4)
    _cmp(sub(noise.grad)[0], "fdgen_dnoise", 1e-4)
    params = dict(net.named_parameters())
    for key in GOLD.files:
        if key.startswith("fdgen_g_"):
            _cmp(sub(params[key[len("fdgen_g_"):]].grad)[0], key, 2e-4)


def test_dptn_generator():
    net, (xs, ps, pt) = C.dptn_case()
    t, s_ = net(xs, ps, pt)
    _cmp(sub(t)[0], "dptn_fwd_t")
    _cmp(sub(s_)[0], "dptn_fwd_s")
    g = torch.Generator().manual_seed(6)
    dt, ds = torch.randn(t.shape, generator=g), torch.randn(s_.shape, generator=g)
    ((t * dt).sum() + (s_ * ds).sum()).backward()
    params = dict(net.named_parameters())
    for key in GOLD.files:
        if key.startswith("dptn_g_"):
            _cmp(sub(params[key[len("dptn_g_"):]].grad)[0], key, 2e-4)
