"""dual_gan modules on the MI355X (HIP tape programs of reid-gan_amd/dual_gan) against the oracle restatement
(oracle/ref_dualgan.py, pinned on the reference by tests/golden/reference_dualgan.npz) on identical seeded weights and
inputs.  Forward 1e-3 max-norm (measured ~1e-5); gradients by the flip-robust L2 metrics of test_modules_gpu.py."""
import argparse
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


def test_state_dict_keys_match_reference_layout():
    """the oracle's keys ARE the reference's (make_golden_dualgan loads one into the other)"""
    from dual_gan.models import networks as N
    from tests.golden import cases_dualgan as C
    og, _ = C.posegen1_case()
    rg = N.PoseGenerator1(64, 18, 256, 3, 'instance', 'LeakyReLU', False, False, 3, True, 2, 2, 2)
    assert list(rg.state_dict().keys()) == list(og.state_dict().keys())
    od, _ = C.resdisc_case()
    rd = N.ResDiscriminator(3, 32, 128, 3, 'none', 'LeakyReLU', True)
    assert sorted(rd.state_dict().keys()) == sorted(od.state_dict().keys())
    for k, v in od.state_dict().items():
        assert rd.state_dict()[k].shape == v.shape, k


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
