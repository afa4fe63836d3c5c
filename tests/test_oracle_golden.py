"""CPU: the oracle restatement (oracle/ref_torch.py) against the golden vectors that tests/golden/make_golden.py
recorded from the reference's own modules.  No reference and no GPU needed."""
import os

import numpy as np
import pytest
import torch

from oracle import ref_torch as O
from tests.golden import cases as C
from tests.golden.cases import sub

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_modules.npz"))


def _cmp(got, key, tol=2e-5):
    ref = GOLD[key]
    got = np.asarray(got, dtype=np.float64).reshape(ref.shape)
    scale = max(np.abs(ref).max(), 1e-12)
    err = np.abs(got - ref).max()
    assert err <= tol * scale, "%s: %.3e vs scale %.3e" % (key, err, scale)


@pytest.mark.parametrize("cl", [0, 2])
def test_generator(cl):
    og, (pose, feat, z) = C.generator_case(cl)
    y = og(pose, feat, z)
    s, st = sub(y)
    _cmp(s, "g_fwd_cl%d" % cl)
    _cmp(st, "g_fwd_cl%d_stats" % cl)


@pytest.mark.parametrize("norm,key", [("batch", "dp_fwd"), ("instance", "dp_in_fwd")])
def test_discriminator(norm, key):
    od, x = C.discriminator_case(norm)
    s, st = sub(od(x))
    _cmp(s, key)
    _cmp(st, key + "_stats")


def test_ganloss():
    pred = C.ganloss_case()
    _cmp([O.o_gan_loss(pred, True).item(), O.o_gan_loss(pred, False).item()], "ganloss")


def test_embed():
    oe, (f1, f2) = C.embed_case()
    for mode in ("train", "eval"):          # eval runs on the running statistics left by the train-mode call
        getattr(oe, mode)()
        _cmp(oe(f1, f2).detach().numpy(), "embed_" + mode)


def test_resnet50_trunk():
    oreid, trunk_sd, imgs = C.trunk_case()
    for mode in ("eval", "train"):
        getattr(oreid, mode)()
        s, st = sub(oreid(imgs.clone()))
        _cmp(s, "resnet50_trunk_%s" % mode, tol=1e-4)
        _cmp(st, "resnet50_trunk_%s_stats" % mode, tol=1e-4)
        oreid.base.load_state_dict(trunk_sd)


def test_gem():
    og, xg, wsum = C.gem_case()
    x = xg.clone().requires_grad_(True)
    y = og(x)
    (y * wsum).sum().backward()
    _cmp(y.detach().flatten().numpy(), "gem_fwd")
    _cmp(og.p.grad.numpy(), "gem_dp", tol=1e-4)
    _cmp(sub(x.grad)[0], "gem_dx")


@pytest.mark.parametrize("name,fn", [("cm", O.OCM), ("cm_hard", O.OCMHard)])
def test_cluster_memory_functions(name, fn):
    bank, feats, labels, gout = C.cm_case()
    b = bank.clone()
    x = feats.clone().requires_grad_(True)
    y = fn.apply(x, labels, b, torch.Tensor([0.2]))
    y.backward(gout)
    _cmp(sub(y)[0], name + "_logits")
    _cmp(sub(x.grad)[0], name + "_grad")
    s, st = sub(b, 2048)
    _cmp(s, name + "_bank")
    _cmp(st, name + "_bank_stats")
    assert not torch.equal(b, bank)          # the bank really was updated in place


def test_cluster_memory_module_matches_functions():
    """ClusterMemory.forward (cm.py:123-137) is normalize -> cm -> /temp -> CE(reduction='none')."""
    bank, feats, labels, _ = C.cm_case()
    mem = O.OClusterMemory(256, 40, temp=0.05, momentum=0.2)
    mem.features = bank.clone()
    x = (feats * 3.0).requires_grad_(True)
    loss = mem(x, labels)
    assert loss.shape == (24,)
    ref = torch.nn.functional.cross_entropy(torch.nn.functional.normalize(x, dim=1).mm(bank.t()) / 0.05, labels,
                                            reduction="none")
    assert torch.allclose(loss, ref, rtol=1e-6, atol=1e-6)
    loss.mean().backward()
    assert not torch.equal(mem.features, bank)


def test_dropout_mask_restatement_is_deterministic():
    m1 = O.dropout_keep_mask(1 << 16, 0.2, 12345)
    m2 = O.dropout_keep_mask(1 << 16, 0.2, 12345)
    assert torch.equal(m1, m2)
    assert abs(m1.float().mean().item() - 0.8) < 0.01
