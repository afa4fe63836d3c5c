"""Seeded oracle-side construction of every golden case (shared by make_golden.py, which also runs the
reference on the same weights and inputs, and by tests/test_oracle_golden.py, which only has the fixtures)."""
from __future__ import absolute_import

import torch
import torch.nn as nn
import torch.nn.functional as F

from oracle import ref_torch as O


def generator_case(connect_layers):
    torch.manual_seed(10 + connect_layers)
    og = O.OPoseGenerator(128, 2048, 256, dropout=0.0, norm='batch', connect_layers=connect_layers)
    og.apply(O.o_weights_init_normal)
    og.train()
    pose = O.synth_posemaps(2, seed=3)
    g = torch.Generator().manual_seed(4)
    feat = torch.randn(2, 2048, 1, 1, generator=g).abs()
    z = torch.randn(2, 256, 1, 1, generator=g)
    return og, (pose, feat, z)


def discriminator_case(norm):
    torch.manual_seed(20 if norm == 'batch' else 21)
    od = O.OPatchDiscriminator(21, norm)
    od.apply(O.o_weights_init_normal)
    x = torch.cat((O.synth_posemaps(2, seed=5), O.synth_images(2, seed=6)), 1)
    return od, x


def ganloss_case():
    g = torch.Generator().manual_seed(7)
    return torch.randn(4, 1, 30, 14, generator=g) * 3


def embed_case():
    torch.manual_seed(30)
    oe = O.OEltwiseSubEmbed(use_batch_norm=True, use_classifier=True, num_features=2048, num_classes=2)
    oe.classifier.weight.data.normal_(0, 0.05)
    oe.bn.running_mean.normal_(0.5, 0.1)
    oe.bn.running_var.uniform_(0.5, 1.5)
    g = torch.Generator().manual_seed(31)
    f1, f2 = torch.randn(6, 2048, generator=g), torch.randn(6, 2048, generator=g)
    return oe, (f1, f2)


def trunk_case():
    torch.manual_seed(40)
    ot = O.OTVResNet(50)
    O._reid_reset_params(ot)
    for m in ot.modules():                    # non-trivial BN statistics so eval mode is a real test
        if isinstance(m, nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.1)
            m.running_var.uniform_(0.8, 1.2)
            m.weight.data.uniform_(0.4, 0.6)
            m.bias.data.normal_(0, 0.1)
    oreid = O.OReidResNet(50, cut_at_pooling=True)
    oreid.base.load_state_dict(ot.state_dict())
    return oreid, ot.state_dict(), O.synth_images(2, 128, 64, seed=8)


def gem_case():
    g = torch.Generator().manual_seed(50)
    return O.OGeM(), torch.randn(3, 16, 16, 8, generator=g), torch.randn(3, 16, 1, 1, generator=g)


def cm_case():
    g = torch.Generator().manual_seed(60)
    K, D, B = 40, 256, 24
    bank = F.normalize(torch.randn(K, D, generator=g), dim=1)
    feats = F.normalize(torch.randn(B, D, generator=g), dim=1)
    labels = torch.tensor([3, 3, 7, 3, 9, 7, 7, 3] * 3)
    gout = torch.randn(B, K, generator=g)
    return bank, feats, labels, gout


def sub(t, n=512):
    """deterministic sub-sample of a tensor (flattened, fixed stride) + global statistics."""
    import numpy as np
    f = t.detach().reshape(-1).double()
    step = max(1, f.numel() // n)
    return f[::step][:n].numpy().astype(np.float64), np.array([f.mean().item(), f.abs().mean().item(), f.numel()])
