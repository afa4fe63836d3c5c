"""Seeded oracle-side construction of the DPTNModel golden cases (shared by make_golden_dptn.py, which also runs the
reference's own DPTNModel / GANLoss / cal_gradient_penalty / VGGLoss on the same weights and inputs, and by the tests,
which only have the fixtures)."""
from __future__ import absolute_import

import torch

from oracle import ref_dualgan as D
from tests.golden.cases_dualgan import _perturb

GAN_MODES = ('lsgan', 'vanilla', 'hinge', 'wgangp')
H, W, B = 64, 32, 2
LAMBDAS = dict(lambda_rec=2.0, lambda_g=5.0, lambda_style=500.0, lambda_content=0.5, t_s_ratio=0.8)


def ganloss_case():
    g = torch.Generator().manual_seed(195)
    return torch.randn(3, 1, 8, 4, generator=g) * 1.5


def nets(dis_layers=3):
    """DPTNModel's networks as its constructor builds them (DPTN_model.py:51-53,63) with seeded, perturbed weights."""
    torch.manual_seed(185)
    net_G = D.o_init_weights(D.ODPTNGenerator(3, 18, 64, 512, 3, 'instance', 3, 3, 2, 2, 2))
    _perturb(net_G, 186)
    net_D = D.o_init_weights(D.OResDiscriminator(3, 32, 128, dis_layers, True))
    _perturb(net_D, 187, 0.02)
    net_G.train()
    net_D.train()
    return net_G, net_D


def vgg_features():
    torch.manual_seed(188)
    return D.o_seed_vgg(D.o_tv_vgg19_features(), 189)


def inputs():
    return D.synth_dptn_inputs(B, H, W, seed=190)


def gp_alpha(step):
    g = torch.Generator().manual_seed(300 + step)
    return torch.rand(B, 1, generator=g)


def vgg_pair():
    g = torch.Generator().manual_seed(196)
    return torch.rand(2, 3, 32, 16, generator=g) * 2 - 1, torch.rand(2, 3, 32, 16, generator=g) * 2 - 1


def model(gan_mode, with_vgg):
    net_G, net_D = nets()
    vgg = D.OVGGLoss(D.OVGG19(vgg_features())) if with_vgg else None
    return D.ODPTNModel(net_G, net_D, gan_mode=gan_mode, vgg=vgg, **LAMBDAS)


PROBES_G = ["block0.model.0.weight", "mblock1.conv2.weight", "PTM.decoder.layers.0.multihead_attn.in_proj_weight",
            "source_encoder.encoder1.model.2.weight", "decoder1.model.2.weight", "outconv.conv1.weight"]
PROBES_D = ["block0.model.0.weight_orig", "encoder1.model.3.weight_orig", "encoder0.shortcut.1.weight_orig", "conv.weight_orig"]
