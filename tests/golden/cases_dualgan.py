"""Seeded oracle-side construction of the dual_gan golden cases (shared by make_golden_dualgan.py, which also runs the
reference on the same weights and inputs, and by the tests, which only have the fixtures)."""
from __future__ import absolute_import

import torch

from oracle import ref_dualgan as D


def _perturb(net, seed, scale=0.05):
    """Orthogonal(0.02) init leaves every activation tiny; add seeded noise so norms, biases and affine terms matter."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in net.named_parameters():
            if p.dim() > 1:
                p.add_(torch.randn(p.shape, generator=g) * scale / max(1.0, (p[0].numel()) ** 0.5) * 4)
            else:
                p.add_(torch.randn(p.shape, generator=g) * scale)
    return net


def posegen1_case():
    torch.manual_seed(70)
    net = D.o_init_weights(D.OPoseGenerator1(ngf=64, pose_nc=18, img_f=256, layers=3, norm='instance'))
    _perturb(net, 71)
    net.train()
    inp = D.synth_dualgan_inputs(2, 64, 32, seed=72)
    g = torch.Generator().manual_seed(73)
    feat = torch.nn.functional.normalize(torch.randn(2, 2048, 8, 4, generator=g).abs(), dim=1)
    return net, (feat, inp['Ps'])


def pctm_case():
    torch.manual_seed(80)
    net = D.OPCTM(64, 2, 2, 2, 64)
    _perturb(net, 81)
    g = torch.Generator().manual_seed(82)
    return net, (torch.randn(2, 64, 4, 3, generator=g), torch.randn(2, 64, 5, 2, generator=g))


def resdisc_case():
    torch.manual_seed(90)
    net = D.o_init_weights(D.OResDiscriminator(3, 32, 128, 3, True))
    _perturb(net, 91, 0.02)
    net.train()
    x = D.synth_dualgan_inputs(2, 64, 32, seed=92)['Xs']
    return net, x


def lsgan_case():
    g = torch.Generator().manual_seed(95)
    return torch.randn(3, 1, 16, 8, generator=g) * 2


def bicubic_case():
    g = torch.Generator().manual_seed(96)
    return torch.rand(2, 3, 32, 16, generator=g)


def aegen_case():
    torch.manual_seed(75)
    net = D.o_init_weights(D.OAEGenerator(3, 64, 256, 3, 'instance', 3, 3))
    _perturb(net, 76)
    net.train()
    return net, D.synth_dualgan_inputs(2, 64, 32, seed=77)['Xs']


def dptn_case():
    torch.manual_seed(85)
    net = D.o_init_weights(D.ODPTNGenerator(3, 18, 64, 256, 3, 'instance', 3, 3, 2, 2, 2))
    _perturb(net, 86)
    net.train()
    a = D.synth_dualgan_inputs(2, 64, 32, seed=87)
    b = D.synth_dualgan_inputs(2, 64, 32, seed=88)
    return net, (a['Xs'], a['Ps'], b['Ps'])


def decgen1_case():
    """`--model_gen DEC`: ReID feature map in, image out"""
    torch.manual_seed(60)
    net = D.o_init_weights(D.ODECGenerator1(ngf=64, img_f=256, layers=3, norm='instance', output_nc=3, num_blocks=3))
    _perturb(net, 61)
    net.train()
    g = torch.Generator().manual_seed(62)
    feat = torch.nn.functional.normalize(torch.randn(2, 2048, 8, 4, generator=g).abs(), dim=1)
    return net, feat


def decgen_case():
    torch.manual_seed(64)
    net = D.o_init_weights(D.ODECGenerator(ngf=64, img_f=2048, layers=3, norm='instance', output_nc=3))
    _perturb(net, 65)
    net.train()
    g = torch.Generator().manual_seed(66)
    feat = torch.nn.functional.normalize(torch.randn(2, 2048, 8, 4, generator=g).abs(), dim=1)
    return net, feat


def resize_reid_case():
    """--use_adp: the adaptor from synthesised 128x64 images to ReID inputs"""
    torch.manual_seed(50)
    net = D.o_init_weights(D.OResize_ReID(3, 64))
    _perturb(net, 51, 0.02)
    net.train()
    g = torch.Generator().manual_seed(52)
    return net, torch.tanh(torch.randn(2, 3, 32, 16, generator=g))


def fdgen_case():
    """`--model_gen FD`: define_G's FDGenerator(img_f, ngf, output_nc=3, noise_nc=512, fuse_mode='add') at img_f = 256"""
    torch.manual_seed(54)
    net = D.o_init_weights(D.OFDGenerator(256, 64, noise_nc=512, fuse_mode='add'))
    _perturb(net, 55, 0.02)
    net.train()
    g = torch.Generator().manual_seed(56)
    return net, (torch.randn(3, 256, 1, 1, generator=g), torch.randn(3, 512, generator=g))
