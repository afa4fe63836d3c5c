"""ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product; nothing under reid-gan_amd/ imports it.

CPU restatement (plain torch fp32 ops) of the dual_gan side of the joint training step: the `Pose` generator
(PoseGenerator1 with its Pose Transformer Module), the spectral-normed ResDiscriminator, the lsgan loss, the AEModel
loss/update logic, the bicubic `my_transform`, and the joint ReID + GAN step of the trainer.

Pinning: tests/golden/make_golden_dualgan.py imports the reference's own `dual_gan/models/{base_function,PTM,networks,
external_function}.py` in the build container, loads THIS file's seeded weights into them and stores their outputs in
tests/golden/reference_dualgan.npz; tests/test_oracle_golden.py re-checks this file against the fixtures without the
reference.  `AEModel`, the trainer loop and `diff_augs` are `.cuda()` / torchvision bound in the reference and are restated
from the cited lines (CC/ = /root/reference/cluster-contrast-reid-main/).
"""
from __future__ import absolute_import

import copy

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn.utils import spectral_norm as _torch_spectral_norm


def _norm2d(norm, c):
    """CC/dual_gan/models/base_function.py:38-48"""
    if norm == 'instance':
        return nn.InstanceNorm2d(c, affine=True)
    if norm == 'batch':
        return nn.BatchNorm2d(c, momentum=0.1, affine=True)
    raise NotImplementedError(norm)


def _act():
    """base_function.py:51-63, 'LeakyReLU' -> negative slope 0.1, not in place"""
    return nn.LeakyReLU(0.1)


def _sn(m, use):
    return _torch_spectral_norm(m) if use else m


# ---- generator blocks, base_function.py:236-339, 423-443 ------------------------------------------------------
class OEncoderBlockOptimized(nn.Module):
    def __init__(self, cin, cout, norm):
        super(OEncoderBlockOptimized, self).__init__()
        self.model = nn.Sequential(nn.Conv2d(cin, cout, 4, 2, 1), _norm2d(norm, cout), _act(), nn.Conv2d(cout, cout, 3, 1, 1))

    def forward(self, x):
        return self.model(x)


class OEncoderBlock(nn.Module):
    def __init__(self, cin, cout, norm):
        super(OEncoderBlock, self).__init__()
        self.model = nn.Sequential(_norm2d(norm, cin), _act(), nn.Conv2d(cin, cout, 4, 2, 1),
                                   _norm2d(norm, cout), _act(), nn.Conv2d(cout, cout, 3, 1, 1))

    def forward(self, x):
        return self.model(x)


class OFeatureAdaptBlock1(nn.Module):
    def __init__(self, cin, cout, norm):
        super(OFeatureAdaptBlock1, self).__init__()
        self.model = nn.Sequential(nn.Conv2d(cin, cout, 1), _norm2d(norm, cout), _act())

    def forward(self, x):
        return self.model(x)


class OResBlockDecoder(nn.Module):
    def __init__(self, cin, cout, hidden, norm):
        super(OResBlockDecoder, self).__init__()
        self.model = nn.Sequential(_norm2d(norm, cin), _act(), nn.Conv2d(cin, hidden, 3, 1, 1), _norm2d(norm, hidden), _act(),
                                   nn.ConvTranspose2d(hidden, cout, 3, 2, 1, output_padding=1))
        self.shortcut = nn.Sequential(nn.ConvTranspose2d(cin, cout, 3, 2, 1, output_padding=1))

    def forward(self, x):
        return self.model(x) + self.shortcut(x)


class OOutput(nn.Module):
    def __init__(self, cin, cout, k=3):
        super(OOutput, self).__init__()
        self.conv1 = nn.Conv2d(cin, cout, k, padding=0, bias=True)
        self.model = nn.Sequential(_act(), nn.ReflectionPad2d(k // 2), self.conv1, nn.Tanh())

    def forward(self, x):
        return self.model(x)


class OResBlock(nn.Module):
    """base_function.py:193-233, sample_type 'none'"""

    def __init__(self, cin, cout, hidden, norm):
        super(OResBlock, self).__init__()
        self.conv1 = nn.Conv2d(cin, hidden, 3, 1, 1)
        self.conv2 = nn.Conv2d(hidden, cout, 3, 1, 1)
        self.bypass = nn.Conv2d(cin, cout, 1)
        self.model = nn.Sequential(_norm2d(norm, cin), _act(), self.conv1, _norm2d(norm, hidden), _act(), self.conv2)
        self.shortcut = nn.Sequential(self.bypass)

    def forward(self, x):
        return self.model(x) + self.shortcut(x)


class OAEGenerator(nn.Module):
    """networks.py:278-355"""

    def __init__(self, image_nc=3, ngf=64, img_f=256, layers=3, norm='instance', output_nc=3, num_blocks=3):
        super(OAEGenerator, self).__init__()
        self.layers, self.num_blocks = layers, num_blocks
        self.block0 = OEncoderBlockOptimized(image_nc, ngf, norm)
        mult = 1
        for i in range(layers - 1):
            prev, mult = mult, min(2 ** (i + 1), img_f // ngf)
            setattr(self, 'encoder%d' % i, OEncoderBlock(ngf * prev, ngf * mult, norm))
        for i in range(num_blocks):
            setattr(self, 'mblock%d' % i, OResBlock(ngf * mult, ngf * mult, ngf * mult, norm))
        for i in range(layers):
            prev = mult
            mult = min(2 ** (layers - i - 2), img_f // ngf) if i != layers - 1 else 1
            setattr(self, 'decoder%d' % i, OResBlockDecoder(ngf * prev, ngf * mult, ngf * mult, norm))
        self.outconv = OOutput(ngf, output_nc, 3)

    def forward_enc(self, x):
        x = self.block0(x)
        for i in range(self.layers - 1):
            x = getattr(self, 'encoder%d' % i)(x)
        return x

    def forward_dec(self, f):
        for i in range(self.num_blocks):
            f = getattr(self, 'mblock%d' % i)(f)
        for i in range(self.layers):
            f = getattr(self, 'decoder%d' % i)(f)
        return self.outconv(f)

    def forward(self, x):
        return self.forward_dec(self.forward_enc(x))


class ODECGenerator1(nn.Module):
    """networks.py:401-444 (`--model_gen DEC`): FeatureAdaptBlock1 -> ResBlocks -> ResBlockDecoders -> Output"""

    def __init__(self, ngf=64, img_f=256, layers=3, norm='instance', output_nc=3, num_blocks=3, feat_nc=2048):
        super(ODECGenerator1, self).__init__()
        self.layers, self.num_blocks = layers, num_blocks
        mult = 4
        self.feature_block = OFeatureAdaptBlock1(feat_nc, ngf * mult, norm)
        for i in range(num_blocks):
            setattr(self, 'mblock%d' % i, OResBlock(ngf * mult, ngf * mult, ngf * mult, norm))
        for i in range(layers):
            prev = mult
            mult = min(2 ** (layers - i - 2), img_f // ngf) if i != layers - 1 else 1
            setattr(self, 'decoder%d' % i, OResBlockDecoder(ngf * prev, ngf * mult, ngf * mult, norm))
        self.outconv = OOutput(ngf, output_nc, 3)

    def forward(self, f):
        f = self.feature_block(f)
        for i in range(self.num_blocks):
            f = getattr(self, 'mblock%d' % i)(f)
        for i in range(self.layers):
            f = getattr(self, 'decoder%d' % i)(f)
        return self.outconv(f)


class ODECGenerator(nn.Module):
    """networks.py:356-398: ResBlock(img_f -> 4 ngf) -> ResBlockDecoders -> Output"""

    def __init__(self, ngf=64, img_f=2048, layers=3, norm='instance', output_nc=3):
        super(ODECGenerator, self).__init__()
        self.layers = layers
        mult = 4
        self.resblock = OResBlock(img_f, ngf * mult, ngf * mult, norm)
        for i in range(layers):
            prev = mult
            mult = min(2 ** (layers - i - 2), img_f // ngf) if i != layers - 1 else 1
            setattr(self, 'decoder%d' % i, OResBlockDecoder(ngf * prev, ngf * mult, ngf * mult, norm))
        self.outconv = OOutput(ngf, output_nc, 3)

    def forward(self, f):
        f = self.resblock(f)
        for i in range(self.layers):
            f = getattr(self, 'decoder%d' % i)(f)
        return self.outconv(f)


def o_hard_mix(F_s, reid_f, group_size, lambda_fus):
    """AE_model.py:274-292"""
    fdim = reid_f.shape[1]
    anchor = F.normalize(torch.mean(reid_f.reshape(-1, group_size, fdim), dim=1))
    inst = F.normalize(reid_f)
    sim = torch.exp(torch.einsum('n c, m c -> n m', anchor, inst))
    id_mask = torch.eye(anchor.shape[0]).repeat_interleave(group_size, dim=1)
    in_id = torch.argmin(id_mask * sim + (1 - id_mask) * torch.max(sim), dim=1)
    out_id = torch.argmax((1 - id_mask) * sim, dim=1)
    return lambda_fus * F_s[in_id] + (1 - lambda_fus) * F_s[out_id]


# ---- Pose Transformer Module, PTM.py:6-58, 115-247 ----------------------------------------------------------------
class OCAB(nn.Module):
    def __init__(self, d, nhead, ff):
        super(OCAB, self).__init__()
        self.self_attn = nn.MultiheadAttention(d, nhead)
        self.linear1 = nn.Linear(d, ff)
        self.linear2 = nn.Linear(ff, d)
        self.norm1 = nn.InstanceNorm1d(d, affine=True)
        self.norm2 = nn.InstanceNorm1d(d, affine=True)
        self.activation = _act()

    def forward(self, src):                                  # [L, B, C]
        src = src + self.self_attn(src, src, value=src)[0]
        src = self.norm1(src.permute(1, 2, 0)).permute(2, 0, 1)
        src = src + self.linear2(self.activation(self.linear1(src)))
        return self.norm2(src.permute(1, 2, 0)).permute(2, 0, 1)


class OTTB(nn.Module):
    def __init__(self, d, nhead, ff):
        super(OTTB, self).__init__()
        self.self_attn = nn.MultiheadAttention(d, nhead)
        self.multihead_attn = nn.MultiheadAttention(d, nhead)
        self.linear1 = nn.Linear(d, ff)
        self.linear2 = nn.Linear(ff, d)
        self.norm1 = nn.InstanceNorm1d(d, affine=True)
        self.norm2 = nn.InstanceNorm1d(d, affine=True)
        self.norm3 = nn.InstanceNorm1d(d, affine=True)
        self.activation = _act()

    def forward(self, tgt, memory, val):
        tgt = tgt + self.self_attn(tgt, tgt, value=tgt)[0]
        tgt = self.norm1(tgt.permute(1, 2, 0)).permute(2, 0, 1)
        tgt = tgt + self.multihead_attn(query=tgt, key=memory, value=val)[0]
        tgt = self.norm2(tgt.permute(1, 2, 0)).permute(2, 0, 1)
        tgt = tgt + self.linear2(self.activation(self.linear1(tgt)))
        return self.norm3(tgt.permute(1, 2, 0)).permute(2, 0, 1)


class _Stack(nn.Module):
    def __init__(self, layer, n, norm=None):
        super(_Stack, self).__init__()
        self.layers = nn.ModuleList([copy.deepcopy(layer) for _ in range(n)])
        self.norm = norm


class OPCTM(nn.Module):
    def __init__(self, d, nhead, n_cab, n_ttb, ff):
        super(OPCTM, self).__init__()
        self.encoder = _Stack(OCAB(d, nhead, ff), n_cab, None)
        self.decoder = _Stack(OTTB(d, nhead, ff), n_ttb, nn.InstanceNorm1d(d, affine=True))
        for p in self.parameters():                       # PTM.py:43-46
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)

    def forward(self, query, value):
        bs, c, h, w = query.shape
        q = query.flatten(2).permute(2, 0, 1)
        v = value.flatten(2).permute(2, 0, 1)
        for layer in self.encoder.layers:
            v = layer(v)
        for layer in self.decoder.layers:
            q = layer(q, v, v)
        hs = self.decoder.norm(q.permute(1, 2, 0))          # stays [B, C, L] (PTM.py:158-160)
        return hs.reshape(bs, c, h, w)


class OPTM(nn.Module):
    """PTM, PTM.py:60-112: src -> CABs = memory; tgt attends with key = memory, value = val"""

    def __init__(self, d, nhead, n_cab, n_ttb, ff):
        super(OPTM, self).__init__()
        self.encoder = _Stack(OCAB(d, nhead, ff), n_cab, None)
        self.decoder = _Stack(OTTB(d, nhead, ff), n_ttb, nn.InstanceNorm1d(d, affine=True))
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)

    def forward(self, src, tgt, val):
        bs, c, h, w = src.shape
        m = src.flatten(2).permute(2, 0, 1)
        t = tgt.flatten(2).permute(2, 0, 1)
        v = val.flatten(2).permute(2, 0, 1)
        for layer in self.encoder.layers:
            m = layer(m)
        for layer in self.decoder.layers:
            t = layer(t, m, v)
        return self.decoder.norm(t.permute(1, 2, 0)).reshape(bs, c, h, w)


class OSourceEncoder(nn.Module):
    def __init__(self, image_nc, ngf, img_f, layers, norm):
        super(OSourceEncoder, self).__init__()
        self.encoder_layer = layers
        self.block0 = OEncoderBlockOptimized(image_nc, ngf, norm)
        mult = 1
        for i in range(layers - 1):
            prev, mult = mult, min(2 ** (i + 1), img_f // ngf)
            setattr(self, 'encoder%d' % i, OEncoderBlock(ngf * prev, ngf * mult, norm))

    def forward(self, x):
        x = self.block0(x)
        for i in range(self.encoder_layer - 1):
            x = getattr(self, 'encoder%d' % i)(x)
        return x


class ODPTNGenerator(nn.Module):
    """networks.py:165-275"""

    def __init__(self, image_nc=3, pose_nc=18, ngf=64, img_f=256, layers=3, norm='instance', output_nc=3, num_blocks=3,
                 nhead=2, num_CABs=2, num_TTBs=2):
        super(ODPTNGenerator, self).__init__()
        self.layers, self.num_blocks = layers, num_blocks
        self.block0 = OEncoderBlockOptimized(2 * pose_nc + image_nc, ngf, norm)
        mult = 1
        for i in range(layers - 1):
            prev, mult = mult, min(2 ** (i + 1), img_f // ngf)
            setattr(self, 'encoder%d' % i, OEncoderBlock(ngf * prev, ngf * mult, norm))
        for i in range(num_blocks):
            setattr(self, 'mblock%d' % i, OResBlock(ngf * mult, ngf * mult, ngf * mult, norm))
        self.PTM = OPTM(ngf * mult, nhead, num_CABs, num_TTBs, ngf * mult)
        self.source_encoder = OSourceEncoder(image_nc, ngf, img_f, layers, norm)
        for i in range(layers):
            prev = mult
            mult = min(2 ** (layers - i - 2), img_f // ngf) if i != layers - 1 else 1
            setattr(self, 'decoder%d' % i, OResBlockDecoder(ngf * prev, ngf * mult, ngf * mult, norm))
        self.outconv = OOutput(ngf, output_nc, 3)

    def _encode(self, x):
        x = self.block0(x)
        for i in range(self.layers - 1):
            x = getattr(self, 'encoder%d' % i)(x)
        for i in range(self.num_blocks):
            x = getattr(self, 'mblock%d' % i)(x)
        return x

    def _decode(self, f):
        for i in range(self.layers):
            f = getattr(self, 'decoder%d' % i)(f)
        return self.outconv(f)

    def forward(self, source, source_B, target_B, is_train=True):
        F_s_s = self._encode(torch.cat((source, source_B, source_B), 1))
        F_s_t = self._encode(torch.cat((source, source_B, target_B), 1))
        F_s_t = self.PTM(F_s_s, F_s_t, self.source_encoder(source))
        out_s = self._decode(F_s_s) if is_train else None
        return self._decode(F_s_t), out_s


# ---- PoseGenerator1, networks.py:639-738 ------------------------------------------------------------------
class OPoseGenerator1(nn.Module):
    def __init__(self, ngf=64, pose_nc=18, img_f=256, layers=3, norm='instance', output_nc=3, nhead=2, num_CABs=2,
                 num_TTBs=2, feat_nc=2048):
        super(OPoseGenerator1, self).__init__()
        self.layers = layers
        self.block0 = OEncoderBlockOptimized(pose_nc, ngf, norm)
        mult = 1
        for i in range(layers - 1):
            prev, mult = mult, min(2 ** (i + 1), img_f // ngf)
            setattr(self, 'encoder%d' % i, OEncoderBlock(ngf * prev, ngf * mult, norm))
        self.feature_block = OFeatureAdaptBlock1(feat_nc, ngf * mult, norm)
        self.PCTM = OPCTM(ngf * mult, nhead, num_CABs, num_TTBs, ngf * mult)
        for i in range(layers):
            prev = mult
            mult = min(2 ** (layers - i - 2), img_f // ngf) if i != layers - 1 else 1
            setattr(self, 'decoder%d' % i, OResBlockDecoder(ngf * prev, ngf * mult, ngf * mult, norm))
        self.outconv = OOutput(ngf, output_nc, 3)

    def forward(self, reid_f, source_pose):
        F_p = self.block0(source_pose)
        skips = []
        for i in range(self.layers - 1):
            skips.append(F_p)
            F_p = getattr(self, 'encoder%d' % i)(F_p)
        F_g = self.PCTM(F_p, self.feature_block(reid_f))
        for i in range(self.layers):
            F_g = getattr(self, 'decoder%d' % i)(F_g)
            if i < self.layers - 1:
                F_g = F_g + skips.pop()
        return self.outconv(F_g)


# ---- ResDiscriminator, networks.py:917-955; blocks base_function.py:372-420 ------------------------------------
class OResBlockEncoderOptimized(nn.Module):
    def __init__(self, cin, cout, hidden, sn):
        super(OResBlockEncoderOptimized, self).__init__()
        self.model = nn.Sequential(_sn(nn.Conv2d(cin, hidden, 3, 1, 1), sn), _act(), _sn(nn.Conv2d(hidden, cout, 4, 2, 1), sn))
        self.shortcut = nn.Sequential(nn.AvgPool2d(2, 2), _sn(nn.Conv2d(cin, cout, 1), sn))

    def forward(self, x):
        return self.model(x) + self.shortcut(x)


class OResBlockEncoder(nn.Module):
    def __init__(self, cin, cout, hidden, sn):
        super(OResBlockEncoder, self).__init__()
        self.model = nn.Sequential(_act(), _sn(nn.Conv2d(cin, hidden, 3, 1, 1), sn), _act(),
                                   _sn(nn.Conv2d(hidden, cout, 4, 2, 1), sn))
        self.shortcut = nn.Sequential(nn.AvgPool2d(2, 2), _sn(nn.Conv2d(cin, cout, 1), sn))

    def forward(self, x):
        return self.model(x) + self.shortcut(x)


class OResDiscriminator(nn.Module):
    def __init__(self, input_nc=3, ndf=32, img_f=128, layers=3, use_spect=True):
        super(OResDiscriminator, self).__init__()
        self.layers = layers
        self.nonlinearity = _act()
        self.block0 = OResBlockEncoderOptimized(input_nc, ndf, ndf, use_spect)
        mult = 1
        for i in range(layers - 1):
            prev, mult = mult, min(2 ** (i + 1), img_f // ndf)
            setattr(self, 'encoder%d' % i, OResBlockEncoder(ndf * prev, ndf * mult, ndf * prev, use_spect))
        self.conv = _torch_spectral_norm(nn.Conv2d(ndf * mult, 1, 1))

    def forward(self, x):
        out = self.block0(x)
        for i in range(self.layers - 1):
            out = getattr(self, 'encoder%d' % i)(out)
        return self.conv(self.nonlinearity(out))


def o_init_weights(net, gain=0.02):
    """base_function.py:13-35 with init_type='orthogonal' (AE_model.py:22).  On a spectral-normed conv the reference
    touches only the hook's derived `weight` attribute (re-computed at the next forward) and zeroes the bias."""
    for m in net.modules():
        name = m.__class__.__name__
        if hasattr(m, 'weight') and ('Conv' in name or 'Linear' in name):
            if not hasattr(m, 'weight_orig'):
                nn.init.orthogonal_(m.weight.data, gain=gain)
            if getattr(m, 'bias', None) is not None:
                nn.init.constant_(m.bias.data, 0.0)
    return net


def o_lsgan(pred, target_is_real, is_disc):
    """external_function.py:53-57: MSELoss(reduction='none') against the constant label; .mean() only for the D form."""
    loss = (pred - (1.0 if target_is_real else 0.0)) ** 2
    return loss.mean() if is_disc else loss


def o_my_transform(x, size=(256, 128), mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225), normalize=True):
    """CC/clustercontrast/utils/data/diff_augs.py:6-16: torchvision resize(BICUBIC) of a float tensor =
    F.interpolate(bicubic, align_corners=False) (no antialias when up-sampling), then (x - mean) / std."""
    y = F.interpolate(x, size=size, mode='bicubic', align_corners=False) if tuple(x.shape[2:]) != tuple(size) else x
    if normalize:
        m = torch.tensor(mean, dtype=x.dtype).view(1, -1, 1, 1)
        s = torch.tensor(std, dtype=x.dtype).view(1, -1, 1, 1)
        y = (y - m) / s
    return y


class OFDGenerator(nn.Module):
    """networks.py:449-538 (`--model_gen FD`: fuse_mode='add', noise_nc=512, BatchNorm): W_reid / W_noise -> ReLU -> ConvT(8,4) ->
    norm -> dropout, four [ReLU, ConvT 4x4/2, norm, dropout] blocks, [ReLU, ConvT 4x4/2, Tanh]"""

    def __init__(self, reid_feature_nc, ngf=64, noise_nc=3, output_nc=3, dropout=0.0, fuse_mode='none'):
        super(OFDGenerator, self).__init__()
        self.fuse_mode = fuse_mode
        if fuse_mode == 'cat':
            nc = reid_feature_nc + noise_nc
        elif fuse_mode == 'add':
            nc = max(reid_feature_nc, noise_nc)
            self.W_reid = nn.Linear(reid_feature_nc, nc, bias=False)
            self.W_noise = nn.Linear(noise_nc, nc, bias=False)
        else:
            nc = reid_feature_nc
            self.W_reid = nn.Linear(reid_feature_nc, nc, bias=False)

        def block(cin, cout):
            return nn.Sequential(nn.ReLU(True), nn.ConvTranspose2d(cin, cout, 4, 2, 1, bias=False), nn.BatchNorm2d(cout),
                                 nn.Dropout(dropout))
        self.de_avg = nn.Sequential(nn.ReLU(True), nn.ConvTranspose2d(nc, ngf * 8, (8, 4), bias=False), nn.BatchNorm2d(ngf * 8),
                                    nn.Dropout(dropout))
        self.de_conv5 = block(ngf * 8, ngf * 8)
        self.de_conv4 = block(ngf * 8, ngf * 4)
        self.de_conv3 = block(ngf * 4, ngf * 2)
        self.de_conv2 = block(ngf * 2, ngf)
        self.de_conv1 = nn.Sequential(nn.ReLU(True), nn.ConvTranspose2d(ngf, output_nc, 4, 2, 1, bias=False), nn.Tanh())

    def forward(self, reid_feature, noise=None):
        b = reid_feature.shape[0]
        if self.fuse_mode == 'cat':
            feature = torch.cat((reid_feature, noise), dim=1)
        elif self.fuse_mode == 'add':
            feature = (self.W_reid(reid_feature.view(b, -1)) + self.W_noise(noise.view(b, -1))).view(b, -1, 1, 1)
        else:
            feature = self.W_reid(reid_feature.view(b, -1)).view(b, -1, 1, 1)
        x = self.de_avg(feature)
        for blk in (self.de_conv5, self.de_conv4, self.de_conv3, self.de_conv2, self.de_conv1):
            x = blk(x)
        return x


class OResBlockSN(nn.Module):
    """base_function.py:193-233 as Resize_ReID builds it (networks.py:152-157): BatchNorm, ReLU, spectral-normed convolutions"""

    def __init__(self, cin, cout, norm='batch', sn=True):
        super(OResBlockSN, self).__init__()
        self.conv1 = _sn(nn.Conv2d(cin, cout, 3, 1, 1), sn)
        self.conv2 = _sn(nn.Conv2d(cout, cout, 3, 1, 1), sn)
        self.bypass = _sn(nn.Conv2d(cin, cout, 1), sn)
        self.model = nn.Sequential(_norm2d(norm, cin), nn.ReLU(), self.conv1, _norm2d(norm, cout), nn.ReLU(), self.conv2)
        self.shortcut = nn.Sequential(self.bypass)

    def forward(self, x):
        return self.model(x) + self.shortcut(x)


class OResize_ReID(nn.Module):
    """networks.py:140-162: `my_resize` (bicubic, diff_augs.py:6-7) of the synthesised 128x64 image to 256x128, three residual blocks
    3 -> 64 -> 64 -> 3 added back onto the resized image"""

    def __init__(self, image_nc=3, ngf=64):
        super(OResize_ReID, self).__init__()
        self.resblock1 = OResBlockSN(image_nc, ngf)
        self.resblock2 = OResBlockSN(ngf, ngf)
        self.resblock3 = OResBlockSN(ngf, image_nc)

    def forward(self, inputs):
        x = o_my_transform(inputs, (256, 128), normalize=False)
        return x + self.resblock3(self.resblock2(self.resblock1(x)))


class OAEModel(object):
    """AEModel with model_gen='Pose', lsgan, no VGG loss (CC/dual_gan/models/AE_model.py:58-160, 212-214, 294-376)."""

    def __init__(self, net_G, net_D, gan_lr=2e-4, beta1=0.5, ratio_g2d=0.1, lambda_rec=2.0, lambda_g=5.0):
        self.net_G, self.net_D = net_G, net_D
        self.lambda_rec, self.lambda_g = lambda_rec, lambda_g
        self.optimizer_G = torch.optim.Adam(net_G.parameters(), lr=gan_lr, betas=(beta1, 0.999))
        self.optimizer_D = torch.optim.Adam(net_D.parameters(), lr=gan_lr * ratio_g2d, betas=(beta1, 0.999))

    def set_input(self, inputs):
        self.source_image, self.source_pose = inputs['Xs'], inputs['Ps']

    def synthesize_p(self, features):
        self.fake_image = self.net_G(features, self.source_pose)
        return self.fake_image

    def backward_D(self):
        for p in self.net_D.parameters():
            p.requires_grad = True
        real = o_lsgan(self.net_D(self.source_image), True, True)
        fake = o_lsgan(self.net_D(self.fake_image.detach()), False, True)
        self.loss_D = (real + fake) * 0.5
        self.loss_D.backward()

    def get_loss_G(self, need_cm=False, cluster_features=None):
        """:355-376.  need_cm=False: L1 map * lambda_rec and lsgan map * lambda_g, each averaged.  need_cm=True (:361-372): the same
        objective formed per sample first (`flatten(1).mean(-1)` of both maps, then the mean over the batch) plus `loss_rec`, the
        per-sample mean |net_G(cluster_features, pose) - source| of nn.L1Loss(reduction='none') (:122), not weighted."""
        for p in self.net_D.parameters():
            p.requires_grad = False
        app = (self.fake_image - self.source_image).abs() * self.lambda_rec
        ad = o_lsgan(self.net_D(self.fake_image), True, False) * self.lambda_g
        if need_cm:
            cluster_image = self.net_G(cluster_features, self.source_pose)
            loss_rec = (cluster_image - self.source_image).abs().flatten(1).mean(dim=-1)
            self.loss_G = (app.flatten(1).mean(dim=-1) + ad.flatten(1).mean(dim=-1)).mean()
            return self.loss_G, loss_rec
        self.loss_G = app.mean() + ad.mean()
        return self.loss_G

    def get_L1_loss(self, with_dis=False):
        """:378-390: per-sample mean |fake - source|; with_dis: lambda_rec * that + lambda_g * per-sample mean of the lsgan map"""
        rec = (self.fake_image - self.source_image).abs().flatten(1).mean(dim=-1)
        if not with_dis:
            return rec
        for p in self.net_D.parameters():
            p.requires_grad = False
        dis = o_lsgan(self.net_D(self.fake_image), True, False).flatten(1).mean(dim=-1)
        return rec * self.lambda_rec + dis * self.lambda_g

    def optimize_generated(self):
        """D update then G update on an already synthesised fake (:403-410)."""
        self.optimizer_D.zero_grad()
        self.backward_D()
        self.optimizer_D.step()
        self.optimizer_G.zero_grad()
        self.get_loss_G().backward()
        self.optimizer_G.step()


def o_joint_step(encoder, memory, gan, optimizer, imgs, labels, gan_inputs, conf_mask=None):
    """Live lines of ClusterContrastWithGANTrainer.train_all, CC/clustercontrast/trainers_b.py:617-774: the encoder's
    train-mode output is (bn_x, normalize(feature map)) (CC/clustercontrast/models/resnet.py:73-107); the map drives the
    generator (detached), bn_x the cluster-contrast loss."""
    gan.set_input(gan_inputs)
    f_out, f_gan = encoder(imgs)
    gan.synthesize_p(f_gan.detach())
    loss_G = gan.get_loss_G()
    loss_cl = memory(f_out, labels)
    if conf_mask is not None:
        loss_cl = loss_cl * conf_mask
    loss_cl = loss_cl.mean()
    loss = loss_cl + loss_G
    gan.optimizer_D.zero_grad()
    gan.backward_D()
    gan.optimizer_D.step()
    gan.optimizer_G.zero_grad()
    optimizer.zero_grad()
    loss.backward()
    gan.optimizer_G.step()
    optimizer.step()
    return loss.detach(), loss_cl.detach(), loss_G.detach(), gan.loss_D.detach()


# ---- synthetic dual_gan inputs --------------------------------------------------------------------
def synth_dualgan_inputs(n, h=128, w=64, pose_nc=18, seed=0, sigma=6.0, p_missing=0.1):
    """Xs = (U[0,1) - 0.5) / 0.5; Ps = exp(-((y-y0)^2 + (x-x0)^2) / (2 sigma^2)) per landmark, zero channel if missing
    (CC/clustercontrast/utils/data/pose_utils.py:52-70)."""
    g = torch.Generator().manual_seed(seed)
    xs = (torch.rand(n, 3, h, w, generator=g) - 0.5) / 0.5
    yy = torch.arange(h, dtype=torch.float32).view(1, 1, h, 1)
    xx = torch.arange(w, dtype=torch.float32).view(1, 1, 1, w)
    y0 = torch.randint(0, h, (n, pose_nc, 1, 1), generator=g).float()
    x0 = torch.randint(0, w, (n, pose_nc, 1, 1), generator=g).float()
    ps = torch.exp(-((yy - y0) ** 2 + (xx - x0) ** 2) / (2 * sigma ** 2))
    keep = (torch.rand(n, pose_nc, 1, 1, generator=g) >= p_missing).float()
    return {'Xs': xs, 'Ps': ps * keep}


# =====================================================================================================================
# DPTNModel path (BASELINE config 5): GAN objectives, gradient penalty, VGG perceptual / style loss, the step driver
# =====================================================================================================================
def o_ganloss(pred, target_is_real, is_disc, mode, real_label=1.0, fake_label=0.0):
    """CC/dual_gan/models/external_function.py:46-69 for every gan_mode."""
    if mode in ('lsgan', 'vanilla'):
        labels = torch.full_like(pred, real_label if target_is_real else fake_label)
        if mode == 'lsgan':
            loss = (pred - labels) ** 2                              # MSELoss(reduction='none') (:35)
            return loss.mean() if is_disc else loss
        loss = F.binary_cross_entropy_with_logits(pred, labels)      # BCEWithLogitsLoss() is already a mean (:38)
        return loss.mean() if is_disc else loss
    if mode in ('hinge', 'wgangp'):
        if is_disc:
            if target_is_real:
                pred = -pred
            return F.relu(1 + pred).mean() if mode == 'hinge' else pred.mean()
        return -pred.mean()
    raise NotImplementedError('gan mode %s not implemented' % mode)


def o_cal_gradient_penalty(netD, real, fake, alpha, constant=1.0, lambda_gp=10.0):
    """external_function.py:72-104, type='mixed'; `alpha` [B, 1] is the torch.rand draw of :89 passed in explicitly."""
    a = alpha.expand(real.shape[0], real.nelement() // real.shape[0]).contiguous().view(*real.shape)
    inter = (a * real + (1 - a) * fake).detach().requires_grad_(True)
    out = netD(inter)
    grads = torch.autograd.grad(outputs=out, inputs=inter, grad_outputs=torch.ones_like(out), create_graph=True,
                                retain_graph=True, only_inputs=True)[0].view(real.size(0), -1)
    gp = (((grads + 1e-16).norm(2, dim=1) - constant) ** 2).mean() * lambda_gp
    return gp, grads


def o_tv_vgg19_features():
    """torchvision.models.vgg19().features (configuration 'E', no batch norm): 16 conv3x3(pad 1)+ReLU(inplace) in groups
    of 2, 2, 4, 4, 4 separated by MaxPool2d(2, 2) — 37 modules, indices as external_function.py:250-296 slices them.
    THIRD-PARTY RESTATEMENT: torchvision is not installed here (version unpinned in the reference, CC/setup.py:11); the
    architecture is the published VGG-19 (Simonyan & Zisserman 2014, table 1 column E).  Parity unpinned for this function;
    everything the reference builds on top of it (VGG19 wrapper, VGGLoss) is pinned through it."""
    cfg = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 256, 'M', 512, 512, 512, 512, 'M', 512, 512, 512, 512, 'M']
    layers, cin = [], 3
    for v in cfg:
        if v == 'M':
            layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
        else:
            layers += [nn.Conv2d(cin, v, kernel_size=3, padding=1), nn.ReLU(inplace=True)]
            cin = v
    return nn.Sequential(*layers)


_VGG_SLICES = [('relu1_1', 0, 2), ('relu1_2', 2, 4), ('relu2_1', 4, 7), ('relu2_2', 7, 9), ('relu3_1', 9, 12),
               ('relu3_2', 12, 14), ('relu3_3', 14, 16), ('relu3_4', 16, 18), ('relu4_1', 18, 21), ('relu4_2', 21, 23),
               ('relu4_3', 23, 25), ('relu4_4', 25, 27), ('relu5_1', 27, 30), ('relu5_2', 30, 32), ('relu5_3', 32, 34),
               ('relu5_4', 34, 36)]


class OVGG19(nn.Module):
    """external_function.py:226-347: the feature stack cut into 16 named slices, parameters frozen."""

    def __init__(self, features=None):
        super(OVGG19, self).__init__()
        features = o_tv_vgg19_features() if features is None else features
        for name, a, b in _VGG_SLICES:
            seq = nn.Sequential()
            for x in range(a, b):
                seq.add_module(str(x), features[x])
            setattr(self, name, seq)
        for p in self.parameters():
            p.requires_grad = False

    def forward(self, x):
        out = {}
        for name, _, _ in _VGG_SLICES:
            x = getattr(self, name)(x)
            out[name] = x
        return out


def o_gram(x):
    b, ch, h, w = x.size()
    f = x.view(b, ch, w * h)
    return f.bmm(f.transpose(1, 2)) / (h * w * ch)


class OVGGLoss(nn.Module):
    """external_function.py:107-147 -> (content_loss, style_loss)."""

    def __init__(self, vgg=None, weights=(1.0, 1.0, 1.0, 1.0, 1.0)):
        super(OVGGLoss, self).__init__()
        self.vgg = OVGG19() if vgg is None else vgg
        self.weights = list(weights)

    def forward(self, x, y):
        xv, yv = self.vgg(x), self.vgg(y)
        content = 0.0
        for w, k in zip(self.weights, ('relu1_1', 'relu2_1', 'relu3_1', 'relu4_1', 'relu5_1')):
            content = content + w * F.l1_loss(xv[k], yv[k])
        style = 0.0
        for k in ('relu2_2', 'relu3_4', 'relu4_4', 'relu5_2'):
            style = style + F.l1_loss(o_gram(xv[k]), o_gram(yv[k]))
        return content, style


def o_seed_vgg(vgg, seed=7):
    """seeded stand-in for the ImageNet weights (a download): He-normal filters, small biases"""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for m in vgg.modules():
            if isinstance(m, nn.Conv2d):
                fan_in = m.weight[0].numel()
                m.weight.copy_(torch.randn(m.weight.shape, generator=g) * (2.0 / fan_in) ** 0.5)
                m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.05)
    return vgg


class ODPTNModel(object):
    """DPTNModel training side, CC/dual_gan/models/DPTN_model.py:44-107 (construction), :135-137 (forward), :159-182
    (backward_D_basic / backward_D), :184-214 (backward_G_basic / backward_G), :216-225 (optimize_parameters).
    `vgg=None` drops the perceptual / style terms (the run configuration of BASELINE config 5: no ImageNet weights)."""

    def __init__(self, net_G=None, net_D=None, gan_mode='hinge', gan_lr=2e-4, beta1=0.5, ratio_g2d=0.1, lambda_rec=2.0,
                 lambda_g=5.0, lambda_style=500.0, lambda_content=0.5, t_s_ratio=0.8, dis_layers=3, vgg=None):
        if net_G is None:
            net_G = o_init_weights(ODPTNGenerator(3, 18, 64, 512, 3, 'instance', 3, 3, 2, 2, 2))        # :51-53
        if net_D is None:
            net_D = o_init_weights(OResDiscriminator(3, 32, 128, dis_layers, True))                       # :63
        self.net_G, self.net_D = net_G, net_D
        self.gan_mode, self.t_s_ratio = gan_mode, t_s_ratio
        self.lambda_rec, self.lambda_g = lambda_rec, lambda_g
        self.lambda_style, self.lambda_content = lambda_style, lambda_content
        self.vgg = vgg
        self.optimizer_G = torch.optim.Adam(net_G.parameters(), lr=gan_lr, betas=(beta1, 0.999))
        self.optimizer_D = torch.optim.Adam(net_D.parameters(), lr=gan_lr * ratio_g2d, betas=(beta1, 0.999))
        self.gp_alpha = None                   # wgangp: explicit interpolation draw for the next backward_D

    def set_input(self, inputs):
        self.source_image, self.source_pose = inputs['Xs'], inputs['Ps']
        self.target_image, self.target_pose = inputs['Xt'], inputs['Pt']

    def forward(self):
        self.fake_image_t, self.fake_image_s = self.net_G(self.source_image, self.source_pose, self.target_pose)

    def backward_D(self):
        for p in self.net_D.parameters():
            p.requires_grad = True
        real = o_ganloss(self.net_D(self.target_image), True, True, self.gan_mode)
        fake = o_ganloss(self.net_D(self.fake_image_t.detach()), False, True, self.gan_mode)
        loss = (real + fake) * 0.5
        if self.gan_mode == 'wgangp':
            alpha = self.gp_alpha if self.gp_alpha is not None else torch.rand(self.target_image.shape[0], 1)
            gp, _ = o_cal_gradient_penalty(self.net_D, self.target_image, self.fake_image_t.detach(), alpha)
            loss = loss + gp
        self.loss_dis_img_gen_t = loss
        loss.backward()

    def _g_basic(self, fake, target, use_d):
        app = F.l1_loss(fake, target) * self.lambda_rec
        ad = None
        if use_d:
            for p in self.net_D.parameters():
                p.requires_grad = False
            ad = o_ganloss(self.net_D(fake), True, False, self.gan_mode) * self.lambda_g
        if self.vgg is not None:
            content, style = self.vgg(fake, target)
            style, content = style * self.lambda_style, content * self.lambda_content
        else:
            style = content = torch.zeros(())
        return app, ad, style, content

    def backward_G(self):
        for p in self.net_D.parameters():
            p.requires_grad = True
        self.loss_app_gen_t, self.loss_ad_gen_t, self.loss_style_gen_t, self.loss_content_gen_t = \
            self._g_basic(self.fake_image_t, self.target_image, True)
        self.loss_app_gen_s, _, self.loss_style_gen_s, self.loss_content_gen_s = \
            self._g_basic(self.fake_image_s, self.source_image, False)
        r = self.t_s_ratio
        G_loss = (r * (self.loss_app_gen_t + self.loss_style_gen_t + self.loss_content_gen_t)
                  + (1 - r) * (self.loss_app_gen_s + self.loss_style_gen_s + self.loss_content_gen_s) + self.loss_ad_gen_t)
        G_loss.backward()                      # lsgan: loss_ad_gen_t is a map -> RuntimeError, as in the reference
        for p in self.net_D.parameters():
            p.requires_grad = True

    def optimize_parameters(self):
        self.forward()
        self.optimizer_D.zero_grad()
        self.backward_D()
        self.optimizer_D.step()
        self.optimizer_G.zero_grad()
        self.backward_G()
        self.optimizer_G.step()

    def step(self, inputs):
        self.set_input(inputs)
        self.optimize_parameters()
        return self.get_current_errors()

    def get_current_errors(self):
        import collections
        names = ['app_gen_s', 'content_gen_s', 'style_gen_s', 'app_gen_t', 'ad_gen_t', 'dis_img_gen_t', 'content_gen_t',
                 'style_gen_t']
        return collections.OrderedDict((n, float(torch.as_tensor(getattr(self, 'loss_' + n)).detach())) for n in names)


def synth_dptn_inputs(n, h=128, w=64, seed=0):
    a = synth_dualgan_inputs(n, h, w, seed=seed)
    b = synth_dualgan_inputs(n, h, w, seed=seed + 1000)
    return {'Xs': a['Xs'], 'Ps': a['Ps'], 'Xt': b['Xs'], 'Pt': b['Ps']}
