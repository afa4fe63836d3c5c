"""ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product; nothing under reid-gan_amd/ imports it.

A CPU restatement (plain torch fp32 ops, no HIP, no reference imports) of the reference's arithmetic for
the hot path: the joint FD-GAN + cluster-contrast training step.  Only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import this package, and only as the checker / reported baseline.

Pinning: tests/golden/make_golden.py (run in the build container, where /root/reference is mounted)
imports the reference's own modules, loads THIS file's seeded weights into them and stores their outputs
as fixtures under tests/golden/; tests/test_oracle_golden.py re-checks this file against those fixtures
without the reference.  The ResNet-50 trunk is taken by the reference from torchvision (unpinned version,
not in the reference tree, not installed here); its structure is restated from the torchvision v1.5
layout and pinned against the reference's in-tree Bottleneck (CC/clustercontrast/models/resnet_ibn_a.py:70-109).

Citations: FD/ = /root/reference/FD-GAN-master/, CC/ = /root/reference/cluster-contrast-reid-main/.
"""
from __future__ import absolute_import

import collections
import random

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


# ------------------------------------------------------------------------------------------------
# ResNet trunk, torchvision layout (stride on the 3x3).  Pin: CC/clustercontrast/models/resnet_ibn_a.py:70-159
# ------------------------------------------------------------------------------------------------
class OBottleneck(nn.Module):
    expansion = 4

    def __init__(self, cin, width, stride=1, downsample=None):
        super(OBottleneck, self).__init__()
        self.conv1 = nn.Conv2d(cin, width, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(width)
        self.conv2 = nn.Conv2d(width, width, 3, stride, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(width)
        self.conv3 = nn.Conv2d(width, width * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(width * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        skip = x if self.downsample is None else self.downsample(x)
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.relu(self.bn2(self.conv2(y)))
        y = self.bn3(self.conv3(y))
        return self.relu(y + skip)


class OBasicBlock(nn.Module):
    expansion = 1

    def __init__(self, cin, width, stride=1, downsample=None):
        super(OBasicBlock, self).__init__()
        self.conv1 = nn.Conv2d(cin, width, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(width)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(width, width, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(width)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        skip = x if self.downsample is None else self.downsample(x)
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.bn2(self.conv2(y))
        return self.relu(y + skip)


_DEPTHS = {18: (OBasicBlock, (2, 2, 2, 2)), 34: (OBasicBlock, (3, 4, 6, 3)), 50: (OBottleneck, (3, 4, 6, 3)),
           101: (OBottleneck, (3, 4, 23, 3)), 152: (OBottleneck, (3, 8, 36, 3))}


class OTVResNet(nn.Module):
    def __init__(self, depth=50, num_classes=1000):
        super(OTVResNet, self).__init__()
        block, counts = _DEPTHS[depth]
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        cin = 64
        for li, (width, n, stride) in enumerate(zip((64, 128, 256, 512), counts, (1, 2, 2, 2)), 1):
            blocks = []
            for bi in range(n):
                s = stride if bi == 0 else 1
                ds = None
                if s != 1 or cin != width * block.expansion:
                    ds = nn.Sequential(nn.Conv2d(cin, width * block.expansion, 1, s, bias=False),
                                       nn.BatchNorm2d(width * block.expansion))
                blocks.append(block(cin, width, s, ds))
                cin = width * block.expansion
            setattr(self, "layer%d" % li, nn.Sequential(*blocks))
        self.avgpool = nn.Identity()
        self.fc = nn.Linear(cin, num_classes)


def _reid_reset_params(model):
    """FD/reid/models/resnet.py:90-102 (same in CC/clustercontrast/models/resnet.py:112-127 + BN1d)."""
    for m in model.modules():
        if isinstance(m, nn.Conv2d):
            nn.init.kaiming_normal_(m.weight, mode='fan_out')
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, (nn.BatchNorm2d, nn.BatchNorm1d)):
            nn.init.constant_(m.weight, 1)
            nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.Linear):
            nn.init.normal_(m.weight, std=0.001)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)


class OReidResNet(nn.Module):
    """FD/reid/models/resnet.py:13-88."""

    def __init__(self, depth=50, cut_at_pooling=False, num_features=0, norm=False, dropout=0, num_classes=0):
        super(OReidResNet, self).__init__()
        self.cut_at_pooling = cut_at_pooling
        self.base = OTVResNet(depth)
        if not cut_at_pooling:
            self.num_features, self.norm, self.dropout = num_features, norm, dropout
            self.has_embedding = num_features > 0
            self.num_classes = num_classes
            out_planes = self.base.fc.in_features
            if self.has_embedding:
                self.feat = nn.Linear(out_planes, num_features)
                self.feat_bn = nn.BatchNorm1d(num_features)
            else:
                self.num_features = out_planes
            if dropout > 0:
                self.drop = nn.Dropout(dropout)
            if num_classes > 0:
                self.classifier = nn.Linear(self.num_features, num_classes)
        _reid_reset_params(self)
        if not cut_at_pooling and self.has_embedding:
            nn.init.kaiming_normal_(self.feat.weight, mode='fan_out')

    def forward(self, x):
        b = self.base
        x = b.maxpool(b.relu(b.bn1(b.conv1(x))))
        x = b.layer4(b.layer3(b.layer2(b.layer1(x))))
        x = F.avg_pool2d(x, x.shape[2:]).flatten(1)                    # :71-72
        if self.cut_at_pooling:
            return x
        if self.has_embedding:
            x = self.feat_bn(self.feat(x))
        if self.norm:
            x = F.normalize(x)
        elif self.has_embedding:
            x = F.relu(x)
        if self.dropout > 0:
            x = self.drop(x)
        if self.num_classes > 0:
            x = self.classifier(x)
        return x


class OGeM(nn.Module):
    """GeneralizedMeanPoolingP, CC/clustercontrast/models/pooling.py:57-103."""

    def __init__(self, norm=3.0, eps=1e-6):
        super(OGeM, self).__init__()
        self.p = nn.Parameter(torch.ones(1) * norm)
        self.eps = eps

    def forward(self, x):
        return F.adaptive_avg_pool2d(x.clamp(min=self.eps).pow(self.p), 1).pow(1.0 / self.p)


class OCCResNet(nn.Module):
    """CC/clustercontrast/models/resnet.py:14-110: layer4 stride 1 (:34-35), Sequential base (:36-38), pooling
    factory (:40), feat_bn with frozen bias (:60-61), train-mode tuple return (:107)."""

    def __init__(self, depth=50, cut_at_pooling=False, num_features=0, norm=False, dropout=0, num_classes=0,
                 pooling_type='avg'):
        super(OCCResNet, self).__init__()
        self.cut_at_pooling = cut_at_pooling
        r = OTVResNet(depth)
        r.layer4[0].conv2.stride = (1, 1)
        r.layer4[0].downsample[0].stride = (1, 1)
        self.base = nn.Sequential(r.conv1, r.bn1, r.relu, r.maxpool, r.layer1, r.layer2, r.layer3, r.layer4)
        if pooling_type == 'gem':
            self.gap = OGeM()
        elif pooling_type == 'avg':
            self.gap = nn.AdaptiveAvgPool2d(1)
        else:
            raise KeyError("oracle: pooling %r not restated" % pooling_type)
        if not cut_at_pooling:
            self.num_features, self.norm, self.dropout = num_features, norm, dropout
            self.has_embedding = num_features > 0
            self.num_classes = num_classes
            out_planes = r.fc.in_features
            if self.has_embedding:
                self.feat = nn.Linear(out_planes, num_features)
                self.feat_bn = nn.BatchNorm1d(num_features)
            else:
                self.num_features = out_planes
                self.feat_bn = nn.BatchNorm1d(out_planes)
            self.feat_bn.bias.requires_grad_(False)
            if dropout > 0:
                self.drop = nn.Dropout(dropout)
            if num_classes > 0:
                self.classifier = nn.Linear(self.num_features, num_classes, bias=False)
        _reid_reset_params(self)
        if not cut_at_pooling and self.has_embedding:
            nn.init.kaiming_normal_(self.feat.weight, mode='fan_out')

    def forward(self, x, test_all=False):
        x = self.base(x)
        gan_x = x
        x = self.gap(x).flatten(1)
        if self.cut_at_pooling:
            return x
        bn_x = self.feat_bn(self.feat(x)) if self.has_embedding else self.feat_bn(x)
        if not self.training:
            bn_x = F.normalize(bn_x)
            return (bn_x, F.normalize(gan_x, dim=1)) if test_all else bn_x
        if self.norm:
            bn_x = F.normalize(bn_x)
        elif self.has_embedding:
            bn_x = F.relu(bn_x)
        if self.dropout > 0:
            bn_x = self.drop(bn_x)
        if self.num_classes > 0:
            return self.classifier(bn_x)
        return bn_x, F.normalize(gan_x, dim=1)


# ------------------------------------------------------------------------------------------------
# embedding / siamese   FD/reid/models/embedding.py:7-39, multi_branch.py:6-15
# ------------------------------------------------------------------------------------------------
class OEltwiseSubEmbed(nn.Module):
    def __init__(self, use_batch_norm=False, use_classifier=False, num_features=0, num_classes=0):
        super(OEltwiseSubEmbed, self).__init__()
        self.use_batch_norm, self.use_classifier = use_batch_norm, use_classifier
        if use_batch_norm:
            self.bn = nn.BatchNorm1d(num_features)
        if use_classifier:
            self.classifier = nn.Linear(num_features, num_classes)
            self.classifier.weight.data.normal_(0, 0.001)
            self.classifier.bias.data.zero_()

    def forward(self, x1, x2):
        x = (x1 - x2).pow(2)
        if self.use_batch_norm:
            x = self.bn(x)
        return self.classifier(x.flatten(1)) if self.use_classifier else x.sum(1)


class OSiameseNet(nn.Module):
    def __init__(self, base_model, embed_model):
        super(OSiameseNet, self).__init__()
        self.base_model, self.embed_model = base_model, embed_model

    def forward(self, x1, x2):
        f1, f2 = self.base_model(x1), self.base_model(x2)            # two separate trunk passes (:13)
        if self.embed_model is None:
            return f1, f2
        return f1, f2, self.embed_model(f1, f2)


# ------------------------------------------------------------------------------------------------
# FD-GAN generator / discriminator   FD/fdgan/networks.py:62-237
# ------------------------------------------------------------------------------------------------
def o_weights_init_normal(m):
    """FD/fdgan/networks.py:13-21."""
    name = m.__class__.__name__
    if 'Conv' in name or 'Linear' in name:
        nn.init.normal_(m.weight.data, 0.0, 0.02)
    elif 'BatchNorm2d' in name:
        nn.init.normal_(m.weight.data, 1.0, 0.02)
        nn.init.constant_(m.bias.data, 0.0)


class OPoseGenerator(nn.Module):
    """CustomPoseGenerator.  The in-place LeakyReLU/ReLU modules are kept in place on purpose: they rewrite
    the skip tensors, which matters when connect_layers > 0 (networks.py:141-156,158-162)."""
    _IN_CH = ((8, 8, 4, 2, 1), (16, 8, 4, 2, 1), (16, 16, 4, 2, 1), (16, 16, 8, 2, 1), (16, 16, 8, 4, 1),
              (16, 16, 8, 4, 2))

    def __init__(self, pose_feature_nc, reid_feature_nc, noise_nc, pose_nc=18, output_nc=3, dropout=0.0,
                 norm='batch', fuse_mode='cat', connect_layers=0):
        super(OPoseGenerator, self).__init__()
        g = 64
        self.connect_layers, self.fuse_mode = connect_layers, fuse_mode
        if norm == 'batch':
            mk, bias = (lambda c: nn.BatchNorm2d(c, affine=True)), False
        else:
            mk, bias = (lambda c: nn.InstanceNorm2d(c, affine=False)), True

        def enc(i, o):
            return nn.Sequential(nn.LeakyReLU(0.2, True), nn.Conv2d(i, o, 4, 2, 1, bias=bias), mk(o))

        def dec(i, o):
            return nn.Sequential(nn.ReLU(True), nn.ConvTranspose2d(i, o, 4, 2, 1, bias=bias), mk(o),
                                 nn.Dropout(dropout))
        self.en_conv1 = nn.Conv2d(pose_nc, g, 4, 2, 1, bias=bias)
        self.en_conv2, self.en_conv3 = enc(g, 2 * g), enc(2 * g, 4 * g)
        self.en_conv4, self.en_conv5 = enc(4 * g, 8 * g), enc(8 * g, 8 * g)
        self.en_avg = nn.Sequential(nn.LeakyReLU(0.2, True), nn.Conv2d(8 * g, pose_feature_nc, (8, 4), bias=bias),
                                    mk(pose_feature_nc))
        if fuse_mode == 'cat':
            nin = pose_feature_nc + reid_feature_nc + noise_nc
        else:
            nin = max(pose_feature_nc, reid_feature_nc, noise_nc)
            self.W_pose = nn.Linear(pose_feature_nc, nin, bias=False)
            self.W_reid = nn.Linear(reid_feature_nc, nin, bias=False)
            self.W_noise = nn.Linear(noise_nc, nin, bias=False)
        self.de_avg = nn.Sequential(nn.ReLU(True), nn.ConvTranspose2d(nin, 8 * g, (8, 4), bias=bias), mk(8 * g),
                                    nn.Dropout(dropout))
        ch = self._IN_CH[connect_layers]
        self.de_conv5, self.de_conv4 = dec(g * ch[0], 8 * g), dec(g * ch[1], 4 * g)
        self.de_conv3, self.de_conv2 = dec(g * ch[2], 2 * g), dec(g * ch[3], g)
        self.de_conv1 = nn.Sequential(nn.ReLU(True), nn.ConvTranspose2d(g * ch[4], output_nc, 4, 2, 1, bias=bias),
                                      nn.Tanh())

    def forward(self, posemap, reid_feature, noise):
        n = posemap.shape[0]
        p1 = self.en_conv1(posemap)
        p2 = self.en_conv2(p1)
        p3 = self.en_conv3(p2)
        p4 = self.en_conv4(p3)
        p5 = self.en_conv5(p4)
        pose_feature = self.en_avg(p5)
        if self.fuse_mode == 'cat':
            feat = torch.cat((reid_feature, pose_feature, noise), dim=1)
        else:
            feat = (self.W_reid(reid_feature.view(n, -1)) + self.W_pose(pose_feature.view(n, -1))
                    + self.W_noise(noise.view(n, -1))).view(n, -1, 1, 1)
        x = self.de_avg(feat)
        left = self.connect_layers
        for blk, skip in ((self.de_conv5, p5), (self.de_conv4, p4), (self.de_conv3, p3), (self.de_conv2, p2),
                          (self.de_conv1, p1)):
            if left > 0:
                x, left = blk(torch.cat((x, skip), dim=1)), left - 1
            else:
                x = blk(x)
        return x


class OPatchDiscriminator(nn.Module):
    """NLayerDiscriminator, FD/fdgan/networks.py:194-237 (n_layers = 3, ndf = 64)."""

    def __init__(self, input_nc, norm='batch'):
        super(OPatchDiscriminator, self).__init__()
        if norm == 'batch':
            mk, bias = (lambda c: nn.BatchNorm2d(c, affine=True)), False
        else:
            mk, bias = (lambda c: nn.InstanceNorm2d(c, affine=False)), True
        seq = [nn.Conv2d(input_nc, 64, 4, 2, 1), nn.LeakyReLU(0.2, True)]
        prev = 64
        for mult, stride in ((2, 2), (4, 2), (8, 1)):
            seq += [nn.Conv2d(prev, 64 * mult, 4, stride, 1, bias=bias), mk(64 * mult), nn.LeakyReLU(0.2, True)]
            prev = 64 * mult
        seq += [nn.Conv2d(prev, 1, 4, 1, 1)]
        self.model = nn.Sequential(*seq)

    def forward(self, x):
        return self.model(x)


def o_gan_loss(pred, target_is_real, smooth=False):
    """GANLoss.__call__, FD/fdgan/losses.py:15-32."""
    real, fake = 1.0, 0.0
    if smooth:
        real = random.uniform(0.7, 1.0)
        fake = random.uniform(0.0, 0.3)
    t = torch.full_like(pred, real if target_is_real else fake)
    return F.binary_cross_entropy(torch.sigmoid(pred), t)


# ------------------------------------------------------------------------------------------------
# cluster memory   CC/clustercontrast/models/cm.py:9-137
# ------------------------------------------------------------------------------------------------
class OCM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, inputs, targets, features, momentum):
        ctx.features, ctx.momentum = features, momentum
        ctx.save_for_backward(inputs, targets)
        return inputs.mm(features.t())

    @staticmethod
    def backward(ctx, grad_outputs):
        inputs, targets = ctx.saved_tensors
        grad_inputs = grad_outputs.mm(ctx.features) if ctx.needs_input_grad[0] else None   # pre-update bank (:25-26)
        for x, y in zip(inputs, targets):                                                   # batch order (:29-31)
            ctx.features[y] = ctx.momentum * ctx.features[y] + (1. - ctx.momentum) * x
            ctx.features[y] /= ctx.features[y].norm()
        return grad_inputs, None, None, None


class OCMHard(torch.autograd.Function):
    @staticmethod
    def forward(ctx, inputs, targets, features, momentum):
        ctx.features, ctx.momentum = features, momentum
        ctx.save_for_backward(inputs, targets)
        return inputs.mm(features.t())

    @staticmethod
    def backward(ctx, grad_outputs):
        inputs, targets = ctx.saved_tensors
        grad_inputs = grad_outputs.mm(ctx.features) if ctx.needs_input_grad[0] else None
        groups = collections.OrderedDict()
        for f, idx in zip(inputs, targets.tolist()):
            groups.setdefault(idx, []).append(f)
        for idx, feats in groups.items():                                                   # :62-70
            sims = [float(f.unsqueeze(0).mm(ctx.features[idx].unsqueeze(0).t())[0][0]) for f in feats]
            pick = int(np.argmin(np.array(sims, dtype=np.float32)))
            ctx.features[idx] = ctx.features[idx] * ctx.momentum + (1 - ctx.momentum) * feats[pick]
            ctx.features[idx] /= ctx.features[idx].norm()
        return grad_inputs, None, None, None


class OCMGan(torch.autograd.Function):
    """CM_gan, CC/clustercontrast/models/cm.py:83-108: the ReID bank and a second bank of GAN-side features are updated
    together, in batch order; the second one is re-normalised with F.normalize (eps-clamped)."""

    @staticmethod
    def forward(ctx, inputs, gan_inputs, targets, features, gan_features, momentum):
        ctx.features, ctx.gan_features, ctx.momentum = features, gan_features, momentum
        ctx.save_for_backward(inputs, gan_inputs, targets)
        return inputs.mm(features.t())

    @staticmethod
    def backward(ctx, grad_outputs):
        inputs, gan_inputs, targets = ctx.saved_tensors
        grad_inputs = grad_outputs.mm(ctx.features) if ctx.needs_input_grad[0] else None
        for x, gx, y in zip(inputs, gan_inputs, targets):
            ctx.features[y] = ctx.momentum * ctx.features[y] + (1. - ctx.momentum) * x
            ctx.features[y] /= ctx.features[y].norm()
            ctx.gan_features[y] = ctx.momentum * ctx.gan_features[y] + (1. - ctx.momentum) * gx
            ctx.gan_features[y] = F.normalize(ctx.gan_features[y], dim=0)
        return grad_inputs, None, None, None, None, None


class OClusterMemoryGradient(object):
    """ClusterMemory_Gradient, cm.py:138-193, without the hard-coded .cuda(): centroids are a leaf tensor trained by their
    own SGD; the loss sees the DETACHED normalised copy (so gradients reach the centroids only through other paths)."""

    def __init__(self, temp=0.05):
        self.temp = temp

    def set_clusters(self, clusters, cluster_lr):
        self.trainable_clusters = clusters.detach().clone().requires_grad_(True)
        self.optimizer_cluster = torch.optim.SGD([self.trainable_clusters], lr=cluster_lr)
        self.normed_clusters = F.normalize(self.trainable_clusters)

    def forward(self, inputs, targets, ex_f=None):
        inputs = F.normalize(inputs, dim=1)
        outputs = torch.mm(inputs, self.normed_clusters.detach().clone().t())
        if ex_f is not None:
            ex_f = F.normalize(ex_f, dim=1)
            outputs_ex = torch.mm(inputs, ex_f.t())
            group_size = outputs_ex.shape[0] // outputs_ex.shape[1]
            outputs_ex = outputs_ex + (-10000.0 * torch.eye(ex_f.shape[0])).repeat_interleave(group_size, dim=0)
            outputs = torch.cat([outputs, outputs_ex], dim=1)
        return F.cross_entropy(outputs / self.temp, targets)

    def update_clusters(self, p_ids, eps=1e-16):
        for p_id in p_ids:
            self.trainable_clusters.grad[p_id] /= self.trainable_clusters.grad[p_id].norm() + eps
        self.optimizer_cluster.step()
        self.optimizer_cluster.zero_grad()
        self.normed_clusters = F.normalize(self.trainable_clusters)


class OClusterMemory(nn.Module):
    """ClusterMemory.forward (cm.py:123-137) without the hard-coded .cuda()."""

    def __init__(self, num_features, num_samples, temp=0.05, momentum=0.2, use_hard=False):
        super(OClusterMemory, self).__init__()
        self.momentum, self.temp, self.use_hard = momentum, temp, use_hard
        self.register_buffer('features', torch.zeros(num_samples, num_features))
        self.register_buffer('gan_features', torch.zeros(num_samples, num_features))

    def forward(self, inputs, targets, gan_inputs=None, conf_weight=None):
        inputs = F.normalize(inputs, dim=1)
        m = torch.Tensor([self.momentum])
        fn = OCMHard if self.use_hard else OCM
        outputs = fn.apply(inputs, targets, self.features, m)
        outputs = outputs / self.temp
        return F.cross_entropy(outputs, targets, reduction="none")


# ------------------------------------------------------------------------------------------------
# step drivers
# ------------------------------------------------------------------------------------------------
def o_set_bn_eval(module):
    """set_bn_fix, FD/fdgan/networks.py:57-60."""
    for m in module.modules():
        if 'BatchNorm' in m.__class__.__name__:
            m.eval()


class OFDGANStep(object):
    """FDGANModel step (FD/fdgan/model.py:100-229), stage-2 optimizer wiring by default, on the CPU."""

    def __init__(self, net_E, net_G, net_Di, net_Dp, lr=0.001, stage=2, lambda_recon=1.0, lambda_veri=1.0,
                 lambda_sp=1.0, smooth_label=False):
        self.net_E, self.net_G, self.net_Di, self.net_Dp = net_E, net_G, net_Di, net_Dp
        self.lr, self.stage = lr, stage
        self.smooth = smooth_label                                  # _init_losses, model.py:90-98
        self.rand_list = [True] * 1 + [False] * 10000 if smooth_label else [False]
        self.lam = (lambda_recon, lambda_veri, lambda_sp)
        if stage == 1:
            self.opt_G = torch.optim.Adam(net_G.parameters(), lr=lr * 0.1, betas=(0.5, 0.999))
            self.opt_Di = torch.optim.SGD(net_Di.parameters(), lr=lr * 0.01, momentum=0.9, weight_decay=1e-4)
        else:
            groups = [{'params': net_E.base_model.parameters()}, {'params': net_E.embed_model.parameters()},
                      {'params': net_G.parameters()}]
            self.opt_G = torch.optim.Adam(groups, lr=lr * 0.1, betas=(0.5, 0.999))
            self.opt_Di = torch.optim.SGD(net_Di.parameters(), lr=lr, momentum=0.9, weight_decay=1e-4)
        self.opt_Dp = torch.optim.SGD(net_Dp.parameters(), lr=lr, momentum=0.9, weight_decay=1e-4)
        self.reset_model_status()

    def reset_model_status(self):                                   # model.py:72-85
        self.net_G.train()
        self.net_Dp.train()
        self.net_Di.train()
        o_set_bn_eval(self.net_Di)
        if self.stage == 1:
            self.net_E.eval()
        else:
            self.net_E.train()
            o_set_bn_eval(self.net_E)

    def _d_loss(self, pred_real, pred_fake):
        """model.py:165-171 / :178-184: one `random.choice` (label flip with p = 1/10001 when smoothing), then two GANLoss
        calls, each of which draws (real, fake) smoothed labels (losses.py:20-22)"""
        if random.choice(self.rand_list):
            loss_real = o_gan_loss(pred_fake, True, self.smooth)
            loss_fake = o_gan_loss(pred_real, False, self.smooth)
        else:
            loss_real = o_gan_loss(pred_real, True, self.smooth)
            loss_fake = o_gan_loss(pred_fake, False, self.smooth)
        return (loss_real + loss_fake) * 0.5

    def step(self, origin, target, posemap, labels, noise):
        """Inputs already in the post-set_input form (2b crops; labels [b]; noise [2b, nz])."""
        b2 = origin.shape[0]
        f1, f2, id_score = self.net_E(origin[:b2 // 2], origin[b2 // 2:])           # forward(), :149-157
        a_id = torch.cat((f1, f2))
        fake = self.net_G(posemap, a_id.view(b2, -1, 1, 1), noise.view(b2, -1, 1, 1))

        self.opt_Di.zero_grad()                                                      # backward_Di, :175-186
        _, _, pred_real = self.net_Di(origin, target)
        _, _, pred_fake = self.net_Di(origin, fake.detach())
        loss_Di = self._d_loss(pred_real, pred_fake)
        loss_Di.backward()
        self.opt_Di.step()

        self.opt_Dp.zero_grad()                                                      # backward_Dp, :159-173
        pred_real = self.net_Dp(torch.cat((posemap, target), dim=1))
        pred_fake = self.net_Dp(torch.cat((posemap, fake.detach()), dim=1))
        loss_Dp = self._d_loss(pred_real, pred_fake)
        loss_Dp.backward()
        self.opt_Dp.step()

        self.opt_G.zero_grad()                                                       # backward_G, :188-214
        loss_v = F.cross_entropy(id_score, labels.view(-1))
        loss_r = F.l1_loss(fake, target)
        half = b2 // 2
        fk1, fk2 = fake[:half], fake[half:]
        mask = labels.view(-1, 1, 1, 1).expand_as(fk1) == 1
        loss_sp = F.l1_loss(fk1[mask], fk2[mask])
        _, _, pred_fake_Di = self.net_Di(origin, fake)
        pred_fake_Dp = self.net_Dp(torch.cat((posemap, fake), dim=1))
        g_di, g_dp = o_gan_loss(pred_fake_Di, True), o_gan_loss(pred_fake_Dp, True)
        loss_G = g_di + g_dp + loss_r * self.lam[0] + loss_v * self.lam[1] + loss_sp * self.lam[2]
        loss_G.backward()
        self.opt_G.step()
        return collections.OrderedDict([('G_v', loss_v.item()), ('G_r', loss_r.item()), ('G_sp', loss_sp.item()),
                                        ('G_gan_Di', g_di.item()), ('G_gan_Dp', g_dp.item()),
                                        ('D_i', loss_Di.item()), ('D_p', loss_Dp.item())]), fake.detach()


def o_joint_step_fd(fd, encoder, memory, optimizer, imgs, labels, batch):
    """BASELINE config 4b: the joint step of CC/clustercontrast/trainers_b.py:617-774 with the FD-GAN object (`fd`, an
    OFDGANStep) in the GAN role.  Call order of the trainer: synthesize -> generator loss (discriminators as
    constants) -> cluster-contrast loss -> discriminator gradients -> ONE backward of loss_cl + loss_G -> optimizer
    steps; all four updates use gradients taken at the pre-update weights (the discriminator steps are applied after
    the generator backward, which still needs their forward-pass weights)."""
    origin, target, posemap, lab, noise = batch
    f_out = encoder(imgs)
    if isinstance(f_out, tuple):
        f_out = f_out[0]
    b2 = origin.shape[0]
    f1, f2, id_score = fd.net_E(origin[:b2 // 2], origin[b2 // 2:])
    a_id = torch.cat((f1, f2))
    fake = fd.net_G(posemap, a_id.view(b2, -1, 1, 1), noise.view(b2, -1, 1, 1))
    d_params = list(fd.net_Di.parameters()) + list(fd.net_Dp.parameters())
    for p in d_params:
        p.requires_grad = False
    loss_v = F.cross_entropy(id_score, lab.view(-1))
    loss_r = F.l1_loss(fake, target)
    half = b2 // 2
    fk1, fk2 = fake[:half], fake[half:]
    mask = lab.view(-1, 1, 1, 1).expand_as(fk1) == 1
    loss_sp = F.l1_loss(fk1[mask], fk2[mask])
    _, _, pred_fake_Di = fd.net_Di(origin, fake)
    pred_fake_Dp = fd.net_Dp(torch.cat((posemap, fake), dim=1))
    g_di, g_dp = o_gan_loss(pred_fake_Di, True), o_gan_loss(pred_fake_Dp, True)
    loss_G = g_di + g_dp + loss_r * fd.lam[0] + loss_v * fd.lam[1] + loss_sp * fd.lam[2]
    loss_cl = memory(f_out, labels).mean()
    loss = loss_cl + loss_G
    for p in d_params:
        p.requires_grad = True
    fd.opt_Di.zero_grad()
    fd.opt_Dp.zero_grad()
    _, _, pred_real = fd.net_Di(origin, target)
    _, _, pred_fake = fd.net_Di(origin, fake.detach())
    loss_Di = (o_gan_loss(pred_real, True) + o_gan_loss(pred_fake, False)) * 0.5
    loss_Di.backward()
    pred_real = fd.net_Dp(torch.cat((posemap, target), dim=1))
    pred_fake = fd.net_Dp(torch.cat((posemap, fake.detach()), dim=1))
    loss_Dp = (o_gan_loss(pred_real, True) + o_gan_loss(pred_fake, False)) * 0.5
    loss_Dp.backward()
    fd.opt_G.zero_grad()
    optimizer.zero_grad()
    loss.backward()
    fd.opt_G.step()
    optimizer.step()
    fd.opt_Di.step()
    fd.opt_Dp.step()
    return collections.OrderedDict([('loss', loss.item()), ('loss_cl', loss_cl.item()), ('G', loss_G.item()),
                                    ('D_i', loss_Di.item()), ('D_p', loss_Dp.item())]), fake.detach()


def o_cc_step(encoder, memory, optimizer, imgs, labels):
    """ClusterContrastTrainer.train body, CC/clustercontrast/trainers.py:229-249 (tuple output unpacked as in
    train_all :157)."""
    f_out = encoder(imgs)
    if isinstance(f_out, tuple):
        f_out = f_out[0]
    loss = memory(f_out, labels).mean()
    optimizer.zero_grad()
    loss.backward()
    optimizer.step()
    return loss.item()


# ------------------------------------------------------------------------------------------------
# synthetic inputs and the dropout mask of the HIP path
# ------------------------------------------------------------------------------------------------
def synth_images(n, h=256, w=128, seed=0):
    g = torch.Generator().manual_seed(seed)
    u = torch.rand(n, 3, h, w, generator=g)
    mean = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
    return (u - mean) / std


def synth_posemaps(n, h=256, w=128, seed=0, sigma=5.0, p_missing=0.1):
    """18 landmark channels: unit impulse blurred by a Gaussian (sigma 5) and divided by its max, or all-zero
    for a missing landmark (FD/reid/utils/data/preprocessor.py:114-131), in closed form."""
    rng = np.random.RandomState(seed)
    ys = torch.arange(h, dtype=torch.float32).view(1, 1, h, 1)
    xs = torch.arange(w, dtype=torch.float32).view(1, 1, 1, w)
    cy = torch.from_numpy(rng.randint(0, h, size=(n, 18)).astype(np.float32)).view(n, 18, 1, 1)
    cx = torch.from_numpy(rng.randint(0, w, size=(n, 18)).astype(np.float32)).view(n, 18, 1, 1)
    present = torch.from_numpy((rng.rand(n, 18) >= p_missing).astype(np.float32)).view(n, 18, 1, 1)
    return torch.exp(-((ys - cy) ** 2 + (xs - cx) ** 2) / (2 * sigma * sigma)) * present


def synth_fdgan_batch(b, h=256, w=128, nz=256, seed=0):
    """One post-set_input FD-GAN batch of b pairs: labels pattern (same, diff, diff, diff) as
    RandomPairSampler(neg_pos_ratio=3) yields (FD/reid/utils/data/sampler.py:37-52); same-identity pairs share
    pose map and target (model.py:133-136); the noise is duplicated for both halves (:141)."""
    labels = torch.tensor([1 if i % 4 == 0 else 0 for i in range(b)], dtype=torch.long)
    origin = synth_images(2 * b, h, w, seed)
    target = synth_images(2 * b, h, w, seed + 1)
    pose = synth_posemaps(2 * b, h, w, seed + 2)
    m = labels.view(-1, 1, 1, 1).float()
    target = torch.cat([target[:b], target[:b] * m + target[b:] * (1 - m)])
    pose = torch.cat([pose[:b], pose[:b] * m + pose[b:] * (1 - m)])
    g = torch.Generator().manual_seed(seed + 3)
    z = torch.randn(b, nz, generator=g)
    return origin, target, pose, labels, torch.cat((z, z))


def dropout_keep_mask(n, p, seed):
    """CPU restatement of the counter-based mask of rg_dropout (reid-gan_amd/csrc/eltwise.hip: mix32)."""
    M = (1 << 64) - 1
    idx = np.arange(n, dtype=np.uint64)
    with np.errstate(over='ignore'):
        z = (np.uint64((seed * 0x100000001B3) & M) + idx) + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    r = (z >> np.uint64(32)).astype(np.uint32)
    thr = np.uint32(min(np.float32(p) * np.float32(4294967296.0), np.float32(4294967295.0)))
    return torch.from_numpy((r >= thr))


# ------------------------------------------------------------------------------------------------
# evaluation: pairwise distances and the cascade second stage   FD/reid/evaluators.py:19-43,76-98,198-227
# ------------------------------------------------------------------------------------------------
def o_pairwise_distance(x, y):
    """:90-98: |x|^2 + |y|^2 - 2 x y^T (addmm_)"""
    m, n = x.size(0), y.size(0)
    dist = torch.pow(x, 2).sum(dim=1, keepdim=True).expand(m, n) + torch.pow(y, 2).sum(dim=1, keepdim=True).expand(n, m).t()
    return dist - 2 * x.mm(y.t())


def o_cascade_second_stage(distmat, probe, gallery, embed_model, rerank_topk, embed_dist_fn=None):
    """CascadeEvaluator.evaluate, :202-225, on feature matrices: per query, score the top-k gallery rows with the embedding
    network (extract_embeddings :19-43: one call per query with a [1, D] probe broadcast against [k, D]), overwrite their
    distances, then push everything outside the top-k behind them."""
    import numpy as np
    distmat = distmat.numpy().copy()
    rank_indices = np.argsort(distmat, axis=1, kind="stable")
    Q = distmat.shape[0]
    embed_model.eval()
    scores = []
    with torch.no_grad():
        for i in range(Q):
            g = gallery[torch.as_tensor(rank_indices[i, :rerank_topk].copy())]
            scores.append(embed_model(probe[i].view(1, -1), g))
    embeddings = torch.cat(scores, 0).view(Q * rerank_topk, -1)
    if embed_dist_fn is not None:
        embeddings = embed_dist_fn(embeddings)
    for k, embed in enumerate(embeddings):
        i, j = k // rerank_topk, k % rerank_topk
        distmat[i, rank_indices[i, j]] = float(embed)
    for i, indices in enumerate(rank_indices):
        bar = max(distmat[i][indices[:rerank_topk]])
        gap = max(bar + 1. - distmat[i, indices[rerank_topk]], 0)
        if gap > 0:
            distmat[i][indices[rerank_topk:]] += gap
    return distmat
