"""Cluster-contrast ResNet — restates CC/clustercontrast/models/resnet.py:14-147 on the HIP tape runtime.

torchvision-layout trunk with layer4 stride forced to 1 (:34-35) registered as `base = Sequential(conv1, bn1, relu,
maxpool, layer1..4)` (:36-38, so checkpoints use `base.0.weight`, `base.4.0.conv1.weight`, ...), pooling from the
factory (:40), `feat_bn` with frozen bias (:60-61); in train mode with num_classes == 0 the forward returns the
tuple (bn_x, F.normalize(gan_x, dim=1)) (:96-107), in eval mode the L2-normalised embedding (:90-94).
"""
from __future__ import absolute_import

from torch.nn import init

from rg_hip import nn as rnn
from rg_hip import ops
from rg_hip.resnet_trunk import TVResNet, load_pretrained, trunk_tb, trunk_tf
from rg_hip.tape import RGModule

from .pooling import build_pooling_layer

__all__ = ['ResNet', 'resnet18', 'resnet34', 'resnet50', 'resnet101', 'resnet152']


class ResNet(RGModule):
    _depths = (18, 34, 50, 101, 152)

    def __init__(self, depth, pretrained=True, cut_at_pooling=False,
                 num_features=0, norm=False, dropout=0, num_classes=0, pooling_type='avg'):
        print('pooling_type: {}'.format(pooling_type))
        super(ResNet, self).__init__()
        self.pretrained = pretrained
        self.depth = depth
        self.cut_at_pooling = cut_at_pooling
        if depth not in ResNet._depths:
            raise KeyError("Unsupported depth:", depth)
        resnet = TVResNet(depth)
        if pretrained:
            load_pretrained(resnet, depth)
        resnet.layer4[0].conv2.stride = (1, 1)
        resnet.layer4[0].downsample[0].stride = (1, 1)
        self.base = rnn.Sequential(resnet.conv1, resnet.bn1, resnet.relu, resnet.maxpool,
                                   resnet.layer1, resnet.layer2, resnet.layer3, resnet.layer4)
        self.gap = build_pooling_layer(pooling_type)

        if not self.cut_at_pooling:
            self.num_features = num_features
            self.norm = norm
            self.dropout = dropout
            self.has_embedding = num_features > 0
            self.num_classes = num_classes
            out_planes = resnet.fc.in_features
            if self.has_embedding:
                self.feat = rnn.Linear(out_planes, self.num_features)
                self.feat_bn = rnn.BatchNorm1d(self.num_features)
                init.kaiming_normal_(self.feat.weight, mode='fan_out')
                init.constant_(self.feat.bias, 0)
            else:
                self.num_features = out_planes
                self.feat_bn = rnn.BatchNorm1d(self.num_features)
            self.feat_bn.bias.requires_grad_(False)
            if self.dropout > 0:
                self.drop = rnn.Dropout(self.dropout)
            if self.num_classes > 0:
                self.classifier = rnn.Linear(self.num_features, self.num_classes, bias=False)
                init.normal_(self.classifier.weight, std=0.001)
            init.constant_(self.feat_bn.weight, 1)
            init.constant_(self.feat_bn.bias, 0)

        if not pretrained:
            self.reset_params()

    def forward(self, x, test_all=False):
        self._test_all = bool(test_all)
        return super(ResNet, self).forward(x)

    # ---- tape program ---------------------------------------------------------------------------
    def tf(self, tape, x):
        test_all = getattr(self, "_test_all", False)
        fmap = trunk_tf(tape, list(self.base), x)
        x = self.gap.tf(tape, fmap).reshape(fmap.shape[0], -1)
        if self.cut_at_pooling:
            tape.push(("cut",))
            return x
        if self.has_embedding:
            x = self.feat.tf(tape, x)
        relu_after = self.training and (not self.norm) and self.has_embedding
        bn_x = self.feat_bn.tf(tape, x, act=ops.ACT_RELU if relu_after else ops.ACT_NONE)
        if not self.training:
            y, nrm = ops.l2norm_rows_fwd(bn_x)
            gan = self._normalize_map(fmap)[0] if test_all else None
            tape.push(("eval", y, nrm, fmap.shape))
            return (y, gan) if test_all else y
        mode = ["train"]
        if self.norm:
            y, nrm = ops.l2norm_rows_fwd(bn_x)
            mode += [y, nrm]
            bn_x = y
        else:
            mode += [None, None]
        if self.dropout > 0:
            bn_x = self.drop.tf(tape, bn_x)
        if self.num_classes > 0:
            out = self.classifier.tf(tape, bn_x)
            tape.push(tuple(mode) + (None, fmap.shape))
            return out
        gan, gnorm = self._normalize_map(fmap)
        tape.push(tuple(mode) + ((gan, gnorm), fmap.shape))
        return bn_x, gan

    @staticmethod
    def _normalize_map(fmap):
        """F.normalize(gan_x, dim=1) for [N, C, H, W]: unit L2 norm over the channels at every pixel, in place of layout (one
        kernel on the NCHW map; a row view would cost two permuted copies of the 2048-channel map per step)."""
        return ops.l2norm_channels_fwd(fmap)

    def tb(self, tape, d_bn, d_gan=None, need_dx=True):
        rec = tape.pop()
        d_fmap_extra = None
        if rec[0] == "cut":
            dy = d_bn
        elif rec[0] == "eval":
            _, y, nrm, fshape = rec
            dy = ops.l2norm_rows_bwd(y, d_bn, nrm)
            if d_gan is not None:
                raise RuntimeError("clustercontrast ResNet: no backward through the eval-mode feature map output")
            dy = self.feat_bn.tb(tape, dy)
            if self.has_embedding:
                dy = self.feat.tb(tape, dy)
        else:
            _, y, nrm, gan_rec, fshape = rec
            dy = d_bn
            if self.num_classes > 0:
                dy = self.classifier.tb(tape, dy)
            if self.dropout > 0:
                dy = self.drop.tb(tape, dy)
            if y is not None:
                dy = ops.l2norm_rows_bwd(y, dy, nrm)
            dy = self.feat_bn.tb(tape, dy)
            if self.has_embedding:
                dy = self.feat.tb(tape, dy)
            if d_gan is not None and gan_rec is not None:
                gan, gnorm = gan_rec
                d_fmap_extra = ops.l2norm_channels_bwd(gan, d_gan, gnorm)
        d_fmap = self.gap.tb(tape, dy.reshape(dy.shape[0], -1, 1, 1))
        if d_fmap_extra is not None:
            d_fmap = ops.add(d_fmap, d_fmap_extra)
        return trunk_tb(tape, list(self.base), d_fmap, need_dx)

    def reset_params(self):
        for m in self.modules():
            if isinstance(m, rnn.Conv2d):
                init.kaiming_normal_(m.weight, mode='fan_out')
                if m.bias is not None:
                    init.constant_(m.bias, 0)
            elif isinstance(m, (rnn.BatchNorm2d, rnn.BatchNorm1d)):
                init.constant_(m.weight, 1)
                init.constant_(m.bias, 0)
            elif isinstance(m, rnn.Linear):
                init.normal_(m.weight, std=0.001)
                if m.bias is not None:
                    init.constant_(m.bias, 0)


def resnet18(**kwargs):
    return ResNet(18, **kwargs)


def resnet34(**kwargs):
    return ResNet(34, **kwargs)


def resnet50(**kwargs):
    return ResNet(50, **kwargs)


def resnet101(**kwargs):
    return ResNet(101, **kwargs)


def resnet152(**kwargs):
    return ResNet(152, **kwargs)
