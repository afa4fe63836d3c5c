"""ReID ResNet wrapper — restates FD-GAN-master/reid/models/resnet.py:13-122 on the HIP tape runtime.

`self.base` is the whole torchvision-layout ResNet (including the never-used `fc`, reference :33 and
SURVEY §9.11); forward runs its children up to (not including) `avgpool`, global-average-pools the map
and applies the optional feat / feat_bn / normalize|relu / dropout / classifier head (:65-88).
"""
from __future__ import absolute_import

from torch.nn import init

from rg_hip import nn as rnn
from rg_hip import ops
from rg_hip.resnet_trunk import TVResNet, load_pretrained
from rg_hip.tape import RGModule

__all__ = ['ResNet', 'resnet18', 'resnet34', 'resnet50', 'resnet101', 'resnet152']


class ResNet(RGModule):
    _depths = (18, 34, 50, 101, 152)

    def __init__(self, depth, pretrained=True, cut_at_pooling=False,
                 num_features=0, norm=False, dropout=0, num_classes=0):
        super(ResNet, self).__init__()
        self.depth = depth
        self.pretrained = pretrained
        self.cut_at_pooling = cut_at_pooling
        if depth not in ResNet._depths:
            raise KeyError("Unsupported depth:", depth)
        self.base = TVResNet(depth)
        if pretrained:
            load_pretrained(self.base, depth)

        if not self.cut_at_pooling:
            self.num_features = num_features
            self.norm = norm
            self.dropout = dropout
            self.has_embedding = num_features > 0
            self.num_classes = num_classes
            out_planes = self.base.fc.in_features
            if self.has_embedding:
                self.feat = rnn.Linear(out_planes, self.num_features)
                self.feat_bn = rnn.BatchNorm1d(self.num_features)
                init.kaiming_normal_(self.feat.weight, mode='fan_out')
                init.constant_(self.feat.bias, 0)
                init.constant_(self.feat_bn.weight, 1)
                init.constant_(self.feat_bn.bias, 0)
            else:
                self.num_features = out_planes
            if self.dropout > 0:
                self.drop = rnn.Dropout(self.dropout)
            if self.num_classes > 0:
                self.classifier = rnn.Linear(self.num_features, self.num_classes)
                init.normal_(self.classifier.weight, std=0.001)
                init.constant_(self.classifier.bias, 0)

        if not self.pretrained:
            self.reset_params()

    # ---- tape program -------------------------------------------------------------------------
    def tf(self, tape, x):
        fmap = self.base.tf(tape, x)
        tape.push(fmap.shape)
        x = ops.global_avgpool_fwd(fmap)
        if self.cut_at_pooling:
            return x
        if self.has_embedding:
            x = self.feat_bn.tf(tape, self.feat.tf(tape, x),
                                act=ops.ACT_NONE if self.norm else ops.ACT_RELU)
        if self.norm:
            y, nrm = ops.l2norm_rows_fwd(x)
            tape.push((y, nrm))
            x = y
        if self.dropout > 0:
            x = self.drop.tf(tape, x)
        if self.num_classes > 0:
            x = self.classifier.tf(tape, x)
        return x

    def tb(self, tape, dy, need_dx=True):
        if not self.cut_at_pooling:
            if self.num_classes > 0:
                dy = self.classifier.tb(tape, dy)
            if self.dropout > 0:
                dy = self.drop.tb(tape, dy)
            if self.norm:
                y, nrm = tape.pop()
                dy = ops.l2norm_rows_bwd(y, dy, nrm)
            if self.has_embedding:
                dy = self.feat.tb(tape, self.feat_bn.tb(tape, dy))
        shape = tape.pop()
        return self.base.tb(tape, ops.global_avgpool_bwd(dy, shape), need_dx=need_dx)

    def reset_params(self):
        for m in self.modules():
            if isinstance(m, rnn.Conv2d):
                init.kaiming_normal_(m.weight, mode='fan_out')
                if m.bias is not None:
                    init.constant_(m.bias, 0)
            elif isinstance(m, rnn.BatchNorm2d):
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
