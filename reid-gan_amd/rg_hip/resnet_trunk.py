"""torchvision-layout ResNet (v1.5: the 3x3 carries the stride) as an RGModule container.

The reference takes its trunks from `torchvision.models.resnet{18..152}` (FD/reid/models/resnet.py:15-33,
CC/clustercontrast/models/resnet.py:16-38); torchvision is a third-party dependency that is not in the
reference tree.  The layout (attribute names => state_dict keys, Bottleneck order, stem, downsample) is
pinned in-tree by CC/clustercontrast/models/resnet_ibn_a.py:70-109 (Bottleneck) and :112-159 (stem,
_make_layer), which this file follows so that `base.conv1.weight`, `base.layer1.0.conv1.weight`,
`base.layer1.0.downsample.0.weight`, `base.fc.weight` ... load unchanged.
"""
from __future__ import absolute_import

import os

import torch

from . import nn as rnn
from .ops import ACT_RELU
from .tape import RGModule


class Bottleneck(RGModule):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super(Bottleneck, self).__init__()
        self.conv1 = rnn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = rnn.BatchNorm2d(planes)
        self.conv2 = rnn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn2 = rnn.BatchNorm2d(planes)
        self.conv3 = rnn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = rnn.BatchNorm2d(planes * 4)
        self.relu = rnn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def tf(self, tape, x):
        o = rnn.conv_bn_tf(tape, self.conv1, self.bn1, x, act=ACT_RELU)
        o = rnn.conv_bn_tf(tape, self.conv2, self.bn2, o, act=ACT_RELU)
        if self.downsample is not None:
            idn = rnn.conv_bn_tf(tape, self.downsample[0], self.downsample[1], x)
        else:
            idn = x
        return rnn.conv_bn_tf(tape, self.conv3, self.bn3, o, residual=idn, act=ACT_RELU)   # conv3 + bn3 + add + relu

    def tb(self, tape, dy, need_dx=True, dy_masked=False, mask_input=False):
        """dy_masked: dy already went through this block's final ReLU backward; mask_input: the block's input is the
        previous block's ReLU output, whose backward is applied in conv1's dgrad epilogue (see nn.conv_bn_tb)."""
        d, d_idn = rnn.conv_bn_tb(tape, self.conv3, self.bn3, dy, dy_masked=dy_masked, mask_input=True)
        if self.downsample is not None:
            d_idn = rnn.conv_bn_tb(tape, self.downsample[0], self.downsample[1], d_idn, need_dx=need_dx)
        d = rnn.conv_bn_tb(tape, self.conv2, self.bn2, d, dy_masked=True, mask_input=True)
        # the skip gradient is added (and the ReLU backward of the block below applied) in the dgrad epilogue of conv1
        return rnn.conv_bn_tb(tape, self.conv1, self.bn1, d, need_dx=need_dx, residual=d_idn if need_dx else None,
                              dy_masked=True, mask_input=mask_input)


class BasicBlock(RGModule):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super(BasicBlock, self).__init__()
        self.conv1 = rnn.Conv2d(inplanes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn1 = rnn.BatchNorm2d(planes)
        self.relu = rnn.ReLU(inplace=True)
        self.conv2 = rnn.Conv2d(planes, planes, 3, padding=1, bias=False)
        self.bn2 = rnn.BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride

    def tf(self, tape, x):
        o = rnn.conv_bn_tf(tape, self.conv1, self.bn1, x, act=ACT_RELU)
        if self.downsample is not None:
            idn = rnn.conv_bn_tf(tape, self.downsample[0], self.downsample[1], x)
        else:
            idn = x
        return rnn.conv_bn_tf(tape, self.conv2, self.bn2, o, residual=idn, act=ACT_RELU)

    def tb(self, tape, dy, need_dx=True, dy_masked=False, mask_input=False):
        d, d_idn = rnn.conv_bn_tb(tape, self.conv2, self.bn2, dy, dy_masked=dy_masked, mask_input=True)
        if self.downsample is not None:
            d_idn = rnn.conv_bn_tb(tape, self.downsample[0], self.downsample[1], d_idn, need_dx=need_dx)
        return rnn.conv_bn_tb(tape, self.conv1, self.bn1, d, need_dx=need_dx, residual=d_idn if need_dx else None,
                              dy_masked=True, mask_input=mask_input)


_CFG = {
    18: (BasicBlock, [2, 2, 2, 2]),
    34: (BasicBlock, [3, 4, 6, 3]),
    50: (Bottleneck, [3, 4, 6, 3]),
    101: (Bottleneck, [3, 4, 23, 3]),
    152: (Bottleneck, [3, 8, 36, 3]),
}


class TVResNet(RGModule):
    """Container with torchvision's attribute names; `tf` runs conv1..layer4 (everything before avgpool)."""

    def __init__(self, depth, num_classes=1000):
        super(TVResNet, self).__init__()
        block, layers = _CFG[depth]
        self.inplanes = 64
        self.conv1 = rnn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = rnn.BatchNorm2d(64)
        self.relu = rnn.ReLU(inplace=True)
        self.maxpool = rnn.MaxPool2d(3, 2, 1)
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
        self.layer3 = self._make_layer(block, 256, layers[2], stride=2)
        self.layer4 = self._make_layer(block, 512, layers[3], stride=2)
        self.avgpool = torch.nn.Identity()          # never executed by the ReID wrappers (they stop before it)
        self.fc = rnn.Linear(512 * block.expansion, num_classes)
        for m in self.modules():                    # torchvision's default init
            if isinstance(m, rnn.Conv2d):
                torch.nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")

    def _make_layer(self, block, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = rnn.Sequential(
                rnn.Conv2d(self.inplanes, planes * block.expansion, 1, stride=stride, bias=False),
                rnn.BatchNorm2d(planes * block.expansion))
        layers = [block(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.inplanes, planes))
        return rnn.Sequential(*layers)

    def trunk_modules(self):
        return [self.conv1, self.bn1, self.relu, self.maxpool, self.layer1, self.layer2, self.layer3, self.layer4]

    def tf(self, tape, x):
        return trunk_tf(tape, self.trunk_modules(), x)

    def tb(self, tape, dy, need_dx=True):
        return trunk_tb(tape, self.trunk_modules(), dy, need_dx)


def _conv_bn_pairs(mods):
    pairs = [(mods[0], mods[1])]
    for layer in mods[4:]:
        for blk in layer:
            pairs.append((blk.conv1, blk.bn1))
            pairs.append((blk.conv2, blk.bn2))
            if hasattr(blk, "conv3"):
                pairs.append((blk.conv3, blk.bn3))
            if blk.downsample is not None:
                pairs.append((blk.downsample[0], blk.downsample[1]))
    return pairs


def trunk_tf(tape, mods, x):
    conv1, bn1, _relu, maxpool = mods[0], mods[1], mods[2], mods[3]
    if not bn1.training:
        # frozen statistics: fold every BatchNorm of the trunk into its convolution with one launch per weight version
        grp = getattr(conv1, "_rg_fold_group", None)
        if grp is None:
            grp = conv1._rg_fold_group = rnn.FoldGroup(_conv_bn_pairs(mods))
        if grp.usable():
            grp.prepare()
    x = rnn.conv_bn_tf(tape, conv1, bn1, x, act=ACT_RELU)
    x = maxpool.tf(tape, x)
    for layer in mods[4:]:
        x = layer.tf(tape, x)
    return x


def trunk_tb(tape, mods, dy, need_dx=True):
    """Backward program of the trunk.  `mods[0]._rg_stage_hook(tape, params)` — set by rg_hip.parallel.attach_stage_hooks — is called
    as each stage (layer4 .. layer1, then the stem) has launched its last weight gradient: the data-parallel all-reduce of that
    stage's range of the gradient arena starts there and runs under the backward of the stages below (the arena keeps a stage's
    parameters adjacent, so a stage is one element range)."""
    conv1, bn1, _relu, maxpool = mods[0], mods[1], mods[2], mods[3]
    hook = getattr(conv1, "_rg_stage_hook", None)
    layers = [list(layer) for layer in mods[4:]]
    nblk = sum(len(layer) for layer in layers)
    i = nblk
    for li in range(len(layers) - 1, -1, -1):
        for blk in reversed(layers[li]):
            i -= 1
            # every block's input except the first one's (the max-pool output) is the previous block's ReLU output: its
            # backward runs in this block's conv1 dgrad epilogue, so the block below receives an already masked gradient
            dy = blk.tb(tape, dy, dy_masked=(i != nblk - 1), mask_input=(i != 0))
        if hook is not None:
            hook(tape, [p for blk in layers[li] for p in blk.parameters()])
    dy = maxpool.tb(tape, dy)
    dx = rnn.conv_bn_tb(tape, conv1, bn1, dy, need_dx=need_dx)
    if hook is not None:
        hook(tape, list(conv1.parameters()) + list(bn1.parameters()))
    return dx


def bn_all_eval(module):
    """True when every BatchNorm under `module` normalises with its running statistics (so samples are
    independent and separate forward calls may be batched without changing any result)."""
    for m in module.modules():
        if isinstance(m, rnn._BatchNorm) and (m.training or not m.track_running_stats):
            return False
    return True


def load_pretrained(model, depth):
    """ImageNet initialisation from a LOCAL file only (there is no network on the target machines):
    $RG_RESNET{depth}_WEIGHTS or the torch hub cache."""
    cand = [os.environ.get("RG_RESNET%d_WEIGHTS" % depth)]
    hub = os.path.join(os.environ.get("TORCH_HOME", os.path.expanduser("~/.cache/torch")), "hub", "checkpoints")
    if os.path.isdir(hub):
        cand += [os.path.join(hub, f) for f in sorted(os.listdir(hub)) if f.startswith("resnet%d-" % depth)]
    for path in cand:
        if path and os.path.exists(path):
            model.load_state_dict(torch.load(path, map_location="cpu"))
            return model
    raise RuntimeError(
        "pretrained=True needs ImageNet weights for resnet%d, which the reference downloads through torchvision; "
        "no network here: point RG_RESNET%d_WEIGHTS at a local torchvision state_dict or pass pretrained=False"
        % (depth, depth))
