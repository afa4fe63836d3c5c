"""EltwiseSubEmbed — restates FD-GAN-master/reid/models/embedding.py:7-39: (x1-x2)^2|abs -> BN1d -> Linear."""
from __future__ import absolute_import

import torch

from rg_hip import nn as rnn
from rg_hip import ops
from rg_hip.tape import RGModule


class EltwiseSubEmbed(RGModule):
    def __init__(self, nonlinearity='square', use_batch_norm=False,
                 use_classifier=False, num_features=0, num_classes=0):
        super(EltwiseSubEmbed, self).__init__()
        self.nonlinearity = nonlinearity
        if nonlinearity is not None and nonlinearity not in ['square', 'abs']:
            raise KeyError("Unknown nonlinearity:", nonlinearity)
        if nonlinearity != 'square':
            raise NotImplementedError("rg_hip EltwiseSubEmbed: only nonlinearity='square' (the FD-GAN setting, "
                                      "FD/fdgan/model.py:43,47) has a HIP kernel")
        self.use_batch_norm = use_batch_norm
        self.use_classifier = use_classifier
        if self.use_batch_norm:
            self.bn = rnn.BatchNorm1d(num_features)
            self.bn.weight.data.fill_(1)
            self.bn.bias.data.zero_()
        if self.use_classifier:
            assert num_features > 0 and num_classes > 0
            self.classifier = rnn.Linear(num_features, num_classes)
            self.classifier.weight.data.normal_(0, 0.001)
            self.classifier.bias.data.zero_()

    def tf(self, tape, x1, x2):
        x1 = x1.reshape(x1.size(0), -1)
        x2 = x2.reshape(x2.size(0), -1)
        tape.push((x1, x2))
        x = ops.sub_square_fwd(x1, x2)
        if self.use_batch_norm:
            x = self.bn.tf(tape, x)
        if self.use_classifier:
            x = self.classifier.tf(tape, x)
        else:
            tape.push(x.shape)
            ones = ops.fill_(torch.empty(x.shape[1], 1, dtype=torch.float32, device=x.device), 1.0)
            x = ops.linear_fwd(x, ones.view(1, -1)).view(-1)      # x.sum(1)
        return x

    def tb(self, tape, dy, need_dx=True):
        if self.use_classifier:
            d = self.classifier.tb(tape, dy)
        else:
            shape = tape.pop()
            ones = ops.fill_(torch.empty(1, shape[1], dtype=torch.float32, device=dy.device), 1.0)
            d = ops.linear_dgrad(dy.reshape(-1, 1), ones)
        if self.use_batch_norm:
            d = self.bn.tb(tape, d)
        x1, x2 = tape.pop()
        return ops.sub_square_bwd(x1, x2, d)
