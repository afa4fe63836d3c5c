"""GANLoss — restates FD-GAN-master/fdgan/losses.py:12-32 (sigmoid + BCE against a constant, optionally
label-smoothed, target) as one fused reduction kernel; the full-size target tensor is never built."""
from __future__ import absolute_import

import random

import torch
import torch.nn as nn

from rg_hip import functional as RF


class GANLoss(nn.Module):
    def __init__(self, smooth=False):
        super(GANLoss, self).__init__()
        self.smooth = smooth

    def get_target_value(self, target_is_real):
        real_label = 1.0
        fake_label = 0.0
        if self.smooth:                       # same two draws, in the same order, as the reference (:20-22)
            real_label = random.uniform(0.7, 1.0)
            fake_label = random.uniform(0.0, 0.3)
        return real_label if target_is_real else fake_label

    def get_target_tensor(self, input, target_is_real):
        return torch.zeros_like(input).fill_(self.get_target_value(target_is_real))

    def __call__(self, input, target_is_real):
        return RF.sigmoid_bce_const(input, self.get_target_value(target_is_real))
