"""Losses of the dual_gan models on the HIP kernels.

Mirror of CC/dual_gan/models/external_function.py:14-69 (`GANLoss`).  `lsgan` — the default and only mode the
training scripts run (`--gan_mode lsgan`, examples/options/train_options.py:28) — is built: one fused
(x - label)^2 reduction kernel for the discriminator form, an element-wise kernel for the `reduction='none'` generator
form.  `VGGLoss` / `VGG19` (:107-347) need torchvision's pretrained VGG-19 download; the scripts are run with
`--no_vgg_loss` on the target machines (no network), so constructing one raises.
"""
from __future__ import absolute_import

import torch
from torch import nn

from rg_hip import functional as RF
from rg_hip import ops


class _SquareDiffConst(torch.autograd.Function):
    """(x - c)^2 element-wise (nn.MSELoss(reduction='none') against a constant label)."""

    @staticmethod
    def forward(ctx, x, c):
        t = ops.fill_(torch.empty_like(x), float(c))
        ctx.save_for_backward(x, t)
        return ops.sub_square_fwd(x, t)

    @staticmethod
    def backward(ctx, g):
        x, t = ctx.saved_tensors
        dx, _ = ops.sub_square_bwd(x, t, g.contiguous(), need_a=True, need_b=False)
        return dx, None


class GANLoss(nn.Module):
    """Define different GAN objectives (external_function.py:14-69)."""

    def __init__(self, gan_mode, target_real_label=1.0, target_fake_label=0.0):
        super(GANLoss, self).__init__()
        self.register_buffer('real_label', torch.tensor(target_real_label))
        self.register_buffer('fake_label', torch.tensor(target_fake_label))
        self._real, self._fake = float(target_real_label), float(target_fake_label)
        self.gan_mode = gan_mode
        if gan_mode == 'lsgan':
            self.loss = None
        elif gan_mode in ('vanilla', 'hinge', 'wgangp'):
            raise NotImplementedError("gan mode %s has no HIP kernels yet (the training scripts run lsgan)" % gan_mode)
        else:
            raise NotImplementedError('gan mode %s not implemented' % gan_mode)

    def label(self, target_is_real):
        return self._real if target_is_real else self._fake

    def __call__(self, prediction, target_is_real, is_disc=False):
        if is_disc:
            return RF.mse_const(prediction, self.label(target_is_real))
        return _SquareDiffConst.apply(prediction, self.label(target_is_real))


class VGGLoss(nn.Module):
    def __init__(self, *args, **kwargs):
        super(VGGLoss, self).__init__()
        raise NotImplementedError("VGGLoss needs torchvision's pretrained VGG-19 weights (a network download, "
                                  "external_function.py:229); run with --no_vgg_loss")
