"""Losses of the dual_gan models on the HIP kernels.

Mirror of CC/dual_gan/models/external_function.py:14-69 (`GANLoss`).  All four modes are built: `lsgan` (the default of the
training scripts, examples/options/train_options.py:28: one fused (x - label)^2 reduction for the discriminator form, an
element-wise kernel for the `reduction='none'` generator form), `vanilla` (fused sigmoid-BCE), `hinge` and `wgangp`
(`rg_affine_relu_mean_*`; the wgangp gradient PENALTY of `cal_gradient_penalty` needs a double backward and is not built).  `VGGLoss` / `VGG19` (:107-347) need torchvision's pretrained VGG-19 download; the scripts are run with
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
        if gan_mode not in ('lsgan', 'vanilla', 'hinge', 'wgangp'):
            raise NotImplementedError('gan mode %s not implemented' % gan_mode)
        self.loss = None

    def label(self, target_is_real):
        return self._real if target_is_real else self._fake

    def __call__(self, prediction, target_is_real, is_disc=False):
        if self.gan_mode == 'lsgan':
            if is_disc:
                return RF.mse_const(prediction, self.label(target_is_real))
            return _SquareDiffConst.apply(prediction, self.label(target_is_real))
        if self.gan_mode == 'vanilla':                       # BCEWithLogitsLoss (mean) in both forms (:55-57)
            return RF.sigmoid_bce_const(prediction, self.label(target_is_real))
        if is_disc:                                          # hinge / wgangp, discriminator form (:58-66)
            sign = -1.0 if target_is_real else 1.0
            if self.gan_mode == 'hinge':
                return RF.affine_relu_mean(prediction, 1.0, sign, clamp=True)
            return RF.affine_relu_mean(prediction, 0.0, sign, clamp=False)
        return RF.affine_relu_mean(prediction, 0.0, -1.0, clamp=False)          # generator form: -mean(D(fake)) (:67-68)


class VGGLoss(nn.Module):
    def __init__(self, *args, **kwargs):
        super(VGGLoss, self).__init__()
        raise NotImplementedError("VGGLoss needs torchvision's pretrained VGG-19 weights (a network download, "
                                  "external_function.py:229); run with --no_vgg_loss")
