"""Losses of the dual_gan models on the HIP kernels.

Mirror of CC/dual_gan/models/external_function.py: `GANLoss` (:14-69, all four modes: `lsgan` — one fused (x - label)^2
reduction for the discriminator form, an element-wise kernel for the `reduction='none'` generator form —, `vanilla`
(fused sigmoid-BCE), `hinge` and `wgangp` (`rg_affine_relu_mean_*`)), `cal_gradient_penalty` (:72-104), `VGGLoss` (:107-147)
and `VGG19` (:226-347).

VGG-19 runs on the conv engine (3x3 convolutions with bias + ReLU in the MFMA epilogue, 2x2 max-pool kernels, Gram matrices
on the batched MFMA GEMM).  The reference downloads torchvision's ImageNet weights in the constructor
(`models.vgg19(pretrained=True)`, :229); there is no network on the target machines, so `VGG19(weights=path)` loads a local
torchvision `vgg19` state_dict (`features.N.weight` keys, or this module's own keys) and `VGG19()` without a path starts
from the default initialisation and says so once.

The wgangp gradient penalty needs d/dW of the discriminator's INPUT gradient.  The discriminator is piecewise linear
(convolutions, LeakyReLU, average pooling), so with v = dP/dg that derivative is <v, g(W)> = <1, J(W) v>: a tangent forward
pass of v through the masked-linear network gives each convolution's input tangent, the backward pass that produced g gives
each convolution's output adjoint, and the second-order weight gradient is the ordinary wgrad kernel on that pair — exactly
what autograd's double backward computes, on the kernels that already exist.
"""
from __future__ import absolute_import

import torch
from torch import nn

from rg_hip import functional as RF
from rg_hip import nn as rnn
from rg_hip import ops
from rg_hip.tape import RGModule, Tape


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


def cal_gradient_penalty(netD, real_data, fake_data, type='mixed', constant=1.0, lambda_gp=10.0, alpha=None):
    """external_function.py:72-104.  Returns (gradient_penalty, gradients); the penalty is an attached 0-dim tensor whose
    backward writes the discriminator's second-order weight gradients.  `alpha` [B, 1] overrides the torch.rand draw of
    :89 (tests)."""
    if lambda_gp <= 0.0:
        return 0.0, None
    net = netD.module if hasattr(netD, "module") else netD
    if not hasattr(net, "gp_forward"):
        raise NotImplementedError("cal_gradient_penalty: %s has no tangent / adjoint program (built for ResDiscriminator)"
                                  % net.__class__.__name__)
    if type == 'real':
        inter = real_data
    elif type == 'fake':
        inter = fake_data
    elif type == 'mixed':
        if alpha is None:
            alpha = torch.rand(real_data.shape[0], 1)
        a = alpha.to(real_data.device, non_blocking=True).view(-1, 1, 1, 1).expand_as(real_data)
        inter = a * real_data + (1 - a) * fake_data
    else:
        raise NotImplementedError('{} not implemented'.format(type))
    params = [p for p in net.parameters() if p.requires_grad]
    return _GradPenalty.apply(net, inter.detach().contiguous(), float(constant), float(lambda_gp), *params)


class _GradPenalty(torch.autograd.Function):
    @staticmethod
    def forward(ctx, net, x, constant, lambda_gp, *params):
        tape = Tape(param_grad=True, record=True, needs_input=[True])
        g, adj = net.gp_forward(tape, x)                      # g = d sum(D(x)) / dx, adj = per-conv output adjoints
        B = x.shape[0]
        gv = g.view(B, -1)
        gp_rows, v = ops.grad_penalty_rows(gv, constant, lambda_gp / B)       # per-sample penalty terms, v = dP/dg
        ctx.net, ctx.tape, ctx.adj, ctx.v, ctx.params = net, tape, adj, v.view(x.shape), params
        ctx.mark_non_differentiable(gv)
        return RF._WeightedSum.apply(gp_rows, None, 1.0).detach(), gv

    @staticmethod
    def backward(ctx, g_pen, _g_grads):
        grads = ctx.net.gp_backward(ctx.tape, ctx.adj, ctx.v, g_pen.reshape(1).contiguous())
        out = []
        for p in ctx.params:
            gp = grads.get(id(p))
            view = getattr(p, "_rg_grad", None)
            if gp is not None and view is not None and p.grad is not None and p.grad.data_ptr() == view.data_ptr():
                ops.axpby(view, gp, 1.0, 1.0, out=view)       # accumulate next to the first-order gradients
                gp = None
            out.append(gp)
        return (None, None, None, None) + tuple(out)


# torchvision vgg19().features: (index, kind, channels)
_VGG_CFG = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 256, 'M', 512, 512, 512, 512, 'M', 512, 512, 512, 512, 'M']
_VGG_SLICES = [('relu1_1', 0, 2), ('relu1_2', 2, 4), ('relu2_1', 4, 7), ('relu2_2', 7, 9), ('relu3_1', 9, 12),
               ('relu3_2', 12, 14), ('relu3_3', 14, 16), ('relu3_4', 16, 18), ('relu4_1', 18, 21), ('relu4_2', 21, 23),
               ('relu4_3', 23, 25), ('relu4_4', 25, 27), ('relu5_1', 27, 30), ('relu5_2', 30, 32), ('relu5_3', 32, 34),
               ('relu5_4', 34, 36)]
_VGG_USED = ('relu1_1', 'relu2_1', 'relu3_1', 'relu4_1', 'relu5_1', 'relu2_2', 'relu3_4', 'relu4_4', 'relu5_2')


class _Slice(nn.Module):
    """one named slice of the feature stack; children are registered under their torchvision feature index"""


class VGG19(RGModule):
    """external_function.py:226-347: torchvision's VGG-19 feature stack cut into 16 named slices (same attribute names and
    state_dict keys, `relu1_1.0.weight` ... `relu5_4.34.bias`), parameters frozen; forward returns the dict of the 16
    activations.  One autograd node: a program of 16 conv3x3+bias+ReLU launches and 4 max-pools."""

    _warned = [False]

    def __init__(self, weights=None):
        super(VGG19, self).__init__()
        feats, cin, idx = [], 3, 0
        for v in _VGG_CFG:
            if v == 'M':
                feats.append(('pool', rnn.MaxPool2d(2, 2)))
                idx += 1
            else:
                feats.append(('conv', rnn.Conv2d(cin, v, 3, 1, 1)))
                feats.append(('relu', rnn.ReLU(inplace=True)))
                cin = v
                idx += 2
        self._plan = []
        for name, a, b in _VGG_SLICES:
            sl = _Slice()
            for i in range(a, b):
                sl.add_module(str(i), feats[i][1])
            setattr(self, name, sl)
            self._plan.append((name, [feats[i] for i in range(a, b) if feats[i][0] != 'relu']))
        if weights is not None:
            self.load_torchvision(weights)
        elif not VGG19._warned[0]:
            VGG19._warned[0] = True
            print("VGG19: no weight file given (the reference downloads torchvision's ImageNet weights, which needs a "
                  "network) - using the default initialisation; pass VGG19(weights=<vgg19 state_dict .pth>)")
        for p in self.parameters():
            p.requires_grad = False
        self._upto = None

    def load_torchvision(self, path_or_state):
        sd = torch.load(path_or_state, map_location="cpu") if isinstance(path_or_state, str) else path_or_state
        own = self.state_dict()
        by_index = {k.split('.', 1)[1]: k for k in own}            # '0.weight' -> 'relu1_1.0.weight'
        picked = {}
        for k, v in sd.items():
            kk = k[len("features."):] if k.startswith("features.") else k
            kk = by_index.get(kk, kk)
            if kk in own and tuple(v.shape) == tuple(own[kk].shape):
                picked[kk] = v
        missing = [k for k in own if k not in picked]
        if missing:
            raise KeyError("VGG19.load_torchvision: %d tensors missing, e.g. %s" % (len(missing), missing[:3]))
        self.load_state_dict(picked)

    def tf(self, tape, x):
        outs, h = [], x
        last = self._upto
        for name, layers in self._plan:
            for kind, mod in layers:
                h = mod.tf(tape, h, act=ops.ACT_RELU) if kind == 'conv' else mod.tf(tape, h)
            outs.append(h)
            if name == last:
                break
        tape.push(len(outs))
        return tuple(outs)

    def tb(self, tape, *dys, need_dx=True):
        n = tape.pop()
        d = None
        for i in range(n - 1, -1, -1):
            dy = dys[i] if i < len(dys) else None
            if dy is not None:
                d = dy.contiguous() if d is None else ops.add(d, dy.contiguous())
            layers = self._plan[i][1]
            for j in range(len(layers) - 1, -1, -1):
                kind, mod = layers[j]
                if d is None:
                    tape.pop()                                  # nothing flows back through the deepest unused slices
                else:
                    d = mod.tb(tape, d, need_dx=(need_dx or i > 0 or j > 0))
        return d

    def forward(self, x):
        outs = super(VGG19, self).forward(x)
        return {name: o for (name, _), o in zip(self._plan, outs)}

    def features(self, x, upto='relu5_2'):
        """{name: activation} up to and including slice `upto` (VGGLoss needs nothing deeper than relu5_2)"""
        self._upto = upto
        try:
            outs = super(VGG19, self).forward(x)
        finally:
            self._upto = None
        return {name: o for (name, _), o in zip(self._plan, outs)}


class _Gram(torch.autograd.Function):
    """G[b] = f f^T / (h w ch), f = x.view(b, ch, h*w) (compute_gram, :121-126) on the batched MFMA GEMM."""

    @staticmethod
    def forward(ctx, x):
        B, C, H, W = x.shape
        L = H * W
        x = x.contiguous()
        G = torch.empty((B, C, C), dtype=torch.float32, device=x.device)
        alpha = 1.0 / (H * W * C)
        ops.bgemm(x, x, G, C, C, L, (L, 1), (1, L), (C, 1), (B, 1), (C * L, 0), (C * L, 0), (C * C, 0), alpha=alpha)
        ctx.save_for_backward(x)
        ctx.alpha = alpha
        return G

    @staticmethod
    def backward(ctx, dG):
        (x,) = ctx.saved_tensors
        B, C, H, W = x.shape
        L = H * W
        dG = dG.contiguous()
        dx = torch.empty_like(x)
        # dF = alpha * (dG + dG^T) F
        ops.bgemm(dG, x, dx, C, L, C, (C, 1), (L, 1), (L, 1), (B, 1), (C * C, 0), (C * L, 0), (C * L, 0), alpha=ctx.alpha)
        ops.bgemm(dG, x, dx, C, L, C, (1, C), (L, 1), (L, 1), (B, 1), (C * C, 0), (C * L, 0), (C * L, 0), alpha=ctx.alpha,
                  beta=1.0)
        return dx


class VGGLoss(nn.Module):
    r"""Perceptual loss, VGG-based (external_function.py:107-147): returns (content_loss, style_loss)."""

    def __init__(self, weights=[1.0, 1.0, 1.0, 1.0, 1.0], vgg_weights=None):
        super(VGGLoss, self).__init__()
        self.add_module('vgg', VGG19(vgg_weights))
        self.weights = weights

    def compute_gram(self, x):
        return _Gram.apply(x)

    def __call__(self, x, y):
        x_vgg = self.vgg.features(x)
        with torch.no_grad():
            y_vgg = self.vgg.features(y.detach())
        terms = []
        for w, k in zip(self.weights, ('relu1_1', 'relu2_1', 'relu3_1', 'relu4_1', 'relu5_1')):
            terms.append((RF.l1_loss(x_vgg[k], y_vgg[k]), w))
        content_loss = _wsum(terms)
        terms = []
        for k in ('relu2_2', 'relu3_4', 'relu4_4', 'relu5_2'):
            terms.append((RF.l1_loss(self.compute_gram(x_vgg[k]), self.compute_gram(y_vgg[k])), 1.0))
        style_loss = _wsum(terms)
        return content_loss, style_loss


def _wsum(terms):
    vals = torch.stack([t for t, _ in terms])
    return RF._WeightedSum.apply(vals, RF.const_vector(tuple(float(w) for _, w in terms), vals.device), 1.0)
