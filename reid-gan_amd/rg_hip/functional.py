"""autograd entry points for the loss kernels (each a single fused forward reduction and a single
backward launch).  Scalar losses stay on the device; upstream scalar gradients are read by the
backward kernels from device memory, so nothing here synchronises with the host."""
from __future__ import absolute_import

import torch

from . import ops


_CONST = {}


def const_vector(values, device):
    """a cached fp32 device vector of compile-time constants (loss weights): one host->device copy per distinct tuple, none on
    the step path afterwards (pageable H2D copies are neither asynchronous nor capturable into a hipGraph)"""
    key = (values, device.index)
    t = _CONST.get(key)
    if t is None:
        t = _CONST[key] = torch.tensor(values, dtype=torch.float32).to(device)
    return t


def _scalar_grad(g):
    """a 0-dim upstream gradient as a 1-element contiguous device tensor (or None)."""
    if g is None:
        return None
    return g.reshape(1).contiguous()


class _SigmoidBCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, target):
        ctx.save_for_backward(x)
        ctx.target = float(target)
        return ops.sigmoid_bce_fwd(x, target)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return ops.sigmoid_bce_bwd(x, _scalar_grad(g), ctx.target), None


def sigmoid_bce_const(x, target):
    """mean(binary_cross_entropy(sigmoid(x), full_like(x, target))) — GANLoss, FD/fdgan/losses.py:29-32."""
    return _SigmoidBCE.apply(x, target)


class _MSEConst(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, target):
        ctx.save_for_backward(x)
        ctx.target = float(target)
        return ops.mse_const_fwd(x, target)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return ops.mse_const_bwd(x, _scalar_grad(g), ctx.target), None


def mse_const(x, target):
    return _MSEConst.apply(x, target)


class _L1(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, row_labels):
        out2 = ops.l1_fwd(a, b, row_labels)
        ctx.save_for_backward(a, b, out2)
        ctx.row_labels = row_labels
        return out2[0]

    @staticmethod
    def backward(ctx, g):
        a, b, out2 = ctx.saved_tensors
        need_a, need_b = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        da, db = ops.l1_bwd(a, b, ctx.row_labels, _scalar_grad(g), out2, need_a, need_b)
        return da, db, None


def l1_loss(a, b, row_labels=None):
    """F.l1_loss(a, b); with row_labels (int64 [rows]) only rows whose label == 1 take part, i.e.
    F.l1_loss(a[mask], b[mask]) of FD/fdgan/model.py:191-194 without materialising the gather."""
    return _L1.apply(a, b, row_labels)


class _L1Rows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        ctx.save_for_backward(a, b)
        return ops.l1_rows_fwd(a, b)

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        da, db = ops.l1_rows_bwd(a, b, g.contiguous(), ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        return da, db


def l1_loss_rows(a, b):
    """nn.L1Loss(reduction='none')(a, b).flatten(1).mean(-1): one mean absolute difference per sample"""
    return _L1Rows.apply(a, b)


class _MSEConstRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, target):
        ctx.save_for_backward(x)
        ctx.target = float(target)
        return ops.mse_const_rows_fwd(x, target)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return ops.mse_const_rows_bwd(x, ctx.target, g.contiguous()), None


def mse_const_rows(x, target):
    """((x - target) ** 2).flatten(1).mean(-1)"""
    return _MSEConstRows.apply(x, target)


class _SoftmaxCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels, scale):
        loss, lse = ops.softmax_ce_fwd(logits, labels, scale)
        ctx.save_for_backward(logits, labels, lse)
        ctx.scale = scale
        return loss

    @staticmethod
    def backward(ctx, g):
        logits, labels, lse = ctx.saved_tensors
        return ops.softmax_ce_bwd(logits, labels, lse, g, ctx.scale), None, None


def cross_entropy_rows(logits, labels, scale=1.0):
    """F.cross_entropy(logits * scale, labels, reduction='none')."""
    return _SoftmaxCE.apply(logits, labels, float(scale))


class _WeightedSum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, scale):
        ctx.w, ctx.scale, ctx.n = w, scale, x.numel()
        return ops.weighted_sum_fwd(x, w, scale)

    @staticmethod
    def backward(ctx, g):
        return ops.weighted_sum_bwd(_scalar_grad(g), ctx.w, ctx.n, ctx.scale, g.device), None, None


def weighted_mean(x, w=None):
    """(x * w).mean() for a 1-D tensor of per-sample losses (w optional, treated as a constant)."""
    return _WeightedSum.apply(x, w, 1.0 / x.numel())


def cross_entropy(logits, labels, scale=1.0):
    """F.cross_entropy(logits * scale, labels) (mean reduction)."""
    return weighted_mean(cross_entropy_rows(logits, labels, scale))


class _L2NormRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, eps):
        y, nrm = ops.l2norm_rows_fwd(x, eps)
        ctx.save_for_backward(y, nrm)
        ctx.eps = eps
        return y

    @staticmethod
    def backward(ctx, g):
        y, nrm = ctx.saved_tensors
        return ops.l2norm_rows_bwd(y, g, nrm, ctx.eps), None


def normalize_rows(x, eps=1e-12):
    """F.normalize(x, dim=1) for a [rows, D] tensor."""
    return _L2NormRows.apply(x, eps)


class _AffineReluMean(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, a, b, clamp):
        ctx.save_for_backward(x)
        ctx.a, ctx.b, ctx.clamp = float(a), float(b), bool(clamp)
        return ops.affine_relu_mean_fwd(x, ctx.a, ctx.b, ctx.clamp)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return ops.affine_relu_mean_bwd(x, _scalar_grad(g), ctx.a, ctx.b, ctx.clamp), None, None, None


def affine_relu_mean(x, a, b, clamp=True):
    """mean(relu(a + b * x)) (clamp) or mean(a + b * x): hinge / wgangp GAN objectives in one fused reduction."""
    return _AffineReluMean.apply(x, a, b, clamp)
