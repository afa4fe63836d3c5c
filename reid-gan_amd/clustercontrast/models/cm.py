"""Cluster memory — restates CC/clustercontrast/models/cm.py:9-193 on the HIP kernels.

CM / CM_Hard / CM_gan keep the reference's autograd.Function signatures: forward(ctx, inputs, targets, features,
momentum) = inputs @ features^T on the MFMA GEMM; backward = grad_outputs @ features with the PRE-update bank
(:25-26) and THEN the momentum update of the bank rows, in batch order, in place (:29-31) — one HIP launch
(rg_cm_update) instead of the Python loop of ~6 kernels per sample.  `features` is read at call time and mutated
in place, exactly like `ctx.features` in the reference, so callers may re-assign `memory.features` every epoch.

Under torch.distributed every rank holds a replica of the bank: the (normalised) batch features and labels are
all-gathered before the update so all replicas apply the identical sequential update; the input
gradient stays local.
"""
from __future__ import absolute_import

from abc import ABC

import torch
from torch import nn, autograd

from rg_hip import functional as RF
from rg_hip import ops
from rg_hip.parallel import all_gather_rows_async, world_size


def _momentum_value(m):
    return float(m.reshape(-1)[0].item()) if isinstance(m, torch.Tensor) else float(m)


class _Gather(object):
    """Batch features + labels of every rank, in rank order.  Issued in the autograd Function's FORWARD (the values
    are final there: the normalised batch features) on the process group's stream and collected right before
    rg_cm_update in backward, so the collective hides behind the loss / dgrad instead of blocking the compute stream
    (reference update: CC/clustercontrast/models/cm.py:22-33)."""
    __slots__ = ("hx", "hy")

    def __init__(self, inputs, targets):
        if world_size() == 1:
            self.hx = self.hy = None
        else:
            self.hx = all_gather_rows_async(inputs.detach())
            self.hy = all_gather_rows_async(targets)

    def wait(self, inputs, targets):
        if self.hx is None:
            return inputs, targets
        return self.hx.wait(), self.hy.wait()


class CM(autograd.Function):

    @staticmethod
    def forward(ctx, inputs, targets, features, momentum):
        ctx.features = features
        ctx.momentum = _momentum_value(momentum)
        ctx.save_for_backward(inputs, targets)
        ctx.gather = _Gather(inputs, targets)
        return ops.linear_fwd(inputs, features)

    @staticmethod
    def backward(ctx, grad_outputs):
        inputs, targets = ctx.saved_tensors
        grad_inputs = None
        if ctx.needs_input_grad[0]:
            grad_inputs = ops.linear_dgrad(grad_outputs, ctx.features)
        xs, ys = ctx.gather.wait(inputs, targets)
        ops.cm_update(xs, ys, ctx.features, ctx.momentum)           # after dgrad: it used the pre-update bank
        return grad_inputs, None, None, None


def cm(inputs, indexes, features, momentum=0.5):
    return CM.apply(inputs, indexes, features, momentum)


class CM_Hard(autograd.Function):

    @staticmethod
    def forward(ctx, inputs, targets, features, momentum):
        ctx.features = features
        ctx.momentum = _momentum_value(momentum)
        ctx.save_for_backward(inputs, targets)
        ctx.gather = _Gather(inputs, targets)
        return ops.linear_fwd(inputs, features)

    @staticmethod
    def backward(ctx, grad_outputs):
        inputs, targets = ctx.saved_tensors
        grad_inputs = None
        if ctx.needs_input_grad[0]:
            grad_inputs = ops.linear_dgrad(grad_outputs, ctx.features)
        xs, ys = ctx.gather.wait(inputs, targets)
        ops.cm_update(xs, ys, ctx.features, ctx.momentum, hard=True)
        return grad_inputs, None, None, None


def cm_hard(inputs, indexes, features, momentum=0.5):
    return CM_Hard.apply(inputs, indexes, features, momentum)


class CM_gan(autograd.Function):

    @staticmethod
    def forward(ctx, inputs, gan_inputs, targets, features, gan_features, momentum):
        ctx.features = features
        ctx.gan_features = gan_features
        ctx.momentum = _momentum_value(momentum)
        ctx.save_for_backward(inputs, gan_inputs, targets)
        ctx.gather = _Gather(inputs, targets)
        ctx.gather_gan = _Gather(gan_inputs, targets)
        return ops.linear_fwd(inputs, features)

    @staticmethod
    def backward(ctx, grad_outputs):
        inputs, gan_inputs, targets = ctx.saved_tensors
        grad_inputs = None
        if ctx.needs_input_grad[0]:
            grad_inputs = ops.linear_dgrad(grad_outputs, ctx.features)
        xs, ys = ctx.gather.wait(inputs, targets)
        ops.cm_update(xs, ys, ctx.features, ctx.momentum)
        gs, _ = ctx.gather_gan.wait(gan_inputs, targets)
        ops.cm_update(gs, ys, ctx.gan_features, ctx.momentum, normalize_eps=True)     # F.normalize flavour (:103)
        return grad_inputs, None, None, None, None, None


def cm_gan(inputs, gan_inputs, indexes, features, gan_features, momentum=0.5):
    return CM_gan.apply(inputs, gan_inputs, indexes, features, gan_features, momentum)


class ClusterMemory(nn.Module, ABC):
    def __init__(self, num_features, num_samples, temp=0.05, momentum=0.2, use_hard=False, use_conf=False):
        super(ClusterMemory, self).__init__()
        self.num_features = num_features
        self.num_samples = num_samples

        self.momentum = momentum
        self.temp = temp
        self.use_hard = use_hard

        self.register_buffer('features', torch.zeros(num_samples, num_features))
        self.register_buffer('gan_features', torch.zeros(num_samples, num_features))

    def forward(self, inputs, targets, gan_inputs=None, conf_weight=None):
        if isinstance(inputs, (tuple, list)):          # the encoder's train-mode tuple
            inputs = inputs[0]
        inputs = RF.normalize_rows(inputs)
        if not self.features.is_contiguous():
            self.features = self.features.contiguous()
        if self.use_hard:
            outputs = cm_hard(inputs, targets, self.features, self.momentum)
        else:
            outputs = cm(inputs, targets, self.features, self.momentum)
        # outputs /= temp and F.cross_entropy(..., reduction="none") in one kernel (:134-135)
        return RF.cross_entropy_rows(outputs, targets, 1.0 / self.temp)


class ClusterMemory_Gradient(nn.Module, ABC):
    """Learnable centroids trained by their own SGD (cm.py:140-193)."""

    def __init__(self, num_features, num_samples, temp=0.05):
        super(ClusterMemory_Gradient, self).__init__()
        self.num_features = num_features
        self.num_samples = num_samples
        self.temp = temp
        self.normed_clusters = None

    def set_clusters(self, clusters, cluster_lr):
        from rg_hip import optim as roptim
        self.trainable_clusters = clusters.detach().clone().requires_grad_(True)
        self.optimizer_cluster = roptim.SGD([self.trainable_clusters], lr=cluster_lr)
        # attached, as the reference's F.normalize(self.trainable_clusters) (:150,193): losses built on
        # `normed_clusters` reach trainable_clusters.grad; forward() itself detaches it (:163)
        self.normed_clusters = RF.normalize_rows(self.trainable_clusters)

    def forward(self, inputs, targets, ex_f=None):
        inputs = RF.normalize_rows(inputs)
        outputs = _MatmulT.apply(inputs, self.normed_clusters.detach())
        if ex_f is not None:
            ex_f = RF.normalize_rows(ex_f)
            outputs_ex = _MatmulT.apply(inputs, ex_f)
            group_size = outputs_ex.shape[0] // outputs_ex.shape[1]
            mask = (-10000.0 * torch.eye(ex_f.shape[0], device=inputs.device)).repeat_interleave(group_size, dim=0)
            outputs_ex = _AddConst.apply(outputs_ex, mask)
            outputs = _Cat2.apply(outputs, outputs_ex)
        return RF.cross_entropy(outputs, targets, 1.0 / self.temp)

    def update_clusters(self, p_ids, eps=1e-16):
        g = self.trainable_clusters.grad
        if g is None:
            raise RuntimeError("ClusterMemory_Gradient.update_clusters: trainable_clusters has no gradient — back-"
                               "propagate a loss built on `normed_clusters` (the forward() logits are detached, "
                               "CC/clustercontrast/models/cm.py:163) before calling it")
        ids = torch.as_tensor(p_ids, dtype=torch.int64).to(g.device).reshape(-1).contiguous()
        ops.normalize_listed_rows(g, ids, eps)
        self.optimizer_cluster.step()
        self.optimizer_cluster.zero_grad()
        self.normed_clusters = RF.normalize_rows(self.trainable_clusters)


class _MatmulT(autograd.Function):
    """a @ b^T with gradients for both operands (MFMA GEMM kernels)."""

    @staticmethod
    def forward(ctx, a, b):
        ctx.save_for_backward(a, b)
        return ops.linear_fwd(a, b)

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        da = ops.linear_dgrad(g, b) if ctx.needs_input_grad[0] else None
        db = ops.linear_wgrad(a, g) if ctx.needs_input_grad[1] else None
        return da, db


class _AddConst(autograd.Function):
    @staticmethod
    def forward(ctx, x, c):
        return ops.add(x, c)

    @staticmethod
    def backward(ctx, g):
        return g, None


class _Cat2(autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        ctx.ca = a.shape[1]
        return ops.cat_channels([a, b])

    @staticmethod
    def backward(ctx, g):
        return ops.slice_channels(g, 0, ctx.ca), ops.slice_channels(g, ctx.ca, g.shape[1])
