"""Pooling layers of the cluster-contrast ResNet — restates CC/clustercontrast/models/pooling.py for the entries
the hot path uses: 'avg' (AdaptiveAvgPool2d(1), :193-195) and 'gem' (GeneralizedMeanPoolingP, :57-103).
Both return [N, C, 1, 1] like the reference."""
from __future__ import absolute_import

import torch
from torch import nn

from rg_hip import ops
from rg_hip.tape import RGModule

__all__ = ["GeneralizedMeanPooling", "GeneralizedMeanPoolingP", "AdaptiveAvgPool2d", "avg_pooling",
           "build_pooling_layer", "pooling_names"]


class AdaptiveAvgPool2d(RGModule):
    def __init__(self, output_size=1):
        super(AdaptiveAvgPool2d, self).__init__()
        assert output_size == 1
        self.output_size = output_size

    def tf(self, tape, x):
        tape.push(x.shape)
        return ops.global_avgpool_fwd(x).view(x.shape[0], x.shape[1], 1, 1)

    def tb(self, tape, dy, need_dx=True):
        shape = tape.pop()
        return ops.global_avgpool_bwd(dy.reshape(shape[0], shape[1]), shape)


class GeneralizedMeanPooling(RGModule):
    """f(X) = (mean(clamp(X, eps)^p))^(1/p) over the whole map."""

    def __init__(self, norm, output_size=1, eps=1e-6):
        super(GeneralizedMeanPooling, self).__init__()
        assert norm > 0 and output_size == 1
        self.p = float(norm)
        self.output_size = output_size
        self.eps = eps

    def _p_tensor(self, dev):
        if isinstance(self.p, torch.Tensor):
            return self.p
        return ops.fill_(torch.empty(1, dtype=torch.float32, device=dev), self.p)

    def tf(self, tape, x):
        p = self._p_tensor(x.device)
        y = ops.gem_pool_fwd(x, p, self.eps)
        tape.push((x, p, y))
        return y.view(x.shape[0], x.shape[1], 1, 1)

    def tb(self, tape, dy, need_dx=True):
        x, p, y = tape.pop()
        want_p = isinstance(self.p, nn.Parameter) and tape.wants(self.p)
        dx, dp = ops.gem_pool_bwd(x, p, y, dy.reshape(y.shape), self.eps, need_dp=want_p)
        if want_p:
            tape.add_grad(self.p, dp)
        return dx

    def __repr__(self):
        return self.__class__.__name__ + "(" + str(self.p) + ", output_size=" + str(self.output_size) + ")"


class GeneralizedMeanPoolingP(GeneralizedMeanPooling):
    """Same, but the exponent is trainable (init 3)."""

    def __init__(self, norm=3, output_size=1, eps=1e-6):
        super(GeneralizedMeanPoolingP, self).__init__(norm, output_size, eps)
        self.p = nn.Parameter(torch.ones(1) * norm)


def avg_pooling():
    return AdaptiveAvgPool2d(1)


__pooling_factory = {
    "avg": avg_pooling,
    "gem": GeneralizedMeanPoolingP,
}


def pooling_names():
    return sorted(__pooling_factory.keys())


def build_pooling_layer(name):
    if name not in __pooling_factory:
        raise KeyError("Unknown pooling layer:", name)
    return __pooling_factory[name]()
