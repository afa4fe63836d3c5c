"""Differentiable resize / normalise applied to generated images before they re-enter the encoder.

Mirror of CC/clustercontrast/utils/data/diff_augs.py:6-16 (`my_resize`, `my_normalize`, `my_transform`): there a
torchvision `resize(X, size, BICUBIC)` on a float tensor (= F.interpolate(mode='bicubic', align_corners=False), no
antialias when up-sampling) followed by `(X - mean) / std`; here ONE fused HIP kernel each way
(`rg_bicubic_normalize_fwd/bwd`) so the 256x128 intermediate of `my_transform` is written once.
`pair_rand_flip` / `my_pad` of the same file are plain indexing and stay torch ops.
"""
from __future__ import absolute_import

import torch
from torch import autograd

from rg_hip import ops

_MEAN = (0.485, 0.456, 0.406)
_STD = (0.229, 0.224, 0.225)
_const_cache = {}


def _const(values, device):
    key = (tuple(float(v) for v in values), str(device))
    t = _const_cache.get(key)
    if t is None:
        t = torch.tensor(key[0], dtype=torch.float32, device=device)
        _const_cache[key] = t
    return t


class _ResizeNormalize(autograd.Function):
    @staticmethod
    def forward(ctx, X, size, mean, std):
        ctx.in_hw = (X.shape[2], X.shape[3])
        ctx.std = std
        return ops.bicubic_normalize_fwd(X, size, mean, std)

    @staticmethod
    def backward(ctx, g):
        return ops.bicubic_normalize_bwd(g, ctx.in_hw, ctx.std), None, None, None


def _check(X, n):
    if X.dim() == 3:
        X = X.unsqueeze(0)
    if X.dim() != 4:
        raise TypeError("Tensor is not a torch image.")
    if n is not None and X.shape[1] != n:
        raise ValueError("std/mean have %d channels, the image batch has %d" % (n, X.shape[1]))
    return X


def my_resize(X, size=(256, 128)):
    squeeze = X.dim() == 3
    out = _ResizeNormalize.apply(_check(X, None), (int(size[0]), int(size[1])), None, None)
    return out[0] if squeeze else out


def my_normalize(X, mean=_MEAN, std=_STD):
    squeeze = X.dim() == 3
    Xb = _check(X, len(mean))
    out = _ResizeNormalize.apply(Xb, (Xb.shape[2], Xb.shape[3]), _const(mean, X.device), _const(std, X.device))
    return out[0] if squeeze else out


def my_transform(X, size=(256, 128)):
    squeeze = X.dim() == 3
    Xb = _check(X, len(_MEAN))
    out = _ResizeNormalize.apply(Xb, (int(size[0]), int(size[1])), _const(_MEAN, X.device), _const(_STD, X.device))
    return out[0] if squeeze else out


def my_pad(X, pad=10):
    return torch.nn.functional.pad(X, (pad, pad, pad, pad))


def pair_rand_flip(Xa, Xb, flip_prob=0.5):
    randf = torch.rand(Xa.size(0), 1, 1, 1, device=Xa.device)
    return torch.where(randf < flip_prob, Xa.flip(3), Xa), torch.where(randf < flip_prob, Xb.flip(3), Xb)
