"""Feature extraction and distance matrices of the evaluation / pseudo-labelling loops on the MI355X.

Mirror of the numeric part of CC/clustercontrast/evaluators.py: `extract_cnn_feature` (:16-20), `extract_all_feature`
(:22-27), `pairwise_distance` (:71-88) (FD/reid/evaluators.py:76-98 and FD/reid/feature_extraction/cnn.py:9-16 are the same
functions).  The loader loop `extract_features` (:30-68), CMC / mAP scoring, re-ranking and the `Evaluator` class are host
code and stay the reference's: with its tree behind this one on sys.path they are inherited at the bottom of
this file and call the functions defined here.
"""
from __future__ import print_function, absolute_import

import torch

from rg_hip import ops


def _device():
    return torch.device("cuda", torch.cuda.current_device())


def _to_torch(x):
    return x if torch.is_tensor(x) else torch.as_tensor(x)


def extract_cnn_feature(model, inputs):
    inputs = _to_torch(inputs).to(_device(), non_blocking=True)
    with torch.no_grad():
        outputs = model(inputs)
    return outputs.data.cpu()


def extract_all_feature(model, inputs):
    inputs = _to_torch(inputs).to(_device(), non_blocking=True)
    with torch.no_grad():
        outputs, extra_outputs = model(inputs, test_all=True)
    return outputs.data.cpu(), extra_outputs.data.cpu()


def _dist_block(x, y, xx_scale, with_y_norm):
    """[m, n] block: xx_scale * |x_i|^2 (+ |y_j|^2) - 2 x_i . y_j — the GEMM with -2 folded into the MFMA epilogue."""
    n = y.shape[0]
    minus2 = ops.fill_(torch.empty(n, dtype=torch.float32, device=x.device), -2.0)
    yy = ops.row_sqsum(y) if with_y_norm else None
    d = ops.conv2d_fwd(x.view(x.shape[0], x.shape[1], 1, 1), y.view(n, y.shape[1], 1, 1), scale=minus2, shift=yy)
    d = d.view(x.shape[0], n)
    return ops.add_outer_terms(d, rowv=ops.row_sqsum(x), colv=None, alpha=1.0, a=xx_scale)


def pairwise_distance(features, query=None, gallery=None):
    dev = _device()
    if query is None and gallery is None:
        n = len(features)
        x = torch.cat(list(features.values())).view(n, -1).float().to(dev).contiguous()
        # 2 |x_i|^2 - 2 x_i . x_j  (the reference's expression for the self-distance matrix, :71-77)
        return _dist_block(x, x, 2.0, False).cpu()
    x = torch.cat([features[f].unsqueeze(0) for f, _, _ in query], 0)
    y = torch.cat([features[f].unsqueeze(0) for f, _, _ in gallery], 0)
    m, n = x.size(0), y.size(0)
    x = x.view(m, -1)
    y = y.view(n, -1)
    xd, yd = x.float().to(dev).contiguous(), y.float().to(dev).contiguous()
    dist_m = _dist_block(xd, yd, 1.0, True)
    return dist_m.cpu(), x.numpy(), y.numpy()


# `extract_features`, `Evaluator`, `evaluate_all`, CMC / mAP and re-ranking are the reference's own (CPU) code: when its tree sits behind this
# one on sys.path they are taken from there, and they call the functions above (rg_hip/overlay.py)
from rg_hip.overlay import inherit as _rg_inherit  # noqa: E402
_rg_inherit(globals())
