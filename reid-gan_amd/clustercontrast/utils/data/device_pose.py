"""`cords_to_map` of CC/clustercontrast/utils/data/pose_utils.py:51-70 with the maps written by the GPU.

Same arguments; the per-point host arithmetic (rescale to the image size, optional affine map, `int()` truncation,
MISSING_VALUE test on the unscaled coordinates) is restated from the cited lines, the H x W x J Gaussian evaluation —
the part that costs — is `rg_pose_maps` mode 1.  The reference returns a numpy array [H, W, J]; this returns a device
tensor of the same shape (a permuted view of the [J, H, W] result); `cords_to_map_batch` returns [N, J, H, W].
The drawing helpers of the reference file (skimage / matplotlib) are visualisation only and not part of the hot path.
"""
from __future__ import absolute_import

import numpy as np
import torch

from rg_hip import ops

MISSING_VALUE = -1
_SENTINEL = -2 ** 31


def _centres(cords, img_size, old_size, affine_matrix):
    old_size = img_size if old_size is None else old_size
    cords = np.asarray(cords).astype(float)
    out = np.empty((cords.shape[0], 2), dtype=np.int64)
    for i, point in enumerate(cords):
        if point[0] == MISSING_VALUE or point[1] == MISSING_VALUE:
            out[i] = _SENTINEL
            continue
        p0 = point[0] / old_size[0] * img_size[0]
        p1 = point[1] / old_size[1] * img_size[1]
        if affine_matrix is not None:
            p_ = np.dot(np.asarray(affine_matrix, dtype=float), np.array([p1, p0, 1.0]).reshape(3, 1))
            out[i, 0], out[i, 1] = int(p_[1, 0]), int(p_[0, 0])
        else:
            out[i, 0], out[i, 1] = int(p0), int(p1)
    return out


def cords_to_map_batch(cords, img_size, old_size=None, affine_matrix=None, sigma=6, device=None):
    """cords [N, J, 2] (y, x) -> [N, J, H, W] float32 on the device."""
    cords = np.asarray(cords)
    if cords.ndim != 3 or cords.shape[2] != 2:
        raise ValueError("cords_to_map_batch: cords must be [N, J, 2]")
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    mats = affine_matrix if isinstance(affine_matrix, (list, tuple)) else [affine_matrix] * cords.shape[0]
    c = np.stack([_centres(cords[n], tuple(img_size), old_size, mats[n]) for n in range(cords.shape[0])])
    c = np.clip(c, _SENTINEL, 2 ** 31 - 1).astype(np.int32)
    sig = torch.full((cords.shape[0],), float(sigma), dtype=torch.float32)
    return ops.pose_maps(torch.from_numpy(c).to(dev), sig.to(dev), int(img_size[0]), int(img_size[1]), mode=1)


def cords_to_map(cords, img_size, old_size=None, affine_matrix=None, sigma=6, device=None):
    return cords_to_map_batch(np.asarray(cords)[None], img_size, old_size, affine_matrix, sigma, device)[0].permute(1, 2, 0)
