"""CPU restatement (numpy / scipy) of the reference's per-sample input synthesis — TEST INFRASTRUCTURE ONLY: imported by
tests/ and tests/golden/ alone, never by the product path.

Pinning: `tests/golden/make_golden_datagen.py` runs the reference's own `Preprocessor._generate_pose_map`
(FD/reid/utils/data/preprocessor.py:114-131) and `RandomErasing.__call__` (CC/clustercontrast/utils/data/transforms.py:52-96)
in the build container on seeded inputs and stores their outputs in tests/golden/reference_datagen.npz; this file equals
them bit-for-bit (tests/test_oracle_golden_datagen.py).  `cords_to_map` (CC/.../pose_utils.py:51-70) cannot be imported
here (its module needs skimage and matplotlib, absent from the image): restated from the cited lines, parity unpinned
for that one function beyond the restatement itself (10 lines of numpy).  Pad / RandomCrop / flip are torchvision
(third-party, absent): restated from their documented semantics with numpy padding / slicing.
"""
from __future__ import absolute_import

import math
import random

import numpy as np
from scipy import ndimage

MISSING_VALUE = -1


def o_generate_pose_map(landmark, height, width, pose_aug='no', gauss_sigma=5, rnd=random):
    """FD/reid/utils/data/preprocessor.py:114-131; landmark: integer array [J, 2] (row, col), -1 = missing."""
    landmark = np.asarray(landmark)
    maps = []
    randnum = landmark.shape[0] + 1
    if pose_aug == 'erase':
        randnum = rnd.randrange(landmark.shape[0])
    elif pose_aug == 'gauss':
        gauss_sigma = rnd.randint(gauss_sigma - 1, gauss_sigma + 1)
    for i in range(landmark.shape[0]):
        m = np.zeros([height, width])
        if landmark[i, 0] != -1 and landmark[i, 1] != -1 and i != randnum:
            m[landmark[i, 0], landmark[i, 1]] = 1
            m = ndimage.gaussian_filter(m, sigma=gauss_sigma)
            m = m / m.max()
        maps.append(m)
    return np.stack(maps, axis=0)


def o_pose_item(landmark, height, width, pose_aug='no', rnd=random):
    """maps + flip of `_get_single_item_with_pose` (:84-91): returns (float32 maps [J, H, W], flip flag)."""
    maps = o_generate_pose_map(landmark, height, width, pose_aug, rnd=rnd)
    flip_flag = rnd.choice([True, False])
    if flip_flag:
        maps = np.flip(maps, 2)
    return maps.copy().astype(np.float32), flip_flag


def o_cords_to_map(cords, img_size, old_size=None, affine_matrix=None, sigma=6):
    """CC/clustercontrast/utils/data/pose_utils.py:51-70 -> float32 [H, W, J]."""
    old_size = img_size if old_size is None else old_size
    cords = np.asarray(cords).astype(float)
    result = np.zeros(tuple(img_size) + cords.shape[0:1], dtype='float32')
    for i, point in enumerate(cords):
        if point[0] == MISSING_VALUE or point[1] == MISSING_VALUE:
            continue
        point[0] = point[0] / old_size[0] * img_size[0]
        point[1] = point[1] / old_size[1] * img_size[1]
        if affine_matrix is not None:
            point_ = np.dot(np.asarray(affine_matrix, dtype=float), np.array([point[1], point[0], 1.0]).reshape(3, 1))
            point_0 = int(point_[1, 0])
            point_1 = int(point_[0, 0])
        else:
            point_0 = int(point[0])
            point_1 = int(point[1])
        xx, yy = np.meshgrid(np.arange(img_size[1]), np.arange(img_size[0]))
        result[..., i] = np.exp(-((yy - point_0) ** 2 + (xx - point_1) ** 2) / (2 * sigma ** 2))
    return result


def o_random_erasing(img, probability=0.5, sl=0.02, sh=0.4, r1=0.3, mean=(0.4914, 0.4822, 0.4465), rnd=random):
    """CC/clustercontrast/utils/data/transforms.py:67-96 on a numpy image [C, H, W] (modified in place and returned)."""
    if rnd.uniform(0, 1) >= probability:
        return img
    for _attempt in range(100):
        area = img.shape[1] * img.shape[2]
        target_area = rnd.uniform(sl, sh) * area
        aspect_ratio = rnd.uniform(r1, 1 / r1)
        h = int(round(math.sqrt(target_area * aspect_ratio)))
        w = int(round(math.sqrt(target_area / aspect_ratio)))
        if w < img.shape[2] and h < img.shape[1]:
            x1 = rnd.randint(0, img.shape[1] - h)
            y1 = rnd.randint(0, img.shape[2] - w)
            if img.shape[0] == 3:
                img[0, x1:x1 + h, y1:y1 + w] = mean[0]
                img[1, x1:x1 + h, y1:y1 + w] = mean[1]
                img[2, x1:x1 + h, y1:y1 + w] = mean[2]
            else:
                img[0, x1:x1 + h, y1:y1 + w] = mean[0]
            return img
    return img


def o_flip_pad_crop(x, params, out_hw, pad=0, pad_value=None):
    """x [N, C, Hs, Ws]; params [N, 3] = (flip, top, left): constant padding, crop, then horizontal flip of the crop."""
    x = np.asarray(x)
    N, C = x.shape[:2]
    H, W = out_hw
    out = np.empty((N, C, H, W), dtype=x.dtype)
    for n in range(N):
        flip, top, left = [int(v) for v in params[n]]
        for c in range(C):
            pv = 0.0 if pad_value is None else pad_value[c]
            p = np.pad(x[n, c], pad, mode='constant', constant_values=pv)
            crop = p[top:top + H, left:left + W]
            out[n, c] = crop[:, ::-1] if flip else crop
    return out
