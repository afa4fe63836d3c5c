"""Batch, on-device form of the pose-map half of FD-GAN's `Preprocessor` (FD/reid/utils/data/preprocessor.py).

The reference builds, per sample and on the CPU, 18 maps of 256x128: a unit impulse at the landmark, blurred with
`scipy.ndimage.filters.gaussian_filter(sigma)` and divided by its maximum (`_generate_pose_map`, :114-131), with the
landmark augmentations `pose_aug` in {'no', 'erase', 'gauss'}; then flips maps and target image together with
probability 1/2 (`_get_single_item_with_pose`, :86-91).  Here the landmarks of a whole batch go to the GPU as
[N, 18, 2] integers and one kernel writes the [N, 18, H, W] maps (`rg_pose_maps`, mode 0: the closed form of the
filtered impulse including scipy's 'reflect' boundary and 4-sigma truncation, float64 arithmetic).

The random choices are drawn on the host with Python's `random`, per sample and in the reference's order
(augmentation draw, then the flip flag), so a seeded run makes the same choices as the reference loop.
"""
from __future__ import absolute_import

import random

import torch

from rg_hip import ops


class PoseMapGenerator(object):
    def __init__(self, height=256, width=128, pose_aug='no', device=None):
        self.height, self.width, self.pose_aug = height, width, pose_aug
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)

    def draw(self, n_joints, gauss_sigma=5, with_flip=True):
        """Host-side draws for ONE sample -> (erased joint or None, sigma, flip flag)."""
        erase = None
        if self.pose_aug == 'erase':
            erase = random.randrange(n_joints)                       # preprocessor.py:118-119
        elif self.pose_aug == 'gauss':
            gauss_sigma = random.randint(gauss_sigma - 1, gauss_sigma + 1)      # :120-121
        # any other value behaves like 'no': the reference's `assert ('Unknown ...')` (:122-123) never fires
        flip = random.choice([True, False]) if with_flip else False  # :86
        return erase, gauss_sigma, flip

    def __call__(self, landmarks, with_flip=True, draws=None):
        """landmarks: [N, J, 2] integers (row, col) as `_load_landmark` returns them (-1: joint not detected).
        Returns (maps [N, J, H, W] float32 on the device, flips: list of bool).  `draws` overrides the random draws
        (list of (erase, sigma, flip) per sample)."""
        lm = torch.as_tensor(landmarks).to(torch.int32).clone()
        if lm.dim() != 3 or lm.shape[2] != 2:
            raise ValueError("PoseMapGenerator: landmarks must be [N, J, 2]")
        N, J = lm.shape[0], lm.shape[1]
        if draws is None:
            draws = [self.draw(J, with_flip=with_flip) for _ in range(N)]
        sig = torch.empty(N, dtype=torch.float32)
        for n, (erase, s, _flip) in enumerate(draws):
            sig[n] = float(s)
            if erase is not None:
                lm[n, erase] = -1
        flips = [bool(d[2]) for d in draws]
        maps = ops.pose_maps(lm.to(self.device), sig.to(self.device), self.height, self.width, mode=0)
        if any(flips):
            maps = flip_images(maps, flips)                          # np.flip(maps, 2), :88-89
        return maps, flips


def flip_images(x, flips):
    """Horizontal flip of the samples whose flag is set (device tensor [N, C, H, W])."""
    par = torch.zeros((x.shape[0], 3), dtype=torch.int32)
    par[:, 0] = torch.as_tensor([1 if f else 0 for f in flips], dtype=torch.int32)
    return ops.flip_pad_crop(x, par.to(x.device), (x.shape[2], x.shape[3]), pad=0)
