"""Batch, on-device forms of the training augmentations that the reference applies per image on the CPU
(CC/examples/cluster_contrast_gan_train_usl_infomap.py:110-120): Pad(10) + RandomCrop((h, w)) [+ horizontal flip] and
`RandomErasing` (CC/clustercontrast/utils/data/transforms.py:52-96).  They take NORMALISED device batches
[N, C, H, W]; the padding value is therefore the normalised black pixel (0 - mean) / std per channel (the reference
pads the PIL image before ToTensor / Normalize), and RandomErasing — which the reference also applies after Normalize —
writes `mean` unchanged.

The rectangle / offset draws happen on the host in the reference's order; the kernels apply them to the whole batch.
"""
from __future__ import absolute_import

import math
import random

import torch

from rg_hip import ops


class RandomErasing(object):
    """Same constructor and per-image draw sequence as the reference class (transforms.py:52-96)."""

    def __init__(self, probability=0.5, sl=0.02, sh=0.4, r1=0.3, mean=(0.4914, 0.4822, 0.4465)):
        self.probability, self.mean, self.sl, self.sh, self.r1 = probability, mean, sl, sh, r1

    def draw(self, C, H, W):
        """-> (row0, col0, h, w) for one image, h = 0 when nothing is erased."""
        if random.uniform(0, 1) >= self.probability:
            return (0, 0, 0, 0)
        for _attempt in range(100):
            area = H * W
            target_area = random.uniform(self.sl, self.sh) * area
            aspect_ratio = random.uniform(self.r1, 1 / self.r1)
            h = int(round(math.sqrt(target_area * aspect_ratio)))
            w = int(round(math.sqrt(target_area / aspect_ratio)))
            if w < W and h < H:
                x1 = random.randint(0, H - h)
                y1 = random.randint(0, W - w)
                return (x1, y1, h, w)
        return (0, 0, 0, 0)

    def __call__(self, img, rects=None):
        single = img.dim() == 3
        x = img.unsqueeze(0) if single else img
        N, C, H, W = x.shape
        if rects is None:
            rects = [self.draw(C, H, W) for _ in range(N)]
        if C == 3:
            fill = [float(self.mean[0]), float(self.mean[1]), float(self.mean[2])]
        else:                                   # the reference erases channel 0 only (:92-93)
            fill = [float(self.mean[0])] + [float("nan")] * (C - 1)
        r = torch.tensor(rects, dtype=torch.int32).view(N, 4).to(x.device)
        ops.erase_rects_(x, r, torch.tensor(fill, dtype=torch.float32, device=x.device))
        return img


class PadRandomCropFlip(object):
    """T.Pad(padding) -> T.RandomCrop(size) [-> T.RandomHorizontalFlip(flip_p)] on a normalised device batch.
    torchvision is not in the reference tree; the draws follow its published `RandomCrop.get_params`
    (top = randint(0, h - th), left = randint(0, w - tw) from torch's generator) and `torch.rand(1) < p`."""

    def __init__(self, size, padding=10, flip_p=0.0, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225), generator=None):
        self.size, self.padding, self.flip_p = (int(size[0]), int(size[1])), int(padding), float(flip_p)
        self.pad_value = [(0.0 - m) / s for m, s in zip(mean, std)]
        self.generator = generator

    def draw(self, Hs, Ws):
        th, tw = self.size
        h, w = Hs + 2 * self.padding, Ws + 2 * self.padding
        if h < th or w < tw:
            raise ValueError("Required crop size {} is larger than input image size {}".format((th, tw), (h, w)))
        if w == tw and h == th:
            top = left = 0
        else:
            top = int(torch.randint(0, h - th + 1, size=(1,), generator=self.generator).item())
            left = int(torch.randint(0, w - tw + 1, size=(1,), generator=self.generator).item())
        flip = 0
        if self.flip_p > 0:
            flip = 1 if float(torch.rand(1, generator=self.generator)) < self.flip_p else 0
        return (flip, top, left)

    def __call__(self, x, params=None):
        N, C, Hs, Ws = x.shape
        if params is None:
            params = [self.draw(Hs, Ws) for _ in range(N)]
        p = torch.tensor(params, dtype=torch.int32).view(N, 3).to(x.device)
        pv = torch.tensor((self.pad_value * C)[:C] if C != len(self.pad_value) else self.pad_value, dtype=torch.float32,
                          device=x.device)
        return ops.flip_pad_crop(x, p, self.size, pad=self.padding, pad_value=pv)
