"""Seeded inputs shared by make_golden_datagen.py (reference side) and the tests (oracle / device side)."""
import numpy as np

POSE_STRIDE = 2


def _landmarks(H, W, seed, missing=(3, 11)):
    g = np.random.RandomState(seed)
    lm = np.stack([g.randint(0, H, 18), g.randint(0, W, 18)], 1)
    lm[0] = (0, 0)                      # corner: both reflections
    lm[1] = (H - 1, W - 1)
    lm[2] = (2, W - 3)                  # near two different borders
    lm[4] = (H // 2, 1)
    for i in missing:
        lm[i] = (-1, -1)
    lm[7] = (-1, 5)                     # one coordinate missing
    return lm.astype(np.int64)


# name -> (landmark [18, 2], H, W, pose_aug, seed)
POSE_CASES = {
    "no_64x32": (_landmarks(64, 32, 1), 64, 32, "no", 11),
    "erase_64x32": (_landmarks(64, 32, 2), 64, 32, "erase", 12),
    "gauss_64x32": (_landmarks(64, 32, 3), 64, 32, "gauss", 13),
    "gauss_256x128": (_landmarks(256, 128, 4), 256, 128, "gauss", 15),
}

ERASE_REPEATS = 6
# name -> (image shape, seed, RandomErasing kwargs)
ERASE_CASES = {
    "rgb": ((3, 64, 32), 21, dict(probability=0.5, mean=[0.485, 0.456, 0.406])),
    "rgb_always": ((3, 48, 24), 22, dict(probability=1.0, sl=0.1, sh=0.4, r1=0.3, mean=[0.1, 0.2, 0.3])),
    "one_channel": ((1, 32, 16), 23, dict(probability=1.0)),
    "five_channels": ((5, 32, 16), 24, dict(probability=1.0)),
}
