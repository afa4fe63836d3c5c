"""oracle/ref_datagen.py against the outputs of the reference's own functions (tests/golden/reference_datagen.npz, made by
tests/golden/make_golden_datagen.py in the build container), and the host-side draw order of the device pipeline against
the oracle's consumption of Python's `random` stream.  No GPU, no reference tree needed."""
import os
import random

import numpy as np

from oracle import ref_datagen as OD
from tests.golden import cases_datagen as C

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_datagen.npz"))


def test_pose_maps_equal_reference():
    for name, (landmark, H, W, aug, seed) in C.POSE_CASES.items():
        random.seed(seed)
        m = OD.o_generate_pose_map(landmark, H, W, aug)
        assert m.shape == (18, H, W) and m.dtype == np.float64
        assert np.array_equal(m.astype(np.float32)[:, ::C.POSE_STRIDE, ::C.POSE_STRIDE], GOLD["pose_" + name]), name
        assert np.array_equal(m.sum((1, 2)), GOLD["pose_" + name + "_sum"]), name
        assert np.array_equal(np.array([int(x.argmax()) for x in m]), GOLD["pose_" + name + "_argmax"]), name
        present = [(i, p) for i, p in enumerate(landmark) if p[0] != -1 and p[1] != -1]
        zero = [i for i in range(18) if m[i].max() == 0]
        assert len(zero) == 18 - len(present) + (1 if aug == "erase" else 0) or aug == "erase"
        for i, p in present:
            if m[i].max() > 0:
                assert m[i].max() == 1.0


def test_random_erasing_equals_reference():
    for name, (shape, seed, kw) in C.ERASE_CASES.items():
        img = np.random.RandomState(seed).rand(*shape).astype(np.float32)
        random.seed(seed)
        got = np.stack([OD.o_random_erasing(img.copy(), rnd=random, **kw) for _ in range(C.ERASE_REPEATS)])
        assert np.array_equal(got, GOLD["erase_" + name]), name


def test_host_draw_order_matches_the_oracle_stream():
    """The device pipeline draws on the host; a seeded run must consume `random` exactly like the per-sample reference loop."""
    from reid.utils.data.device_pipeline import PoseMapGenerator
    from clustercontrast.utils.data.device_transforms import RandomErasing
    for aug in ("no", "erase", "gauss", "something-else"):
        gen = PoseMapGenerator(64, 32, aug, device="cpu")
        lm = C.POSE_CASES["no_64x32"][0]
        random.seed(5)
        draws = [gen.draw(18) for _ in range(4)]
        after = random.random()
        random.seed(5)
        flips = [OD.o_pose_item(lm, 64, 32, aug, rnd=random)[1] for _ in range(4)]
        assert random.random() == after and [d[2] for d in draws] == flips, aug
    for name, (shape, seed, kw) in C.ERASE_CASES.items():
        re = RandomErasing(**kw)
        img = np.random.RandomState(seed).rand(*shape).astype(np.float32)
        random.seed(seed)
        rects = [re.draw(*shape) for _ in range(C.ERASE_REPEATS)]
        after = random.random()
        random.seed(seed)
        outs = [OD.o_random_erasing(img.copy(), rnd=random, **kw) for _ in range(C.ERASE_REPEATS)]
        assert random.random() == after, name
        for (r0, c0, h, w), o in zip(rects, outs):
            changed = np.argwhere((o != img).any(0)) if h else np.zeros((0, 2))
            if h == 0:
                assert np.array_equal(o, img)
            else:   # the erased rectangle is exactly where the fill differs from the random image
                assert changed[:, 0].min() == r0 and changed[:, 0].max() == r0 + h - 1
                assert changed[:, 1].min() == c0 and changed[:, 1].max() == c0 + w - 1


def test_cords_to_map_restatement_properties():
    cords = np.array([[10, 5], [-1, -1], [63, 31], [20, -1]], dtype=np.int64)
    m = OD.o_cords_to_map(cords, (64, 32), sigma=6)
    assert m.shape == (64, 32, 4) and m.dtype == np.float32
    assert m[10, 5, 0] == 1.0 and m[..., 1].max() == 0 and m[..., 3].max() == 0 and m[63, 31, 2] == 1.0
    assert abs(m[10, 11, 0] - np.exp(-36 / 72.0)) < 1e-7
    m2 = OD.o_cords_to_map(cords, (64, 32), old_size=(128, 64), sigma=6)       # rescaled: centre (5, 2)
    assert m2[5, 2, 0] == 1.0
