"""On-device input synthesis through the C ABI vs the oracle (oracle/ref_datagen.py, pinned against the
reference's own functions by tests/golden/reference_datagen.npz): pose heat maps in both forms, flip / pad / crop,
RandomErasing; plus properties at the full 32 x 18 x 256 x 128 batch size."""
import random

import numpy as np
import pytest
import torch

from oracle import ref_datagen as OD
from tests.golden import cases_datagen as C

pytestmark = pytest.mark.gpu


def test_fd_pose_maps_match_oracle(dev):
    from reid.utils.data.device_pipeline import PoseMapGenerator
    for name, (landmark, H, W, aug, seed) in C.POSE_CASES.items():
        gen = PoseMapGenerator(H, W, aug, device=dev)
        batch = np.stack([landmark, landmark[::-1].copy(), landmark])
        random.seed(seed)
        maps, flips = gen(batch)
        after = random.random()
        random.seed(seed)
        ref = [OD.o_pose_item(batch[i], H, W, aug, rnd=random) for i in range(3)]
        assert random.random() == after, name                     # same consumption of the random stream
        assert flips == [r[1] for r in ref], name
        want = np.stack([r[0] for r in ref])
        got = maps.cpu().numpy()
        assert got.shape == want.shape == (3, 18, H, W) and got.dtype == np.float32
        assert np.abs(got - want).max() <= 2e-7, (name, np.abs(got - want).max())
        assert np.array_equal(got == 0, want == 0), name              # same support (4-sigma truncation, reflections)


def test_cords_to_map_matches_oracle(dev):
    from clustercontrast.utils.data.device_pose import cords_to_map, cords_to_map_batch, MISSING_VALUE
    g = np.random.RandomState(3)
    cords = np.stack([g.randint(0, 128, 18), g.randint(0, 64, 18)], 1)
    cords[2] = (MISSING_VALUE, MISSING_VALUE)
    cords[9] = (40, MISSING_VALUE)
    A = np.array([[0.9, 0.1, -3.0], [-0.05, 1.1, 2.0]])             # moves some joints outside the image
    for kw in (dict(), dict(old_size=(256, 128)), dict(affine_matrix=A), dict(old_size=(64, 32), affine_matrix=A, sigma=4)):
        want = OD.o_cords_to_map(cords.copy(), (128, 64), **kw)
        got = cords_to_map(cords.copy(), (128, 64), device=dev, **kw)
        assert tuple(got.shape) == (128, 64, 18) and got.dtype == torch.float32
        assert np.abs(got.cpu().numpy() - want).max() <= 2e-7, kw
    b = cords_to_map_batch(np.stack([cords, cords[::-1]]), (128, 64), device=dev)
    assert tuple(b.shape) == (2, 18, 128, 64)
    assert np.abs(b[1].permute(1, 2, 0).cpu().numpy() - OD.o_cords_to_map(cords[::-1].copy(), (128, 64))).max() <= 2e-7


def test_random_erasing_bit_exact(dev):
    from clustercontrast.utils.data.device_transforms import RandomErasing
    for name, (shape, seed, kw) in C.ERASE_CASES.items():
        img = np.random.RandomState(seed).rand(*shape).astype(np.float32)
        batch = torch.from_numpy(np.stack([img] * C.ERASE_REPEATS)).to(dev)
        random.seed(seed)
        out = RandomErasing(**kw)(batch)
        after = random.random()
        assert out is batch                                         # in place, like the reference
        random.seed(seed)
        want = np.stack([OD.o_random_erasing(img.copy(), rnd=random, **kw) for _ in range(C.ERASE_REPEATS)])
        assert random.random() == after, name
        assert np.array_equal(out.cpu().numpy(), want), name
    single = torch.rand(3, 40, 20, device=dev)
    random.seed(1)
    assert RandomErasing(probability=1.0)(single).shape == (3, 40, 20)


def test_pad_crop_flip_bit_exact(dev):
    from rg_hip import ops
    from clustercontrast.utils.data.device_transforms import PadRandomCropFlip
    g = np.random.RandomState(9)
    x = g.randn(5, 3, 40, 20).astype(np.float32)
    params = np.array([[0, 0, 0], [1, 20, 20], [0, 10, 10], [1, 3, 17], [1, 0, 20]], dtype=np.int32)
    pv = [(0.0 - m) / s for m, s in zip((0.485, 0.456, 0.406), (0.229, 0.224, 0.225))]
    want = OD.o_flip_pad_crop(x, params, (40, 20), pad=10, pad_value=np.float32(pv))
    got = PadRandomCropFlip((40, 20), padding=10)(torch.from_numpy(x).to(dev), params=params.tolist())
    assert np.array_equal(got.cpu().numpy(), want)
    # drawn parameters stay inside the padded image; crop to a smaller size without padding
    t = PadRandomCropFlip((32, 16), padding=0, flip_p=0.5, generator=torch.Generator().manual_seed(2))
    y = t(torch.from_numpy(x).to(dev))
    assert tuple(y.shape) == (5, 3, 32, 16)
    # pure flip (np.flip(maps, 2))
    par = torch.tensor([[1, 0, 0]] * 5, dtype=torch.int32, device=dev)
    f = ops.flip_pad_crop(torch.from_numpy(x).to(dev), par, (40, 20))
    assert np.array_equal(f.cpu().numpy(), x[..., ::-1])
    with pytest.raises(ValueError):
        PadRandomCropFlip((64, 20), padding=10)(torch.from_numpy(x).to(dev))


def test_pose_maps_full_batch_properties(dev):
    """config-2 size: 32 samples x 18 joints x 256 x 128."""
    from reid.utils.data.device_pipeline import PoseMapGenerator
    g = np.random.RandomState(17)
    lm = np.stack([g.randint(0, 256, (32, 18)), g.randint(0, 128, (32, 18))], 2)
    lm[:, 5] = -1
    gen = PoseMapGenerator(256, 128, "no", device=dev)
    maps, flips = gen(lm, with_flip=False)
    assert tuple(maps.shape) == (32, 18, 256, 128) and not any(flips)
    assert maps[:, 5].abs().max().item() == 0
    present = [j for j in range(18) if j != 5]
    peak = maps[:, present].amax((2, 3))
    assert (peak == 1).all()                                         # divided by the map maximum
    idx = maps[:, present].flatten(2).argmax(2).cpu().numpy()
    # the maximum sits at the landmark; within a few pixels of a border the reflected tail pulls it towards the border
    rows, cols = idx // 128, idx % 128
    dr, dc = np.abs(rows - lm[:, present, 0]), np.abs(cols - lm[:, present, 1])
    assert (dr <= 4).all() and (dc <= 4).all()
    interior = (lm[:, present, 0] >= 20) & (lm[:, present, 0] < 236) & (lm[:, present, 1] >= 20) & (lm[:, present, 1] < 108)
    assert (dr[interior] == 0).all() and (dc[interior] == 0).all()
    assert (maps >= 0).all() and torch.isfinite(maps).all()
    flipped = gen(lm, draws=[(None, 5, True)] * 32)[0]
    assert torch.equal(flipped, maps.flip(3))


def test_datagen_error_paths(dev):
    from rg_hip import ops
    with pytest.raises(ValueError):
        ops.pose_maps(torch.zeros(2, 18, 2, device=dev), torch.ones(2, device=dev), 64, 32)          # not int32
    with pytest.raises(ValueError):
        ops.pose_maps(torch.zeros(2, 18, 2, dtype=torch.int32, device=dev), torch.ones(3, device=dev), 64, 32)
    with pytest.raises(RuntimeError):
        ops.pose_maps(torch.zeros(1, 1, 2, dtype=torch.int32, device=dev), torch.ones(1, device=dev), 2048, 32)
    with pytest.raises(ValueError):
        ops.erase_rects_(torch.zeros(2, 3, 8, 8, device=dev), torch.zeros(2, 4, dtype=torch.int32, device=dev),
                         torch.zeros(2, device=dev))
