"""Generates tests/golden/reference_datagen.npz by running the REFERENCE's own data-pipeline functions (build container
only; /root/reference does not exist on the GPU box):

  * FD/reid/utils/data/preprocessor.py  `Preprocessor._generate_pose_map` (:114-131) for pose_aug in no / erase / gauss
  * CC/clustercontrast/utils/data/transforms.py  `RandomErasing.__call__` (:67-96)

Both modules do `from torchvision.transforms import *` at import time although the two functions use none of it;
torchvision is not installed in this image, so an empty placeholder module is registered for that import only (the recipe
of SURVEY.md §8c / make_golden.py).  Nothing from the reference is copied: the file stores inputs and the reference's
outputs, and asserts on the way that oracle/ref_datagen.py reproduces them exactly.
`pose_utils.cords_to_map` is not covered: its module imports skimage and matplotlib, which the image lacks.

Usage:  python tests/golden/make_golden_datagen.py
"""
from __future__ import absolute_import, print_function

import importlib.util
import os
import random
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
from oracle import ref_datagen as OD  # noqa: E402
from tests.golden import cases_datagen as C  # noqa: E402

FD = "/root/reference/FD-GAN-master"
CC = "/root/reference/cluster-contrast-reid-main"


def _placeholders():
    tv = types.ModuleType("torchvision")
    tvt = types.ModuleType("torchvision.transforms")
    tv.transforms = tvt
    sys.modules.setdefault("torchvision", tv)
    sys.modules.setdefault("torchvision.transforms", tvt)


def load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def main():
    _placeholders()
    # the FD preprocessor imports `reid.utils.data.transforms` from its own tree
    for name, path in (("reid", FD + "/reid"), ("reid.utils", FD + "/reid/utils"), ("reid.utils.data", FD + "/reid/utils/data")):
        pkg = types.ModuleType(name)
        pkg.__path__ = [path]
        sys.modules[name] = pkg
    load(FD + "/reid/utils/data/transforms.py", "reid.utils.data.transforms")
    sys.modules["reid.utils.data"].transforms = sys.modules["reid.utils.data.transforms"]
    pre = load(FD + "/reid/utils/data/preprocessor.py", "reid.utils.data.preprocessor")
    cct = load(CC + "/clustercontrast/utils/data/transforms.py", "cc_ref_transforms")

    out = {}
    for name, (landmark, H, W, aug, seed) in C.POSE_CASES.items():
        ns = types.SimpleNamespace(height=H, width=W, pose_aug=aug)
        random.seed(seed)
        ref = pre.Preprocessor._generate_pose_map(ns, torch.as_tensor(landmark).long())
        state_after = random.random()
        random.seed(seed)
        mine = OD.o_generate_pose_map(landmark, H, W, aug)
        assert random.random() == state_after, name
        assert ref.shape == mine.shape and np.array_equal(ref, mine), name
        out["pose_" + name] = ref.astype(np.float32)[:, ::C.POSE_STRIDE, ::C.POSE_STRIDE]
        out["pose_" + name + "_sum"] = ref.sum((1, 2))
        out["pose_" + name + "_argmax"] = np.array([int(m.argmax()) for m in ref])
        print("pose", name, ref.shape, "== oracle")
    for name, (shape, seed, kw) in C.ERASE_CASES.items():
        g = np.random.RandomState(seed)
        img = g.rand(*shape).astype(np.float32)
        re = cct.RandomErasing(**kw)
        random.seed(seed)
        refs, mines = [], []
        for _ in range(C.ERASE_REPEATS):
            refs.append(re(torch.from_numpy(img.copy())).numpy())
        state_after = random.random()
        random.seed(seed)
        for _ in range(C.ERASE_REPEATS):
            mines.append(OD.o_random_erasing(img.copy(), rnd=random, **kw))
        assert random.random() == state_after, name
        assert all(np.array_equal(a, b) for a, b in zip(refs, mines)), name
        out["erase_" + name] = np.stack(refs)
        print("erase", name, np.stack(refs).shape, "== oracle;", sum(int(not np.array_equal(r, img)) for r in refs), "erased")
    path = os.path.join(HERE, "reference_datagen.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
