"""Throughput of the on-device input synthesis kernels at config-2 batch size, with the oracle
(numpy / scipy, what the reference's DataLoader workers run per sample) timed beside it."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "reid-gan_amd"))
sys.path.insert(0, REPO)
import numpy as np
import torch
from rg_hip import ops
from reid.utils.data.device_pipeline import PoseMapGenerator
from clustercontrast.utils.data.device_transforms import RandomErasing, PadRandomCropFlip
from oracle import ref_datagen as OD

dev = torch.device("cuda:0")


def timeit(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def main():
    N, J, H, W = 32, 18, 256, 128
    g = np.random.RandomState(0)
    lm = np.stack([g.randint(0, H, (N, J)), g.randint(0, W, (N, J))], 2)
    c = torch.from_numpy(lm.astype(np.int32)).to(dev)
    sig = torch.full((N,), 5.0, device=dev)
    for mode, name in ((0, "pose maps, FD-GAN form"), (1, "pose maps, Gaussian form")):
        t = timeit(lambda: ops.pose_maps(c, sig, H, W, mode))
        gb = N * J * H * W * 4 / 1e9
        print("%-28s %d x %d x %dx%d: %7.1f us  %6.0f GB/s written  %8.0f samples/s" % (name, N, J, H, W, t * 1e6, gb / t, N / t))
    gen = PoseMapGenerator(H, W, "gauss", device=dev)
    t = timeit(lambda: gen(lm), reps=5)
    print("%-28s incl. host draws + upload: %7.1f us  %8.0f samples/s" % ("PoseMapGenerator (gauss)", t * 1e6, N / t))
    t0 = time.perf_counter()
    for n in range(4):
        OD.o_pose_item(lm[n], H, W, "gauss")
    tc = (time.perf_counter() - t0) / 4
    print("%-28s one sample on one host core: %7.1f ms  %8.1f samples/s" % ("oracle (scipy)", tc * 1e3, 1 / tc))
    x = torch.randn(N, 3, H, W, device=dev)
    re = RandomErasing(probability=1.0, mean=[0.485, 0.456, 0.406])
    rects = [re.draw(3, H, W) for _ in range(N)]
    r = torch.tensor(rects, dtype=torch.int32, device=dev)
    fill = torch.tensor([0.485, 0.456, 0.406], device=dev)
    t = timeit(lambda: ops.erase_rects_(x, r, fill))
    print("%-28s %d x 3 x %dx%d: %7.1f us" % ("RandomErasing fill", N, H, W, t * 1e6))
    pc = PadRandomCropFlip((H, W), padding=10, flip_p=0.5)
    par = torch.tensor([pc.draw(H, W) for _ in range(N)], dtype=torch.int32, device=dev)
    pv = torch.tensor(pc.pad_value, device=dev)
    t = timeit(lambda: ops.flip_pad_crop(x, par, (H, W), 10, pv))
    gb = 2 * x.numel() * 4 / 1e9
    print("%-28s %d x 3 x %dx%d: %7.1f us  %6.0f GB/s (read + write)" % ("Pad + RandomCrop + flip", N, H, W, t * 1e6, gb / t))


if __name__ == "__main__":
    main()
