import os, sys, math
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "reid-gan_amd"))
import torch, torch.nn.functional as F
from rg_hip import ops
dev = torch.device("cuda:0")
for (N, C, H, W, K, k, s, p) in [(2, 64, 16, 8, 64, 1, 1, 0), (2, 64, 16, 8, 256, 1, 1, 0), (3, 32, 16, 8, 32, 3, 3, 1)]:
    g = torch.Generator().manual_seed(1)
    x = torch.randn(N, C, H, W, generator=g); w = torch.randn(K, C, k, k, generator=g) / math.sqrt(C * k * k)
    ref = F.conv2d(x.double(), w.double(), stride=s, padding=p)
    y = ops.conv2d_fwd(x.to(dev), w.to(dev), s, p).cpu().double()
    err = (y - ref).abs()
    print((N, C, H, W, K, k, s, p), "max err", err.max().item())
    bad = (err > 1e-3).nonzero()
    print("  bad count", bad.shape[0], "of", err.numel())
    if bad.shape[0]:
        print("  first bad idx", bad[:5].tolist(), "last", bad[-3:].tolist())
        # which output channels / pixels are bad
        print("  bad channels", sorted(set(bad[:, 1].tolist()))[:20], " bad imgs", sorted(set(bad[:, 0].tolist())))
        print("  bad rows(h)", sorted(set(bad[:, 2].tolist()))[:20], "cols(w)", sorted(set(bad[:, 3].tolist())))
        i = bad[0].tolist(); print("  got", y[tuple(i)].item(), "ref", ref[tuple(i)].item())
