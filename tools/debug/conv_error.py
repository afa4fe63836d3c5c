"""Error of the conv kernels against an fp64 reference (CPU), per reduction length: relative L2 error, worst element relative to
the output RMS, and the mean SIGNED error relative to sum|a b| (a bias would show there).  Development aid for the arithmetic
notes in DESIGN.md (split-bf16 vs fp32 MFMA builds)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "reid-gan_amd"))
import torch
import torch.nn.functional as F
from rg_hip import ops

dev = torch.device("cuda:0")
torch.manual_seed(0)
print("%-26s %10s %10s %12s | %10s %10s" % ("shape", "relL2", "max/rms", "bias/sum|ab|", "torch relL2", "torch bias"))
for (C, k, positive) in ((64, 1, False), (512, 1, False), (2048, 1, False), (256, 3, False), (512, 3, False), (512, 3, True), (2048, 1, True)):
    N, H, W, K = 4, 16, 8, 64
    x = torch.randn(N, C, H, W)
    w = torch.randn(K, C, k, k) * 0.05
    if positive:                       # all products positive: truncation / dropped-term biases add up coherently
        x, w = x.abs(), w.abs()
    y64 = F.conv2d(x.double(), w.double(), padding=k // 2)
    sabs = F.conv2d(x.double().abs(), w.double().abs(), padding=k // 2)
    y = ops.conv2d_fwd(x.to(dev), w.to(dev), 1, k // 2).double().cpu()
    yt = F.conv2d(x, w, padding=k // 2).double()       # oneDNN fp32 on the host

    def stats(v):
        e = v - y64
        return (e.norm() / y64.norm()).item(), (e.abs().max() / y64.pow(2).mean().sqrt()).item(), (e / sabs).mean().item()
    a, b = stats(y), stats(yt)
    print("%-26s %10.2e %10.2e %12.2e | %10.2e %10.2e" % ("C=%d k=%d%s" % (C, k, " (+)" if positive else ""), a[0], a[1], a[2], b[0], b[2]))
