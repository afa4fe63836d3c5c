import os, sys, math
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "reid-gan_amd"))
import torch
from rg_hip import ops
dev = torch.device("cuda:0")
N, C, H, W = 2, 64, 16, 8
x = torch.zeros(N, C, H, W)
for n in range(N):
    for c in range(C):
        x[n, c] = 1000 * n + c + 0.001 * torch.arange(H * W).view(H, W)
w = torch.eye(64).view(64, 64, 1, 1).contiguous()
y = ops.conv2d_fwd(x.to(dev), w.to(dev), 1, 0).cpu()
print("y[0,:8,0,0]", y[0, :8, 0, 0].tolist())
print("y[0,5,0,:8]", y[0, 5, 0, :8].tolist())
print("y[1,5,3,:4]", y[1, 5, 3, :4].tolist(), "expect", x[1, 5, 3, :4].tolist())
w2 = torch.zeros(64, 64, 1, 1); w2[3, 7] = 1.0
y2 = ops.conv2d_fwd(x.to(dev), w2.to(dev), 1, 0).cpu()
nz = y2.abs().sum(dim=(0, 2, 3)).nonzero().flatten().tolist()
print("w[3,7]=1 -> nonzero out channels", nz, " y2[0,3,0,:4]", y2[0, 3, 0, :4].tolist() if len(nz) else None)
for ch in nz[:4]:
    print("   ch", ch, y2[0, ch, 0, :4].tolist())
