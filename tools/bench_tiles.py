"""stride-2 dgrad shapes under the tile forced by RG_CONV_FORCE (read once per process): development aid."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "reid-gan_amd"))
import torch
from rg_hip import ops
dev = torch.device("cuda:0")
def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
shapes = [(32,128,64,32,128,3,2,1), (32,256,32,16,256,3,2,1), (32,512,16,8,512,3,2,1), (32,256,64,32,512,1,2,0),
          (32,512,32,16,1024,1,2,0), (32,1024,16,8,2048,1,2,0), (96,512,16,8,512,3,2,1), (96,1024,16,8,2048,1,2,0)]
for (N,C,H,W,K,k,s,p) in shapes:
    x = torch.randn(N, C, H, W, device=dev); w = torch.randn(K, C, k, k, device=dev) * 0.05
    y = ops.conv2d_fwd(x, w, s, p); dy = torch.randn_like(y)
    wk = ops.weights_to_krsc(w) if k > 1 and C % 4 == 0 else None
    gf = 2.0 * N * y.shape[2] * y.shape[3] * K * C * k * k / 1e9
    us = t(lambda: ops.conv2d_dgrad(dy, w, (H, W), s, p, w_krsc=wk))
    print(os.environ.get("RG_CONV_FORCE", "plan"), (N,C,H,W,K,k,s,p), "%.2f GF  %.0f us  %.1f TF" % (gf, us, gf / us * 1e3))
