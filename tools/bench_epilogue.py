"""conv fwd: plain store vs fused scale/shift(+residual)+ReLU epilogue, per ResNet-50 shape (development aid)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "reid-gan_amd"))
import torch
from rg_hip import ops
from bench_conv import resnet_shapes, uniq, timeit
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 96
tot = [0.0, 0.0, 0.0, 0.0]
for name, cnt, C, H, W, K, k, s, p in uniq(resnet_shapes(256, 128)):
    x = torch.randn(N, C, H, W, device=dev)
    w = torch.randn(K, C, k, k, device=dev) * 0.05
    wk = ops.weights_to_krsc(w) if (k > 1 and C % 16 == 0) else None
    sc, sh = torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev)
    y = ops.conv2d_fwd(x, w, s, p, w_krsc=wk)
    res = torch.randn_like(y)
    t0 = timeit(lambda: ops.conv2d_fwd(x, w, s, p, w_krsc=wk))
    t1 = timeit(lambda: ops.conv2d_fwd(x, w, s, p, scale=sc, shift=sh, act=ops.ACT_RELU, w_krsc=wk))
    t2 = timeit(lambda: ops.conv2d_fwd(x, w, s, p, scale=sc, shift=sh, residual=res, act=ops.ACT_RELU, w_krsc=wk))
    t3 = timeit(lambda: ops.bn_apply_fwd(y, sh, sc, sc, sh, None, True, 1e-5, ops.ACT_RELU, 0.0))
    mb = y.numel() * 4 / 1e6
    print("%-12s x%d out %6.1f MB | plain %6.1f us  +affine+relu %6.1f us  +res %6.1f us | bn pass %6.1f us" %
          (name, cnt, mb, t0 * 1e3, t1 * 1e3, t2 * 1e3, t3 * 1e3))
    for i, t in enumerate((t0, t1, t2, t3)):
        tot[i] += t * cnt
print("TOTAL ms: plain %.2f  affine %.2f  +res %.2f  bn passes %.2f" % tuple(tot))
