"""SURVEY §8f ranks 1-2 at the reference's sizes: kNN over 12 936 x 2048 features (k = 15) and the 3 368 x 15 913 x 2048
evaluation distance matrix (Market-1501).  Times exclude the final device->host copies of the results."""
import os, sys, time, json
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "reid-gan_amd"))
import torch
import torch.nn.functional as F
from rg_hip import ops
from clustercontrast.evaluators import _dist_block
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
x = F.normalize(torch.randn(12936, 2048, generator=g, device=dev), dim=1)

def knn(k=15, block=2048):
    n = x.shape[0]
    for r0 in range(0, n, block):
        r1 = min(n, r0 + block)
        ops.topk_rows(ops.linear_fwd(x[r0:r1], x), k)

def timeit(fn, reps=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps

t_knn = timeit(knn)
q = F.normalize(torch.randn(3368, 2048, generator=g, device=dev), dim=1)
gal = F.normalize(torch.randn(15913, 2048, generator=g, device=dev), dim=1)
t_dist = timeit(lambda: _dist_block(q, gal, 1.0, True))
print(json.dumps({"knn_12936x2048_k15_ms": round(t_knn * 1e3, 2), "knn_gemm_tflops": round(2 * 12936 ** 2 * 2048 / t_knn / 1e12, 1),
                  "pairwise_3368x15913x2048_ms": round(t_dist * 1e3, 2),
                  "pairwise_tflops": round(2 * 3368 * 15913 * 2048 / t_dist / 1e12, 1)}))
