"""Per-shape throughput of the conv implicit-GEMM kernels for every geometry of the FD-GAN / cluster-contrast
steps (development aid; prints TFLOP/s for fwd / dgrad / wgrad)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "reid-gan_amd"))
import torch
from rg_hip import ops

dev = torch.device("cuda:0")


def resnet_shapes(H, W, l4_stride=2):
    out = [("stem7x7", 3, H, W, 64, 7, 2, 3)]
    h, w = H // 4, W // 4
    cin = 64
    for li, (width, n, s) in enumerate(((64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, l4_stride)), 1):
        for bi in range(n):
            st = s if bi == 0 else 1
            out.append(("l%d.%d.conv1" % (li, bi), cin, h, w, width, 1, 1, 0))
            out.append(("l%d.%d.conv2" % (li, bi), width, h, w, width, 3, st, 1))
            if bi == 0:
                out.append(("l%d.%d.down" % (li, bi), cin, h, w, width * 4, 1, st, 0))
            h, w = h // st, w // st
            out.append(("l%d.%d.conv3" % (li, bi), width, h, w, width * 4, 1, 1, 0))
            cin = width * 4
    return out


def uniq(shapes):
    seen, out = {}, []
    for s in shapes:
        k = s[1:]
        if k in seen:
            seen[k][1] += 1
        else:
            seen[k] = [s[0], 1]
            out.append(k)
    return [(seen[k][0], seen[k][1]) + k for k in out]


def timeit(fn, reps=5):
    """ms per call.  Default: the library's per-launch HIP events (rg::ProfScope, on the launch stream) — the kernels of the small
    layers take 10-30 us, less than one Python call, so host-side timing of back-to-back calls measures the host (RG_BENCH_HOST=1)."""
    fn()
    torch.cuda.synchronize()
    if os.environ.get("RG_BENCH_HOST") != "1":
        ops.profile_reset()
        ops.profile_enable(True)
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        ops.profile_enable(False)
        return sum(f["ms"] for f in ops.profile_collect().values()) / reps
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    which = sys.argv[2] if len(sys.argv) > 2 else "resnet"
    if which == "resnet":
        shapes = uniq(resnet_shapes(256, 128))
    elif which == "resnet_cc":
        shapes = uniq(resnet_shapes(256, 128, l4_stride=1))
    else:
        shapes = [("en1", 1, 18, 256, 128, 64, 4, 2, 1), ("en2", 1, 64, 128, 64, 128, 4, 2, 1),
                  ("en3", 1, 128, 64, 32, 256, 4, 2, 1), ("en4", 1, 256, 32, 16, 512, 4, 2, 1),
                  ("en5", 1, 512, 16, 8, 512, 4, 2, 1), ("dp1", 1, 21, 256, 128, 64, 4, 2, 1),
                  ("dp4", 1, 256, 32, 16, 512, 4, 1, 1), ("dp5", 1, 512, 31, 15, 1, 4, 1, 1),
                  ("de1(convT as conv 3->64)", 1, 3, 256, 128, 64, 4, 2, 1)]
    tot = {"fwd": [0.0, 0.0], "dgrad": [0.0, 0.0], "wgrad": [0.0, 0.0]}
    print("%-28s %3s %22s %9s | %8s %8s %8s (TFLOP/s)   ms f/d/w" % ("layer", "cnt", "N,C,H,W->K k s p", "GFLOP", "fwd", "dgrad", "wgrad"))
    only = os.environ.get("RG_BENCH_ONLY")
    for name, cnt, C, H, W, K, k, s, p in shapes:
        if only and name not in only.split(","):
            continue
        x = torch.randn(N, C, H, W, device=dev)
        w = torch.randn(K, C, k, k, device=dev) * 0.05
        y = ops.conv2d_fwd(x, w, s, p)
        dy = torch.randn_like(y)
        fl = 2.0 * y.numel() * C * k * k
        # the (r,s)-major filter copy is built once per optimizer step for a whole network (rg_hip.nn.KrscGroup), not per call:
        # pass it in, as the layers do (rounds 1-3 let every timed call rebuild it: 7-20 us of re-layout inside the multi-tap rows)
        wk = ops.weights_to_krsc(w) if k > 1 and C % 4 == 0 else None
        wkf = wk if C % 16 == 0 else None
        tf = timeit(lambda: ops.conv2d_fwd(x, w, s, p, w_krsc=wkf))
        td = timeit(lambda: ops.conv2d_dgrad(dy, w, (H, W), s, p, w_krsc=wk))
        tw = timeit(lambda: ops.conv2d_wgrad(x, dy, (K, C, k, k), s, p))
        for key, t in (("fwd", tf), ("dgrad", td), ("wgrad", tw)):
            tot[key][0] += cnt * t
            tot[key][1] += cnt * fl
        print("%-28s %3d %22s %9.2f | %8.1f %8.1f %8.1f   %.3f/%.3f/%.3f" % (
            name, cnt, "%d,%d,%d,%d->%d k%d s%d p%d" % (N, C, H, W, K, k, s, p), fl / 1e9,
            fl / tf / 1e9, fl / td / 1e9, fl / tw / 1e9, tf, td, tw))
    for key, (t, f) in tot.items():
        print("TOTAL %-6s %.2f ms  %.1f TFLOP/s" % (key, t, f / t / 1e9))


if __name__ == "__main__":
    main()
