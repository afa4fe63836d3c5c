"""Host cost of one step (diagnostic, GPU box):  python tools/debug/host_cost.py KEY [crops]

Builds bench.py's workload KEY at a tiny batch (default 4 crops), where the kernels take far less time than their launches, so
the step time IS the host's enqueue cost; then profiles the Python side of the same steps (cProfile, own time) to show where
that cost sits: ctypes calls into the C ABI, tensor allocation, tape bookkeeping, autograd, ATen.
"""
import cProfile
import functools
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    key = sys.argv[1] if len(sys.argv) > 1 else "2"
    crops = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    cls = bench.WORKLOADS[key]
    cls.crops = crops
    if key in ("2", "4b"):
        base = bench.fdgan_opt
        bench.fdgan_opt = functools.partial(base, batch_size=crops // 2) if key == "2" else base
    w = cls()
    w.build(dev, 0)
    for _ in range(5):
        w.step()
    torch.cuda.synchronize()
    n = 10
    t0 = time.perf_counter()
    for _ in range(n):
        w.step()
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print("config %s at %d crops: %.2f ms/step, host enqueue %.2f ms/step" % (key, crops, 1e3 * t_all / n, 1e3 * t_host / n))
    from rg_hip import lib as L
    if os.environ.get("RG_PROFILE_BACKWARD") == "1":
        # the autograd engine runs backward() on its own device thread, which cProfile does not see: single-threaded mode runs the
        # network backward programs in the calling thread, so their Python cost shows up in the table below
        torch.autograd.set_multithreading_enabled(False)
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(n):
        w.step()
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("tottime")
    rows = []
    for (fn, line, name), (cc, nc, tt, ct, callers) in st.stats.items():
        rows.append((tt, ct, nc, "%s:%d(%s)" % (fn.replace(ROOT + "/", ""), line, name)))
    rows.sort(reverse=True)
    tot = sum(r[0] for r in rows)
    print("profiled host time %.2f ms/step (cProfile overhead included); top own-time entries, per step:" % (1e3 * tot / n))
    for tt, ct, nc, where in rows[:45]:
        print("  %7.3f ms own  %7.3f ms cum  %7.1f calls  %s" % (1e3 * tt / n, 1e3 * ct / n, nc / n, where[-110:]))
    del L


if __name__ == "__main__":
    main()
