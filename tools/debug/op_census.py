"""Which torch ops (and which HIP runtime calls) does one step of a bench workload issue?  (diagnostic, GPU box)
python tools/debug/op_census.py KEY   — torch.profiler over two steps: counts per step of every CPU-side op name, of the runtime
calls (hipMemcpyAsync, hipLaunchKernel, ...) and of the device-side memcpy / memset records with their direction."""
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

key = sys.argv[1] if len(sys.argv) > 1 else "2"
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
w = bench.WORKLOADS[key]()
w.build(dev, 0)
for _ in range(4):
    w.step()
torch.cuda.synchronize()
steps = 2
acts = [torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA]
with torch.profiler.profile(activities=acts, with_stack=True) as prof:
    for _ in range(steps):
        w.step()
    torch.cuda.synchronize()
by = collections.Counter()
for ev in prof.events():
    n = ev.name
    if n.startswith("void ") or "kernel" in n.lower() and "(" in n:
        continue
    by[(str(ev.device_type).split(".")[-1], n[:90])] += 1
for (dt, name), n in by.most_common(60):
    print("%7.1f / step  %-6s %s" % (n / steps, dt, name))
# python frames above the memcpy runtime calls
sites = collections.Counter()
for ev in prof.events():
    if "emcpy" in ev.name and ev.stack:
        fr = [f for f in ev.stack if "reid-gan_amd" in f or "bench.py" in f]
        sites[(ev.name[:40], (fr[0] if fr else ev.stack[0]).replace(ROOT + "/", "")[:120])] += 1
for (name, site), n in sites.most_common(30):
    print("%7.1f / step  %-40s %s" % (n / steps, name, site))
