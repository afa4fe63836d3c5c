"""Who issues the device-to-device copies of a step?  (diagnostic, GPU box)   python tools/debug/copy_callers.py KEY
torch.profiler (CPU activities, with stacks — it sees the autograd engine's thread too) over two steps of bench.py's workload KEY;
prints every aten::copy_ / clone / contiguous / cat call site (innermost repo frame) with its count per step."""
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
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU], with_stack=True) as prof:
    for _ in range(steps):
        w.step()
    torch.cuda.synchronize()
by = collections.Counter()
for ev in prof.events():
    if ev.name not in ("aten::copy_", "aten::clone", "aten::contiguous", "aten::cat", "aten::_to_copy", "aten::add_", "aten::add"):
        continue
    site = "?"
    for fr in ev.stack or ():
        if "reid-gan_amd" in fr or "bench.py" in fr:
            site = fr.replace(ROOT + "/", "")
            break
    else:
        site = (ev.stack[0] if ev.stack else "(no python frame: autograd engine / C++)")
    by[(ev.name, site)] += 1
for (name, site), n in by.most_common(40):
    print("%6.1f / step  %-18s %s" % (n / steps, name, site[:150]))
