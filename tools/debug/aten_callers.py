"""Which Python lines of a step still go through ATen?  (diagnostic, GPU box)

    python tools/debug/aten_callers.py 5 [steps]

Runs bench.py's workload for the given BASELINE configuration under torch.profiler with Python stacks and prints, per ATen
operator that launches device work (copy_, fill_, cat, clone, contiguous, add, mul, ...), the innermost repo source lines
that called it with their call counts per step.  The step path is meant to be hand-written kernels only; every line listed
here is either host bookkeeping that should not touch the device or a candidate for one of the fused kernels.
"""
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    key = sys.argv[1] if len(sys.argv) > 1 else "5"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    w = bench.WORKLOADS[key]()
    w.build(dev, 0)
    for _ in range(3):
        w.step()
    torch.cuda.synchronize()
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
        for _ in range(steps):
            w.step()
        torch.cuda.synchronize()
    by_op = collections.defaultdict(collections.Counter)
    for ev in prof.events():
        name = ev.name
        if not name.startswith("aten::"):
            continue
        if ev.cpu_parent is not None and ev.cpu_parent.name.startswith("aten::"):
            continue                                            # count the outermost ATen call only
        where = "?"
        for fr in (ev.stack or []):
            if ROOT in fr and "/torch/" not in fr and "aten_callers" not in fr:
                where = fr.replace(ROOT + "/", "")
                break
        by_op[name][where] += 1
    skip = {"aten::empty", "aten::empty_like", "aten::view", "aten::reshape", "aten::as_strided", "aten::empty_strided",
            "aten::detach", "aten::alias", "aten::slice", "aten::select", "aten::narrow", "aten::unsqueeze", "aten::squeeze",
            "aten::t", "aten::transpose", "aten::permute", "aten::expand", "aten::size", "aten::stride", "aten::is_contiguous",
            "aten::numel", "aten::dim", "aten::_unsafe_view", "aten::view_as", "aten::flatten", "aten::unflatten",
            "aten::is_same_size", "aten::result_type", "aten::lift_fresh", "aten::resize_", "aten::set_", "aten::chunk",
            "aten::split", "aten::unbind", "aten::requires_grad_", "aten::is_leaf", "aten::item", "aten::_local_scalar_dense"}
    total = 0
    for name, c in sorted(by_op.items(), key=lambda kv: -sum(kv[1].values())):
        if name in skip:
            continue
        n = sum(c.values())
        total += n
        print("%-32s %7.1f / step" % (name, n / steps))
        for where, k in c.most_common(12):
            print("      %6.1f  %s" % (k / steps, where))
    print("total device-touching ATen calls per step: %.1f" % (total / steps))


if __name__ == "__main__":
    main()
