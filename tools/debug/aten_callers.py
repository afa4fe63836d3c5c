"""Which Python lines of a step still go through ATen?  (diagnostic, GPU box)

    python tools/debug/aten_callers.py 5 [steps]

Runs bench.py's workload for the given BASELINE configuration under a TorchDispatchMode that records the Python stack of every
ATen call and prints, per ATen operator that launches device work (copy_, fill_, cat, clone, contiguous, add, mul, ...), the innermost repo source lines
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
    import traceback
    from torch.utils._python_dispatch import TorchDispatchMode

    by_op = collections.defaultdict(collections.Counter)

    class Recorder(TorchDispatchMode):
        """every ATen call the main thread dispatches (the autograd engine's worker thread — custom Function backward bodies —
        is outside a dispatch mode; those bodies are the tape programs, which call the C ABI directly)"""

        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            where = "?"
            for fr in reversed(traceback.extract_stack(limit=24)):
                fn = fr.filename
                if fn.startswith(ROOT) and "aten_callers" not in fn and "/torch/" not in fn:
                    where = "%s:%d (%s)" % (fn.replace(ROOT + "/", ""), fr.lineno, fr.name)
                    break
            by_op["aten::" + func.__name__.split(".")[0]][where] += 1
            return func(*args, **(kwargs or {}))

    with Recorder():
        for _ in range(steps):
            w.step()
        torch.cuda.synchronize()
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
