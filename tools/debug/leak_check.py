"""Does a long run grow device memory?  python tools/debug/leak_check.py KEY [steps]  (diagnostic, GPU box)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    key = sys.argv[1] if len(sys.argv) > 1 else "2"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    sys.stdout, out = sys.stderr, sys.stdout
    w = bench.WORKLOADS[key]()
    w.build(dev, 0)
    marks = []
    for i in range(steps):
        w.step()
        if i in (9, steps // 2, steps - 1):
            torch.cuda.synchronize()
            marks.append((i + 1, torch.cuda.memory_allocated() >> 20, torch.cuda.memory_reserved() >> 20))
    print("config %s: (step, allocated MiB, reserved MiB) %s  losses %s" % (key, marks, w.losses()), file=out)


if __name__ == "__main__":
    main()
