#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel trace and/or PMC counter collection) into small per-kernel summaries.

usage: prof_summary.py <rocprof output dir> <summary.csv> [--steps K]

Kernel names are reduced to their template head (conv_fwd_kernel<128,128,...> stays distinct per instantiation);
counter values are summed per kernel and divided by launches.  Written for the box: the raw traces are tens of MB,
the summaries a few KB (profiles/ holds the summaries)."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name[:120]


def main():
    d, out = sys.argv[1], sys.argv[2]
    steps = 1
    if "--steps" in sys.argv:
        steps = int(sys.argv[sys.argv.index("--steps") + 1])
    rows = []
    tr = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    cc = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if cc:
        agg = defaultdict(lambda: defaultdict(float))
        cnt = defaultdict(set)
        with open(cc[0]) as f:
            for r in csv.DictReader(f):
                k = short(r["Kernel_Name"])
                agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
                cnt[k].add(r["Dispatch_Id"])
        names = sorted({c for v in agg.values() for c in v})
        rows.append(["kernel", "launches"] + ["%s_total" % n for n in names] + ["%s_per_launch" % n for n in names])
        for k in sorted(agg, key=lambda k: -sum(agg[k].values())):
            n = len(cnt[k])
            rows.append([k, n] + ["%.1f" % agg[k][c] for c in names] + ["%.2f" % (agg[k][c] / n) for c in names])
    elif tr:
        agg = defaultdict(lambda: [0, 0.0, 1e30, 0.0])
        with open(tr[0]) as f:
            for r in csv.DictReader(f):
                k = short(r["Kernel_Name"])
                dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
                a = agg[k]
                a[0] += 1
                a[1] += dur
                a[2] = min(a[2], dur)
                a[3] = max(a[3], dur)
        tot = sum(a[1] for a in agg.values())
        rows.append(["kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct", "calls_per_step", "ms_per_step"])
        for k in sorted(agg, key=lambda k: -agg[k][1]):
            a = agg[k]
            rows.append([k, a[0], "%.1f" % a[1], "%.2f" % (a[1] / a[0]), "%.2f" % a[2], "%.2f" % a[3],
                         "%.2f" % (100 * a[1] / tot), "%.1f" % (a[0] / steps), "%.3f" % (a[1] / steps / 1e3)])
    else:
        sys.exit("no rocprofv3 csv found under %s" % d)
    with open(out, "w", newline="") as f:
        csv.writer(f).writerows(rows)
    print("wrote", out, len(rows) - 1, "kernels")


if __name__ == "__main__":
    main()
