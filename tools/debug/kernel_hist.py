"""Per-grid-size histogram of one kernel in a rocprofv3 --kernel-trace csv directory: python tools/debug/kernel_hist.py <dir> <kernel substring>"""
import collections
import csv
import glob
import sys

d, pat = sys.argv[1], sys.argv[2]
files = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
h = collections.defaultdict(list)
for f in files:
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            wg = int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1) // 256
            h[wg].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(len(v) for v in h.values())
print("%s: %d dispatches" % (pat, tot))
for wg in sorted(h):
    v = h[wg]
    print("  %8d workgroups  x %5d   avg %7.2f us  min %7.2f  total %8.1f us" % (wg, len(v), sum(v) / len(v), min(v), sum(v)))
