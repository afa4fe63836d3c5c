"""Kernels of a rocprofv3 --kernel-trace csv directory whose launches are long for their size: per kernel the dispatch count, total
time, and the average duration of its SMALL launches (< 512 workgroups) — those should sit near the ~4 us launch floor; more means a
latency chain inside the kernel (one load in flight per thread, serial phases).  python tools/debug/kernel_suspects.py <dir>"""
import collections
import csv
import glob
import sys

d = sys.argv[1]
rows = collections.defaultdict(list)
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        wg = int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1) // max(
            1, int(r["Workgroup_Size_X"]) * int(r.get("Workgroup_Size_Y", 1) or 1) * int(r.get("Workgroup_Size_Z", 1) or 1))
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        rows[name.split("(")[0][:70]].append((wg, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
out = []
for k, v in rows.items():
    small = [t for wg, t in v if wg < 512]
    out.append((sum(t for _, t in v), k, len(v), small))
out.sort(reverse=True)
print("%-70s %7s %10s | small launches (< 512 workgroups): n, avg us, total us" % ("kernel", "n", "total us"))
for tot, k, n, small in out[:60]:
    s = "%6d %8.2f %10.1f" % (len(small), sum(small) / len(small), sum(small)) if small else "     -"
    print("%-70s %7d %10.1f | %s" % (k, n, tot, s))
