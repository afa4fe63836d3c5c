#!/usr/bin/env python3
"""Idle time between kernels in a rocprofv3 kernel trace: union of busy intervals vs wall time per step."""
import csv, glob, os, sys
d = sys.argv[1]; steps = int(sys.argv[2])
f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(f)))
# keep the last `steps` steps worth: drop the first 40% (warm-up / build) of the timeline
# kernels per step are constant: keep the last `steps` of the `total` steps by kernel count
total = int(sys.argv[3]) if len(sys.argv) > 3 else steps
iv = iv[len(iv) - (len(iv) * steps) // total:]
busy = 0; gaps = []; cur_s, cur_e = iv[0]
for s, e in iv[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; gaps.append(s - cur_e); cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
wall = iv[-1][1] - iv[0][0]
gaps.sort()
n = len(gaps)
print("per step: wall %.2f ms busy %.2f ms |" % (wall / 1e6 / steps, busy / 1e6 / steps), end=" ")
print("kernels %d  wall %.2f ms  busy %.2f ms (%.1f%%)  idle %.2f ms in %d gaps; median gap %.2f us, p90 %.2f us, >20us: %d (%.2f ms)" % (
    len(iv), wall / 1e6, busy / 1e6, 100.0 * busy / wall, (wall - busy) / 1e6, n, gaps[n // 2] / 1e3, gaps[int(n * 0.9)] / 1e3,
    sum(1 for g in gaps if g > 20000), sum(g for g in gaps if g > 20000) / 1e6))
