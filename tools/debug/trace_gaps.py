"""GPU idle time and per-stream occupancy of the timed steps from a rocprofv3 kernel trace (rocpd sqlite):
  cd /tmp && rocprofv3 --kernel-trace -d OUT -o t -- python3 bench.py --no-others --no-cpu-baseline --steps 10 --warmup 5
  python tools/debug/trace_gaps.py OUT/t_results.db
Looks at the last 40 % of the dispatches (steady state): union of kernel intervals (busy), idle gaps with the kernels on either
side, and busy time per stream."""
import collections
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select start, end, name, stream_id from kernels order by start"))
sub = rows[int(len(rows) * 0.6):]
span = max(r[1] for r in sub) - sub[0][0]
busy, cur_e, prev, gaps = 0, sub[0][0], None, []
per = collections.Counter()
for s, e, nm, st in sub:
    per[st] += e - s
    if s > cur_e:
        gaps.append((s - cur_e, prev, nm, st))
    if e > cur_e:
        busy += e - max(s, cur_e)
        cur_e, prev = e, nm
print("dispatches %d  span %.2f ms  busy %.2f ms  idle %.2f ms (%.1f %%)" % (len(sub), span / 1e6, busy / 1e6, (span - busy) / 1e6,
                                                                           100.0 * (span - busy) / span))
for st, t in per.most_common():
    print("stream %s: kernel time %.2f ms (%.1f %% of span)" % (st, t / 1e6, 100.0 * t / span))
print("gaps > 5 us: %d, %.2f ms;  gaps <= 5 us: %d, %.2f ms" % (sum(1 for g in gaps if g[0] > 5e3), sum(g[0] for g in gaps if g[0] > 5e3) / 1e6,
                                                             sum(1 for g in gaps if g[0] <= 5e3), sum(g[0] for g in gaps if g[0] <= 5e3) / 1e6))
agg = collections.Counter()
cnt = collections.Counter()
for g, a, b, st in gaps:
    k = ((a or "")[:48].split("(")[0], b[:48].split("(")[0])
    agg[k] += g
    cnt[k] += 1
for k, t in agg.most_common(25):
    print("%8.1f us in %4d gaps  after %-50s before %s" % (t / 1e3, cnt[k], k[0], k[1]))
