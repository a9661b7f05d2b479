#!/usr/bin/env python
"""Per-kernel summary (calls, total, avg, min, max, %) of a rocprofv3 rocpd SQLite result (`*_results.db`), the
format this image's rocprofv3 writes for --kernel-trace --stats.  usage: rocpd_stats.py results.db [header text]
MADTP_STATS_SKIP=n: start at the (n+1)-th forward (a forward starts with patchify_kernel) - the first forward of a process also
prepares every weight (fused-projection concatenations, max|w| reductions, casts: ~600 one-time launches that are no part of a step)."""
import sqlite3
import sys

db = sys.argv[1]
c = sqlite3.connect(db)
# kernels in front of the first forward (patchify_kernel opens one) are model construction - e.g. the ~10 k integer launches of the
# on-device synthetic weight generator - and are left out
import os
skip = int(os.environ.get("MADTP_STATS_SKIP", "0"))
starts = [r[0] for r in c.execute("select start from kernels where name like '%patchify%' order by start").fetchall()]
t0 = starts[min(skip, len(starts) - 1)] if starts else 0
rows = c.execute("select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3 "
                 "from kernels where start >= ? group by name order by 3 desc", (t0,)).fetchall()
tot = sum(r[2] for r in rows)
for h in sys.argv[2:]:
    print("# " + h)
if skip:
    print(f"# (the first {skip} forward(s) - weight preparation - left out: {len(starts) - min(skip, len(starts) - 1)} forwards below)")
print(f"# total kernel time {tot / 1e3:.3f} ms")
print(f"{'kernel':110s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>9s} {'min_us':>8s} {'max_us':>8s} {'pct':>6s}")
for r in rows:
    print(f"{r[0][:110]:110s} {r[1]:6d} {r[2] / 1e3:10.3f} {r[3]:9.1f} {r[4]:8.1f} {r[5]:8.1f} {100 * r[2] / tot:6.2f}")
