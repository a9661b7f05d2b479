#!/usr/bin/env python
"""L2 hit rate and fabric read requests per kernel from one rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum pass
(rocpd SQLite): where the big GEMMs' fabric traffic above their algorithmic bytes comes from (each of the 8 XCDs' L2 streams
its own copy of W; MI355X_MICROARCH.md: the per-XCD L2s are not coherent with each other).   usage: rocpd_l2.py PASS.db"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info(pmc_events)").fetchall()]
key = "start" if "start" in cols else "dispatch_id"
t0 = c.execute(f"select min({key}) from pmc_events where name like '%patchify%'").fetchone()[0] or 0
rows = c.execute(f"select name, counter_name, count(*), sum(counter_value) from pmc_events where {key} >= ? group by name, counter_name",
                 (t0,)).fetchall()
per = {}
for name, ctr, n, v in rows:
    per.setdefault(name, {})[ctr] = (n, v)
print(f"{'kernel':72s} {'calls':>6s} {'L2 hit rate':>12s} {'hits/call':>12s} {'misses/call':>12s} {'EA rdreq/call':>14s}")
for name in sorted(per, key=lambda k: -sum(v[1] for v in per[k].values())):
    d = per[name]
    h, m, r = d.get("TCC_HIT_sum", (1, 0)), d.get("TCC_MISS_sum", (1, 0)), d.get("TCC_EA0_RDREQ_sum", (1, 0))
    n = h[0]
    if h[1] + m[1] == 0:
        continue
    print(f"{name[:72]:72s} {n:6d} {h[1] / (h[1] + m[1]):12.3f} {h[1] / n:12.0f} {m[1] / n:12.0f} {r[1] / max(r[0], 1):14.0f}")
