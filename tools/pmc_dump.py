"""Per-dispatch PMC values of the GEMM kernels from a rocprofv3 --pmc rocpd database: kernel, grid, duration, counters."""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, counter_name, counter_value, duration, dispatch_id from pmc_events where name like '%gemm_%' order by dispatch_id").fetchall()
agg = {}
for name, cn, v, dur, did in rows:
    agg.setdefault(did, {"name": name[:60], "dur": dur})[cn] = agg.get(did, {}).get(cn, 0) + v
for did in sorted(agg):
    r = agg[did]
    print(did, r["name"], f"{r['dur']/1e3:8.1f} us", {k: v for k, v in r.items() if k not in ("name", "dur")})
