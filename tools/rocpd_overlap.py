#!/usr/bin/env python
"""Concurrency of a rocprofv3 kernel trace with several forwards in flight (bench.py --inflight N): per-stream kernel time, the
union of busy intervals, how long two streams' kernels actually overlap, and a merged timeline with the stream of every kernel.
usage: rocpd_overlap.py results.db [n_lines=200] [skip_fraction=0.5]"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
nlines = int(sys.argv[2]) if len(sys.argv) > 2 else 200
skip = float(sys.argv[3]) if len(sys.argv) > 3 else 0.5
cols = [r[1] for r in c.execute("pragma table_info(kernels)").fetchall()]
sid = next((x for x in ("stream_id", "queue_id", "stream", "queue") if x in cols), None)
print("columns:", cols, "-> stream column:", sid)
rows = c.execute(f"select start, end, name, grid_x, {sid or '0'} from kernels order by start").fetchall()
rows = rows[int(len(rows) * skip):]
t0, t1 = rows[0][0], max(r[1] for r in rows)
streams = sorted({r[4] for r in rows})
print(f"{len(rows)} kernels over {(t1 - t0) / 1e6:.2f} ms on streams {streams}")
for s in streams:
    ks = [r for r in rows if r[4] == s]
    print(f"  stream {s}: {len(ks)} kernels, {sum(r[1] - r[0] for r in ks) / 1e6:.2f} ms of kernel time")
# union / overlap by sweeping
ev = sorted([(r[0], 1) for r in rows] + [(r[1], -1) for r in rows])
busy = over = 0
depth, last = 0, ev[0][0]
for t, d in ev:
    if depth >= 1: busy += t - last
    if depth >= 2: over += t - last
    depth += d; last = t
print(f"busy (>=1 kernel running) {busy / 1e6:.2f} ms = {100 * busy / (t1 - t0):.1f} % of the span; >=2 kernels running {over / 1e6:.2f} ms")
names = {s: chr(ord('A') + i) for i, s in enumerate(streams)}
prev_end = {}
for r in rows[:nlines]:
    s = r[4]
    wait = (r[0] - prev_end.get(s, r[0])) / 1e3
    print(f"{(r[0] - t0) / 1e3:9.1f} {names[s]} dur={(r[1] - r[0]) / 1e3:7.1f} since_prev_on_stream={wait:7.1f} grid={r[3]:7d} {r[2][:70]}")
    prev_end[s] = r[1]
