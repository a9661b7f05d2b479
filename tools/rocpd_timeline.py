#!/usr/bin/env python
"""Kernel timeline (start offset, duration, grid, name) around the N-th last occurrence of a kernel-name pattern in a rocpd DB.
usage: rocpd_timeline.py results.db pattern [nth_from_end=3] [before=3] [after=30]"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
pat = sys.argv[2]
nth = int(sys.argv[3]) if len(sys.argv) > 3 else 3
before = int(sys.argv[4]) if len(sys.argv) > 4 else 3
after = int(sys.argv[5]) if len(sys.argv) > 5 else 30
seq = c.execute("select name, grid_x, (end-start)/1e3, start, end from kernels order by start").fetchall()
idx = [i for i, r in enumerate(seq) if pat in r[0]]
i = idx[-nth]
t0 = seq[max(i - before, 0)][3]
prev_end = None
for r in seq[max(i - before, 0):i + after]:
    gap = (r[3] - prev_end) / 1e3 if prev_end else 0.0
    print(f"{(r[3] - t0) / 1e3:9.1f} dur={r[2]:7.1f} gap={gap:6.1f} grid={r[1]:7d} {r[0][:80]}")
    prev_end = r[4]
