#!/usr/bin/env python
"""Kernel durations of a rocprofv3 rocpd result in LAUNCH ORDER, consecutive identical (name, grid) dispatches grouped:
the kernel-only time of each microbenchmark case (event timing of short kernels measures the launch rate instead)."""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
pat = sys.argv[2] if len(sys.argv) > 2 else ""
rows = c.execute("select name, grid_x, (end-start)/1e3 from kernels order by start").fetchall()
cur, acc = None, []
def flush():
    if cur and (pat in cur[0]) and len(acc) >= 5:
        a = sorted(acc)
        print(f"{cur[0][:64]:64s} grid={cur[1]:7d} n={len(acc):3d} median={a[len(a)//2]:7.1f} min={a[0]:7.1f} us")
for n, g, d in rows:
    if (n, g) != cur:
        flush(); cur, acc = (n, g), []
    acc.append(d)
flush()
