#!/usr/bin/env python
"""Per-forward wall and kernel-busy time from a rocprofv3 kernel trace of bench.py (a forward starts at patchify_kernel):
the busy time is the noise-free figure of merit for kernel work (bench.py's wall clock varies +-2.5 % between boxes)."""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select start, end, name from kernels order by start").fetchall()
starts = [r[0] for r in rows if "patchify" in r[2]]
for a, b in zip(starts[1:-1], starts[2:]):
    ks = [r for r in rows if a <= r[0] < b]
    print(f"forward: wall {(b - a) / 1e6:6.2f} ms  kernel-busy {sum(r[1] - r[0] for r in ks) / 1e6:6.2f} ms  kernels {len(ks)}")
