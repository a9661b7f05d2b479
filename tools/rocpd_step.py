#!/usr/bin/env python
"""Per-forward wall and kernel-busy time from a rocprofv3 kernel trace of bench.py (a forward starts at patchify_kernel):
the busy time is the noise-free figure of merit for kernel work (bench.py's wall clock varies +-2.5 % between boxes).
With a second argument "phases": the same per phase (vision = patchify .. bert_embed, text = bert_embed .. last kernel, tail =
idle until the next forward's patchify), with the union-busy time and the idle time of each (serial traces)."""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
phases = len(sys.argv) > 2 and sys.argv[2] == "phases"
rows = c.execute("select start, end, name from kernels order by start").fetchall()
starts = [r[0] for r in rows if "patchify" in r[2]]


def union(ks):
    busy, cur_s, cur_e = 0, None, None
    for s, e, _ in ks:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        busy += cur_e - cur_s
    return busy


for a, b in zip(starts[1:-1], starts[2:]):
    ks = [r for r in rows if a <= r[0] < b]
    print(f"forward: wall {(b - a) / 1e6:6.2f} ms  kernel-busy {sum(r[1] - r[0] for r in ks) / 1e6:6.2f} ms  kernels {len(ks)}")
    if phases:
        te = [r[0] for r in ks if "bert_embed" in r[2]]
        if not te:
            continue
        t = te[0]
        last = max(r[1] for r in ks)
        vis, txt = [r for r in ks if r[0] < t], [r for r in ks if r[0] >= t]
        print(f"   vision {(t - a) / 1e3:7.1f} us (busy {union(vis) / 1e3:7.1f}, {len(vis)} kernels)   text {(last - t) / 1e3:7.1f} us "
              f"(busy {union(txt) / 1e3:7.1f}, {len(txt)} kernels)   tail idle {(b - last) / 1e3:6.1f} us")
