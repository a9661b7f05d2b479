#!/usr/bin/env python
"""profiles/*_pmc_gemm_bf16.json from the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; rocpd SQLite): HBM-side bytes per
launch averaged over ALL bf16 GEMM launches (gemm_ws_kernel + gemm_kernel<bf16>) - the kernel bench.py's roofline object
describes.  gfx950 correction as in tools/rocpd_pmc.py / MI355X_MICROARCH.md: FETCH_SIZE x2 for 16-B/lane streams.
usage: pmc_gemm_json.py FETCH.db WRITE.db 'command line' > profiles/rNN_pmc_gemm_bf16.json"""
import json, sqlite3, sys


def total(db, counter):
    c = sqlite3.connect(db)
    rows = c.execute("select name, count(*), sum(counter_value) from pmc_events where counter_name=? group by name", (counter,)).fetchall()
    n = v = 0
    per = {}
    for name, cnt, val in rows:
        if "gemm_ws_kernel" in name or ("gemm_kernel" in name and "unsigned short" in name):
            n += cnt; v += val
            per[name[:80]] = {"launches": cnt, "kib_per_launch": val / cnt}
    return n, v, per


fn, fv, fper = total(sys.argv[1], "FETCH_SIZE")
wn, wv, wper = total(sys.argv[2], "WRITE_SIZE")
out = {
    "kernel": "bf16 GEMM (gemm_ws_kernel + gemm_kernel<bf16>): all madtp_gemm launches of the command below",
    "launches": fn,
    "fetch_size_kib_per_launch_raw": fv / fn,
    "write_size_kib_per_launch": wv / wn,
    "hbm_bytes_per_launch": int((2 * fv / fn + wv / wn) * 1024),
    "correction": "FETCH_SIZE x2 (gfx950 counts 128-B requests at 64 B for 16-B/lane streams, MI355X_MICROARCH.md HBM section); "
                  "FETCH and WRITE collected in separate --pmc passes",
    "command": sys.argv[3] if len(sys.argv) > 3 else "",
    "per_kernel_fetch": fper, "per_kernel_write": wper,
}
print(json.dumps(out, indent=1))
