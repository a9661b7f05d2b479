#!/usr/bin/env python
"""Per-kernel register / spill / LDS table of one .hip source (hipcc -Rpass-analysis=kernel-resource-usage, gfx950).
usage: python tools/kernel_resources.py madtp_amd/csrc/gemm.hip [name filter]"""
import re
import subprocess
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-function",
                      "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"], capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"remark: +(.*?) \[-Rpass", line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = t.split(":", 1)[1].strip()
        rows[cur] = {}
    elif cur and ":" in t:
        k, v = t.rsplit(":", 1)
        rows[cur][k.strip()] = v.strip()
for name, r in rows.items():
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    if flt and flt not in dem:
        continue
    print(f"{dem[:100]:100s} vgpr {r.get('VGPRs')} agpr {r.get('AGPRs')} sgpr {r.get('SGPRs')} scratch {r.get('ScratchSize [bytes/lane]')} "
          f"occ {r.get('Occupancy [waves/SIMD]')} vspill {r.get('VGPRs Spill')} sspill {r.get('SGPRs Spill')} lds {r.get('LDS Size [bytes/block]')}")
