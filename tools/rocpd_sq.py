#!/usr/bin/env python
"""Per-kernel SQ counters from rocprofv3 --pmc passes (rocpd SQLite): the MFMA-busy evidence north_star names.

    rocpd_sq.py PASS1.db [PASS2.db ...]

Each pass holds up to 8 SQ counters (+ GRBM_GUI_ACTIVE in its own block); the passes are merged per kernel name (same command,
same launches per pass).  Units (MI355X_MICROARCH.md, cycle-constants table): SQ_BUSY_CYCLES and SQ_VALU_MFMA_BUSY_CYCLES count
cycles summed over the chip's SQs / SIMDs, SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles per wave, so every
figure printed here is a RATIO of counters of the same unit:
  mfma_busy    = SQ_VALU_MFMA_BUSY_CYCLES (chip total) / (GRBM_GUI_ACTIVE x 1024 SIMDs) - share of the launch during which a SIMD's
                 matrix pipe is busy (rocprofv3's derived 'MfmaUtil' falls back to gfx94x formulas on this ROCm, so the raw ratio)
  mops/clk/CU  = SQ_INSTS_VALU_MFMA_MOPS_{F16,BF16,F32} x 512 flop / (GRBM_GUI_ACTIVE x CUs)   vs 4096 (f16/bf16) flop/clk/CU at peak
  wait / issue-stall / active = SQ_WAIT_ANY / SQ_WAIT_INST_ANY / SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES (disjoint, sum ~ 1)
  clock        = GRBM_GUI_ACTIVE / kernel duration (effective shader clock, GHz)
"""
import sqlite3
import sys

CUS, SIMDS = 256, 1024


def load(db):
    """-> {kernel: {counter: (dispatches, chip total per dispatch, instance rows per dispatch, avg duration ns)}}.  pmc_events holds one
    row per (dispatch, counter, hardware instance - XCC / SE / ...): the chip total of a counter is the sum over its instance rows."""
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(pmc_events)").fetchall()]
    key = "dispatch_id" if "dispatch_id" in cols else "start"
    t0 = c.execute(f"select min({key}) from pmc_events where name like '%patchify%'").fetchone()[0] or 0
    rows = c.execute(f"select name, counter_name, count(distinct {key}), count(*), sum(counter_value), avg(duration) from pmc_events "
                     f"where {key} >= ? group by name, counter_name", (t0,)).fetchall()
    per = {}
    for name, ctr, nd, nrows, v, dur in rows:
        per.setdefault(name, {})[ctr] = (nd, float(v) / nd, nrows / nd, float(dur))
    return per


per = {}
for db in sys.argv[1:]:
    for name, d in load(db).items():
        per.setdefault(name, {}).update(d)


def tot(d, k):
    """chip total per dispatch"""
    return d[k][1] if k in d else None


def inst_avg(d, k):
    """per-instance average per dispatch (GRBM_GUI_ACTIVE: one free-running cycle count per instance)"""
    return d[k][1] / d[k][2] if k in d else None


any_k = next(iter(per.values()))
print("# instance rows per dispatch: " + ", ".join(f"{k} {v[2]:.0f}" for k, v in sorted(any_k.items())))
print(f"{'kernel':64s} {'calls':>5s} {'us':>7s} {'clk GHz':>7s} {'mfma_busy':>9s} {'flop/clk/CU':>11s} {'of peak':>7s} "
      f"{'wait':>6s} {'stall':>6s} {'active':>6s} {'waves/SIMD':>10s} {'sq_busy':>8s}")
order = sorted(per, key=lambda k: -max((v[0] * v[3] for v in per[k].values()), default=0))
for name in order:
    d = per[name]
    n = max(v[0] for v in d.values())
    dur = max(v[3] for v in d.values()) / 1e3  # us per call
    gui, busy, mfma = inst_avg(d, "GRBM_GUI_ACTIVE"), tot(d, "SQ_BUSY_CYCLES"), tot(d, "SQ_VALU_MFMA_BUSY_CYCLES")
    wc, wa, wi, ac = tot(d, "SQ_WAVE_CYCLES"), tot(d, "SQ_WAIT_ANY"), tot(d, "SQ_WAIT_INST_ANY"), tot(d, "SQ_ACTIVE_INST_ANY")
    m16 = (tot(d, "SQ_INSTS_VALU_MFMA_MOPS_F16") or 0) + (tot(d, "SQ_INSTS_VALU_MFMA_MOPS_BF16") or 0)
    m32 = tot(d, "SQ_INSTS_VALU_MFMA_MOPS_F32") or 0
    mops = m16 + m32
    peakf = 256.0 if (m32 > 0 and m16 == 0) else 4096.0
    f = lambda x, w=6, p=3: (f"{x:{w}.{p}f}" if x is not None else " " * (w - 1) + "-")  # noqa: E731
    clk = gui / (dur * 1e3) if gui else None
    mb = mfma / (gui * SIMDS) if (mfma is not None and gui) else None
    fpc = mops * 512.0 / (gui * CUS) if (mops and gui) else None
    occ = wc * 4.0 / (gui * SIMDS) if (wc and gui) else None  # quad-cycles of resident waves per SIMD-cycle
    sqb = busy / (gui * d["SQ_BUSY_CYCLES"][2]) if (busy and gui) else None  # share of the launch with any wave on an SQ instance
    print(f"{name[:64]:64s} {n:5d} {dur:7.1f} {f(clk, 7, 2)} {f(mb, 9)} {f(fpc, 11, 0)} {f(fpc / peakf if fpc else None, 7)} "
          f"{f(wa / wc if wa is not None and wc else None)} {f(wi / wc if wi is not None and wc else None)} {f(ac / wc if ac is not None and wc else None)} "
          f"{f(occ, 10, 2)} {f(sqb, 8)}")
