"""Benchmark configurations and their calibrated temperatures.

The reference's only runtime knob is `temperature`; "p" is a GFLOPs-reduction target reached by a per-epoch
controller on real data (compress_nlvr_dtp.py:162-201).  With synthetic weights the temperature is calibrated by
tools/calibrate_temperature.py (CPU oracle, bisection on the analytic FLOP ratio of the observed token counts).
Entries: (task, samples per GPU, p) -> calibration record (stdout of the tool, committed verbatim).
"""

CALIBRATED = {
    ("nlvr", 64, 0.5): {"temperature": 8.612223847001898, "flops_ratio": 0.49703609682247707, "seed": 0, "size": 224,
                        "len": 20, "vit_lens": [134, 112, 97, 92, 87, 84, 83, 82, 82, 81, 81, 81],
                        "txt_lens": [20] * 12, "full_gflops_per_sample": 88.041483264},
}


def temperature_for(task, batch, p):
    rec = CALIBRATED.get((task, batch, p))
    if rec is None:
        # nearest calibrated batch for the task/p: the FLOP ratio then deviates (k = batch max) and bench.py reports it
        cands = [(abs(b - batch), r) for (t, b, pp), r in CALIBRATED.items() if t == task and pp == p]
        if not cands:
            raise KeyError(f"no calibrated temperature for {(task, batch, p)}; run tools/calibrate_temperature.py")
        rec = min(cands, key=lambda c: c[0])[1]
    return rec["temperature"], rec
