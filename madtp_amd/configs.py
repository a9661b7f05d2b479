"""Benchmark configurations and their calibrated temperatures.

The reference's only runtime knob is `temperature`; "p" is a GFLOPs-reduction target reached by a per-epoch
controller on real data (compress_nlvr_dtp.py:162-201).  With synthetic weights the temperature is calibrated by
tools/calibrate_temperature.py --task <config> (CPU oracle, bisection on the analytic FLOP ratio of the observed token
counts of the workload in madtp_amd/workloads.py).  Entries: (task, samples per GPU, p) -> the JSON record the tool printed
(committed verbatim).  Random-weight pruning patterns differ from trained checkpoints (none are available offline): the
importance rule saturates early, which is why retrieval's p = 0.75 needs a temperature in the thousands.
"""

CALIBRATED = {
    ('nlvr', 64, 0.5): {"task": "nlvr", "batch": 64, "p": 0.5, "size": 224, "seed": 0, "temperature": 8.612223847001898, "flops_ratio": 0.49703609682247707, "lens": {"vit": [134, 112, 97, 92, 87, 84, 83, 82, 82, 81, 81, 81], "text": [20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20]}, "full_gflops_per_sample": 88.041483264},
    ('retrieval', 128, 0.75): {"task": "retrieval", "batch": 128, "p": 0.75, "size": 224, "seed": 0, "temperature": 2449.489742783178, "flops_ratio": 0.24920149086372603, "lens": {"vit": [102, 59, 39, 25, 18, 14, 11, 10, 9, 8, 8, 8], "text": [31, 27, 24, 22, 20, 18, 16, 15, 14, 13, 12, 11], "mm": [31, 28, 26, 24, 22, 20, 18, 17, 16, 15, 14, 14]}, "full_gflops_per_sample": 54.902956032},
    ('clip', 128, 0.5): {"task": "clip", "batch": 128, "p": 0.5, "size": 224, "seed": 0, "temperature": 6.062487508511834, "flops_ratio": 0.5047922411833896, "lens": {"vit": [148, 123, 102, 92, 89, 84, 82, 79, 77, 74, 70, 69], "text": [77, 47, 47, 47, 47, 47, 47, 47, 47, 47, 47, 47]}, "full_gflops_per_sample": 45.579595776},
    ('vqa', 32, 0.5): {"task": "vqa", "batch": 32, "p": 0.5, "size": 480, "seed": 0, "temperature": 2.800457982639407, "flops_ratio": 0.5028383825942014, "lens": {"vit": [605, 524, 486, 458, 454, 445, 435, 429, 425, 423, 420, 418], "mm": [20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20]}, "full_gflops_per_sample": 217.579327488},
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
