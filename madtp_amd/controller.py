"""Host-side compression controller of the reference's drivers: the per-epoch temperature
step of compress_nlvr_dtp.py:172-200 and a fvcore-free GFLOPs figure for it.

The reference measures `Cur_Gflops` with fvcore's FlopCountAnalysis (compress_nlvr_dtp.py:93-99; one multiply-accumulate
counts as one "flop") and steps the pruning temperature towards `Target_Gflops = Ori_Gflops * (1 - p)`.  fvcore is not
available on this stack, so the figure comes from the analytic counter of madtp_amd.harness evaluated on the OBSERVED
per-layer token counts (Block.last_prune / BertLayer.last_prune): for the unpruned BLIP-NLVR model at 384x384 and 20 text
tokens it gives 132.30 against the reference's hard-coded Ori_Gflops = 132.54 (compress_nlvr_dtp.py:162)."""
from . import harness

ORI_GFLOPS = {"nlvr": 132.54, "caption": 65.7, "retrieval_clip": 395.7}  # compress_*_dtp.py constants (fvcore, per sample)


def nlvr_gflops(vit_lens, txt_lens, image_size=384, text_len=20):
    """GFLOPs per NLVR sample (two images + text) in fvcore's convention (1 MAC = 1 flop) from the token counts entering
    each layer, e.g. harness.token_lengths(trace['vit'], n0)."""
    n0 = (image_size // 16) ** 2 + 1
    return harness.nlvr_forward_flops(vit_lens, txt_lens, n0, text_len) / 2.0 / 1e9


def step_temperature(temperature, cur_gflops, target_gflops):
    """compress_nlvr_dtp.py:175-200 (identical in the other compress_*_dtp.py drivers): one controller step."""
    d = cur_gflops - target_gflops
    sign = 1.0 if d > 0 else -1.0  # `if Cur_Gflops > Target_Gflops: ... else: ...`
    a = abs(d)
    if a > 30:
        step = 1.0
    elif a > 10:
        step = 0.5
    elif a > 5:
        step = 0.25
    elif a > 1:
        step = 0.1
    else:
        step = 0.01
    return temperature + sign * step


def run_controller(measure_gflops, temperature, p, ori_gflops, epochs):
    """The epoch loop around it (:169-201): measure_gflops(T) -> Cur_Gflops of an epoch run at temperature T.
    Returns the list of (epoch, temperature, cur_gflops)."""
    target = ori_gflops * (1 - p)
    cur = ori_gflops
    log = []
    for epoch in range(epochs):
        if epoch > 0:
            temperature = step_temperature(temperature, cur, target)
        cur = measure_gflops(temperature)
        log.append((epoch, temperature, cur))
    return log
