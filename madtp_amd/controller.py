"""Host-side compression controller of the reference's drivers: the per-epoch temperature
step of every compress_*_dtp.py driver, their `calculate_temperature()` search, and a fvcore-free GFLOPs figure for both.

The reference measures `Cur_Gflops` with fvcore's FlopCountAnalysis (one multiply-accumulate counts as one "flop"; e.g.
compress_nlvr_dtp.py:93-99) and steps the pruning temperature towards `Target_Gflops = Ori_Gflops * (1 - p)`.  fvcore is not
available on this stack, so the figure comes from the analytic counters of madtp_amd.harness / madtp_amd.workloads evaluated
on the OBSERVED per-layer token counts (Block.last_prune / BertLayer.last_prune): for the unpruned BLIP-NLVR model at 384x384
and 20 text tokens it gives 132.30 against the reference's hard-coded Ori_Gflops = 132.54 (compress_nlvr_dtp.py:162).

The drivers do NOT share one step rule: each has its own ladder of (|Cur - Target| threshold, step) pairs, restated below as
data with the reference span.  `Ori_Gflops` of the retrieval / VQA / CLIP drivers are fvcore counts of the TRAINING forward
(momentum encoders, ITM negatives, answer decoder), which is out of this path's scope: the eval-path counters of
madtp_amd.workloads are therefore compared with their own unpruned value (ratio = 1 - p), as bench.py's `flops_ratio_vs_unpruned`
does, and the constants are kept only as the drivers' defaults."""
from . import harness

# compress_*_dtp.py constants (fvcore, per sample): nlvr :162, caption :216, retrieval (coco / flickr) :383 / :380,
# retrieval_clip :281, vqa :239
ORI_GFLOPS = {"nlvr": 132.54, "caption": 65.7, "retrieval": 153.2, "retrieval_clip": 395.7, "vqa": 186.1}

# per-epoch step ladders: ((threshold, step), ...) tried in order on d = |Cur_Gflops - Target_Gflops| with `d > threshold`,
# then the final `else` step (None: the driver has no else branch - the temperature stays)
EPOCH_LADDERS = {
    "nlvr": (((30, 1.0), (10, 0.5), (5, 0.25), (1, 0.1)), 0.01),                                   # compress_nlvr_dtp.py:175-200
    "retrieval": (((50, 0.5), (30, 0.3), (20, 0.2), (10, 0.1), (5, 0.05), (2, 0.02)), 0.01),       # compress_retrieval_dtp.py:402-434
    "retrieval_clip": (((50, 0.5), (30, 0.3), (20, 0.2), (10, 0.1), (5, 0.05), (2, 0.02)), 0.01),  # compress_retrieval_clip_dtp.py:300-332
    "caption": (((50, 0.5), (30, 0.3), (20, 0.2), (10, 0.1), (5, 0.05), (2, 0.02)), 0.01),         # compress_caption_dtp.py:235-267
    "vqa": (((50, 0.25), (30, 0.15), (10, 0.1), (5, 0.05), (2, 0.01)), None),                      # compress_vqa_dtp.py:245-268
}

# calculate_temperature(): (start temperature, tolerance of the while condition, ladder, else step, batches averaged,
#                           extra break `Cur - Target < tolerance` after a measurement (VQA only))
SEARCH = {
    # compress_retrieval_dtp.py:257-300 (the flickr driver is identical); the `> 5` rung is unreachable inside the loop
    "retrieval": (0.0, 10, ((100, 1.0), (50, 0.5), (30, 0.3), (20, 0.2), (10, 0.1), (5, 0.05)), 0.02, 20, False),
    # compress_caption_dtp.py:107-150
    "caption": (1.0, 10, ((100, 1.0), (50, 0.5), (30, 0.3), (20, 0.2), (10, 0.1), (5, 0.05)), 0.02, 20, False),
    # compress_retrieval_clip_dtp.py:174-227
    "retrieval_clip": (1.0, 5, ((100, 0.5), (50, 0.25), (30, 0.15), (20, 0.1), (10, 0.05)), 0.02, 30, False),
    # compress_vqa_dtp.py:120-165 (no else rung; breaks as soon as Cur - Target < 10)
    "vqa": (0.0, 10, ((100, 1.0), (50, 0.5), (30, 0.3), (20, 0.2), (10, 0.1)), None, 20, True),
}


def nlvr_gflops(vit_lens, txt_lens, image_size=384, text_len=20):
    """GFLOPs per NLVR sample (two images + text) in fvcore's convention (1 MAC = 1 flop) from the token counts entering
    each layer, e.g. harness.token_lengths(trace['vit'], n0)."""
    n0 = (image_size // 16) ** 2 + 1
    return harness.nlvr_forward_flops(vit_lens, txt_lens, n0, text_len) / 2.0 / 1e9


def workload_gflops(workload, lens):
    """GFLOPs per sample (fvcore convention) of one of madtp_amd.workloads' configurations at the observed token counts
    `lens` (workload.lens(model) after a forward), or unpruned for lens=None."""
    return workload.flops(lens) / 2.0 / 1e9


def _ladder_step(d, ladder, else_step):
    for thr, step in ladder:
        if d > thr:
            return step
    return else_step


def step_temperature(temperature, cur_gflops, target_gflops, task="nlvr"):
    """One per-epoch controller step of the named driver (EPOCH_LADDERS)."""
    ladder, else_step = EPOCH_LADDERS[task]
    up = cur_gflops > target_gflops  # `if Cur_Gflops > Target_Gflops: ... else: ...`
    step = _ladder_step(abs(cur_gflops - target_gflops), ladder, else_step)
    if step is None:
        return temperature
    return temperature + step if up else temperature - step


def run_controller(measure_gflops, temperature, p, ori_gflops, epochs, task="nlvr"):
    """The epoch loop around it (compress_nlvr_dtp.py:169-201): measure_gflops(T) -> Cur_Gflops of an epoch run at
    temperature T.  Returns the list of (epoch, temperature, cur_gflops)."""
    target = ori_gflops * (1 - p)
    cur = ori_gflops
    log = []
    for epoch in range(epochs):
        if epoch > 0:
            temperature = step_temperature(temperature, cur, target, task)
        cur = measure_gflops(temperature)
        log.append((epoch, temperature, cur))
    return log


def calculate_temperature(measure_gflops, cur_gflops, target_gflops, task="retrieval", max_iters=10000):
    """The drivers' initial temperature search (SEARCH): step the temperature by the driver's ladder and re-measure until
    Cur_Gflops is within the driver's tolerance of Target_Gflops.  measure_gflops(T) stands for the drivers' fvcore loop over
    `count_num` batches (:292-300) and returns the averaged figure.  Returns (cur_gflops, temperature) like the reference;
    max_iters bounds the loop (the reference's is unbounded)."""
    t, tol, ladder, else_step, _count_num, vqa_break = SEARCH[task]
    for _ in range(max_iters):
        if not (target_gflops - cur_gflops > tol or cur_gflops - target_gflops > tol):
            break
        step = _ladder_step(abs(cur_gflops - target_gflops), ladder, else_step)
        if step is not None:
            t = t + step if cur_gflops > target_gflops else t - step
        cur_gflops = measure_gflops(t)
        if vqa_break and cur_gflops - target_gflops < tol:
            break
    return cur_gflops, t
