"""Several forwards in flight on ONE GPU (MI355X-first throughput design, no reference counterpart: the reference's eval loops
run one batch at a time on the default stream, compress_nlvr_dtp.py:73-99).

Why: the forward has two regimes.  The vision encoder's GEMMs fill all 256 CUs; the text encoder is ~170 dependent launches
of 5-20 us on 1280 rows that leave most of the chip idle (a third of the step at the headline batch).  Consecutive batches
are independent (k = max_b count couples samples only WITHIN a batch), so batch i+1's vision encoder can run under batch i's
text encoder.  Each in-flight forward gets its own host thread, HIP stream, scratch workspace and model replica (same
deterministic weights; per-module records such as `last_prune` stay private to a replica); the per-layer host read of k spins
inside the library with the GIL released, the k hand-over slots are claimed per call (csrc/prune.hip), and no kernel of the
path waits on another workgroup being resident, so concurrent streams cannot deadlock each other.
Measured on MI355X (profiles/r03_inflight.txt, same box per line group; final figures in (i)): NLVR2 headline 19-20 k serial
-> 22.6-23.1 k images/s with two and 23.8-24.2 k with THREE forwards in flight (four: 21.6-22.5 k), retrieval 17.5 k -> 27.2 k
with three, BLIP-VQA 3.3 k -> 4.7 k with three, CLIP (two chip-filling towers) 17.1-18.0 k -> 15.9-22.8 k with two from run to run
(three: 15 k): (final: FOUR in flight, two of them on high-priority streams: NLVR 25.1-25.5 k, retrieval 27.5-29.0 k, VQA 4.75 k)
bench.py takes 4 in flight, 1 for CLIP.  Three details matter: (1) the workers run the encoder-level C entry points
(madtp_vit_encoder / madtp_bert_encoder) - on the per-layer Python path their progress hinges on GIL hand-overs and the result
swings between 17 k and 22 k from run to run; (2) half of the workers' streams have HIGH priority, the others normal: with equal
priorities three in flight give 20.7 k (the forwards contend kernel by kernel), with one preferred stream 24 k; (3) a host lock
that keeps the workers' vision encoders from overlapping (forced anti-phase) was tried and dropped: no gain on NLVR, retrieval
22.8 k -> 14.5 k.
"""
import threading

import torch


class InflightRunner:
    """n_inflight workers, each = (model replica, resident inputs, HIP stream, host thread).  run(steps) executes `steps` forwards
    in total, step i on worker i % n, every worker's steps in order on its own stream; returns after all of them completed."""

    def __init__(self, workload, n_inflight, temperature, batch, device="cuda", seed0=0, models=None):
        self.w, self.T, self.n = workload, temperature, int(n_inflight)
        self.device = torch.device(device)
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.models = list(models) if models is not None else [workload.build(self.device) for _ in range(self.n)]
        self.inputs = [workload.inputs(batch, seed=seed0 + i) for i in range(self.n)]
        import os
        # stream priorities: the first n // 2 workers (at least one) HIGH, the others normal (HIP has these two levels;
        # MADTP_INFLIGHT_PRIO="p0,p1,..." overrides).  Measured at the headline (profiles/r03_inflight.txt (e), (i), (j)): with equal
        # priorities the forwards contend kernel by kernel (three in flight: 20.7 k images/s), with preferred streams they settle
        # into a stable interleave: two (h,n) 22.6-23.1 k, three (h,n,n) 23.8-24.5 k, four (h,h,n,n) 25.1-25.5 k, four (h,n,n,n)
        # 21.6-22.5 k, five (h,h,n,n,n) 24.1-24.3 k, six (h,h,h,n,n,n) 22.0 k
        env = os.environ.get("MADTP_INFLIGHT_PRIO", "")
        if env.strip():
            prio = [int(x) for x in env.split(",") if x.strip()]
        else:
            prio = [-1] * max(1, self.n // 2)
        self.streams = [torch.cuda.Stream(device=self.device, priority=(prio[i] if i < len(prio) else 0)) for i in range(self.n)]
        # GEMM dispatch hint while the workers run (hip.gemm_set_sq_cost; per workload, measured): None = leave the default
        self.sq_cost = getattr(workload, "inflight_sq_cost", None)
        self.small_tile = getattr(workload, "inflight_small_tile", None)
        if os.environ.get("MADTP_INFLIGHT_SQ_COST"):  # A/B runs: "0" = no hint
            self.sq_cost = float(os.environ["MADTP_INFLIGHT_SQ_COST"]) or None
        if os.environ.get("MADTP_INFLIGHT_SMALL_TILE"):  # A/B runs: "-1" = no hint
            self.small_tile = int(os.environ["MADTP_INFLIGHT_SMALL_TILE"])
            self.small_tile = None if self.small_tile < 0 else self.small_tile
        self.errors = []
        self.last = [None] * self.n  # output of each worker's most recent step

    def _work(self, i, steps, mode):
        try:
            from . import runtime
            runtime.set_precision(mode)  # the precision mode is thread-local state: the workers run the caller's
            runtime.set_encoder_call_preference(True)  # layer loops in C: progress must not hinge on GIL hand-overs
            torch.cuda.set_device(self.device)
            with torch.cuda.stream(self.streams[i]), torch.no_grad():
                for _ in range(steps):
                    self.last[i] = self.w.step(self.models[i], self.inputs[i], self.T)
            self.streams[i].synchronize()
        except BaseException as e:  # surfaced by run()
            self.errors.append(e)

    def run(self, steps, workers=None):
        """workers (optional): use only the first `workers` of the n in-flight slots (bench.py times the headline with two and
        the parity mode with three of the same runner: a second runner's fresh streams may share hardware queues with the
        first one's - measured 9.1 k instead of 10.8 k images/s)."""
        n = self.n if workers is None else max(1, min(int(workers), self.n))
        per = [steps // n + (1 if i < steps % n else 0) if i < n else 0 for i in range(self.n)]
        main = torch.cuda.current_stream(self.device)
        for s in self.streams:
            s.wait_stream(main)
        from . import runtime
        mode = runtime.get_precision()
        threads = [threading.Thread(target=self._work, args=(i, per[i], mode), name=f"madtp-inflight-{i}")
                   for i in range(self.n) if per[i]]
        from . import hip
        prev_cost = prev_small = None
        if len(threads) > 1:  # dispatch hints for a GPU shared by several forwards (include/madtp_hip.h); results do not change
            if self.sq_cost:
                prev_cost = hip.gemm_set_sq_cost(self.sq_cost)
            if self.small_tile is not None:
                prev_small = hip.gemm_set_small_tile(self.small_tile)
        try:
            for t in threads:
                t.start()
            for t in threads:
                t.join()
        finally:
            if prev_cost is not None:
                hip.gemm_set_sq_cost(prev_cost)
            if prev_small is not None:
                hip.gemm_set_small_tile(prev_small)
        for s in self.streams:
            main.wait_stream(s)
        if self.errors:
            err, self.errors = self.errors[0], []
            raise err
