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


# (Rounds 3-5; since round 6 the hints are per-STREAM attributes - madtp_stream_set_sched - and this block only serves
# MADTP_INFLIGHT_GLOBAL_HINTS=1, the A/B switch.)  The GEMM dispatch hints were process-wide state of the library
# (include/madtp_hip.h): runners that overlap in time share ONE
# setting - the first one in sets it and remembers the previous values, the last one out restores them (a nested or concurrent
# runner neither re-applies its own hints nor restores stale ones).
_hint_lock = threading.Lock()
_hint_depth = 0
_hint_prev = (None, None)


def _hints_enter(sq_cost, small_tile):
    global _hint_depth, _hint_prev
    from . import hip
    with _hint_lock:
        if _hint_depth == 0:
            prev_cost = hip.gemm_set_sq_cost(sq_cost) if sq_cost else None
            prev_small = hip.gemm_set_small_tile(small_tile) if small_tile is not None else None
            _hint_prev = (prev_cost, prev_small)
        _hint_depth += 1
    return True


def _hints_exit():
    global _hint_depth, _hint_prev
    from . import hip
    with _hint_lock:
        _hint_depth -= 1
        if _hint_depth == 0:
            prev_cost, prev_small = _hint_prev
            if prev_cost is not None:
                hip.gemm_set_sq_cost(prev_cost)
            if prev_small is not None:
                hip.gemm_set_small_tile(prev_small)
            _hint_prev = (None, None)


def shared_replica(model):
    """A second instance of `model` for another forward in flight that SHARES every parameter, buffer and prepared (compute-dtype)
    weight with it and owns only the per-call records the forward leaves on its modules (`last_prune`, `score_side`, the encoder
    runs, ...).  Round 6 (the review's item 1): four in-flight forwards used to hold four full replicas - 4 x 1.04 GB of f32
    parameters plus their prepared copies, and four address ranges of the same W competing for the L2s / the Infinity Cache.
    The module tree is copied shallowly: each module object is new (own attribute slots), `_parameters` / `_buffers` are the SAME
    dicts as the original's (a load_state_dict or a parameter re-assignment on either is seen by both), `_modules` maps to the
    copied children, PreparedCache objects are shared (and marked so: a weight prepared by one thread is complete before another
    thread's stream reads it)."""
    import copy
    import torch.nn as nn
    from .runtime import PreparedCache
    memo = {}

    def clone(m):
        c = memo.get(id(m))
        if c is not None:
            return c
        c = copy.copy(m)  # new object, shallow copy of __dict__
        memo[id(m)] = c
        c._modules = type(m._modules)((k, (clone(v) if v is not None else None)) for k, v in m._modules.items())
        for stale in ("_enc_weights", "_last_run", "_prepared_weights", "_madtp_params", "_madtp_params_epoch"):
            c.__dict__.pop(stale, None)  # per-instance derived state: rebuilt on first use (from the shared caches)
        return c

    root = clone(model)

    def remap(v):
        if isinstance(v, nn.Module):
            return memo.get(id(v), v)
        if isinstance(v, list):
            return [remap(x) for x in v] if any(isinstance(x, nn.Module) for x in v) else v
        if isinstance(v, tuple):
            return tuple(remap(x) for x in v) if any(isinstance(x, nn.Module) for x in v) else v
        return v

    for c in list(memo.values()):
        for k, v in list(c.__dict__.items()):
            if k in ("_modules", "_parameters", "_buffers"):
                continue
            if isinstance(v, PreparedCache):
                v.shared = True
            else:
                nv = remap(v)
                if nv is not v:
                    c.__dict__[k] = nv
    return root


def partition_from_env(n, default=None):
    """MADTP_INFLIGHT_CUMASK="8,8,8,8" (CUs per XCD of worker 0, 1, ...; they are laid out one after the other: worker i owns CUs
    [sum(c[:i]), sum(c[:i+1])) of every XCD), "0" / "off" = no partition (priority streams), unset = `default`."""
    import os
    env = os.environ.get("MADTP_INFLIGHT_CUMASK", "").strip().lower()
    if not env:
        return default
    if env in ("0", "off", "none"):
        return None
    part = [int(x) for x in env.split(",") if x.strip()]
    if len(part) < n or sum(part[:n]) > 32 or min(part) < 1:
        raise ValueError(f"MADTP_INFLIGHT_CUMASK={env!r}: need {n} positive CU counts per XCD with a sum <= 32")
    return part[:n]


class InflightRunner:
    """n_inflight workers, each = (model replica, resident inputs, HIP stream, host thread).  run(steps) executes `steps` forwards
    in total, step i on worker i % n, every worker's steps in order on its own stream; returns after all of them completed."""

    def __init__(self, workload, n_inflight, temperature, batch, device="cuda", seed0=0, models=None, partition=None):
        """models: one model per worker, or ONE model (the workers then run shared_replica()s of it: one set of weights).
        partition: None = all workers on the whole chip, told apart by stream priority (rounds 3-5), or a list of CUs per XCD
        per worker - every worker then runs on a CU-masked stream that owns its slice of every XCD (hip.MaskedStream; an
        MI355X XCD cannot be masked out as a whole, profiles/r06_cumask_probe.txt)."""
        self.w, self.T, self.n = workload, temperature, int(n_inflight)
        self.device = torch.device(device)
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        if models is not None and not isinstance(models, (list, tuple)):
            models = [models] + [shared_replica(models) for _ in range(self.n - 1)]
        self.models = list(models) if models is not None else [workload.build(self.device) for _ in range(self.n)]
        self.partition = list(partition) if partition else None
        self._masked = {}  # (worker count, worker) -> hip.MaskedStream
        self.inputs = [workload.inputs(batch, seed=seed0 + i) for i in range(self.n)]
        import os
        # stream priorities: the first n // 2 workers (at least one) HIGH, the others normal (HIP has these two levels;
        # MADTP_INFLIGHT_PRIO="p0,p1,..." overrides).  Measured at the headline (profiles/r03_inflight.txt (e), (i), (j)): with equal
        # priorities the forwards contend kernel by kernel (three in flight: 20.7 k images/s), with preferred streams they settle
        # into a stable interleave: two (h,n) 22.6-23.1 k, three (h,n,n) 23.8-24.5 k, four (h,h,n,n) 25.1-25.5 k, four (h,n,n,n)
        # 21.6-22.5 k, five (h,h,n,n,n) 24.1-24.3 k, six (h,h,h,n,n,n) 22.0 k
        # The split follows the number of workers a run() actually uses (the first workers // 2 of THEM are high), so every slot
        # owns one stream of each level and run() picks.
        env = os.environ.get("MADTP_INFLIGHT_PRIO", "")
        self.prio_env = [int(x) for x in env.split(",") if x.strip()] if env.strip() else None
        self._by_prio = [dict() for _ in range(self.n)]  # slot -> {priority: stream}, created on first use (a run() with all n
        #                                                   workers creates exactly n streams, as many as are in use)
        self.streams = [None] * self.n                    # the streams of the most recent run(), by worker
        self.n_high = 0
        # GEMM dispatch hint while the workers run (hip.gemm_set_sq_cost; per workload, measured): None = leave the default
        self.sq_cost = getattr(workload, "inflight_sq_cost", None)
        self.small_tile = getattr(workload, "inflight_small_tile", None)
        if os.environ.get("MADTP_INFLIGHT_SQ_COST"):  # A/B runs: "0" = no hint
            self.sq_cost = float(os.environ["MADTP_INFLIGHT_SQ_COST"]) or None
        if os.environ.get("MADTP_INFLIGHT_SMALL_TILE"):  # A/B runs: "-1" = no hint
            self.small_tile = int(os.environ["MADTP_INFLIGHT_SMALL_TILE"])
            self.small_tile = None if self.small_tile < 0 else self.small_tile
        self.errors = []
        self.stop = threading.Event()  # set by the first worker that fails: the others stop at their next step
        self.last = [None] * self.n  # output of each worker's most recent step

    def _work(self, i, steps, mode):
        try:
            from . import runtime
            runtime.set_precision(mode)  # the precision mode is thread-local state: the workers run the caller's
            runtime.set_encoder_call_preference(True)  # layer loops in C: progress must not hinge on GIL hand-overs
            torch.cuda.set_device(self.device)
            with torch.cuda.stream(self.streams[i]), torch.no_grad():
                for _ in range(steps):
                    if self.stop.is_set():
                        break
                    self.last[i] = self.w.step(self.models[i], self.inputs[i], self.T)
            self.streams[i].synchronize()
        except BaseException as e:  # surfaced by run()
            self.errors.append(e)
            self.stop.set()

    def _partition_for(self, n):
        """CUs per XCD of the n workers of a run(): the configured partition when it names exactly n workers, an even split of its
        total otherwise (a leg that uses fewer workers than slots still owns the same CUs)."""
        if self.partition is None or n < 2:
            return None
        if len(self.partition) == n:
            return self.partition
        total = min(32, sum(self.partition))
        return [total // n + (1 if i < total % n else 0) for i in range(n)]

    def run(self, steps, workers=None):
        """workers (optional): use only the first `workers` of the n in-flight slots (bench.py times the headline with two and
        the parity mode with three of the same runner: a second runner's fresh streams may share hardware queues with the
        first one's - measured 9.1 k instead of 10.8 k images/s)."""
        n = self.n if workers is None else max(1, min(int(workers), self.n))
        per = [steps // n + (1 if i < steps % n else 0) if i < n else 0 for i in range(self.n)]
        main = torch.cuda.current_stream(self.device)
        if self.prio_env is not None:
            high = [i < len(self.prio_env) and self.prio_env[i] < 0 for i in range(self.n)]
        else:
            high = [i < max(1, n // 2) for i in range(self.n)]
        part = self._partition_for(n)
        import os
        per_stream_hints = os.environ.get("MADTP_INFLIGHT_GLOBAL_HINTS", "0") != "1"  # ("1": the rounds-3-5 process-wide setters, A/B)
        for i in range(n):
            if part is not None:
                key = (tuple(part), i)
                if key not in self._masked:
                    from . import hip
                    self._masked[key] = hip.MaskedStream(sum(part[:i]), part[i], device=self.device,
                                                         sq_cost=self.sq_cost or 0.0, small_tile=-2 if self.small_tile is None else self.small_tile)
                self.streams[i] = self._masked[key].stream
                continue
            pr = -1 if high[i] else 0
            if pr not in self._by_prio[i]:
                self._by_prio[i][pr] = torch.cuda.Stream(device=self.device, priority=pr)
                if per_stream_hints and (self.sq_cost or self.small_tile is not None):
                    # the dispatch hints travel with the STREAM (madtp_stream_set_sched, round 6): no process-wide library state, two
                    # runners (or another user of the library) in one process do not see each other's hints
                    from . import hip
                    hip.stream_set_sched(self._by_prio[i][pr], 0, self.sq_cost or 0.0, -2 if self.small_tile is None else self.small_tile)
            self.streams[i] = self._by_prio[i][pr]
        self.n_high = 0 if part is not None else sum(1 for i in range(n) if high[i])
        self.last_partition = part
        self.stop.clear()
        used = [self.streams[i] for i in range(n)]
        for s in used:
            s.wait_stream(main)
        from . import runtime
        mode = runtime.get_precision()
        threads = [threading.Thread(target=self._work, args=(i, per[i], mode), name=f"madtp-inflight-{i}")
                   for i in range(self.n) if per[i]]
        # (CU-masked streams carry their hints themselves - madtp_stream_set_sched - and leave the process-wide state alone)
        hinted = len(threads) > 1 and part is None and not per_stream_hints and _hints_enter(self.sq_cost, self.small_tile)
        try:
            for t in threads:
                t.start()
            for t in threads:
                t.join()
        finally:
            if hinted:
                _hints_exit()
        for s in used:
            main.wait_stream(s)
        if self.errors:
            err, self.errors = self.errors[0], []
            raise err
