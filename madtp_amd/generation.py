"""Beam-search decoding for the BLIP text decoder: the `text_decoder.generate(num_beams=...)` call sites of the reference
(models/blip_vqa.py:134-140, models/blip.py:189-196).  The search itself is transformers 4.15's (environment.yml:266, a
dependency that is not vendored in the reference): generation_utils.py `beam_search` + generation_beam_search.py
`BeamSearchScorer` / `BeamHypotheses` + `MinLengthLogitsProcessor`; oracle/madtp_oracle.py::beam_search restates it for the tests.

Split of work: per step the decoder runs ONE new token per live beam against its layers' self-attention key / value cache
(round 5: incremental decoding, models/med.py:1071-1094; step_fn receives the beam re-ordering of the step before - the
`_reorder_cache` of :1091-1094 - through `beam_src`; a step_fn without the parameter re-runs the whole prefix), the LM head at the
last position only; `madtp_beam_topk` (csrc/lmhead.hip) does the
log-softmax, the beam-score addition, the EOS suppression below min_length and the top-2k selection over num_beams * V
candidates per item on the GPU; the 2 * num_beams winners per item come to the host (one small copy per step - the search is
inherently sequential in the step) where the hypothesis book-keeping runs, as it does in the library."""
import torch

from . import hip


class BeamHypotheses:
    """generation_beam_search.py BeamHypotheses (4.15): score = sum_logprobs / len(hyp) ** length_penalty."""

    def __init__(self, num_beams, length_penalty, early_stopping):
        self.num_beams, self.length_penalty, self.early_stopping = num_beams, length_penalty, early_stopping
        self.beams, self.worst_score = [], 1e9

    def __len__(self):
        return len(self.beams)

    def add(self, hyp, sum_logprobs):
        score = sum_logprobs / (len(hyp) ** self.length_penalty)
        if len(self) < self.num_beams or score > self.worst_score:
            self.beams.append((score, hyp))
            if len(self) > self.num_beams:
                ranked = sorted([(s, idx) for idx, (s, _) in enumerate(self.beams)])
                del self.beams[ranked[0][1]]
                self.worst_score = ranked[1][0]
            else:
                self.worst_score = min(score, self.worst_score)

    def is_done(self, best_sum_logprobs, cur_len):
        if len(self) < self.num_beams:
            return False
        if self.early_stopping:
            return True
        return self.worst_score >= best_sum_logprobs / cur_len ** self.length_penalty



# Pinned host buffers of the asynchronous done / unfinished flag copies: one grow-only buffer per (thread, dtype) instead of a
# hipHostMalloc + free per generate() call (an allocation of pinned memory is expensive and implicitly synchronising, which partly
# undid the sync-free loop).  A call's view is refilled before use; calls of one thread do not overlap.
import threading as _threading

_PINNED = _threading.local()


def _pinned_flags(shape, dtype, fill):
    n = 1
    for d in shape:
        n *= int(d)
    pool = getattr(_PINNED, "pool", None)
    if pool is None:
        pool = _PINNED.pool = {}
    buf = pool.get(dtype)
    if buf is None or buf.numel() < n:
        buf = pool[dtype] = torch.empty((max(n, 4096),), dtype=dtype).pin_memory()
    view = buf[:n].view(*shape)
    view.fill_(fill)
    return view

def beam_search(step_fn, input_ids, num_beams, max_length, min_length, eos_token_id, pad_token_id, n_vocab,
                repetition_penalty=1.0, length_penalty=1.0, early_stopping=False):
    """step_fn(input_ids [B * num_beams, t] on the GPU[, beam_src]) -> f32 last-position scores [B * num_beams, >= n_vocab] (GPU,
    unit column stride); beam_src (when step_fn takes it): int64 [B * num_beams] on the GPU - row i of input_ids continues row
    beam_src[i] of the PREVIOUS call (None on the first call); input_ids: the prompt already repeated num_beams times per item.
    -> int64 [B, <= max_length] on the GPU."""
    import inspect
    import os
    takes_src = "beam_src" in inspect.signature(step_fn).parameters
    if os.environ.get("MADTP_BEAM_DEVICE", "1") != "0" and input_ids.shape[1] < max_length:
        return _beam_search_device(step_fn, takes_src, input_ids, num_beams, max_length, min_length, eos_token_id, pad_token_id, n_vocab,
                                   repetition_penalty, length_penalty, early_stopping)
    beam_src = None
    dev = input_ids.device
    n, cur_len = input_ids.shape
    B = n // num_beams
    hyps = [BeamHypotheses(num_beams, length_penalty, early_stopping) for _ in range(B)]
    done = [False] * B
    beam_scores = torch.zeros((B, num_beams), dtype=torch.float32)
    beam_scores[:, 1:] = -1e9
    beam_scores = beam_scores.view(-1)
    ids_host = input_ids.cpu()
    while True:
        logits = step_fn(input_ids, beam_src=beam_src) if takes_src else step_fn(input_ids)
        suppress = eos_token_id if (min_length is not None and min_length > -1 and cur_len < min_length) else -1
        sc, ix = hip.beam_topk(logits, beam_scores.to(dev), num_beams, n_vocab, suppress_token=suppress,
                               prev_ids=input_ids.contiguous() if repetition_penalty != 1.0 else None,
                               repetition_penalty=repetition_penalty)
        sc, ix = sc.cpu(), ix.cpu().to(torch.int64)
        next_indices, next_tokens = ix // n_vocab, ix % n_vocab
        nb_scores = torch.zeros((B, num_beams), dtype=torch.float32)
        nb_tokens = torch.zeros((B, num_beams), dtype=torch.int64)
        nb_rows = torch.zeros((B, num_beams), dtype=torch.int64)
        for b in range(B):
            if done[b]:
                nb_tokens[b, :] = pad_token_id
                continue
            slot = 0
            for rank in range(2 * num_beams):
                if int(ix[b, rank]) < 0:
                    continue
                tok, s, row = int(next_tokens[b, rank]), float(sc[b, rank]), b * num_beams + int(next_indices[b, rank])
                if tok == eos_token_id:
                    if rank >= num_beams:
                        continue
                    hyps[b].add(ids_host[row].tolist(), s)
                else:
                    nb_scores[b, slot], nb_tokens[b, slot], nb_rows[b, slot] = s, tok, row
                    slot += 1
                if slot == num_beams:
                    break
            if slot < num_beams:
                raise RuntimeError("beam search: fewer than num_beams open continuations among the 2 * num_beams candidates")
            done[b] = done[b] or hyps[b].is_done(float(sc[b].max()), cur_len)
        beam_scores = nb_scores.view(-1)
        ids_host = torch.cat([ids_host[nb_rows.view(-1), :], nb_tokens.view(-1, 1)], dim=-1)
        input_ids = ids_host.to(dev)
        beam_src = nb_rows.view(-1).to(dev) if takes_src else None
        cur_len += 1
        if all(done) or cur_len >= max_length:
            break
    return _finalize(hyps, done, ids_host, beam_scores, num_beams, max_length, eos_token_id, pad_token_id, dev)


def _finalize(hyps, done, ids_host, beam_scores, num_beams, max_length, eos_token_id, pad_token_id, dev):
    """BeamSearchScorer.finalize: the open beams of unfinished items become hypotheses, the best hypothesis per item is returned."""
    B = len(hyps)
    for b in range(B):
        if done[b]:
            continue
        for j in range(num_beams):
            row = b * num_beams + j
            hyps[b].add(ids_host[row].tolist(), float(beam_scores[row]))
    best = [sorted(h.beams, key=lambda x: x[0])[-1][1] for h in hyps]
    lens = [len(h) for h in best]
    out = torch.full((B, min(max(lens) + 1, max_length)), pad_token_id, dtype=torch.int64)
    for b, h in enumerate(best):
        out[b, :lens[b]] = torch.tensor(h, dtype=torch.int64)
        if lens[b] < max_length:
            out[b, lens[b]] = eos_token_id
    return out.to(dev)


def _beam_search_device(step_fn, takes_src, input_ids, num_beams, max_length, min_length, eos_token_id, pad_token_id, n_vocab,
                        repetition_penalty, length_penalty, early_stopping):
    """beam_search with the per-step book-keeping on the device (round 5, madtp_beam_update): sequences, beam scores, source rows and
    the hypothesis lists never leave the GPU during the search - no host round trip per step, the host queues step after step; it
    looks at the items' done flags through asynchronous copies (an event per step, polled without blocking) and stops queueing once a
    completed copy shows every item finished - steps queued past that point only emit pad tokens for finished items, the result
    does not depend on when the host notices.  One synchronisation at the end brings the hypotheses to the host for the library's
    finalize.  MADTP_BEAM_DEVICE=0: the host book-keeping above."""
    dev = input_ids.device
    n, cur_len = input_ids.shape
    B = n // num_beams
    st = hip.BeamState(B, num_beams, max_length, pad_token_id, dev)
    cur = 0
    st.ids[0][:, :cur_len] = input_ids
    flags = _pinned_flags((max_length, B), torch.int32, 0)
    pending = []
    beam_src = None
    while True:
        ids = st.ids[cur][:, :cur_len]
        logits = step_fn(ids, beam_src=beam_src) if takes_src else step_fn(ids.contiguous())
        suppress = eos_token_id if (min_length is not None and min_length > -1 and cur_len < min_length) else -1
        sc, ix = hip.beam_topk(logits, st.beam_scores, num_beams, n_vocab, suppress_token=suppress,
                               prev_ids=ids if repetition_penalty != 1.0 else None, repetition_penalty=repetition_penalty)
        hip.beam_update(st, cur, sc, ix, n_vocab, cur_len, float(cur_len) ** length_penalty, eos_token_id, pad_token_id, early_stopping)
        cur = 1 - cur
        beam_src = st.beam_src.clone() if takes_src else None  # (the step reads it after the next update has been queued)
        flags[cur_len].copy_(st.done, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        pending.append((ev, cur_len))
        cur_len += 1
        stop = cur_len >= max_length
        while pending and pending[0][0].query():
            _, i = pending.pop(0)
            stop = stop or bool(flags[i].all())
        if stop:
            break
    torch.cuda.synchronize()
    if int(st.err.cpu()):
        raise RuntimeError("beam search: fewer than num_beams open continuations among the 2 * num_beams candidates")
    hn, ho, hs, hl, ht = st.hyp_n.cpu(), st.hyp_order.cpu(), st.hyp_score.cpu(), st.hyp_len.cpu(), st.hyp_tok.cpu()
    worst, done = st.worst.cpu(), [bool(d) for d in st.done.cpu()]
    hyps = []
    for b in range(B):
        h = BeamHypotheses(num_beams, length_penalty, early_stopping)
        for i in range(int(hn[b])):
            p = int(ho[b, i])
            h.beams.append((float(hs[b, p]), ht[b, p, :int(hl[b, p])].tolist()))
        h.worst_score = float(worst[b])
        hyps.append(h)
    return _finalize(hyps, done, st.ids[cur][:, :cur_len].cpu(), st.beam_scores.cpu(), num_beams, max_length, eos_token_id, pad_token_id, dev)


def sample(step_fn, input_ids, max_length, min_length, eos_token_id, pad_token_id, n_vocab, top_p, top_k=50, repetition_penalty=1.0,
           generator=None):
    """Nucleus sampling: transformers 4.15 `sample` (generation_utils.py) as models/blip.py:175-186 reaches it - processors
    (repetition penalty on the raw scores, EOS = -inf below min_length), warpers (top-k 50 = config.top_k, then top-p), one draw per
    row; a row that has produced EOS keeps emitting pad_token_id; stops when every row is finished or at max_length.  The draw is
    `madtp_sample_top_p` at one uniform number per row and step from `generator` (a torch.Generator on the rows' device; None: the
    default generator) - the same distribution as torch.multinomial on the warped scores, not the same random stream.
    step_fn as for beam_search (beam_src is always None here).  -> int64 [B, <= max_length] on the GPU."""
    import inspect
    takes_src = "beam_src" in inspect.signature(step_fn).parameters
    dev = input_ids.device
    B, cur_len = input_ids.shape
    unfinished = torch.ones((B,), dtype=torch.int64, device=dev)
    # no host round trip per step (round 5): the rows' unfinished flags reach the host through asynchronous copies polled without
    # blocking; steps queued after every row has finished only append pad tokens, which are cut off below
    flags = _pinned_flags((max_length + 1, B), torch.int64, 1)
    pending, n_steps = [], None
    while True:
        logits = step_fn(input_ids, beam_src=None) if takes_src else step_fn(input_ids)
        suppress = eos_token_id if (min_length is not None and min_length > -1 and cur_len < min_length) else -1
        u = torch.rand((B,), device=dev, dtype=torch.float32, generator=generator)
        nxt = hip.sample_top_p(logits, u, n_vocab, top_p, top_k=top_k, suppress_token=suppress,
                               prev_ids=input_ids.contiguous() if repetition_penalty != 1.0 else None,
                               repetition_penalty=repetition_penalty)
        nxt = nxt * unfinished + pad_token_id * (1 - unfinished)
        input_ids = torch.cat([input_ids, nxt[:, None]], dim=-1)
        cur_len += 1
        unfinished = unfinished * (nxt != eos_token_id).to(torch.int64)
        flags[cur_len].copy_(unfinished, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        pending.append((ev, cur_len))
        stop = cur_len >= max_length
        while pending and pending[0][0].query():
            _, i = pending.pop(0)
            if n_steps is None and int(flags[i].max()) == 0:
                n_steps = i
        if stop or n_steps is not None:
            break
    torch.cuda.synchronize()
    for _, i in pending:  # the length at which the library's loop stops: the first step after which no row is unfinished
        if n_steps is None and int(flags[i].max()) == 0:
            n_steps = i
    return input_ids[:, :n_steps] if n_steps is not None else input_ids
