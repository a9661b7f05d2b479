"""Kept-token index match of the HIP path against the CPU oracle (shared by bench.py's index_match leg and the -m gpu tests).
The oracle is the checker here, never the thing measured."""
import time

import torch


def set_match(mine, ref):
    """mine / ref: per-layer lists of per-sample id sets (None = layer not pruned) -> (pairs, exact matches, sum of Jaccard)."""
    pairs = eq = 0
    jac = 0.0
    for a, b in zip(mine, ref):
        if a is None and b is None:
            continue
        if a is None or b is None:
            pairs += len(a or b)
            continue
        for x, y in zip(a, b):
            pairs += 1
            eq += int(x == y)
            jac += len(x & y) / max(1, len(x | y))
    return pairs, eq, jac


def nlvr_index_match(model, T, modes, B=64, seed=11, teacher_forced=True):
    """Free-running and teacher-forced (every ViT block fed the ORACLE's input of that layer, so one early flip does not
    cascade) kept-set match of each precision mode vs oracle/madtp_oracle.py on B samples of the synthetic NLVR workload."""
    from madtp_amd import harness, runtime, specs
    from oracle import madtp_oracle as O
    images, text, targets = harness.nlvr_inputs(B, 224, 20, seed)
    W = specs.synth_weights(specs.blip_nlvr_shapes(224), 0)
    tr = {}
    t0 = time.perf_counter()
    with torch.no_grad():
        ref_logits = O.blip_nlvr_forward(W, images.cpu(), text["input_ids"].cpu(), text["attention_mask"].cpu(), T, trace=tr)
    rep = {"batch": B, "temperature": T, "oracle": "oracle/madtp_oracle.py (CPU fp32 restatement of the reference)",
           "oracle_forward_s": round(time.perf_counter() - t0, 2)}
    for mode in modes:
        with runtime.precision(mode):
            logits, trace = harness.run_nlvr(model, images, text, targets, T)
        pairs = eq = 0
        jac = 0.0
        for side, n0 in (("vit", 196), ("text", 19)):
            p, e, j = set_match(harness.compose_ids(trace[side], n0), O.compose_ids(tr[side], n0))
            pairs, eq, jac = pairs + p, eq + e, jac + j
        rep[mode] = {"kept_set_exact_match": round(eq / max(1, pairs), 4), "mean_jaccard": round(jac / max(1, pairs), 4),
                     "sample_layer_pairs": pairs, "max_abs_dlogit": round((logits.cpu() - ref_logits).abs().max().item(), 6)}
    if not teacher_forced:
        return rep
    xs, vtr = [], []
    with torch.no_grad():
        O.vit_forward(W, "visual_encoder.", images.cpu(), W["space_dict"], T, trace=vtr, layer_inputs=xs)
    venc = model.visual_encoder
    for mode in modes:
        pairs = eq = 0
        jac = 0.0
        with runtime.precision(mode), torch.no_grad():
            for l, blk in enumerate(venc.blocks):
                if vtr[l] is None or not vtr[l]["pruned"]:
                    continue
                x = xs[l].cuda().contiguous()
                ta, _, _ = venc.img_query_model(x[:, 1:, :], model.space_dict, return_token_att=True)
                blk(x, False, 0, T, ta)
                mine = blk.last_prune
                if mine is None or not mine["pruned"]:
                    pairs += x.shape[0]
                    continue
                a, b = mine["indices"].cpu().numpy(), vtr[l]["indices"].numpy()
                for r in range(x.shape[0]):
                    sa, sb = set(a[r].tolist()), set(b[r].tolist())
                    pairs += 1
                    eq += int(sa == sb)
                    jac += len(sa & sb) / max(1, len(sa | sb))
        rep[mode]["vit_layerwise_exact_match"] = round(eq / max(1, pairs), 4)
        rep[mode]["vit_layerwise_jaccard"] = round(jac / max(1, pairs), 4)
    return rep
