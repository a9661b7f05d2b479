"""Build-your-own harness around the mirrors: synthetic-weight model construction, one forward with a per-layer
pruning trace, token-identity bookkeeping, analytic FLOP counting.  Used by tests/, bench.py and
__graft_entry__.py (product side: never imports oracle/)."""
import torch

from . import specs, synth
from .blip_nlvr import BLIP_NLVR

MERGED_BASE = 100000


def build_nlvr(image_size=224, seed=0, device="cuda"):
    """BLIP_NLVR mirror with the deterministic synthetic weights (same bits as the golden generator used)."""
    model = BLIP_NLVR(image_size=image_size, evaluate=True).eval().to(device)
    # generated ON the target device (same bits as the numpy generator, synth.uniform_pm1_torch): the ranks of a multi-GPU job
    # do not queue for the host cores with 259 M hashes each
    model.load_state_dict(specs.synth_weights(specs.blip_nlvr_shapes(image_size), seed, device=device), strict=True)
    return model


def padded_mask(B, L, pad_tail=0):
    """attention_mask [B,L] of ones with a zero tail of b % (pad_tail+1) tokens in sample b (padded captions); pad_tail may also
    be a list of per-sample tail lengths (ragged captions: a short one next to a full-length one)."""
    att = torch.ones(B, L, dtype=torch.long)
    if isinstance(pad_tail, (list, tuple)):
        for b, n in enumerate(pad_tail):
            if n:
                att[b, L - int(n):] = 0
        return att
    if pad_tail:
        for b in range(B):
            att[b, L - (b % (pad_tail + 1)):] = 0
    return att


def nlvr_inputs(B, image_size=224, L=20, seed=0, device="cuda", pad_tail=0):
    images = synth.synth_images(2 * B, image_size, seed, device=device)  # (generated ON the device: same bits, no large H2D copy)
    ids = synth.synth_token_ids(B, L, seed).to(device)
    return images, {"input_ids": ids, "attention_mask": padded_mask(B, L, pad_tail).to(device)}, \
        torch.zeros(B, dtype=torch.long, device=device)


def build_retrieval(image_size=224, seed=0, device="cuda"):
    """BLIP_Retrieval mirror with the deterministic synthetic weights (same bits as the golden generator used)."""
    from .blip_retrieval import BLIP_Retrieval
    model = BLIP_Retrieval(image_size=image_size, evaluate=True).eval().to(device)
    msg = model.load_state_dict(specs.synth_weights(specs.blip_retrieval_shapes(image_size), seed, device=device), strict=False)
    assert not msg.unexpected_keys and all("query_model" in k or "position_ids" in k for k in msg.missing_keys), msg
    return model


class RetrievalLoader:
    """Stand-in for the reference's data loader over a synthetic evaluation set: iterating yields (image batch, captions,
    image ids); .dataset.text slices to {'input_ids','attention_mask'}; .dataset.image has one entry per image."""

    def __init__(self, batches, ids, att):
        self.batches = batches
        outer = self

        class _Text:
            def __len__(self):
                return ids.shape[0]

            def __getitem__(self, sl):
                return {"input_ids": ids[sl], "attention_mask": att[sl]}

        class _DS:
            text = _Text()
            image = list(range(sum(b.shape[0] for b in batches)))
        self.dataset = _DS()
        del outer

    def __len__(self):
        return len(self.batches)

    def __iter__(self):
        n = 0
        for b in self.batches:
            yield b, ["caption"] * b.shape[0], torch.arange(n, n + b.shape[0])
            n += b.shape[0]


def retrieval_inputs(n_img, img_bs, n_txt, image_size=224, L=35, seed=0, device="cpu"):
    """Synthetic retrieval evaluation set: image loader batches (list of [<=img_bs,3,S,S]), caption ids/masks [n_txt,L] padded
    to max_length 35 with ragged true lengths (pad id 0), as tokenizer(padding='max_length') yields (compress_retrieval_dtp.py:102)."""
    images = synth.synth_images(n_img, image_size, seed, device=device)
    batches = [images[i:i + img_bs] for i in range(0, n_img, img_bs)]
    ids = synth.synth_token_ids(n_txt, L, seed + 1, first_id=101)
    att = torch.ones_like(ids)
    for t in range(n_txt):
        n = 6 + (7 * t + 3 * seed) % (L - 5)  # 6 .. L true tokens
        ids[t, n - 1] = 102  # [SEP]
        ids[t, n:] = 0
        att[t, n:] = 0
    return batches, ids.to(device), att.to(device)


def _cpu_info(info):
    if info is None:
        return None
    return {k: (v.detach().cpu() if torch.is_tensor(v) else v) for k, v in info.items()}


@torch.no_grad()
def run_nlvr(model, images, text, targets, temperature):
    """-> (logits, {'vit': [info]*12, 'text': [info]*12}) with infos on the CPU."""
    logits = model(images, text, targets, temperature=temperature, train=False)
    trace = {"vit": [_cpu_info(b.last_prune) for b in model.visual_encoder.blocks],
             "text": [_cpu_info(l.last_prune) for l in model.text_encoder.encoder.layer]}
    return logits, trace


def compose_ids(trace, n0):
    """per-layer `indices` (positions in the CURRENT patch sequence, any order) -> per-layer, per-sample SETS of
    original token ids; the merged token created at layer l gets id MERGED_BASE+l."""
    out, ids = [], None
    for l, info in enumerate(trace):
        if info is None or not info.get("pruned"):
            out.append(None)
            continue
        idx = info["indices"]
        idx = idx.numpy() if torch.is_tensor(idx) else idx
        B, k = idx.shape
        if ids is None:
            ids = [list(range(n0)) for _ in range(B)]
        new_ids = [[ids[b][j] for j in idx[b]] + [MERGED_BASE + l] for b in range(B)]
        out.append([set(r[:-1]) for r in new_ids])
        ids = new_ids
    return out


def token_lengths(trace, n0):
    """sequence length (incl. CLS) AFTER each layer."""
    out, n = [], n0
    for info in trace:
        if info is not None and info.get("pruned"):
            n = info["k"] + 2
        out.append(n)
    return out


# ---- analytic FLOPs; true FLOPs = 2 x MAC ---------------------------------------------------
D, K_SD, FFN = 768, 100, 3072


def vit_flops(n_in, n_out, att_ft=True):
    """one ViT layer, one image: n_in tokens into attention, n_out into the MLP."""
    f = 6 * n_in * D * D + 4 * n_in * n_in * D + 2 * n_in * D * D + 2 * (n_in - 1) * D * K_SD + 16 * n_out * D * D
    if att_ft:
        f += 2 * K_SD * (n_in - 1) * D
    return f


def nlvr_forward_flops(vit_lens, txt_lens, n0=197, l0=20, att_ft=True):
    """whole BLIP_NLVR.forward for ONE sample (2 images + text).  vit_lens/txt_lens: lengths after each layer."""
    total = 2 * (2 * (n0 - 1) * D * D)  # patch embeds
    n = n0
    for n_out in vit_lens:
        total += 2 * vit_flops(n, n_out, att_ft)
        n = n_out
    n_img = vit_lens[-1]
    l = l0
    for i, l_out in enumerate(txt_lens):
        total += 8 * l * D * D + 4 * l * l * D + 2 * (l - 1) * D * K_SD + (2 * K_SD * (l - 1) * D if att_ft else 0)
        cross = 2 * l_out * D * D + 4 * n_img * D * D + 4 * l_out * n_img * D + 2 * l_out * D * D
        total += 2 * cross + (2 * l_out * 2 * D * D if i >= 6 else 0)
        total += 16 * l_out * D * D
        l = l_out
    total += 2 * D * D + 2 * D * 2
    return total
