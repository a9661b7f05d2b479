"""Mirror of the reference's retrieval evaluation path:
models/blip_retrieval.py BLIP_Retrieval (:19-66, the modules evaluation uses) and compress_retrieval_dtp.py evaluate()
(:84-207) - text features, image features with the cross-batch CLS-repeat padding, similarity matrix, ITM re-ranking of
the top k_test candidates in both directions with the multimodal MED encoder.

Same names, constructor arguments, state-dict keys and call signatures.  The momentum encoders / queues / `temp` of the
training loss (:67-93) are not built (their checkpoint keys are ignored by load_state_dict(strict=False)); forward()
- the training loss - raises.  Everything heavy runs through the HIP library (pruned ViT, MED BERT in text and
multimodal mode, projections, ITM head); torch only does the glue the reference does in torch as well (topk, row
gathers, normalisation of [n,256] features) - on the GPU, the image tokens never visit the host."""
import os

import torch
import torch.nn.functional as F
from torch import nn

from . import hip
from .bert import BertConfig
from .blip_nlvr import ENC_TOKEN_ID, create_vit
from .bert import EncoderKVCache
from .med import BertModel
from .runtime import PreparedCache, compute_dtype, lin_of, require_gpu, to_compute


class BLIP_Retrieval(nn.Module):
    def __init__(self, med_config='configs/med_config.json', image_size=384, vit='base', vit_grad_ckpt=False,
                 vit_ckpt_layer=0, embed_dim=256, queue_size=57600, momentum=0.995, negative_all_rank=False,
                 evaluate=False, config=None):
        super().__init__()
        if config is None:
            self.sd_num, self.sd_dim = 100, 768
        else:
            self.sd_num, self.sd_dim = config['sd_num'], config['sd_dim']
        self.space_dict = nn.Parameter(torch.randn(self.sd_num, self.sd_dim))
        self.world_size = int(os.environ.get('WORLD_SIZE', 1))
        self.layers = 12
        self.visual_encoder, vision_width = create_vit(vit, image_size, vit_grad_ckpt, vit_ckpt_layer, 0,
                                                       evaluate=evaluate, sd_dim=self.sd_dim)
        self.tokenizer = None  # set to a BertTokenizer-like callable to pass raw strings, as the reference does
        if isinstance(med_config, str):
            med_config = BertConfig.from_json_file(med_config) if os.path.exists(med_config) else BertConfig.med_default()
        med_config.encoder_width = vision_width
        med_config.evaluate = evaluate
        self.text_encoder = BertModel(config=med_config, add_pooling_layer=False, sd_dim=self.sd_dim)
        text_width = self.text_encoder.config.hidden_size
        self.vision_proj = nn.Linear(vision_width, embed_dim)
        self.text_proj = nn.Linear(text_width, embed_dim)
        self.itm_head = nn.Linear(text_width, 2)
        self.queue_size, self.momentum, self.negative_all_rank = queue_size, momentum, negative_all_rank
        self._cache = PreparedCache()

    def forward(self, image, caption, alpha, idx, temperature=0, train=True):
        raise NotImplementedError("BLIP_Retrieval.forward is the ITC/ITM training loss (momentum encoders, queues, negative "
                                  "mining): out of scope; use blip_retrieval.evaluate() for the evaluation path")

    # ---- the small Linears of the evaluation path on the library GEMM ----
    def _linear(self, key, lin, x32):
        l = lin_of(self._cache, key, [lin])
        return hip.gemm(to_compute(x32.contiguous()), l.w, l.b, out_dtype=torch.float32, n=l.n)

    def project_image(self, cls_rows):
        return F.normalize(self._linear("vp", self.vision_proj, cls_rows), dim=-1)  # compress_retrieval_dtp.py:121-122

    def project_text(self, cls_rows):
        return F.normalize(self._linear("tp", self.text_proj, cls_rows))  # :104

    def itm_score(self, cls_rows):
        return self._linear("itm", self.itm_head, cls_rows)[:, 1]  # :172


def blip_retrieval(pretrained='', **kwargs):
    model = BLIP_Retrieval(**kwargs)
    if pretrained:  # blip_retrieval.py (blip_retrieval) -> models/blip.py:254-278
        from .checkpoint import load_checkpoint
        model, msg = load_checkpoint(model, pretrained)
        print("missing keys:")
        print(msg.missing_keys)
    return model


def _tokens(model, text, device):
    if isinstance(text, dict) or hasattr(text, "input_ids"):
        ids = text["input_ids"] if isinstance(text, dict) else text.input_ids
        att = text["attention_mask"] if isinstance(text, dict) else text.attention_mask
    elif model.tokenizer is not None:
        t = model.tokenizer(text, padding='max_length', truncation=True, max_length=35, return_tensors="pt")
        ids, att = t.input_ids, t.attention_mask
    else:
        raise TypeError("dataset.text must yield {'input_ids','attention_mask'} tensors or model.tokenizer must be set "
                        "(no vocabulary offline)")
    return ids.to(device), att.to(device)


def all_reduce_scores(score_i2t, score_t2i):
    """compress_retrieval_dtp.py:200-203: SUM all-reduce of the two score matrices over the ranks (every rank filled only the
    rows of its slice, the rest is -100), so all entries re-ranked by some rank end up shifted by the same -100*(world-1)
    and the rankings of itm_eval() are those of a single-rank run.  numpy in, numpy out; no-op without a process group."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return score_i2t, score_t2i
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    a, b = torch.from_numpy(score_i2t.copy()).to(dev), torch.from_numpy(score_t2i.copy()).to(dev)
    dist.all_reduce(a, op=dist.ReduceOp.SUM)
    dist.all_reduce(b, op=dist.ReduceOp.SUM)
    return a.cpu().numpy(), b.cpu().numpy()


def rank_rows(n, rank, world_size):
    """Row slice [start, end) of an n-row score matrix that `rank` fills (compress_retrieval_dtp.py:158-162 / :181-183)."""
    step = n // world_size + 1
    return min(n, rank * step), min(n, rank * step + step)


def all_gather_scores(score_i2t, score_t2i):
    """the exchange the path actually needs - every rank contributes only the ROWS it re-ranked (rank_rows) and
    all ranks end up with exactly the matrices of a single-rank evaluation (no -100*(world-1) shift, 1/world of the
    all-reduce's traffic: at COCO 5k x 25k the two SUM all-reduces move 2 x 500 MB per rank).  numpy in, numpy out; no-op
    without a process group.  Slices are padded to the common step so one all_gather per matrix suffices."""
    import numpy as np
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return score_i2t, score_t2i
    world, rank = dist.get_world_size(), dist.get_rank()
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    outs = []
    for m in (score_i2t, score_t2i):
        n, step = m.shape[0], m.shape[0] // world + 1
        s, e = rank_rows(n, rank, world)
        mine = np.full((step, m.shape[1]), -100.0, dtype=m.dtype)
        mine[:e - s] = m[s:e]
        parts = [torch.empty((step, m.shape[1]), dtype=torch.from_numpy(mine).dtype, device=dev) for _ in range(world)]
        dist.all_gather(parts, torch.from_numpy(mine).to(dev))
        full = np.full_like(m, -100.0)
        for r, part in enumerate(parts):
            rs, re = rank_rows(n, r, world)
            full[rs:re] = part[:re - rs].cpu().numpy()
        outs.append(full)
    return outs[0], outs[1]


@torch.no_grad()
def evaluate(model, data_loader, device, config, temperature=0, rank=0, world_size=1, text_bs=256, kv_cache=True):
    """compress_retrieval_dtp.py evaluate() :84-207 -> (score_matrix_i2t, score_matrix_t2i) as numpy arrays and the GFLOPs
    placeholder (0.0: the reference's fvcore count of the TRAINING forward is out of scope, harness.nlvr_forward_flops is
    the analytic counter of this repo).  rank / world_size select this process's row slices exactly as :158-162 / :181-183
    do; the caller all-reduces the two matrices (SUM) when world_size > 1, as :200-203.
    kv_cache: project the image tokens to every layer's cross-attention [k|v] ONCE (EncoderKVCache) and let the re-ranking
    batches index that cache; False reproduces the reference's data flow (each pair re-projects its image)."""
    k_test = config['k_test']
    sd = model.space_dict
    texts = data_loader.dataset.text
    num_text = len(texts)
    text_ids, text_embeds, text_atts = [], [], []
    for i in range(0, num_text, text_bs):  # :100-110
        ids, att = _tokens(model, texts[i:min(num_text, i + text_bs)], device)
        out, _ = model.text_encoder(ids, attention_mask=att, mode='text', space_dict=sd, temperature=temperature)
        text_embeds.append(model.project_text(out.last_hidden_state[:, 0, :]))
        text_ids.append(ids)
        text_atts.append(att)
    text_embeds = torch.cat(text_embeds, 0)
    text_ids = torch.cat(text_ids, 0).clone()
    text_atts = torch.cat(text_atts, 0)
    text_ids[:, 0] = ENC_TOKEN_ID  # :114

    image_feats, image_embeds = [], []
    for image, _caption, _img_id in data_loader:  # :118-125 (the tokens stay on the GPU)
        feat, _ = model.visual_encoder(require_gpu(image.to(device), "image"), space_dict=sd, temperature=temperature)
        image_embeds.append(model.project_image(feat[:, 0, :]))
        image_feats.append(feat)
    image_embeds = torch.cat(image_embeds, 0)
    n = max(f.shape[1] for f in image_feats)  # :141-153: batches pruned to different lengths, padded with their CLS row
    image_feats = torch.cat([torch.cat([f, f[:, 0:1, :].expand(-1, n - f.shape[1], -1)], 1) if f.shape[1] < n else f
                             for f in image_feats], 0)

    cache = EncoderKVCache.build(model.text_encoder, image_feats) if kv_cache else None

    def rerank(ids, att, img_index):
        if cache is not None:
            out = model.text_encoder(ids, attention_mask=att, return_dict=True, space_dict=sd, temperature=temperature,
                                     encoder_kv_cache=cache.select(img_index))[0]
        else:
            enc = image_feats[img_index].contiguous()
            enc_att = torch.ones(enc.shape[:-1], dtype=torch.long, device=device)
            out = model.text_encoder(ids, attention_mask=att, encoder_hidden_states=enc, encoder_attention_mask=enc_att,
                                     return_dict=True, space_dict=sd, temperature=temperature)[0]
        return model.itm_score(out.last_hidden_state[:, 0, :])

    sims_matrix = image_embeds @ text_embeds.t()  # :155
    n_img = sims_matrix.shape[0]
    score_i2t = torch.full((n_img, num_text), -100.0, device=device)
    start, end = rank_rows(n_img, rank, world_size)
    for i in range(start, end):  # :164-174
        topk_sim, topk_idx = sims_matrix[i].topk(k=k_test, dim=0)
        score_i2t[i, topk_idx] = rerank(text_ids[topk_idx], text_atts[topk_idx],
                                        torch.full((k_test,), i, dtype=torch.long, device=device)) + topk_sim

    sims_t = sims_matrix.t()
    score_t2i = torch.full((num_text, n_img), -100.0, device=device)
    start, end = rank_rows(num_text, rank, world_size)
    for i in range(start, end):  # :186-198
        topk_sim, topk_idx = sims_t[i].topk(k=k_test, dim=0)
        score_t2i[i, topk_idx] = rerank(text_ids[i].repeat(k_test, 1), text_atts[i].repeat(k_test, 1), topk_idx) + topk_sim
    return score_i2t.cpu().numpy(), score_t2i.cpu().numpy(), 0.0
