"""The four GPU benchmark workloads of BASELINE.json (configs 2-5) behind one interface, shared by bench.py, the calibration
tool and the per-config tests.  Each workload is one pass of the pruned forward path over a batch of synthetic inputs that are
resident in HBM; nothing here touches the oracle.

    w = workloads.get("vqa"); model = w.build("cuda"); inp = w.inputs(B, seed, "cuda"); out = w.step(model, inp, T)
    lens = w.lens(model)          # per-encoder token counts after each layer, from the modules' last_prune records
    w.flops(lens), w.flops(None)  # analytic FLOPs per SAMPLE (2 x MAC) at those counts / unpruned

name        reference path (file:line)                                              sample                     images/sample
nlvr        models/blip_nlvr.py:63-100 BLIP_NLVR.forward(train=False)               2 images 224^2 + 20 tokens       2
retrieval   compress_retrieval_dtp.py:101-125,166-178 pieces on matched pairs       image 224^2/384^2 + 35 tokens    1
            (ViT, MED text mode, MED multimodal ITM + itm_head)
clip        clip/model.py:482-503 encode_image + encode_text, similarity matmul     image 224^2 + 77 tokens          1
            (compress_retrieval_clip_dtp.py:92,100,121-122)
vqa         models/blip_vqa.py:59-64,118-125 encoder leg                            image 480^2 + 20 tokens          1
"""
import torch
import torch.nn.functional as F

from . import harness, hip, specs, synth

D, K_SD = harness.D, harness.K_SD


def _traces(layers):
    return [harness._cpu_info(l.last_prune) for l in layers]


def med_layer_flops(l_in, l_out, n_img=0, att_ft=True):
    """one MED BertLayer (med.py:393-462) for one sample: self-attention on l_in tokens, query model, optional single
    cross-attention to n_img image tokens and the FFN on the l_out surviving tokens."""
    f = 8 * l_in * D * D + 4 * l_in * l_in * D + 2 * (l_in - 1) * D * K_SD + (2 * K_SD * (l_in - 1) * D if att_ft else 0)
    if n_img:
        f += 2 * l_out * D * D + 4 * n_img * D * D + 4 * l_out * n_img * D + 2 * l_out * D * D
    return f + 16 * l_out * D * D


def med_flops(lens, l0, n_img=0):
    f, l = 0, l0
    for l_out in lens:
        f += med_layer_flops(l, l_out, n_img)
        l = l_out
    return f


def vit_tower_flops(lens, n0):
    f, n = 2 * (n0 - 1) * D * D, n0  # patch embedding
    for n_out in lens:
        f += harness.vit_flops(n, n_out)
        n = n_out
    return f


def clip_tower_flops(lens, n0, width, sd_dim=768, patch_in=0):
    """clip/model.py ResidualAttentionBlock stack: as the ViT layer plus the q_map Linear(width -> sd_dim) of its query model."""
    f, n = 2 * (n0 - 1) * patch_in * width, n0
    for n_out in lens:
        f += 6 * n * width * width + 4 * n * n * width + 2 * n * width * width + 16 * n_out * width * width
        f += 2 * (n - 1) * width * sd_dim + 2 * (n - 1) * sd_dim * K_SD + 2 * K_SD * (n - 1) * sd_dim
        n = n_out
    return f


class Workload:
    name = ""
    images_per_sample = 1
    default_batch = 64
    p = 0.5
    # forwards in flight per GPU in bench.py (madtp_amd/pipeline.py: half of the workers on high-priority streams); measured per
    # configuration (profiles/r03_inflight.txt (i), (j)): NLVR 22.6-23.1 k with two, 23.8-24.5 k with three, 25.1-25.5 k with four,
    # 24.1-24.3 k with five
    default_inflight = 4
    # GEMM dispatch hint while several forwards are in flight (include/madtp_hip.h madtp_gemm_set_sq_cost); measured with four
    # in flight: NLVR 25.2 -> 25.7 k images/s, VQA 4.77 -> 4.87 k, retrieval 29.4 -> 28.4 k (keeps the default)
    inflight_sq_cost = 0.9
    # ... and the 128x128 tile for the small problems (madtp_gemm_set_small_tile): NLVR 25.5 -> 26.2 k, retrieval 28.5 -> 30.4 k,
    # VQA 4.81 -> 4.85 k with four in flight
    inflight_small_tile = 0

    def build(self, device="cuda"): raise NotImplementedError
    def inputs(self, B, seed=0, device="cuda"): raise NotImplementedError
    def step(self, model, inp, T): raise NotImplementedError
    def lens(self, model): raise NotImplementedError
    def flops(self, lens): raise NotImplementedError
    def describe(self, B): raise NotImplementedError


class Nlvr(Workload):
    name, images_per_sample, default_batch, p = "nlvr", 2, 64, 0.5
    size, L = 224, 20

    def build(self, device="cuda"):
        return harness.build_nlvr(self.size, 0, device)

    def inputs(self, B, seed=0, device="cuda"):
        return harness.nlvr_inputs(B, self.size, self.L, seed, device)

    def step(self, model, inp, T):
        images, text, targets = inp
        return model(images, text, targets, temperature=T, train=False)

    def lens(self, model):
        n0 = (self.size // 16) ** 2 + 1
        return {"vit": harness.token_lengths(_traces(model.visual_encoder.blocks), n0),
                "text": harness.token_lengths(_traces(model.text_encoder.encoder.layer), self.L)}

    def flops(self, lens):
        n0 = (self.size // 16) ** 2 + 1
        if lens is None:
            lens = {"vit": [n0] * 12, "text": [self.L] * 12}
        return harness.nlvr_forward_flops(lens["vit"], lens["text"], n0, self.L)

    def describe(self, B):
        return (f"BLIP-base NLVR2 forward (BLIP_NLVR.forward(train=False)), p={self.p}, {B} samples = {2 * B} images "
                f"{self.size}x{self.size} + {self.L} text tokens per GPU, random-init weights")


class Retrieval(Workload):
    name, default_batch, p = "retrieval", 128, 0.75
    L = 35
    inflight_sq_cost = None  # (29.4 -> 28.4 k images/s with the hint: stays on the default dispatch)

    def __init__(self, size=224):
        self.size = size

    def build(self, device="cuda"):
        return harness.build_retrieval(self.size, 0, device)

    def inputs(self, B, seed=0, device="cuda"):
        batches, ids, att = harness.retrieval_inputs(B, B, B, self.size, self.L, seed, device)
        ids_mm = ids.clone()
        ids_mm[:, 0] = 30523  # compress_retrieval_dtp.py:114: [ENC] token for the multimodal pass
        return batches[0], ids, att, ids_mm

    def step(self, model, inp, T):
        images, ids, att, ids_mm = inp
        sd = model.space_dict
        img, _ = model.visual_encoder(images, space_dict=sd, temperature=T)                       # :118
        img_emb = model.project_image(img[:, 0, :])                                              # :121-122
        txt, _ = model.text_encoder(ids, attention_mask=att, mode='text', space_dict=sd, temperature=T)   # :101-103
        txt_emb = model.project_text(txt.last_hidden_state[:, 0, :])                             # :104
        # the multimodal pass below overwrites the layers' last_prune: keep the text-mode records (references only, read by lens())
        model.__dict__["_madtp_text_prune"] = [l.last_prune for l in model.text_encoder.encoder.layer]
        atts = torch.ones(img.shape[:-1], dtype=torch.long, device=img.device)
        mm, _ = model.text_encoder(ids_mm, attention_mask=att, encoder_hidden_states=img, encoder_attention_mask=atts,
                                   return_dict=True, space_dict=sd, temperature=T)               # :166-171 on matched pairs
        return model.itm_score(mm.last_hidden_state[:, 0, :]), (img_emb * txt_emb).sum(-1)       # :172, diagonal of :155

    def lens(self, model):
        n0 = (self.size // 16) ** 2 + 1
        out = {"vit": harness.token_lengths(_traces(model.visual_encoder.blocks), n0),
               "mm": harness.token_lengths(_traces(model.text_encoder.encoder.layer), self.L)}
        tp = model.__dict__.get("_madtp_text_prune")
        if tp is not None:
            out["text"] = harness.token_lengths([harness._cpu_info(i) for i in tp], self.L)
        return out

    def flops(self, lens):
        n0 = (self.size // 16) ** 2 + 1
        if lens is None:
            lens = {"vit": [n0] * 12, "mm": [self.L] * 12, "text": [self.L] * 12}
        txt = lens.get("text", lens["mm"])  # (step() keeps the text-mode pass's own records: Retrieval.lens()["text"])
        return vit_tower_flops(lens["vit"], n0) + med_flops(txt, self.L) + med_flops(lens["mm"], self.L, lens["vit"][-1])

    def describe(self, B):
        return (f"BLIP-base retrieval forward pieces of evaluate() (ViT, MED text mode, MED multimodal ITM on matched pairs), "
                f"p={self.p}, {B} image-caption pairs {self.size}x{self.size} + {self.L} tokens per GPU, random-init weights")


class Vqa(Workload):
    name, default_batch, p = "vqa", 32, 0.5
    size, L = 480, 20

    def build(self, device="cuda"):
        from .blip_vqa import BLIP_VQA
        model = BLIP_VQA(image_size=self.size, evaluate=True, decoder=False).eval().to(device)  # encoder leg (config 5)
        msg = model.load_state_dict(specs.synth_weights(specs.blip_vqa_shapes(self.size), 0, device=device), strict=False)
        assert not msg.unexpected_keys and all("query_model" in k or "position_ids" in k for k in msg.missing_keys), msg
        return model

    def inputs(self, B, seed=0, device="cuda"):
        images = synth.synth_images(B, self.size, seed, device=device)
        ids = synth.synth_token_ids(B, self.L, seed).to(device)
        return images, {"input_ids": ids, "attention_mask": torch.ones_like(ids)}

    def step(self, model, inp, T):
        return model(inp[0], inp[1], temperature=T, train=False)

    def lens(self, model):
        n0 = (self.size // 16) ** 2 + 1
        out = {"vit": harness.token_lengths(_traces(model.visual_encoder.blocks), n0),
               "mm": harness.token_lengths(_traces(model.text_encoder.encoder.layer), self.L)}
        tp = model.__dict__.get("_madtp_text_prune")
        if tp is not None:
            out["text"] = harness.token_lengths([harness._cpu_info(i) for i in tp], self.L)
        return out

    def flops(self, lens):
        n0 = (self.size // 16) ** 2 + 1
        if lens is None:
            lens = {"vit": [n0] * 12, "mm": [self.L] * 12}
        return vit_tower_flops(lens["vit"], n0) + med_flops(lens["mm"], self.L, lens["vit"][-1])

    def describe(self, B):
        return (f"BLIP-base VQA encoder leg (BLIP_VQA.forward(train=False) up to question_output), p={self.p}, {B} images "
                f"{self.size}x{self.size} (901 visual tokens) + {self.L} question tokens per GPU, random-init weights")


class Clip(Workload):
    name, default_batch, p = "clip", 128, 0.5
    size, ctx = 224, 77
    default_inflight = 1  # both towers fill the chip: 17.1-18.0 k serial; two in flight gave 15.9 k to 22.8 k from run to run, three 15 k

    def build(self, device="cuda"):
        from .clip_model import build_model
        return build_model(specs.synth_weights(specs.clip_shapes(self.size), 0, device=device), evaluate=True).eval().to(device)

    def inputs(self, B, seed=0, device="cuda"):
        return synth.synth_images(B, self.size, seed, device=device), synth.synth_clip_tokens(B, self.ctx, seed).to(device)

    def step(self, model, inp, T):
        images, text = inp
        fi, _ = model.encode_image(images, model.space_dict, T)   # compress_retrieval_clip_dtp.py:100
        ft, _ = model.encode_text(text, model.space_dict, T)      # :92
        fi, ft = F.normalize(fi, dim=-1), F.normalize(ft, dim=-1)
        pad = (-ft.shape[0]) % 128
        w = torch.cat([ft, ft.new_zeros(pad, ft.shape[1])], 0) if pad else ft
        return hip.gemm(fi.contiguous(), w.contiguous(), n=ft.shape[0])  # :121-122 sims = image_feats @ text_feats.T

    def lens(self, model):
        return {"vit": harness.token_lengths(_traces(model.visual.transformer.resblocks), (self.size // 16) ** 2 + 1),
                "text": harness.token_lengths(_traces(model.transformer.resblocks), self.ctx)}

    def flops(self, lens):
        n0 = (self.size // 16) ** 2 + 1
        if lens is None:
            lens = {"vit": [n0] * 12, "text": [self.ctx] * 12}
        return clip_tower_flops(lens["vit"], n0, 768, patch_in=768) + clip_tower_flops(lens["text"], self.ctx, 512)

    def describe(self, B):
        return (f"CLIP ViT-B/16 retrieval forward (encode_image + encode_text + similarity), p={self.p}, {B} image-text pairs "
                f"{self.size}x{self.size} + {self.ctx} tokens per GPU, random-init weights")


def get(name, **kw):
    return {"nlvr": Nlvr, "retrieval": Retrieval, "vqa": Vqa, "clip": Clip}[name](**kw)


NAMES = ("nlvr", "retrieval", "clip", "vqa")
