"""State-dict key/shape specifications of the reference task models (checkpoint-compatible names).

The key names are an API (SURVEY.md section 8(b) "state keys"): reference checkpoints are loaded BY NAME
(models/blip_nlvr.py:146-158, models/vit.py:380-395), so the mirrors in this package and the synthetic
weight generator both use exactly these names.
"""
from collections import OrderedDict

D = 768
FFN = 3072
SD_NUM = 100
VOCAB = 30524  # configs/med_config.json
MAX_POS = 512


def _linear(sd, name, out_f, in_f, bias=True):
    sd[name + ".weight"] = (out_f, in_f)
    if bias:
        sd[name + ".bias"] = (out_f,)


def _ln(sd, name, dim=D):
    sd[name + ".weight"] = (dim,)
    sd[name + ".bias"] = (dim,)


def vit_shapes(prefix="visual_encoder.", img_size=224, patch=16, depth=12, dim=D):
    """models/vit.py VisionTransformer (BLIP create_vit('base'), models/blip.py:228-238)."""
    sd = OrderedDict()
    n = (img_size // patch) ** 2
    sd[prefix + "cls_token"] = (1, 1, dim)
    sd[prefix + "pos_embed"] = (1, n + 1, dim)
    sd[prefix + "patch_embed.proj.weight"] = (dim, 3, patch, patch)
    sd[prefix + "patch_embed.proj.bias"] = (dim,)
    for i in range(depth):
        p = f"{prefix}blocks.{i}."
        _ln(sd, p + "norm1", dim)
        _linear(sd, p + "attn.qkv", 3 * dim, dim)
        _linear(sd, p + "attn.proj", dim, dim)
        _ln(sd, p + "norm2", dim)
        _linear(sd, p + "mlp.fc1", 4 * dim, dim)
        _linear(sd, p + "mlp.fc2", dim, 4 * dim)
    _ln(sd, prefix + "norm", dim)
    return sd


def _bert_self(sd, p, kv_in=D):
    _linear(sd, p + "query", D, D)
    _linear(sd, p + "key", D, kv_in)
    _linear(sd, p + "value", D, kv_in)


def bert_shapes(prefix="text_encoder.", variant="med", layers=12, cross=True):
    """models/med.py BertModel(add_pooling_layer=False) ('med') or models/nlvr_encoder.py ('nlvr')."""
    sd = OrderedDict()
    e = prefix + "embeddings."
    sd[e + "position_ids"] = ("int64", 1, MAX_POS)
    sd[e + "word_embeddings.weight"] = (VOCAB, D)
    sd[e + "position_embeddings.weight"] = (MAX_POS, D)
    _ln(sd, e + "LayerNorm")
    for i in range(layers):
        p = f"{prefix}encoder.layer.{i}."
        _bert_self(sd, p + "attention.self.")
        _linear(sd, p + "attention.output.dense", D, D)
        _ln(sd, p + "attention.output.LayerNorm")
        if cross:
            c = p + "crossattention."
            if variant == "nlvr":
                _bert_self(sd, c + "self0.")
                _bert_self(sd, c + "self1.")
                _ln(sd, c + "output.LayerNorm")
                _linear(sd, c + "output.dense0", D, D)
                _linear(sd, c + "output.dense1", D, D)
                if i >= 6:
                    _linear(sd, c + "output.merge_layer", D, 2 * D)
            else:
                _bert_self(sd, c + "self.")
                _linear(sd, c + "output.dense", D, D)
                _ln(sd, c + "output.LayerNorm")
        _linear(sd, p + "intermediate.dense", FFN, D)
        _linear(sd, p + "output.dense", D, FFN)
        _ln(sd, p + "output.LayerNorm")
    return sd


def blip_nlvr_shapes(img_size=224):
    """models/blip_nlvr.py BLIP_NLVR.__init__ :20-61."""
    sd = OrderedDict()
    sd["space_dict"] = (SD_NUM, D)
    sd.update(vit_shapes("visual_encoder.", img_size))
    sd.update(bert_shapes("text_encoder.", "nlvr"))
    _linear(sd, "cls_head.0", D, D)
    _linear(sd, "cls_head.2", 2, D)
    return sd


def blip_retrieval_shapes(img_size=384, embed_dim=256):
    """models/blip_retrieval.py BLIP_Retrieval.__init__ :20-66 - the modules the evaluation path uses (the momentum
    encoders, queues and `temp` of :67-93 only feed the training loss)."""
    sd = OrderedDict()
    sd["space_dict"] = (SD_NUM, D)
    sd.update(vit_shapes("visual_encoder.", img_size))
    sd.update(bert_shapes("text_encoder.", "med"))
    _linear(sd, "vision_proj", embed_dim, D)
    _linear(sd, "text_proj", embed_dim, D)
    _linear(sd, "itm_head", 2, D)
    return sd


def lm_head_shapes(prefix="text_decoder."):
    """models/med.py BertLMHeadModel :933-947: `bert` (BertModel, no pooler) + `cls.predictions` (BertOnlyMLMHead :616-657)."""
    sd = bert_shapes(prefix + "bert.", "med")
    c = prefix + "cls.predictions."
    sd[c + "bias"] = (VOCAB,)
    _linear(sd, c + "transform.dense", D, D)
    _ln(sd, c + "transform.LayerNorm")
    _linear(sd, c + "decoder", VOCAB, D)
    return sd


# state-dict entries that alias ONE parameter in the reference (transformers ties the LM head to the input embeddings,
# PreTrainedModel.tie_weights; BertLMPredictionHead links decoder.bias to its own bias, med.py:629-637): the synthetic
# generator gives both names the same values, as any saved checkpoint has them
TIED_KEYS = {"cls.predictions.decoder.weight": "bert.embeddings.word_embeddings.weight",
             "cls.predictions.decoder.bias": "cls.predictions.bias"}


def blip_decoder_shapes(img_size=384):
    """models/blip.py BLIP_Decoder.__init__ :71-109: space_dict, visual_encoder, text_decoder (BertLMHeadModel)."""
    sd = OrderedDict()
    sd["space_dict"] = (SD_NUM, D)
    sd.update(vit_shapes("visual_encoder.", img_size))
    sd.update(lm_head_shapes("text_decoder."))
    return sd


def blip_vqa_shapes(img_size=480, decoder=False):
    """models/blip_vqa.py BLIP_VQA.__init__ :15-55: space_dict, visual_encoder, text_encoder and - decoder=True - the answer
    decoder `text_decoder` (:53-55) that rank_answer (:156-203) runs teacher-forced."""
    sd = OrderedDict()
    sd["space_dict"] = (SD_NUM, D)
    sd.update(vit_shapes("visual_encoder.", img_size))
    sd.update(bert_shapes("text_encoder.", "med"))
    if decoder:
        sd.update(lm_head_shapes("text_decoder."))
    return sd


def clip_vit_shapes(prefix="", img_size=224, patch=16, width=768, layers=12, out_dim=512, sd_dim=768):
    """clip/model.py VisionTransformer (:275-313) with ResidualAttentionBlock (:174-261); ViT-B/16 geometry."""
    sd = OrderedDict()
    n = (img_size // patch) ** 2
    sd[prefix + "class_embedding"] = (width,)
    sd[prefix + "positional_embedding"] = (n + 1, width)
    sd[prefix + "proj"] = (width, out_dim)
    sd[prefix + "conv1.weight"] = (width, 3, patch, patch)
    _ln(sd, prefix + "ln_pre", width)
    for i in range(layers):
        p = f"{prefix}transformer.resblocks.{i}."
        sd[p + "attn.in_proj_weight"] = (3 * width, width)
        sd[p + "attn.in_proj_bias"] = (3 * width,)
        _linear(sd, p + "attn.out_proj", width, width)
        _ln(sd, p + "ln_1", width)
        _linear(sd, p + "mlp.c_fc", 4 * width, width)
        _linear(sd, p + "mlp.c_proj", width, 4 * width)
        _ln(sd, p + "ln_2", width)
        _linear(sd, p + "query_model.q_map.0", sd_dim, width)
    _ln(sd, prefix + "ln_post", width)
    return sd


def clip_text_shapes(prefix="", width=512, layers=12, ctx=77, vocab=49408, embed_dim=512, sd_dim=768):
    """clip/model.py CLIP text side (:379-392): token / positional embeddings, the causal Transformer (:264-272) of
    ResidualAttentionBlocks with their query_model.q_map, ln_final, text_projection."""
    sd = OrderedDict()
    sd[prefix + "token_embedding.weight"] = (vocab, width)
    sd[prefix + "positional_embedding"] = (ctx, width)
    sd[prefix + "text_projection"] = (width, embed_dim)
    for i in range(layers):
        p = f"{prefix}transformer.resblocks.{i}."
        sd[p + "attn.in_proj_weight"] = (3 * width, width)
        sd[p + "attn.in_proj_bias"] = (3 * width,)
        _linear(sd, p + "attn.out_proj", width, width)
        _ln(sd, p + "ln_1", width)
        _linear(sd, p + "mlp.c_fc", 4 * width, width)
        _linear(sd, p + "mlp.c_proj", width, 4 * width)
        _ln(sd, p + "ln_2", width)
        _linear(sd, p + "query_model.q_map.0", sd_dim, width)
    _ln(sd, prefix + "ln_final", width)
    return sd


def clip_shapes(img_size=224, patch=16, vision_width=768, vision_layers=12, embed_dim=512, text_width=512, text_layers=12,
                ctx=77, vocab=49408, sd_num=100, sd_dim=768):
    """clip/model.py CLIP (ViT-B/16 geometry by default): visual.*, the text side, logit_scale, space_dict (the momentum
    copies and queues of :395-436 are training state and are not part of the evaluation state dict)."""
    sd = OrderedDict()
    sd["space_dict"] = (sd_num, sd_dim)
    sd["logit_scale"] = ()
    sd.update(clip_vit_shapes("visual.", img_size, patch, vision_width, vision_layers, embed_dim, sd_dim))
    sd.update(clip_text_shapes("", text_width, text_layers, ctx, vocab, embed_dim, sd_dim))
    return sd


def synth_weights(shapes, seed=0, device=None):
    """{key: tensor} from a shape spec using the deterministic generator (device: generate there - same bits, synth.py)."""
    import torch
    from . import synth
    out = {}
    for k, shp in shapes.items():
        if shp and shp[0] == "int64":
            out[k] = torch.arange(shp[-1], device=device).expand(shp[1:]).clone()
        else:
            out[k] = synth.synth_tensor(k, shp, seed, device=device)
    tie_keys(out)
    return out


def tie_keys(sd):
    """give the aliased state-dict names of TIED_KEYS the values of their source entry (in place)."""
    for k in list(sd.keys()):
        for dst, src in TIED_KEYS.items():
            if k.endswith(dst) and (k[:-len(dst)] + src) in sd:
                sd[k] = sd[k[:-len(dst)] + src].clone()
    return sd
