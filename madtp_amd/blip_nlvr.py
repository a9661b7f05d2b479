"""Mirror of the reference's models/blip_nlvr.py BLIP_NLVR (:19-100) - the caller of the hot path whose
forward(train=False) is the end-to-end forward BASELINE.json's metric times."""
import os

import torch
from torch import nn

from . import hip
from .bert import BertConfig
from .nlvr_encoder import BertModel
from .runtime import PreparedCache, compute_dtype, lin_of, require_gpu, to_compute
from .vit import VisionTransformer

ENC_TOKEN_ID = 30523  # tokenizer.enc_token_id ('[ENC]', models/blip.py:221-224)


def create_vit(vit, image_size, use_grad_checkpointing=False, ckpt_layer=0, drop_path_rate=0, evaluate=False,
               sd_dim=768, map_func=False):
    """models/blip.py:228-252."""
    if vit != 'base':
        # the reference's own 'large' branch (models/blip.py:238-246: width 1024) cannot run the pruned forward: Query_model uses the
        # tokens themselves as queries (models/utils.py:161-170, map_func=False at every call site), so ONE space_dict [100, sd_dim] has
        # to match the 1024-wide image tokens AND the 768-wide text tokens - sd_dim 768 fails in the ViT, 1024 in the text encoder
        # ("mat1 and mat2 shapes cannot be multiplied (8x1024 and 768x100)", utils.py:170; run here)
        raise NotImplementedError("vit='large': the reference's large branch cannot run its own pruned forward (image tokens 1024 wide, text "
                                  "tokens 768 wide, one space_dict; models/utils.py:161-170) - there is nothing to mirror")
    vision_width = 768
    visual_encoder = VisionTransformer(img_size=image_size, patch_size=16, embed_dim=vision_width, depth=12,
                                       num_heads=12, use_grad_checkpointing=use_grad_checkpointing,
                                       ckpt_layer=ckpt_layer, drop_path_rate=0 or drop_path_rate, evaluate=evaluate,
                                       sd_dim=sd_dim, map_func=map_func)
    return visual_encoder, vision_width


class BLIP_NLVR(nn.Module):
    def __init__(self, med_config='configs/med_config.json', image_size=480, vit='base', vit_grad_ckpt=False,
                 vit_ckpt_layer=0, evaluate=False, config=None):
        super().__init__()
        self.layers = 12 if vit == 'base' else 24
        if config is None:
            self.sd_num, self.sd_dim, self.batch_size = 100, 768, 16
        else:
            self.sd_num, self.sd_dim, self.batch_size = config['sd_num'], config['sd_dim'], config['batch_size_train']
        self.space_dict = nn.Parameter(torch.randn(self.sd_num, self.sd_dim))
        self.world_size = int(os.environ.get('WORLD_SIZE', 1))
        self.visual_encoder, vision_width = create_vit(vit, image_size, vit_grad_ckpt, vit_ckpt_layer,
                                                       drop_path_rate=0.1, evaluate=evaluate, sd_dim=self.sd_dim)
        self.tokenizer = None  # set to a BertTokenizer-like callable to pass raw strings, as the reference does
        if isinstance(med_config, str):
            med_config = BertConfig.from_json_file(med_config) if os.path.exists(med_config) else BertConfig.med_default()
        med_config.encoder_width = vision_width
        med_config.evaluate = evaluate
        self.text_encoder = BertModel(config=med_config, add_pooling_layer=False, sd_dim=self.sd_dim)
        self.cls_head = nn.Sequential(nn.Linear(self.text_encoder.config.hidden_size, self.text_encoder.config.hidden_size),
                                      nn.ReLU(),
                                      nn.Linear(self.text_encoder.config.hidden_size, 2))
        self._cache = PreparedCache()
        self.compute_sd_ft = True  # sd_*_ft only feed the training loss (:86-96); set False to skip them in eval

    def _tokens(self, text, device):
        if isinstance(text, dict) or hasattr(text, "input_ids"):
            ids = text["input_ids"] if isinstance(text, dict) else text.input_ids
            att = text["attention_mask"] if isinstance(text, dict) else text.attention_mask
        elif self.tokenizer is not None:
            t = self.tokenizer(text, padding='longest', return_tensors="pt")
            ids, att = t.input_ids, t.attention_mask
        else:
            raise TypeError("pass {'input_ids','attention_mask'} tensors or set model.tokenizer (no vocabulary offline)")
        ids = ids.to(device).clone()
        ids[:, 0] = ENC_TOKEN_ID  # :69
        return ids, att.to(device)

    def _losses(self, prediction, targets, temperature, sd_img_ft, sd_txt_ft):
        """blip_nlvr.py:84-98: (loss_ori, loss_fdt) = (cross-entropy of the two-way prediction, CosineEmbeddingLoss between the
        l2-normalised dictionary features of the image pair (averaged) and of the text).  The two loss heads are a few torch ops
        on [B,2] and [100 B, sd_dim] tensors - everything upstream of them is the HIP path and its autograd.Functions."""
        import torch.nn.functional as F
        loss_ori = F.cross_entropy(prediction, targets)
        loss_fdt = loss_ori
        if temperature != 0 and sd_img_ft is not None and sd_txt_ft is not None:
            sd_img0_ft, sd_img1_ft = torch.split(sd_img_ft, targets.size(0))
            sd_img = (sd_img0_ft + sd_img1_ft) / 2
            sd_img = sd_img / (sd_img.norm(dim=-1, keepdim=True) + 1e-10)
            sd_txt = sd_txt_ft / (sd_txt_ft.norm(dim=-1, keepdim=True) + 1e-10)
            sd_img, sd_txt = sd_img.reshape(-1, self.sd_dim), sd_txt.reshape(-1, self.sd_dim)
            labels = torch.ones(sd_img.shape[0], device=sd_txt.device).long()
            loss_fdt = F.cosine_embedding_loss(sd_img, sd_txt, labels)
        return loss_ori, loss_fdt

    def forward(self, image, text, targets, temperature=0, train=True):
        """blip_nlvr.py:63-100.  train=True returns (loss_ori, loss_fdt) as the reference does.  In model.train() mode the BERT layers
        drop hidden states and attention probabilities (p from the config, 0.1) and the ViT blocks apply DropPath, with counter-based
        masks (runtime.set_dropout_seed, madtp_amd/backward.py); model.eval() is the deterministic forward the reference-recorded
        gradient fixtures check.  Gradients need the fp32 (or, opted in, the f16x3) precision mode."""
        require_gpu(image, "image")
        self.visual_encoder.img_query_model.compute_att_ft = self.compute_sd_ft
        self.text_encoder.encoder.txt_query_model.compute_att_ft = self.compute_sd_ft
        # sd_img_ft only feeds the training loss (:86-96): its (fast-mode) sum over the layers runs on the auxiliary stream,
        # under the text encoder, and is joined before this forward returns
        pending = []
        ids, att = self._tokens(text, image.device)  # (:68-69; before the vision encoder: host work that does not depend on it)
        self.text_encoder.encoder.prepare_encoder_call(ids.shape[0])
        image_embeds, sd_img_ft = self.visual_encoder(image, space_dict=self.space_dict, temperature=temperature,
                                                      _pending=pending)  # :64
        # :65 image_atts = ones: every image token is valid, so the cross-attention masks are all zero - passed as None (same
        # values, without the six mask-building launches per forward)
        image0_embeds, image1_embeds = torch.split(image_embeds, targets.size(0))  # :67
        lp = getattr(image_embeds, "_madtp_lp", None)
        if lp is not None and lp[1] == image_embeds._version:  # the compute-dtype copy of the final LayerNorm, split alike
            lp0, lp1 = torch.split(lp[0], targets.size(0))
            image0_embeds._madtp_lp, image1_embeds._madtp_lp = (lp0, image0_embeds._version), (lp1, image1_embeds._version)
        output, sd_txt_ft = self.text_encoder(ids, attention_mask=att,
                                              encoder_hidden_states=[image0_embeds, image1_embeds],
                                              encoder_attention_mask=[None, None],
                                              return_dict=True, space_dict=self.space_dict, temperature=temperature)
        hidden_state = output.last_hidden_state[:, 0, :]  # :80
        if torch.is_grad_enabled() and hidden_state.requires_grad:
            # training use (fp32 mode, madtp_amd/backward.py): the head as two autograd Linears on the exact-f32 GEMM
            from .backward import LinearFunction
            h = LinearFunction.apply(hidden_state.contiguous(), self.cls_head[0].weight, self.cls_head[0].bias, hip.ACT_RELU)
            logits = LinearFunction.apply(h, self.cls_head[2].weight, self.cls_head[2].bias, hip.ACT_NONE)
            for p in pending:
                p.sync()
            self.last_sd_ft = (sd_img_ft, sd_txt_ft)
            return self._losses(logits, targets, temperature, sd_img_ft, sd_txt_ft) if train else logits
        l0 = lin_of(self._cache, "c0", [self.cls_head[0]])
        l2 = lin_of(self._cache, "c2", [self.cls_head[2]])
        lp = getattr(output.last_hidden_state, "_madtp_lp", None)
        if (lp is not None and lp[1] == output.last_hidden_state._version and lp[0].dtype == compute_dtype()
                and compute_dtype() == torch.bfloat16):
            h = lp[0][:, 0, :]  # strided [B, dim] view of the compute-dtype copy: the GEMM reads the CLS rows in place
        else:
            h = to_compute(hidden_state.contiguous())
        h = hip.gemm(h, l0.w, l0.b, act=hip.ACT_RELU, n=l0.n)
        logits = hip.gemm(h, l2.w, l2.b, out_dtype=torch.float32, n=l2.n)  # :81
        for p in pending:
            p.sync()
        self.last_sd_ft = (sd_img_ft, sd_txt_ft)
        return self._losses(logits, targets, temperature, sd_img_ft, sd_txt_ft) if train else logits


def blip_nlvr(pretrained='', **kwargs):
    model = BLIP_NLVR(**kwargs)
    if pretrained:  # blip_nlvr.py:122-128, load_checkpoint :130-159
        from .checkpoint import load_checkpoint_nlvr
        model, msg = load_checkpoint_nlvr(model, pretrained)
        print("missing keys:")
        print(msg.missing_keys)
    return model
