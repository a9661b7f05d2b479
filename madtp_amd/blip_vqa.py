"""Mirror of the reference's models/blip_vqa.py BLIP_VQA, encoder leg (BASELINE config 5: 480x480 images = 901 visual tokens,
the heaviest ragged compaction): ViT on the image (:59), the MED text encoder in multimodal mode cross-attending to the pruned
image tokens (:118-125).  Same constructor arguments and sub-module names (`visual_encoder`, `text_encoder`, `space_dict`:
checkpoint keys) as the reference; the answer decoder `text_decoder` (BertLMHeadModel, :53-55) with rank_answer / beam search
(:127-215) is out of scope (SURVEY.md 8 "out of scope": decoders), so forward(train=False) returns the encoder output the
decoder would consume unless a decoder is attached by the caller."""
import os

import torch
from torch import nn

from . import hip  # noqa: F401
from .bert import BertConfig
from .med import BertModel
from .runtime import require_gpu
from .vit import VisionTransformer

ENC_TOKEN_ID = 30523  # tokenizer.additional_special_tokens_ids[0] after init_tokenizer() (models/blip.py:219-225)


class BLIP_VQA(nn.Module):
    def __init__(self, med_config=None, image_size=480, vit='base', vit_grad_ckpt=False, vit_ckpt_layer=0, evaluate=True,
                 config=None):
        super().__init__()
        if vit != 'base':
            raise NotImplementedError("the gfx950 kernels are tuned for ViT-B (768 wide, 12 heads)")
        self.sd_num = 100 if config is None else config['sd_num']
        self.sd_dim = 768 if config is None else config['sd_dim']
        self.space_dict = nn.Parameter(torch.randn(self.sd_num, self.sd_dim))  # :40
        self.world_size = int(os.environ.get('WORLD_SIZE', 1))
        self.layers = 12
        self.visual_encoder = VisionTransformer(img_size=image_size, patch_size=16, embed_dim=768, depth=12, num_heads=12,
                                                use_grad_checkpointing=vit_grad_ckpt, ckpt_layer=vit_ckpt_layer,
                                                drop_path_rate=0.1, evaluate=evaluate, sd_dim=self.sd_dim)  # :46
        enc_cfg = BertConfig.from_json_file(med_config) if isinstance(med_config, str) else BertConfig.med_default()
        enc_cfg.encoder_width = 768
        enc_cfg.evaluate = evaluate
        self.text_encoder = BertModel(config=enc_cfg, add_pooling_layer=False, sd_dim=self.sd_dim)  # :51
        self.text_decoder = None  # BertLMHeadModel in the reference (:55): off the pruned encoder path
        self.tokenizer = None     # callers pass {'input_ids', 'attention_mask'} tensors (max_length 35, :63)

    def _tokens(self, question, device):
        if self.tokenizer is not None and not isinstance(question, dict) and not hasattr(question, "input_ids"):
            question = self.tokenizer(question, padding='longest', truncation=True, max_length=35, return_tensors="pt")
        ids = question["input_ids"] if isinstance(question, dict) else question.input_ids
        att = question["attention_mask"] if isinstance(question, dict) else question.attention_mask
        ids = ids.to(device).clone()
        ids[:, 0] = ENC_TOKEN_ID  # :64
        return ids, att.to(device)

    def encode_question(self, image, question, temperature=0):
        """:59-64 + :118-125 -> (question_states [B,L',768], image_embeds, (sd_img_ft, sd_txt_ft))."""
        require_gpu(image, "image")
        image_embeds, sd_img_ft = self.visual_encoder(image, space_dict=self.space_dict, temperature=temperature)  # :59
        image_atts = torch.ones(image_embeds.size()[:-1], dtype=torch.long, device=image.device)  # :60
        ids, att = self._tokens(question, image.device)
        out, sd_txt_ft = self.text_encoder(ids, attention_mask=att, encoder_hidden_states=image_embeds,
                                           encoder_attention_mask=image_atts, return_dict=True,
                                           space_dict=self.space_dict, temperature=temperature)  # :118-124
        return out.last_hidden_state, image_embeds, (sd_img_ft, sd_txt_ft)

    def forward(self, image, question, answer=None, temperature=0, train=True, n=None, weights=None, inference='rank',
                k_test=128):
        if train:
            raise NotImplementedError("BLIP_VQA training (answer decoder loss, :66-115) is out of scope: evaluation forward only")
        question_states, _, _ = self.encode_question(image, question, temperature)
        if self.text_decoder is None:
            return question_states  # the tensor rank_answer / generate (:127-180) would consume
        raise NotImplementedError("answer decoding (:127-215) is out of scope; attach your own decoder to `question_states`")


def blip_vqa(pretrained='', **kwargs):
    model = BLIP_VQA(**kwargs)
    if pretrained:  # blip_vqa.py:218-223 -> models/blip.py:254-278
        from .checkpoint import load_checkpoint
        model, msg = load_checkpoint(model, pretrained)
        print("missing keys:")
        print(msg.missing_keys)
    return model
