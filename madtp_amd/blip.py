"""Mirror of the reference's models/blip.py BLIP_Decoder (:71-202), the captioning model of compress_caption_dtp.py: ViT on the
image with the alignment-guided pruning (:162), then beam-search generation with the MED text decoder (BertLMHeadModel)
cross-attending to the pruned image tokens (:164-196).  Same constructor arguments and sub-module names (checkpoint keys) as
the reference; the training forward (:111-158) and nucleus sampling (:175-186) are not implemented.  The reference's OWN
models/blip.py also constructs on the mirrors through madtp_amd.dropin (its `text_decoder.generate(...)` call lands in
madtp_amd.bert.BertLMHeadModel.generate)."""
import os

import torch
from torch import nn

from .bert import BertConfig
from .med import BertLMHeadModel
from .runtime import require_gpu
from .vit import VisionTransformer

BOS_TOKEN_ID = 30522  # tokenizer.bos_token_id ("[DEC]", models/blip.py:222)
SEP_TOKEN_ID = 102
PAD_TOKEN_ID = 0
# tokenizer('a picture of ').input_ids of bert-base-uncased = [CLS] a picture of [SEP]; generate() replaces [CLS] by [DEC] and
# drops [SEP] (:171-173).  Used when no tokenizer is attached (there is no vocabulary file offline).
PROMPT_IDS = {'a picture of ': (101, 1037, 3861, 1997, 102)}


class BLIP_Decoder(nn.Module):
    def __init__(self, med_config=None, image_size=384, vit='base', vit_grad_ckpt=False, vit_ckpt_layer=0,
                 prompt='a picture of ', evaluate=True, config=None):
        super().__init__()
        if vit != 'base':
            raise NotImplementedError("vit='large': the reference's large branch cannot run its own pruned forward (see "
                                      "madtp_amd/blip_nlvr.py create_vit)")
        self.sd_num = 100 if config is None else config['sd_num']  # :90-95
        self.sd_dim = 768 if config is None else config['sd_dim']
        self.space_dict = nn.Parameter(torch.randn(self.sd_num, self.sd_dim))  # :96
        self.world_size = int(os.environ.get('WORLD_SIZE', 1))
        self.layers = 12
        self.visual_encoder = VisionTransformer(img_size=image_size, patch_size=16, embed_dim=768, depth=12, num_heads=12,
                                                use_grad_checkpointing=vit_grad_ckpt, ckpt_layer=vit_ckpt_layer,
                                                drop_path_rate=0, evaluate=evaluate, sd_dim=self.sd_dim)  # :101
        cfg = BertConfig.from_json_file(med_config) if isinstance(med_config, str) else BertConfig.med_default()
        cfg.encoder_width = 768
        cfg.evaluate = evaluate
        self.text_decoder = BertLMHeadModel(config=cfg, sd_dim=self.sd_dim)  # :106
        self.tokenizer = None
        self.prompt = prompt
        self.prompt_ids = PROMPT_IDS.get(prompt)
        self.prompt_length = len(self.prompt_ids) - 1 if self.prompt_ids else None  # :109

    def forward(self, image, caption, temperature=0, train=False):
        """models/blip.py:111-158.  caption: {'input_ids', 'attention_mask'} tensors (or a list of strings with a tokenizer
        attached); train=True -> (loss_lm, loss_fdt) - the decoder is not given space_dict, so sd_txt_ft is None and loss_fdt IS
        loss_lm (:146-147) - train=False -> the decoder's output object.  Gradients need the fp32 precision mode."""
        require_gpu(image, "image")
        image_embeds, _ = self.visual_encoder(image, space_dict=self.space_dict, temperature=temperature)  # :112
        image_atts = torch.ones(image_embeds.size()[:-1], dtype=torch.long, device=image.device)  # :113
        if isinstance(caption, dict) or hasattr(caption, "input_ids"):
            ids = (caption["input_ids"] if isinstance(caption, dict) else caption.input_ids).to(image.device).clone()
            att = (caption["attention_mask"] if isinstance(caption, dict) else caption.attention_mask).to(image.device)
        elif self.tokenizer is not None:
            t = self.tokenizer(caption, padding='longest', truncation=True, max_length=40, return_tensors="pt")  # :115
            ids, att = t.input_ids.to(image.device).clone(), t.attention_mask.to(image.device)
        else:
            raise TypeError("pass {'input_ids','attention_mask'} tensors or set model.tokenizer (no vocabulary offline)")
        ids[:, 0] = BOS_TOKEN_ID  # :117
        if self.prompt_length is None:
            raise TypeError("prompt_length is unknown for this prompt: set model.prompt_length (= tokens of the prompt - 1, :109)")
        targets = ids.masked_fill(ids == PAD_TOKEN_ID, -100)  # :119
        targets[:, :self.prompt_length] = -100  # :120
        if not train:
            return self.text_decoder(ids, attention_mask=att, encoder_hidden_states=image_embeds,
                                     encoder_attention_mask=image_atts, labels=None, return_dict=True)  # :136-146
        out = self.text_decoder(ids, attention_mask=att, encoder_hidden_states=image_embeds, encoder_attention_mask=image_atts,
                                labels=targets, return_dict=True)  # :123-132 (reduction 'mean')
        return out.loss, out.loss  # :133, :147: no text-side dictionary features -> loss_fdt = loss_lm

    def _prompt(self, B, device):
        if self.tokenizer is not None:
            ids = self.tokenizer([self.prompt] * B, return_tensors="pt").input_ids.to(device)  # :170-171
        elif self.prompt_ids is not None:
            ids = torch.tensor([self.prompt_ids] * B, dtype=torch.int64, device=device)
        else:
            raise TypeError(f"no tokenizer attached and no known token ids for the prompt {self.prompt!r}: set model.tokenizer "
                            "or model.prompt_ids")
        ids = ids.clone()
        ids[:, 0] = BOS_TOKEN_ID  # :172
        return ids[:, :-1]  # :173

    def generate(self, image, sample=False, num_beams=3, max_length=30, min_length=10, top_p=0.9, temperature=0,
                 repetition_penalty=1.0):
        """models/blip.py:161-202 -> the decoded captions without the prompt when a tokenizer is attached (:198-202), else the
        generated token ids int64 [B, <= max_length] (prompt included)."""
        require_gpu(image, "image")
        image_embeds = self.visual_encoder(image, space_dict=self.space_dict, temperature=temperature)[0]  # :162
        if not sample:
            image_embeds = image_embeds.repeat_interleave(num_beams, dim=0)  # :164-165
        image_atts = torch.ones(image_embeds.size()[:-1], dtype=torch.long, device=image.device)  # :167
        input_ids = self._prompt(image.size(0), image.device)
        if sample:  # nucleus sampling :175-186 (repetition_penalty 1.1 is the reference's constant there)
            outputs = self.text_decoder.generate(input_ids=input_ids, max_length=max_length, min_length=min_length, do_sample=True,
                                                 top_p=top_p, num_return_sequences=1, eos_token_id=SEP_TOKEN_ID,
                                                 pad_token_id=PAD_TOKEN_ID, repetition_penalty=1.1,
                                                 encoder_hidden_states=image_embeds, encoder_attention_mask=image_atts,
                                                 generator=getattr(self, "sample_generator", None))
            if self.tokenizer is not None and hasattr(self.tokenizer, "decode"):
                return [self.tokenizer.decode(o, skip_special_tokens=True)[len(self.prompt):] for o in outputs]
            return outputs
        outputs = self.text_decoder.generate(input_ids=input_ids, max_length=max_length, min_length=min_length,
                                             num_beams=num_beams, eos_token_id=SEP_TOKEN_ID, pad_token_id=PAD_TOKEN_ID,
                                             repetition_penalty=repetition_penalty, encoder_hidden_states=image_embeds,
                                             encoder_attention_mask=image_atts)  # :189-196
        if self.tokenizer is not None and hasattr(self.tokenizer, "decode"):
            return [self.tokenizer.decode(o, skip_special_tokens=True)[len(self.prompt):] for o in outputs]  # :198-202
        return outputs


def blip_decoder(pretrained='', **kwargs):
    model = BLIP_Decoder(**kwargs)
    if pretrained:  # models/blip.py:204-210 -> load_checkpoint :254-278
        from .checkpoint import load_checkpoint
        model, msg = load_checkpoint(model, pretrained)
        print("missing keys:")
        print(msg.missing_keys)
    return model
