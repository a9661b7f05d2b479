"""Mirror of the reference's models/blip_vqa.py BLIP_VQA: the encoder leg (BASELINE config 5: 480x480 images = 901 visual
tokens, the heaviest ragged compaction) - ViT on the image (:59), the MED text encoder in multimodal mode cross-attending to
the pruned image tokens (:118-125) - and, with decoder=True, the answer decoder `text_decoder` (BertLMHeadModel, :53-55) run
teacher-forced by rank_answer (:156-203; SURVEY.md 8(f) rank 4, inference half).  Same constructor arguments and sub-module
names (checkpoint keys) as the reference.  Beam-search generation (inference="generate", :127-147) goes through BertLMHeadModel.generate (madtp_amd/generation.py); training (:66-115) is not implemented."""
import os

import torch
from torch import nn

from . import hip  # noqa: F401
from .bert import BertConfig, EncoderKVCache
from .med import BertLMHeadModel, BertModel
from .runtime import require_gpu
from .vit import VisionTransformer

ENC_TOKEN_ID = 30523  # tokenizer.additional_special_tokens_ids[0] after init_tokenizer() (models/blip.py:219-225)
PAD_TOKEN_ID = 0
BOS_TOKEN_ID = 30522  # tokenizer.bos_token_id ("[DEC]", models/blip.py:222)
SEP_TOKEN_ID = 102  # tokenizer.sep_token_id of bert-base-uncased


class BLIP_VQA(nn.Module):
    def __init__(self, med_config=None, image_size=480, vit='base', vit_grad_ckpt=False, vit_ckpt_layer=0, evaluate=True,
                 config=None, decoder=True):
        super().__init__()
        if vit != 'base':
            raise NotImplementedError("vit='large': the reference's large branch cannot run its own pruned forward (see "
                                      "madtp_amd/blip_nlvr.py create_vit)")
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
        # :53-55 (decoder=False, extension: the encoder-leg benchmark / fixtures skip its 137 M parameters)
        self.text_decoder = None
        if decoder:
            dec_cfg = BertConfig.from_json_file(med_config) if isinstance(med_config, str) else BertConfig.med_default()
            dec_cfg.evaluate = evaluate
            self.text_decoder = BertLMHeadModel(config=dec_cfg, sd_dim=self.sd_dim)
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
            return self._train_forward(image, question, answer, temperature, n, weights)
        question_states, _, _ = self.encode_question(image, question, temperature)
        if self.text_decoder is None:
            return question_states  # the tensor rank_answer / generate (:127-180) would consume
        if inference == 'generate':  # :127-147 (the reference decodes the ids to strings with its tokenizer, :143-146)
            num_beams = 3
            qs = question_states.repeat_interleave(num_beams, dim=0)  # :128
            qa = torch.ones(qs.size()[:-1], dtype=torch.long, device=qs.device)  # :129
            bos_ids = torch.full((image.size(0), 1), fill_value=BOS_TOKEN_ID, device=image.device)  # :132
            outputs = self.text_decoder.generate(input_ids=bos_ids, max_length=10, min_length=1, num_beams=num_beams,
                                                 eos_token_id=SEP_TOKEN_ID, pad_token_id=PAD_TOKEN_ID,
                                                 encoder_hidden_states=qs, encoder_attention_mask=qa)  # :134-140
            tok = getattr(self, "tokenizer", None)
            if tok is not None and hasattr(tok, "decode"):
                return [tok.decode(o, skip_special_tokens=True) for o in outputs]
            return outputs
        if inference != 'rank':
            raise ValueError("inference must be 'rank' or 'generate'")
        a_ids = answer["input_ids"] if isinstance(answer, dict) else answer.input_ids
        a_att = answer["attention_mask"] if isinstance(answer, dict) else answer.attention_mask
        q_att = question["attention_mask"] if isinstance(question, dict) else question.attention_mask
        return self.rank_answer(question_states, q_att.to(image.device), a_ids.to(image.device), a_att.to(image.device), k_test)  # :151-153

    def _train_forward(self, image, question, answer, temperature, n, weights):
        """blip_vqa.py:66-115: (loss_vqa, loss_fdt).  answer: {'input_ids', 'attention_mask'} of all answers (n[b] per question, in
        question order; position 0 is set to the BOS id as :72 does), weights: one per answer.  The answer decoder runs
        teacher-forced on the question states repeated n[b] times; loss_fdt is the cosine embedding loss of the l2-normalised
        dictionary features (:102-113).  Gradients need the fp32 precision mode (madtp_amd/backward.py); the reference's dropout /
        DropPath apply in model.train() mode (counter-based masks, runtime.set_dropout_seed)."""
        import torch.nn.functional as F
        if self.text_decoder is None:
            raise RuntimeError("BLIP_VQA(decoder=False) has no answer decoder to train")
        question_states, _, (sd_img_ft, sd_txt_ft) = self.encode_question(image, question, temperature)
        dev = image.device
        a_ids = (answer["input_ids"] if isinstance(answer, dict) else answer.input_ids).to(dev).clone()
        a_att = (answer["attention_mask"] if isinstance(answer, dict) else answer.attention_mask).to(dev)
        q_att = (question["attention_mask"] if isinstance(question, dict) else question.attention_mask).to(dev)
        a_ids[:, 0] = BOS_TOKEN_ID  # :72
        targets = a_ids.masked_fill(a_ids == PAD_TOKEN_ID, -100)  # :73
        rep = torch.repeat_interleave(torch.arange(len(n), device=dev), torch.as_tensor(list(n), device=dev))
        qs = question_states.index_select(0, rep)  # :84-90 (each question's states n[b] times)
        qa = q_att.index_select(0, rep)
        out = self.text_decoder(a_ids, attention_mask=a_att, encoder_hidden_states=qs, encoder_attention_mask=qa, labels=targets,
                                return_dict=True, reduction='none')  # :92-99
        loss_vqa = (torch.as_tensor(weights, device=dev, dtype=torch.float32) * out.loss).sum() / image.size(0)  # :101-102
        loss_fdt = loss_vqa
        if temperature != 0 and sd_img_ft is not None and sd_txt_ft is not None:
            si = sd_img_ft / (sd_img_ft.norm(dim=-1, keepdim=True) + 1e-10)
            st = sd_txt_ft / (sd_txt_ft.norm(dim=-1, keepdim=True) + 1e-10)
            si, st = si.reshape(-1, self.sd_dim), st.reshape(-1, self.sd_dim)
            loss_fdt = F.cosine_embedding_loss(si, st, torch.ones(si.shape[0], device=dev).long())
        return loss_vqa, loss_fdt

    def rank_answer(self, question_states, question_atts, answer_ids, answer_atts, k, detail=None):
        """blip_vqa.py:156-203.  Differences in HOW, not in what: the question states are projected to every decoder layer's
        cross-attention [k|v] ONCE (EncoderKVCache) and candidate (q, j) reads block q in place - the reference tiles the
        states k times (:186-187) and re-projects them in every layer; the first-token probabilities come from one kernel
        (madtp_token_prob); the per-candidate sequence losses from madtp_lm_loss.  detail (optional dict): intermediate
        tensors for the parity tests."""
        require_gpu(question_states, "question_states")
        dec = self.text_decoder
        num_ques = question_states.size(0)
        dev = question_states.device
        cache = EncoderKVCache.build(dec.bert, question_states)
        start_ids = answer_ids[0, 0].repeat(num_ques, 1)  # bos token :159
        start = dec(start_ids, encoder_hidden_states=None, encoder_attention_mask=None, return_dict=True, reduction='none',
                    encoder_kv_cache=cache.select(torch.arange(num_ques, device=dev)))  # :161-165
        logits = start.logits[:, 0, :]  # first token's logit :166
        answer_first_token = answer_ids[:, 1].contiguous()
        prob_first_token = hip.token_prob(logits, answer_first_token, logits.shape[-1])  # softmax + index_select :170-171
        topk_probs, topk_ids = prob_first_token.topk(k, dim=1)  # :172
        flat = topk_ids.reshape(-1)
        input_ids = answer_ids.index_select(0, flat)  # :175-181 ([num_ques * k, answer_len], question-major)
        input_atts = answer_atts.index_select(0, flat)
        targets_ids = input_ids.masked_fill(input_ids == PAD_TOKEN_ID, -100)  # :183
        block = torch.arange(num_ques, device=dev).repeat_interleave(k)  # tile(question_states, 0, k) :186 as an index
        out = dec(input_ids, attention_mask=input_atts, encoder_hidden_states=None, encoder_attention_mask=None,
                  labels=targets_ids, return_dict=True, reduction='none', encoder_kv_cache=cache.select(block))  # :189-195
        log_probs_sum = (-out.loss).view(num_ques, k)  # :197-198
        max_topk_ids = log_probs_sum.argmax(dim=1)
        max_ids = topk_ids[max_topk_ids >= 0, max_topk_ids]  # :200-201
        if detail is not None:
            detail.update(first_logits=logits, prob_first_token=prob_first_token, topk_ids=topk_ids, topk_probs=topk_probs,
                          log_probs_sum=log_probs_sum)
        return max_ids


def blip_vqa(pretrained='', **kwargs):
    model = BLIP_VQA(**kwargs)
    if pretrained:  # blip_vqa.py:218-223 -> models/blip.py:254-278
        from .checkpoint import load_checkpoint
        model, msg = load_checkpoint(model, pretrained)
        print("missing keys:")
        print(msg.missing_keys)
    return model
