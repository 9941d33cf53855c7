"""Tokenizer wrapper (host side, CPU text work -- out of the kernel scope) with the call
surface of the reference's chattts_plus/models/tokenizer.py:19-137: wraps the pickled BertTokenizerFast,
left-pads a batch to [B,T,num_vq] ids + attention/text masks, and exposes the special-token ids the
pipeline needs.  The speaker / prompt codecs live in chatttsplus_amd/codec.py."""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch

from . import codec


class Tokenizer:
    def __init__(self, model_path=None, tokenizer=None, **kwargs):
        if tokenizer is None:
            # the reference stores a pickled BertTokenizerFast object (tokenizer.py:27-30): needs weights_only=False
            tokenizer = torch.load(model_path, map_location="cpu", mmap=True, weights_only=False)
        self._tokenizer = tokenizer
        self.len = len(tokenizer)
        self.spk_emb_ids = tokenizer.convert_tokens_to_ids("[spk_emb]")
        self.break_0_ids = tokenizer.convert_tokens_to_ids("[break_0]")
        self.eos_token = tokenizer.convert_tokens_to_ids("[Ebreak]")
        self.decode = self._tokenizer.batch_decode

    @torch.inference_mode()
    def encode(self, text: List[str], num_vq: int, prompt_str: Optional[str] = None, device="cpu") -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """tokenizer.py:49-137: ids replicated over num_vq, LEFT padding, optional audio-prompt codes appended
        (text_mask = 0 there)."""
        prompt = codec.decode_prompt(prompt_str) if prompt_str is not None else None
        prompt_size = 0
        if prompt is not None:
            assert prompt.size(0) == num_vq, "prompt dim 0 must equal to num_vq"
            prompt_size = prompt.size(1)
        ids_l, att_l = [], []
        for t in text:
            x = self._tokenizer(t, return_tensors="pt", add_special_tokens=False, padding=True)   # == encode_plus (tokenizer.py:70-72)
            ids_l.append(x["input_ids"].squeeze(0))
            att_l.append(x["attention_mask"].squeeze(0))
        T = max(i.size(0) for i in ids_l) + prompt_size
        B = len(ids_l)
        input_ids = torch.zeros(B, T, dtype=ids_l[0].dtype)
        attention_mask = torch.zeros(B, T, dtype=att_l[0].dtype)
        for i in range(B):
            n = ids_l[i].size(0)
            input_ids[i, T - prompt_size - n: T - prompt_size] = ids_l[i]
            attention_mask[i, T - prompt_size - n: T - prompt_size] = att_l[i]
            if prompt_size:
                attention_mask[i, T - prompt_size:] = 1
        text_mask = attention_mask.bool()
        new_input_ids = input_ids.unsqueeze(-1).expand(-1, -1, num_vq).clone()
        if prompt_size:
            text_mask[:, T - prompt_size:] = False
            new_input_ids[:, T - prompt_size:] = prompt.t().unsqueeze(0).expand(B, -1, -1)
        return new_input_ids.to(device), attention_mask.to(device), text_mask.to(device)

    def apply_spk_emb(self, emb, spk_emb, input_ids, device=None):
        return codec.apply_spk_emb(emb, spk_emb, input_ids, self.spk_emb_ids)

    _decode_spk_emb = staticmethod(codec.decode_spk_emb)
    _encode_spk_emb = staticmethod(codec.encode_spk_emb)
    _decode_prompt = staticmethod(codec.decode_prompt)
    _encode_prompt = staticmethod(codec.encode_prompt)
