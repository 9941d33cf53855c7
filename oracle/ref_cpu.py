"""CPU restatement (torch fp32) of the reference ChatTTS hot path.

TEST INFRASTRUCTURE -- NOT PART OF THE PRODUCT.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import this module, and only as the *checker* / reported CPU baseline.  The
product path (``chatttsplus_amd``) never imports ``oracle`` and raises if the
HIP library is missing.

Pinning status
  * GPT decoder + sampler + DVAE decoder: pinned against the imported reference
    (``oracle/make_golden.py`` -> ``tests/golden/*.npz``; ``tests/test_oracle_*``).
  * Vocos (third-party ``vocos`` 0.1.0, absent from /root/reference and from this
    image): restated from the upstream algorithm (VocosBackbone + ISTFTHead).  The
    ISTFT head is pinned on the port of vocos' head inside ``transformers``
    (``Xcodec2ISTFTHead``), ``torch.istft`` and a direct fp64 definition; the backbone's
    ConvNeXt block is the reference's own (pinned through the DVAE goldens); the
    backbone's wiring stays **parity unpinned**.
  * GFSQ / mel extractor of the zero-shot branch (``vector_quantize_pytorch``,
    ``torchaudio``: absent): the FSQ layer and the residual loop are pinned on
    ``Xcodec2FiniteScalarQuantization`` (a port of that library's layer), the mel filterbank
    on ``transformers.audio_utils.mel_filter_bank`` (tests/test_third_party_pins.py); the
    grouped wrapper stays a restatement.
  * Device noise stream: ``oracle/device_noise.py`` (Philox4x32-10, Random123 known answers).

Every function cites the reference file:line it follows (paths relative to
/root/reference/chattts_plus/).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn.functional as F

RMS_EPS = 1e-6          # LlamaConfig default rms_norm_eps (SURVEY F1 probe)
ROPE_BASE = 10000.0     # llama.py:252
HEAD_DIM = 64


def _t(x) -> torch.Tensor:
    if isinstance(x, torch.Tensor):
        return x
    return torch.from_numpy(np.ascontiguousarray(x))


# ----------------------------------------------------------------------------------------------
# sampler chain (models/processors.py + transformers TopP/TopK warpers + gpt.py:469-481)
# ----------------------------------------------------------------------------------------------

def penalty_table(penalty: float, window: int = 16) -> torch.Tensor:
    """alpha[n] = penalty**n as torch computes it (processors.py:28: torch.pow(float, int64 tensor))."""
    return torch.pow(float(penalty), torch.arange(0, window + 1, dtype=torch.int64))


def repetition_penalty(logits_token: torch.Tensor, scores: torch.Tensor, penalty: float,
                       max_input_ids: int, past_window: int) -> torch.Tensor:
    """models/processors.py:18-34 (incl. the row-count quirk SURVEY F8, :23-27)."""
    ids = logits_token
    if ids.size(1) > past_window:
        ids = ids.narrow(1, -past_window, past_window)
    freq = F.one_hot(ids, scores.size(1)).sum(1)
    if freq.size(0) > max_input_ids:
        freq.narrow(0, max_input_ids, freq.size(0) - max_input_ids).zero_()
    alpha = torch.pow(float(penalty), freq)
    scores = scores.contiguous()
    return torch.where(scores < 0, scores.multiply(alpha), scores.divide(alpha))


def top_p_warp(scores: torch.Tensor, top_p: float, min_keep: int, stable: bool = False) -> torch.Tensor:
    """transformers TopPLogitsWarper.__call__ (built at processors.py:45): ascending sort,
    remove while cumsum(softmax) <= 1 - top_p, always keep the last ``min_keep``.
    ``stable``: torch.sort is not stable for large groups of exactly equal logits (the reference's result is then
    implementation-defined); stable=True pins the tie order the HIP sampler documents (stable ascending)."""
    sorted_logits, sorted_indices = torch.sort(scores, descending=False, stable=stable)
    cumulative_probs = sorted_logits.softmax(dim=-1).cumsum(dim=-1)
    remove = cumulative_probs <= (1 - top_p)
    remove[..., -min_keep:] = 0
    remove = remove.scatter(1, sorted_indices, remove)
    return scores.masked_fill(remove, -float("inf"))


def top_k_warp(scores: torch.Tensor, top_k: int, min_keep: int) -> torch.Tensor:
    """transformers TopKLogitsWarper.__call__ (built at processors.py:47); ties with the k-th value are kept."""
    k = min(max(top_k, min_keep), scores.size(-1))
    kth = torch.topk(scores, k)[0][..., -1, None]
    return scores.masked_fill(scores < kth, -float("inf"))


@dataclass
class SamplerParams:
    """Scalars read off the HF objects that gen_logits() builds (processors.py:37-57)."""
    temperature: List[float] = field(default_factory=lambda: [0.3] * 4)
    top_p: Optional[float] = 0.7
    top_k: Optional[int] = 20
    min_keep: int = 3
    repetition_penalty: Optional[float] = 1.05
    past_window: int = 16
    max_input_ids: int = 625
    eos_token: int = 625
    min_new_token: int = 0


def sample_step(logits: torch.Tensor, logits_token: torch.Tensor, q: torch.Tensor, step: int,
                sp: SamplerParams, temperature_col: torch.Tensor, stable_sort: bool = False) -> torch.Tensor:
    """One sampling step for rows [B*num_vq, V]  (gpt.py:469-481).

    ``q`` is Exp(1) noise of the same shape; ``argmax(p / q)`` is what
    ``torch.multinomial(p, 1)`` computes on CPU from the same generator.
    Returns idx_next [rows] int64.
    """
    logits = logits / temperature_col                                   # gpt.py:469
    if sp.repetition_penalty is not None and sp.repetition_penalty != 1:
        logits = repetition_penalty(logits_token, logits, sp.repetition_penalty, sp.max_input_ids, sp.past_window)
    if sp.top_p is not None:
        logits = top_p_warp(logits, sp.top_p, sp.min_keep, stable=stable_sort)             # gpt.py:474-475 (top-p first)
    if sp.top_k is not None:
        logits = top_k_warp(logits, sp.top_k, sp.min_keep)
    if step < sp.min_new_token:
        logits = logits.clone()
        logits[:, sp.eos_token] = -torch.inf                           # gpt.py:477-478
    scores = F.softmax(logits, dim=-1)                                 # gpt.py:480
    return torch.argmax(scores / q, dim=-1)                            # gpt.py:481 (multinomial == exp. race)


class TorchExpNoise:
    """Draws q exactly as torch.multinomial does internally: empty_like(p).exponential_(1) on the
    default CPU generator -- so the oracle reproduces the *unpatched* reference under torch.manual_seed."""

    def next(self, rows: int, vocab: int) -> torch.Tensor:
        return torch.empty(rows, vocab, dtype=torch.float32).exponential_(1)


class SeededNoise:
    """Counter-based noise stream from chatttsplus_amd.synth.exp_noise (regenerable on the GPU box).
    All noise sources are sequential streams: one draw per generate-loop iteration, continuing across
    an ensure_non_empty regenerate exactly like torch's generator does in the reference."""

    def __init__(self, seed: int):
        self.seed = seed
        self.counter = 0

    def next(self, rows: int, vocab: int) -> torch.Tensor:
        from chatttsplus_amd.synth import exp_noise
        q = torch.from_numpy(exp_noise(self.seed, self.counter, rows, vocab))
        self.counter += 1
        return q


class ArrayNoise:
    """Replays pre-drawn noise q[draw, rows, vocab] sequentially."""

    def __init__(self, q):
        self.q = _t(q)
        self.counter = 0

    def next(self, rows: int, vocab: int) -> torch.Tensor:
        q = self.q[self.counter]
        self.counter += 1
        return q


# ----------------------------------------------------------------------------------------------
# GPT (models/gpt.py + models/llama.py live path: RMSNorm, RoPE, SDPA attention, SwiGLU)
# ----------------------------------------------------------------------------------------------

@dataclass
class GenerationOutputs:            # gpt.py:280-284
    ids: List[torch.Tensor]
    attentions: list
    hiddens: List[torch.Tensor]
    # extras for tests (not in the reference):
    steps: int = 0
    logits_trace: Optional[List[torch.Tensor]] = None


class OracleGPT:
    """Weights by reference state-dict key.  Pre-allocated KV cache instead of the
    reference's per-step torch.cat (llama.py:633 / DynamicCache) -- same values, different storage."""

    def __init__(self, sd: Dict[str, np.ndarray], num_heads: int, num_vq: int = 4):
        self.sd = {k: _t(v).float() for k, v in sd.items()}
        self.H = self.sd["gpt.norm.weight"].numel()
        self.nh = num_heads
        assert self.H // self.nh == HEAD_DIM
        self.L = 1 + max([int(k.split(".")[2]) for k in self.sd if k.startswith("gpt.layers.")] + [-1])
        self.num_vq = num_vq
        self.V = self.sd["emb_code.0.weight"].shape[0]
        # fold weight-norm once: W = g * v / ||v||_row  (gpt.py:57-77; torch weight_norm dim=0);
        # the reference re-materialises it every step under P.cached() (gpt.py:424).
        self.head_code = []
        for i in range(num_vq):
            g = self.sd[f"head_code.{i}.parametrizations.weight.original0"]
            v = self.sd[f"head_code.{i}.parametrizations.weight.original1"]
            self.head_code.append(torch._weight_norm(v, g, 0))
        self.head_text = None
        if "head_text.parametrizations.weight.original0" in self.sd:
            self.head_text = torch._weight_norm(self.sd["head_text.parametrizations.weight.original1"],
                                                self.sd["head_text.parametrizations.weight.original0"], 0)      # gpt.py:57-64
        self.inv_freq = 1.0 / (ROPE_BASE ** (torch.arange(0, HEAD_DIM, 2, dtype=torch.int64).float() / HEAD_DIM))  # llama.py:100

    # -- embedding ----------------------------------------------------------------------------
    def embed(self, input_ids: torch.Tensor, text_mask: torch.Tensor) -> torch.Tensor:
        """GPT.forward (gpt.py:125-149)."""
        input_ids = _t(input_ids); text_mask = _t(text_mask).bool()
        emb_text = F.embedding(input_ids[text_mask][:, 0], self.sd["emb_text.weight"])
        inv = ~text_mask
        mids = input_ids[inv]
        emb_code = torch.stack([F.embedding(mids[:, i], self.sd[f"emb_code.{i}.weight"]) for i in range(self.num_vq)], 2).sum(2)
        emb = torch.zeros(input_ids.shape[:-1] + (self.H,), dtype=torch.float32)
        emb[text_mask] = emb_text
        emb[inv] = emb_code
        return emb

    @staticmethod
    def apply_spk_emb(emb: torch.Tensor, spk: torch.Tensor, input_ids: torch.Tensor, spk_emb_id: int) -> torch.Tensor:
        """Tokenizer.apply_spk_emb (tokenizer.py:150-178): rows whose id == [spk_emb] <- L2-normalised vector."""
        n = F.normalize(_t(spk).float(), p=2.0, dim=0, eps=1e-12)
        cond = _t(input_ids)[..., 0:1].eq(spk_emb_id).expand(emb.shape)
        return torch.where(cond, n.expand(emb.shape), emb)

    def embed_code(self, ids: torch.Tensor) -> torch.Tensor:
        """decode re-embed (gpt.py:403-407): sum over the num_vq code embeddings. ids [B,1,4] -> [B,1,H]."""
        return torch.stack([F.embedding(ids[:, :, i], self.sd[f"emb_code.{i}.weight"]) for i in range(self.num_vq)], 3).sum(3)

    # -- transformer --------------------------------------------------------------------------
    def _rms(self, x, w):
        """LlamaRMSNorm.forward (llama.py:82-87)."""
        var = x.pow(2).mean(-1, keepdim=True)
        return w * (x * torch.rsqrt(var + RMS_EPS))

    def _rope(self, pos):
        """LlamaRotaryEmbedding.forward (llama.py:106-119): fp32 freqs = inv_freq x pos; emb = cat(freqs, freqs)."""
        freqs = pos[:, :, None].float() * self.inv_freq[None, None, :]
        emb = torch.cat((freqs, freqs), dim=-1)
        return emb.cos(), emb.sin()

    @staticmethod
    def _rotate_half(x):
        x1 = x[..., : x.shape[-1] // 2]
        x2 = x[..., x.shape[-1] // 2:]
        return torch.cat((-x2, x1), dim=-1)                           # llama.py:151-155

    def alloc_cache(self, B: int, Lmax: int):
        self.kc = torch.zeros(self.L, B, self.nh, Lmax, HEAD_DIM)
        self.vc = torch.zeros(self.L, B, self.nh, Lmax, HEAD_DIM)
        self.kv_len = 0

    def forward(self, x: torch.Tensor, attn_mask: torch.Tensor, position_ids: torch.Tensor) -> torch.Tensor:
        """LlamaModel.forward for q_len new tokens appended at cache slot self.kv_len
        (llama.py:905-1019; layer :719-749; attention :590-668; mask :1021-1099).

        x [B,q,H]; attn_mask [B,Ltot] (1 = attend) covering cached + new tokens; position_ids [B,q].
        """
        B, q, _ = x.shape
        past = self.kv_len
        Ltot = past + q
        # additive mask: causal AND key-padding, filled with finfo.min (llama.py:1073-1087)
        minv = torch.finfo(torch.float32).min
        cache_position = torch.arange(past, Ltot)
        causal = torch.full((q, Ltot), minv)
        if q != 1:
            causal = torch.triu(causal, diagonal=1)
        causal = causal * (torch.arange(Ltot) > cache_position.reshape(-1, 1))
        mask4 = causal[None, None].expand(B, 1, -1, -1).clone()
        pad = (mask4 + attn_mask[:, None, None, :Ltot].float()) == 0
        mask4 = mask4.masked_fill(pad, minv)
        cos, sin = self._rope(position_ids)
        cos = cos[:, None]; sin = sin[:, None]
        for l in range(self.L):
            p = f"gpt.layers.{l}."
            res = x
            h = self._rms(x, self.sd[p + "input_layernorm.weight"])
            qs = F.linear(h, self.sd[p + "self_attn.q_proj.weight"]).view(B, q, self.nh, HEAD_DIM).transpose(1, 2)
            ks = F.linear(h, self.sd[p + "self_attn.k_proj.weight"]).view(B, q, self.nh, HEAD_DIM).transpose(1, 2)
            vs = F.linear(h, self.sd[p + "self_attn.v_proj.weight"]).view(B, q, self.nh, HEAD_DIM).transpose(1, 2)
            qs = qs * cos + self._rotate_half(qs) * sin                 # llama.py:158-182
            ks = ks * cos + self._rotate_half(ks) * sin
            self.kc[l, :, :, past:Ltot] = ks
            self.vc[l, :, :, past:Ltot] = vs
            K = self.kc[l, :, :, :Ltot]; V = self.vc[l, :, :, :Ltot]
            att = torch.matmul(qs, K.transpose(-1, -2)) / math.sqrt(HEAD_DIM) + mask4   # SDPA, scale=None -> 1/sqrt(d)
            att = torch.softmax(att, dim=-1)
            o = torch.matmul(att, V).transpose(1, 2).reshape(B, q, self.H)
            x = res + F.linear(o, self.sd[p + "self_attn.o_proj.weight"])
            res = x
            h = self._rms(x, self.sd[p + "post_attention_layernorm.weight"])
            g = F.linear(h, self.sd[p + "mlp.gate_proj.weight"])
            u = F.linear(h, self.sd[p + "mlp.up_proj.weight"])
            x = res + F.linear(F.silu(g) * u, self.sd[p + "mlp.down_proj.weight"])   # llama.py:214
        self.kv_len = Ltot
        return self._rms(x, self.sd["gpt.norm.weight"])                 # llama.py:1002

    def code_logits(self, hidden_last: torch.Tensor) -> torch.Tensor:
        """4 folded heads on the last position -> [B*num_vq, V] (gpt.py:429-447)."""
        lg = torch.stack([F.linear(hidden_last, w) for w in self.head_code], 1)   # [B,4,V]
        return lg.reshape(-1, self.V)

    # -- generate (refine-text pass: infer_text=True) -------------------------------------------
    @torch.no_grad()
    def generate_text(self, emb, inputs_ids, temperature: float, eos_token: int, top_p=0.7, top_k=20, attention_mask=None,
                      max_new_token=384, min_new_token=0, noise=None, ensure_non_empty=True) -> GenerationOutputs:
        """GPT.generate with infer_text=True (gpt.py:400-401,425-426,458-467,489-494; called by pipeline:237-277):
        one 21178-way head, a single temperature, next input = emb_text[id], ids returned as [n] (first column).
        repetition_penalty must be 1 (the reference's processor path mis-broadcasts the [B,n,1] history for this mode)."""
        emb = _t(emb).float(); inputs_ids = _t(inputs_ids)
        B, T = inputs_ids.shape[0], inputs_ids.shape[1]
        noise = noise or TorchExpNoise()
        Vt = self.head_text.shape[0]
        end_idx = torch.zeros(B, dtype=torch.long); finish = torch.zeros(B, dtype=torch.bool)
        mask_cache = torch.ones(B, T + max_new_token, dtype=torch.bool)
        if attention_mask is not None:
            mask_cache[:, :T] = _t(attention_mask).bool()
        ids_buf = torch.zeros(B, T + max_new_token, dtype=torch.long)
        self.alloc_cache(B, T + max_new_token)
        progress = T
        for i in range(max_new_token):
            m = mask_cache[:, :progress]
            pos = m.long().cumsum(-1) - 1
            pos.masked_fill_(m.eq(0), 1)
            if i == 0:
                x, p = emb, pos
            else:
                x, p = F.embedding(ids_buf[:, progress - 1:progress], self.sd["emb_text.weight"]), pos[:, -1:]     # gpt.py:400-401
            last = self.forward(x, m, p)[:, -1]
            logits = F.linear(last, self.head_text) / torch.tensor(float(temperature), dtype=torch.float32)          # gpt.py:426,469
            if top_p is not None:
                logits = top_p_warp(logits, top_p, 3)
            if top_k is not None:
                logits = top_k_warp(logits, top_k, 3)
            if i < min_new_token:
                logits = logits.clone(); logits[:, eos_token] = -torch.inf
            idx = torch.argmax(F.softmax(logits, dim=-1) / noise.next(B, Vt), dim=-1)
            finish |= idx.eq(eos_token)                                                                             # gpt.py:490-491
            ids_buf[:, progress] = idx
            if i == 0 and finish.any():
                if ensure_non_empty:
                    return self.generate_text(emb, inputs_ids, temperature, eos_token, top_p, top_k, attention_mask, max_new_token,
                                              min_new_token, noise, ensure_non_empty)
                return GenerationOutputs([], [], [], 1, None)
            progress += 1
            end_idx += (~finish).long()
            if finish.all():
                break
        return GenerationOutputs([ids_buf[b, T:T + int(end_idx[b])] for b in range(B)], [], [], 0, None)

    # -- generate -----------------------------------------------------------------------------
    @torch.no_grad()
    def generate(self, emb, inputs_ids, sp: SamplerParams, attention_mask=None, max_new_token=2048,
                 noise=None, return_hidden=True, ensure_non_empty=True, trace_logits=False,
                 forced_ids=None) -> GenerationOutputs:
        """GPT.generate, infer_text=False branch (gpt.py:313-569).

        ``forced_ids`` [N,B,4] teacher-forces the sampled ids (test aid); everything else follows the reference.
        """
        emb = _t(emb).float(); inputs_ids = _t(inputs_ids)
        B, T = inputs_ids.shape[0], inputs_ids.shape[1]
        noise = noise or TorchExpNoise()
        start_idx = T
        end_idx = torch.zeros(B, dtype=torch.long)
        finish = torch.zeros(B, dtype=torch.bool)
        temperature = torch.tensor(sp.temperature, dtype=torch.float32).unsqueeze(0).expand(B, -1).contiguous().view(-1, 1)
        mask_cache = torch.ones(B, T + max_new_token, dtype=torch.bool)                # gpt.py:353-364
        if attention_mask is not None:
            mask_cache[:, :T] = _t(attention_mask).bool()
        ids_buf = torch.zeros(B, T + max_new_token, self.num_vq, dtype=torch.long)
        ids_buf[:, :T] = inputs_ids
        self.alloc_cache(B, T + max_new_token)
        hiddens, trace = [], []
        progress = T
        steps = 0
        for i in range(max_new_token):
            m = mask_cache[:, :progress]
            pos = m.long().cumsum(-1) - 1                                              # gpt.py:238-245
            pos.masked_fill_(m.eq(0), 1)
            if i == 0:
                x, p = emb, pos
            else:
                x, p = self.embed_code(ids_buf[:, progress - 1:progress]), pos[:, -1:]
            hidden = self.forward(x, m, p)
            last = hidden[:, -1]
            if return_hidden:
                hiddens.append(last)                                                   # gpt.py:422-423
            logits = self.code_logits(last)
            if trace_logits:
                trace.append(logits.clone())
            logits_token = ids_buf[:, start_idx:progress].permute(0, 2, 1).reshape(B * self.num_vq, -1)
            q = noise.next(B * self.num_vq, self.V)
            idx_next = sample_step(logits, logits_token, q, i, sp, temperature)
            if forced_ids is not None:
                idx_next = _t(forced_ids)[i].reshape(-1)
            idx_next = idx_next.view(-1, self.num_vq)
            finish |= idx_next.eq(sp.eos_token).any(1)                                 # gpt.py:486-487
            ids_buf[:, progress] = idx_next
            steps += 1
            if i == 0 and finish.any() and ensure_non_empty:                           # gpt.py:496-525
                return self.generate(emb, inputs_ids, sp, attention_mask, max_new_token, noise, return_hidden,
                                     ensure_non_empty, trace_logits, forced_ids)
            if i == 0 and finish.any():
                return GenerationOutputs([], [], [], steps, trace)                      # gpt.py:525 bare return
            progress += 1
            end_idx += (~finish).long()                                                # gpt.py:530-531
            if finish.all():
                break
        ids = [ids_buf[b, start_idx:start_idx + int(end_idx[b])] for b in range(B)]    # gpt.py:295-297
        hid = []
        if hiddens:
            hs = torch.stack(hiddens, 1)
            hid = [hs[b, :int(end_idx[b])] for b in range(B)]                          # gpt.py:301-305
        return GenerationOutputs(ids, [], hid, steps, trace if trace_logits else None)


# ----------------------------------------------------------------------------------------------
# DVAE decoder (models/dvae.py)
# ----------------------------------------------------------------------------------------------

def _convnext(sd, p, x, dilation, kernel=7):
    """ConvNeXtBlock.forward (dvae.py:48-63; Vocos' block is the dilation=1 case)."""
    C = x.shape[1]
    y = F.conv1d(x, sd[p + "dwconv.weight"], sd[p + "dwconv.bias"], padding=dilation * (kernel // 2),
                 dilation=dilation, groups=C)
    y = y.transpose(1, 2)
    y = F.layer_norm(y, (C,), sd[p + "norm.weight"], sd[p + "norm.bias"], eps=1e-6)
    y = F.linear(y, sd[p + "pwconv1.weight"], sd[p + "pwconv1.bias"])
    y = F.gelu(y)
    y = F.linear(y, sd[p + "pwconv2.weight"], sd[p + "pwconv2.bias"])
    y = y * sd[p + "gamma"]
    return y.transpose(1, 2) + x


@torch.no_grad()
def dvae_decode(sd: Dict[str, np.ndarray], hidden: torch.Tensor) -> torch.Tensor:
    """DVAE.forward(mode="decode") with vq_layer=None (dvae.py:272-291).
    hidden [n, 768] (one utterance's GPT hiddens) -> mel [100, 2n].  The pipeline passes
    hiddens.permute(1,0)[None] = [1,768,n] (pipeline:298-300)."""
    sd = {k: _t(v).float() for k, v in sd.items()}
    inp = _t(hidden).float().permute(1, 0)[None]                                       # [1,768,n]
    x = inp.view(inp.size(0), 2, inp.size(1) // 2, inp.size(2)).permute(0, 2, 3, 1).flatten(2)   # [1,384,2n]
    y = F.conv1d(x, sd["decoder.conv_in.0.weight"], sd["decoder.conv_in.0.bias"], padding=1)
    y = F.gelu(y)
    y = F.conv1d(y, sd["decoder.conv_in.2.weight"], sd["decoder.conv_in.2.bias"], padding=1)
    n_layer = 1 + max(int(k.split(".")[2]) for k in sd if k.startswith("decoder.decoder_block."))
    for i in range(n_layer):
        y = _convnext(sd, f"decoder.decoder_block.{i}.", y, dilation=2)
    y = F.conv1d(y, sd["decoder.conv_out.weight"])
    y = F.conv1d(y, sd["out_conv.weight"], padding=1)
    return (y * sd["coef"])[0]                                                         # [100, 2n]


# ----------------------------------------------------------------------------------------------
# DVAE encode branch = zero-shot speaker prompt (models/dvae.py:171-199,263-270; pipeline:279-284,486-499)
#   mel extractor:  third-party torchaudio.transforms.MelSpectrogram -- ABSENT here: PARITY UNPINNED, restated from its
#                   documented algorithm (Spectrogram(power=1, center, reflect, periodic hann) -> MelScale(htk, norm=None))
#   conv stack:     the reference's own modules -- pinned by tests/golden/dvae_encode_real.npz
#   GFSQ:           third-party vector_quantize_pytorch==1.17.8 GroupedResidualFSQ -- ABSENT here: PARITY UNPINNED, restated
#                   from the published FSQ / ResidualFSQ algorithm (see gfsq_indices)
# ----------------------------------------------------------------------------------------------

def mel_filterbank(n_freqs: int = 513, n_mels: int = 100, sample_rate: int = 24000) -> torch.Tensor:
    """torchaudio.functional.melscale_fbanks(f_min=0, f_max=sr/2, norm=None, mel_scale="htk") -> [n_freqs, n_mels]."""
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m_min, m_max = 0.0, 2595.0 * math.log10(1.0 + (sample_rate / 2.0) / 700.0)
    m_pts = torch.linspace(m_min, m_max, n_mels + 2)
    f_pts = 700.0 * (10 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    down = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return torch.clamp(torch.min(down, up), min=0.0)


@torch.no_grad()
def mel_features(wav: torch.Tensor, n_fft: int = 1024, hop: int = 256, n_mels: int = 100, sample_rate: int = 24000) -> torch.Tensor:
    """MelSpectrogramFeatures.forward (dvae.py:171-199): log(clip(mel_spec(audio), 1e-5)); wav [n] -> [n_mels, 1 + n // hop]."""
    wav = _t(wav).float()
    window = torch.hann_window(n_fft, periodic=True)
    spec = torch.stft(wav, n_fft, hop_length=hop, win_length=n_fft, window=window, center=True, pad_mode="reflect",
                      normalized=False, onesided=True, return_complex=True).abs()                        # power = 1
    mel = torch.matmul(spec.transpose(-1, -2), mel_filterbank(n_fft // 2 + 1, n_mels, sample_rate)).transpose(-1, -2)
    return torch.log(torch.clip(mel, min=1e-5))


def fsq_bound(z: torch.Tensor, levels: torch.Tensor, eps: float = 1e-3) -> torch.Tensor:
    """FSQ.bound: (z + shift).tanh() * half_l - offset."""
    half_l = (levels - 1) * (1 + eps) / 2
    offset = torch.where(levels % 2 == 0, 0.5, 0.0)
    shift = (offset / half_l).atanh()
    return (z + shift).tanh() * half_l - offset


def gfsq_indices(x: torch.Tensor, sd, levels=(5, 5, 5, 5), G: int = 2, R: int = 2, pre_bound: bool = True) -> torch.Tensor:
    """GFSQ.forward (dvae.py:94-126) on x [T, G*dim_g] -> indices [G*R, T] (ind.permute(1,2,0,3).view(...).transpose).
    Per group: ResidualFSQ = project_in (Linear dim_g -> len(levels)), then R FSQ layers on residual / scale_r with
    scale_r = (levels - 1) ** -r; FSQ: codes = round(bound(z)) / (levels // 2); index = sum((codes * hw + hw) * basis),
    basis = cumprod([1] + levels[:-1]).  `pre_bound`: the 1.1x releases bound the projected input once before the residual
    loop (`residual = first(self.layers).bound(x)`); later releases replaced that by an optional soft clamp and start from
    `residual = x`.  1.17.8 cannot be inspected offline: True is our best knowledge of that release, both variants are
    implemented (oracle and HIP) and frozen in tests/golden/dvae_encode_real.npz (`ids_pre_bound` / `ids`)."""
    return gfsq_quantize(x, sd, levels, G, R, pre_bound)[0]


def gfsq_quantize(x: torch.Tensor, sd, levels=(5, 5, 5, 5), G: int = 2, R: int = 2, pre_bound: bool = True):
    """gfsq_indices plus the quantised latent the indices stand for: (indices [G*R, T], latent [G, T, len(levels)]) with
    latent_g = sum_r codes_r * scale_r in the projected (FSQ) space -- what ResidualFSQ returns before project_out and what
    get_output_from_indices rebuilds from the indices alone (GFSQ._embed, dvae.py:85-96)."""
    lv = torch.tensor(levels, dtype=torch.float32)
    hw = torch.tensor([l // 2 for l in levels], dtype=torch.float32)
    basis = torch.cumprod(torch.tensor([1] + list(levels[:-1]), dtype=torch.float32), 0)
    per = x.shape[-1] // G
    out, lat = [], []
    for g in range(G):
        w, b = _t(sd[f"vq_layer.quantizer.rvqs.{g}.project_in.weight"]).float(), _t(sd[f"vq_layer.quantizer.rvqs.{g}.project_in.bias"]).float()
        z = F.linear(x[..., g * per:(g + 1) * per], w, b)
        residual = fsq_bound(z, lv) if pre_bound else z
        acc = torch.zeros_like(z)
        for r in range(R):
            scale = (lv - 1) ** (-r)
            codes = torch.round(fsq_bound(residual / scale, lv)) / hw
            idx = ((codes * hw + hw) * basis).sum(-1).to(torch.int32)
            residual = residual - codes * scale
            acc = acc + codes * scale
            out.append(idx)
        lat.append(acc)
    return torch.stack(out, 0), torch.stack(lat, 0)                              # [G*R, T] (row = g * R + r), [G, T, D]


def gfsq_latent_from_indices(ids: torch.Tensor, levels=(5, 5, 5, 5), G: int = 2, R: int = 2) -> torch.Tensor:
    """ResidualFSQ.get_output_from_indices restated up to project_out (GFSQ._embed, dvae.py:85-96): index -> per-dimension level
    (index // basis) % levels -> code (level - half_width) / half_width, summed over the R quantizers with scale (levels-1)^-r."""
    lv = torch.tensor(levels, dtype=torch.int64)
    hw = torch.tensor([l // 2 for l in levels], dtype=torch.float32)
    basis = torch.cumprod(torch.tensor([1] + list(levels[:-1]), dtype=torch.int64), 0)
    ids = _t(ids).to(torch.int64)
    lat = []
    for g in range(G):
        acc = 0
        for r in range(R):
            level = (ids[g * R + r][:, None] // basis[None, :]) % lv[None, :]
            acc = acc + ((level.float() - hw) / hw) * (lv.float() - 1) ** (-r)
        lat.append(acc)
    return torch.stack(lat, 0)


@torch.no_grad()
def dvae_decode_codes(sd: Dict[str, np.ndarray], ids: torch.Tensor, levels=(5, 5, 5, 5), G: int = 2, R: int = 2) -> torch.Tensor:
    """DVAE.forward decode WITH a quantiser (dvae.py:272-291; the use_decoder=False branch, pipeline:292,435-439): ids [n, G*R] (one
    utterance's generated code ids) -> GFSQ._embed (dvae.py:85-96 -> GroupedResidualFSQ.get_output_from_indices, vector_quantize_pytorch
    1.17.8, third party: parity unpinned) = per group project_out(sum of the R residual codes), groups concatenated -> the decoder
    stack (pinned on the reference's module: tests/golden/dvae_full_decode_real.npz) -> mel [100, 2n]."""
    ids = _t(ids).to(torch.int64)
    lat = gfsq_latent_from_indices(ids.t().contiguous(), levels, G, R)                  # [G, n, 4]; row g*R + r of ids.T = codebook r of group g
    feats = [F.linear(lat[g], _t(sd[f"vq_layer.quantizer.rvqs.{g}.project_out.weight"]).float(),
                      _t(sd[f"vq_layer.quantizer.rvqs.{g}.project_out.bias"]).float()) for g in range(G)]
    feat = torch.cat(feats, dim=-1)                                                     # [n, dim * G] (= feat.transpose(1, 2) of _embed, per frame)
    dec = {k: v for k, v in sd.items() if k.startswith("decoder.") or k in ("out_conv.weight", "coef")}
    return dvae_decode(dec, feat)


@torch.no_grad()
def dvae_encoder_features(sd: Dict[str, np.ndarray], mel: torch.Tensor) -> torch.Tensor:
    """mel [100, F] -> encoder output [1024, F // 2]: div by coef, downsample_conv, DVAEDecoder-as-encoder (dvae.py:263-268)."""
    sd = {k: _t(v).float() for k, v in sd.items()}
    x = (_t(mel).float() / sd["coef"].view(-1, 1))[None]
    x = F.gelu(F.conv1d(x, sd["downsample_conv.0.weight"], sd["downsample_conv.0.bias"], stride=1, padding=1))
    x = F.gelu(F.conv1d(x, sd["downsample_conv.2.weight"], sd["downsample_conv.2.bias"], stride=2, padding=1))
    y = F.gelu(F.conv1d(x, sd["encoder.conv_in.0.weight"], sd["encoder.conv_in.0.bias"], padding=1))
    y = F.conv1d(y, sd["encoder.conv_in.2.weight"], sd["encoder.conv_in.2.bias"], padding=1)
    n_layer = 1 + max(int(k.split(".")[2]) for k in sd if k.startswith("encoder.decoder_block."))
    for i in range(n_layer):
        y = _convnext(sd, f"encoder.decoder_block.{i}.", y, dilation=2)
    return F.conv1d(y, sd["encoder.conv_out.weight"])[0]


@torch.no_grad()
def dvae_encode(sd: Dict[str, np.ndarray], wav: torch.Tensor, pre_bound: bool = True) -> torch.Tensor:
    """DVAE.forward(mode="encode") (dvae.py:263-270): wav [n] @ 24 kHz -> audio-prompt codes [4, T] (values < 625)."""
    feat = dvae_encoder_features(sd, mel_features(wav))
    return gfsq_indices(feat.transpose(0, 1), {k: _t(v).float() for k, v in sd.items()}, pre_bound=pre_bound)


# ----------------------------------------------------------------------------------------------
# Vocos (third-party vocos 0.1.0 -- PARITY UNPINNED, see module docstring)
# ----------------------------------------------------------------------------------------------

@torch.no_grad()
def vocos_backbone(sd, mel: torch.Tensor) -> torch.Tensor:
    """VocosBackbone.forward: embed conv k7 p3 -> LN -> 8 ConvNeXt (dil 1) -> final LN.  mel [100,F] -> [F,512]."""
    x = F.conv1d(mel[None], sd["backbone.embed.weight"], sd["backbone.embed.bias"], padding=3)
    C = x.shape[1]
    x = F.layer_norm(x.transpose(1, 2), (C,), sd["backbone.norm.weight"], sd["backbone.norm.bias"], eps=1e-6).transpose(1, 2)
    n_layer = 1 + max(int(k.split(".")[2]) for k in sd if k.startswith("backbone.convnext."))
    for i in range(n_layer):
        x = _convnext(sd, f"backbone.convnext.{i}.", x, dilation=1)
    x = F.layer_norm(x.transpose(1, 2), (C,), sd["backbone.final_layer_norm.weight"], sd["backbone.final_layer_norm.bias"], eps=1e-6)
    return x[0]


@torch.no_grad()
def vocos_head_spec(sd, feats: torch.Tensor):
    """ISTFTHead.forward up to the complex spectrogram: Linear(512->n_fft+2); mag = clip(exp(.), max=1e2);
    S = mag * (cos p + i sin p).  Returns (real, imag) each [n_fft/2+1, F]."""
    x = F.linear(feats, sd["head.out.weight"], sd["head.out.bias"]).transpose(0, 1)    # [1026, F]
    mag, p = x.chunk(2, dim=0)
    mag = torch.clip(torch.exp(mag), max=1e2)
    return mag * torch.cos(p), mag * torch.sin(p)


@torch.no_grad()
def vocos_decode(sd: Dict[str, np.ndarray], mel: torch.Tensor, n_fft: int = 1024, hop: int = 256) -> torch.Tensor:
    """vocos.Vocos.decode(mel[1,100,F]) -> wav [hop*(F-1)]  (called at pipeline:303).
    ISTFT with padding="center" == torch.istft(S, n_fft, hop, n_fft, window, center=True)."""
    sd = {k: _t(v).float() for k, v in sd.items()}
    re, im = vocos_head_spec(sd, vocos_backbone(sd, _t(mel).float()))
    S = torch.complex(re, im)[None]
    return torch.istft(S, n_fft, hop, n_fft, sd["head.istft.window"], center=True)[0]


@torch.no_grad()
def istft_direct(re: torch.Tensor, im: torch.Tensor, window: torch.Tensor, n_fft: int = 1024, hop: int = 256) -> torch.Tensor:
    """Direct definition of the centred ISTFT (irfft per frame, windowed overlap-add, divide by the
    window-square envelope, trim n_fft/2) in fp64 -- used to cross-check torch.istft and the HIP DFT-GEMM path."""
    Fr = re.shape[1]
    frames = torch.fft.irfft(torch.complex(re.double(), im.double()).transpose(0, 1), n=n_fft, dim=-1)  # [F, n_fft]
    frames = frames * window.double()
    total = n_fft + hop * (Fr - 1)
    out = torch.zeros(total, dtype=torch.float64)
    env = torch.zeros(total, dtype=torch.float64)
    w2 = window.double() ** 2
    for f in range(Fr):
        out[f * hop: f * hop + n_fft] += frames[f]
        env[f * hop: f * hop + n_fft] += w2
    out = out[n_fft // 2: total - n_fft // 2]
    env = env[n_fft // 2: total - n_fft // 2]
    return (out / env).float()
