"""hip_models.GPT -- drop-in for chattts_plus.models.GPT (reference chattts_plus/models/gpt.py) whose
decoder, heads, sampler and generate-loop bookkeeping run in libctts_hip.so on an MI355X.

Same constructor kwargs, `__call__(input_ids, text_mask) -> emb`, `generate(...)` generator of
GenerationOutputs, `num_vq`, `emb_code[i].num_embeddings`, `.eval()`, `.to()`, `from_pretrained`.
torch is used for device memory, the current stream and the (negligible) embedding gather only.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import List, Optional, Union

import numpy as np
import torch
import torch.nn.functional as F

from .. import _lib

RMS_EPS = 1e-6
ROPE_BASE = 10000.0


def rope_table(n_pos: int) -> np.ndarray:
    """[n_pos][64] = cos[32] | sin[32] with the reference's fp32 arithmetic (llama.py:100,106-119)."""
    inv_freq = 1.0 / (ROPE_BASE ** (torch.arange(0, 64, 2, dtype=torch.int64).float() / 64))
    freqs = torch.arange(n_pos, dtype=torch.int64)[:, None].float() * inv_freq[None, :]
    return torch.cat((freqs.cos(), freqs.sin()), dim=-1).contiguous().numpy()


class _EmbShim:
    def __init__(self, n):
        self.num_embeddings = n


@dataclass(repr=False, eq=False)
class GenerationOutputs:          # gpt.py:280-284
    ids: List[torch.Tensor]
    attentions: list
    hiddens: List[torch.Tensor]


class Context:                    # gpt.py:87-95
    def __init__(self):
        self._interrupt = False

    def set(self, v: bool):
        self._interrupt = v

    def get(self) -> bool:
        return self._interrupt


def sampler_cfg_from_objects(temperature, eos_token, max_new_token, min_new_token, logits_warpers, logits_processors,
                             num_vq=4, infer_text=False) -> _lib.SamplerCfg:
    """Reads the scalars off the HF warpers / reference processor objects that processors.gen_logits builds
    (models/processors.py:37-57).  Unknown object types are rejected (no silent host fallback)."""
    sc = _lib.SamplerCfg()
    t = torch.as_tensor(temperature, dtype=torch.float32).flatten().tolist()
    if len(t) == 1:
        t = t * num_vq
    for i in range(num_vq):
        sc.temperature[i] = t[i]
    sc.top_p_threshold = -1.0
    sc.top_k = 0
    sc.min_tokens_to_keep = 1
    # the kernel applies top-p FIRST and top-k second -- the order processors.gen_logits builds (models/processors.py:43-48) and the
    # reference's loop applies them in (gpt.py:474-475); the two do not commute, so any other list is rejected rather than re-ordered
    seen = []
    for w in logits_warpers or []:
        if hasattr(w, "top_p"):
            if seen:
                raise _lib.HipBackendError("logits_warpers: the hip sampler applies [top-p, top-k] in that order (processors.gen_logits); "
                                           f"got a top-p warper after {seen}")
            # torch compares the fp32 cumsum with the python double (1 - top_p) cast to fp32
            sc.top_p_threshold = float(np.float32(1 - float(w.top_p)))
            sc.min_tokens_to_keep = int(w.min_tokens_to_keep)
            seen.append("top_p")
        elif hasattr(w, "top_k"):
            if "top_k" in seen:
                raise _lib.HipBackendError("logits_warpers: more than one top-k warper")
            sc.top_k = int(w.top_k)          # already max(top_k, min_tokens_to_keep)
            seen.append("top_k")
        else:
            raise _lib.HipBackendError(f"unsupported logits warper for the hip backend: {type(w).__name__}")
    sc.use_penalty = 0
    sc.past_window = 16
    sc.max_input_ids = 1 << 30
    tab = torch.ones(17)
    for p in logits_processors or []:
        if hasattr(p, "penalty") and hasattr(p, "past_window"):
            sc.use_penalty = 1
            sc.past_window = int(p.past_window)
            sc.max_input_ids = int(p.max_input_ids)
            tab = torch.pow(float(p.penalty), torch.arange(0, 17, dtype=torch.int64))     # processors.py:28
        else:
            raise _lib.HipBackendError(f"unsupported logits processor for the hip backend: {type(p).__name__}")
    for i in range(17):
        sc.penalty_table[i] = float(tab[i])
    sc.eos_token = int(eos_token)
    sc.min_new_token = int(min_new_token)
    sc.max_new_token = int(max_new_token)
    sc.infer_text = 1 if infer_text else 0
    if infer_text and sc.use_penalty:
        raise _lib.HipBackendError("infer_text=True supports repetition_penalty == 1 only: the reference's processor receives a [B,n,1] "
                                   "history in this mode and mis-broadcasts it (models/processors.py:18-34 with gpt.py:458-467)")
    return sc


def compact_size(n_live: int) -> int:
    """Decode-batch size used for `n_live` unfinished sequences (finished-row compaction, ctts_gpt_compact): every size up to 16, then
    multiples of 2 up to 32, of 4 up to 64, of 8 beyond -- a captured decode graph exists per batch size, so the sizes are quantised;
    the padding rows are finished sequences left in the batch."""
    n = max(int(n_live), 1)
    if n <= 16:
        return n
    q = 2 if n <= 32 else (4 if n <= 64 else 8)
    return (n + q - 1) // q * q


class RowBook:
    """Continuous batching bookkeeping (GPT.generate_many): which utterance sits in which decode row.  Every seating gets a ticket; the row
    states the engine reports ({fin, end} per row, ctts_gpt_rows_enqueue) arrive one or two chunks late and are interpreted through the
    ticket layout that was current when the report was ENQUEUED -- a report taken before an admission or a compaction can neither finish the
    row's new occupant nor be confused by rows having moved."""

    def __init__(self):
        self.row_tk: List[Optional[int]] = []      # ticket per current decode row (None: the row's utterance is done, the row is free)
        self.tickets = {}                          # live ticket -> [utterance, attempt, current row]
        self._next = 0

    def seat(self, row: int, utt: int, attempt: int = 0) -> int:
        tk = self._next
        self._next += 1
        if row == len(self.row_tk):
            self.row_tk.append(tk)
        else:
            assert self.row_tk[row] is None, "seat: the row is occupied"
            self.row_tk[row] = tk
        self.tickets[tk] = [utt, attempt, row]
        return tk

    def layout(self) -> list:
        return list(self.row_tk)

    def free_rows(self) -> List[int]:
        return [r for r, tk in enumerate(self.row_tk) if tk is None]

    def live_rows(self) -> List[int]:
        return [r for r, tk in enumerate(self.row_tk) if tk is not None]

    def report(self, layout, states, ensure_non_empty: bool, max_restarts: int):
        """`states` = [(fin, end)] per row of `layout`.  Returns ([(utterance, tokens generated)] completed, [(utterance, next attempt)] to admit
        again: their first token was EOS, gpt.py:496-525)."""
        done, again = [], []
        for tk, (fin, end) in zip(layout, states):
            if tk is None or not fin or tk not in self.tickets:
                continue
            utt, att, row = self.tickets.pop(tk)
            self.row_tk[row] = None
            if (fin & 2) and end == 0 and ensure_non_empty and att + 1 < max_restarts:
                again.append((utt, att + 1))
            else:
                done.append((utt, int(end)))
        return done, again

    def compact(self, keep: List[int]) -> None:
        """rows `keep` (ascending) become rows 0..len(keep)-1 (ctts_gpt_compact)"""
        self.row_tk = [self.row_tk[r] for r in keep]
        for nr, tk in enumerate(self.row_tk):
            if tk is not None:
                self.tickets[tk][2] = nr


class _BusyToken:
    """One generate() at a time per KV cache: shared by every engine bound to the same KV tensor (LoRA-merged siblings, with_lora)."""
    def __init__(self):
        self.owner = None


class GPT:
    """See module docstring.  Extra kwargs (ride in the YAML `kwargs`):
    max_batch (<=128), max_seq_len, weight_dtype "fp32" (default, parity) | "fp16" (fast), chunk_steps."""

    Context = Context
    GenerationOutputs = GenerationOutputs

    def __init__(self, gpt_config: dict, num_audio_tokens: int = 626, num_text_tokens: int = 21178, num_vq=4,
                 use_flash_attn=False, **kwargs):
        self.num_vq = num_vq
        self.num_audio_tokens = num_audio_tokens
        self.num_text_tokens = num_text_tokens
        self.gpt_config = dict(gpt_config)
        self.model_dim = int(gpt_config["hidden_size"])
        self.emb_code = [_EmbShim(num_audio_tokens) for _ in range(num_vq)]
        self.max_batch = int(kwargs.get("max_batch", 4))
        self.max_seq = int(kwargs.get("max_seq_len", 4096))
        wd = kwargs.get("weight_dtype", "fp32")        # parity mode by default; "fp16" = fast mode (Modes)
        self.dtype_code = {"fp16": _lib.DTYPE_F16, "float16": _lib.DTYPE_F16, "fp32": _lib.DTYPE_F32, "float32": _lib.DTYPE_F32}[str(wd)]
        self.chunk_steps = int(kwargs.get("chunk_steps", 32))
        self.use_graph = bool(kwargs.get("use_graph", True))
        self.device = torch.device(kwargs.get("device", "cuda"))
        self.gpt = None              # attribute the pipeline swaps for LoRA (pipeline:420-432); unused here
        self._lib = _lib.load()
        if not torch.cuda.is_available():
            raise _lib.HipBackendError("infer_type='hip' needs a visible MI355X (torch.cuda.is_available() is False); no CPU fallback")
        self._h = C.c_void_p()
        cfg = _lib.GptCfg(hidden=self.model_dim, inter=int(gpt_config["intermediate_size"]),
                          heads=int(gpt_config["num_attention_heads"]), layers=int(gpt_config["num_hidden_layers"]),
                          vocab_code=num_audio_tokens, num_vq=num_vq, max_batch=self.max_batch, max_seq=self.max_seq,
                          dtype=self.dtype_code)
        with torch.cuda.device(self.device):
            _lib.check(self._lib.ctts_gpt_create(C.byref(cfg), C.byref(self._h)), "ctts_gpt_create")
        self._finalized = False
        self._kv = None
        self._busy_token = _BusyToken()
        self._lora = []
        # engine options (ctts_gpt_set_option; include/ctts_hip.h lists them): {"prefill_split_rows": 0, ...}; applied before the weights are packed
        self.options = dict(kwargs.get("options") or {})
        for k, v in self.options.items():
            self.set_option(k, v)
        self.compact = bool(kwargs.get("compact", True))      # finished-row compaction at chunk boundaries (batches of >= 8 sequences)
        self.compact_chunk = int(kwargs.get("compact_chunk", 8))   # ... whose chunks are this short: a finished row leaves the batch 1-2 chunks later
        self.model_path = kwargs.get("model_path", None)
        if self.model_path:
            self.from_pretrained(self.model_path)

    def set_option(self, name: str, value: int) -> None:
        """One named engine option (ctts_gpt_set_option): an explicit call, never the environment.  Unknown names are an error."""
        with torch.cuda.device(self.device):
            _lib.check(self._lib.ctts_gpt_set_option(self._h, str(name).encode(), int(value)), f"set_option({name})")
        self.options[str(name)] = int(value)

    def get_option(self, name: str) -> int:
        """The effective value of an engine option (ctts_gpt_get_option)."""
        v = C.c_int(0)
        _lib.check(self._lib.ctts_gpt_get_option(self._h, str(name).encode(), C.byref(v)), f"get_option({name})")
        return int(v.value)

    # -- nn.Module-like surface ------------------------------------------------------------------
    def eval(self):
        return self

    def to(self, device=None, dtype=None):
        return self

    @property
    def busy(self) -> bool:
        """A generate() generator is live on this engine or on one that shares its KV cache."""
        return getattr(self, "_busy_token", None) is not None and self._busy_token.owner is not None

    def close(self, _force: bool = False):
        """Destroys the engine handle (packed weights, workspaces, graphs) now instead of at garbage collection; the KV
        tensor is released with the last engine that shares it.  Refused while a generate() generator of THIS engine is live (its
        native handle is in use): exhaust or close the generator first."""
        if not _force and getattr(self, "_busy_token", None) is not None and self._busy_token.owner is self:
            raise _lib.HipBackendError("GPT.close(): a generate() generator is still live on this engine; exhaust or close it first")
        if getattr(self, "_h", None) and self._h.value:
            self._lib.ctts_gpt_destroy(self._h)
            self._h = C.c_void_p()
        self._finalized = False
        self._kv = None

    def __del__(self):
        try:
            self.close(_force=True)
        except Exception:
            pass

    def from_pretrained(self, file_path: str):
        self.load_state_dict(torch.load(file_path, weights_only=True, mmap=True))      # gpt.py:84-85

    def add_lora(self, layer: int, target: str, A, B, scale: float):
        """peft merge rule W += scale * B @ A (pipeline:420-432) applied before finalize."""
        self._lora.append((layer, target, np.ascontiguousarray(A, dtype=np.float32), np.ascontiguousarray(B, dtype=np.float32), float(scale)))

    @staticmethod
    def _known_key(k: str, layers: int, num_vq: int) -> bool:
        """The 196 keys of the reference's GPT state dict; the engine's ctts_gpt_set_weight is strict about them."""
        import re
        m = re.fullmatch(r"gpt\.layers\.(\d+)\.(self_attn\.[qkvo]_proj|mlp\.(gate|up|down)_proj|input_layernorm|post_attention_layernorm)\.weight", k)
        if m:
            return int(m.group(1)) < layers
        if k in ("gpt.norm.weight", "emb_text.weight", "head_text.parametrizations.weight.original0", "head_text.parametrizations.weight.original1"):
            return True
        m = re.fullmatch(r"emb_code\.(\d+)\.weight", k) or re.fullmatch(r"head_code\.(\d+)\.parametrizations\.weight\.original[01]", k)
        return bool(m) and int(m.group(1)) < num_vq

    def load_state_dict(self, sd, strict: bool = True, _share_kv: Optional[torch.Tensor] = None, _busy_token=None):
        """strict=True (default, like the reference's load_state_dict, gpt.py:84-85): an unexpected key is an error.  strict=False: keys the
        engine does not know (e.g. `...rotary_emb.inv_freq` buffers of older transformers exports) are skipped and returned in
        `self.unexpected_keys`; missing keys are an error either way (the engine cannot run without them)."""
        if self._finalized:
            raise _lib.HipBackendError("weights already loaded")
        keep = []
        self.unexpected_keys = [k for k in sd if not self._known_key(k, int(self.gpt_config["num_hidden_layers"]), self.num_vq)]
        if self.unexpected_keys and strict:
            raise _lib.HipBackendError(f"load_state_dict(strict=True): unexpected key(s) {self.unexpected_keys[:4]}{' ...' if len(self.unexpected_keys) > 4 else ''}")
        if _busy_token is not None:
            self._busy_token = _busy_token
        for k, v in sd.items():
            if k in self.unexpected_keys:
                continue
            a = v.detach().cpu().float().numpy() if isinstance(v, torch.Tensor) else np.asarray(v, dtype=np.float32)
            a = np.ascontiguousarray(a)
            keep.append(a)
            _lib.check(self._lib.ctts_gpt_set_weight(self._h, k.encode(), a.ctypes.data_as(C.c_void_p), a.size), f"set_weight({k})")
        for (layer, target, A, B, scale) in self._lora:
            _lib.check(self._lib.ctts_gpt_merge_lora(self._h, layer, target.encode(), A.ctypes.data_as(C.c_void_p),
                                                     B.ctypes.data_as(C.c_void_p), A.shape[0], scale), "merge_lora")
        with torch.cuda.device(self.device):
            _lib.check(self._lib.ctts_gpt_finalize(self._h), "ctts_gpt_finalize")
            rope = rope_table(self.max_seq)
            _lib.check(self._lib.ctts_gpt_set_rope(self._h, rope.ctypes.data_as(C.c_void_p), self.max_seq), "set_rope")
            nbytes = self._lib.ctts_gpt_kv_bytes(self._h)
            # torch owns the KV cache; LoRA-merged siblings borrow the base engine's (one generate() runs at a time)
            self._kv = _share_kv if (_share_kv is not None and _share_kv.numel() >= nbytes) else torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            _lib.check(self._lib.ctts_gpt_bind_kv(self._h, self._kv.data_ptr(), nbytes), "bind_kv")
        self._finalized = True
        self._ctor = dict(gpt_config=self.gpt_config, num_audio_tokens=self.num_audio_tokens, num_text_tokens=self.num_text_tokens,
                          num_vq=self.num_vq, max_batch=self.max_batch, max_seq_len=self.max_seq,
                          weight_dtype="fp16" if self.dtype_code == _lib.DTYPE_F16 else "fp32", chunk_steps=self.chunk_steps,
                          use_graph=self.use_graph, device=str(self.device), options=dict(self.options))
        self._sd_host = {k: v for k, v in sd.items() if k not in self.unexpected_keys}        # kept for LoRA-merged siblings (pipeline:420-432)
        return self

    def with_lora(self, adapters) -> "GPT":
        """A sibling engine whose q/k/v/o weights carry W += scale * B @ A (peft merge_and_unload, pipeline:420-432);
        the base engine stays untouched, like the reference's gpt_org swap (pipeline:424,465-470)."""
        g = GPT(**self._ctor)
        for (layer, target, A, B, scale) in adapters:
            g.add_lora(layer, target, A, B, scale)
        g.load_state_dict(self._sd_host, _share_kv=self._kv, _busy_token=self._busy_token)     # one KV cache -> one busy token
        return g

    # -- per-utterance LoRA: adapters resident beside the packed weights, one slot (or none) per sequence -------------
    def load_adapter(self, slot: int, adapters) -> None:
        """`adapters` = [(layer, target, A[r,in], B[out,r], scale)] as returned by pipeline.load_lora_adapter; replaces whatever the slot held."""
        if not self._finalized:
            raise _lib.HipBackendError("weights not loaded")
        with torch.cuda.device(self.device):
            _lib.check(self._lib.ctts_gpt_clear_adapter(self._h, int(slot)), "clear_adapter")
            for (layer, target, A, B, scale) in adapters:
                A = np.ascontiguousarray(A, dtype=np.float32); B = np.ascontiguousarray(B, dtype=np.float32)
                _lib.check(self._lib.ctts_gpt_set_adapter(self._h, int(slot), int(layer), target.encode(), A.ctypes.data_as(C.c_void_p),
                                                          B.ctypes.data_as(C.c_void_p), int(A.shape[0]), float(scale)), "set_adapter")

    def set_row_adapters(self, slots) -> None:
        """Adapter slot (or -1 / None) per sequence for the following generate() calls; None switches the per-row path off.  With it
        every q/k/v/o projection evaluates W x + scale * B (A x) per row (decode: inside the projection launches, lora_worker.h; prompt pass: two more launches per layer)."""
        with torch.cuda.device(self.device):
            if slots is None:
                _lib.check(self._lib.ctts_gpt_set_row_adapters(self._h, None, 0), "set_row_adapters")
                return
            arr = np.ascontiguousarray([(-1 if s is None else int(s)) for s in slots], dtype=np.int32)
            _lib.check(self._lib.ctts_gpt_set_row_adapters(self._h, arr.ctypes.data_as(C.c_void_p), int(arr.size)), "set_row_adapters")

    # -- get_emb (gpt.py:125-149) ----------------------------------------------------------------
    def __call__(self, input_ids: torch.Tensor, text_mask: torch.Tensor, spk_emb=None, spk_emb_ids: Optional[int] = None) -> torch.Tensor:
        """emb[B,T,H] fp32 on the device, computed by embed_prompt_kernel (ctts_gpt_embed).  With `spk_emb` (a [H] vector, a
        [B,H] table or a base16384 string) the rows whose id == spk_emb_ids are overwritten by F.normalize(spk) in the same
        launch (Tokenizer.apply_spk_emb, tokenizer.py:150-178)."""
        if not self._finalized:
            raise _lib.HipBackendError("weights not loaded")
        ids = input_ids.to(self.device).to(torch.int32).contiguous()
        tm = text_mask.to(self.device).to(torch.int32).contiguous()
        B, T = int(ids.shape[0]), int(ids.shape[1])
        # nn.Embedding raises on an id outside its table (gpt.py:125-149: emb_text on the text rows' first column, emb_code[i] on the code rows);
        # the gather kernel does not check, so the same IndexError is raised here (one reduction + one flag read per prompt)
        tmb = tm.bool()
        bad = ((ids[..., 0] < 0) | (ids[..., 0] >= self.num_text_tokens)) & tmb
        bad = bad | (((ids < 0) | (ids >= self.num_audio_tokens)).any(-1) & ~tmb)
        if bool(bad.any()):
            raise IndexError("index out of range in self")
        emb = torch.empty(B, T, self.model_dim, dtype=torch.float32, device=self.device)
        spk_t, sid = None, -1
        if spk_emb is not None:
            from .. import codec
            v = codec.speaker_to_vector(spk_emb) if (isinstance(spk_emb, str) or torch.as_tensor(spk_emb).dim() == 1) else torch.as_tensor(spk_emb).float()
            v = v.reshape(-1, self.model_dim)
            v = F.normalize(v, p=2.0, dim=1, eps=1e-12)                    # == normalize(dim=0) of each [H] vector
            spk_t = v.expand(B, -1).contiguous().to(self.device)
            sid = int(spk_emb_ids)
        with torch.cuda.device(self.device):
            _lib.check(self._lib.ctts_gpt_embed(self._h, ids.data_ptr(), tm.data_ptr(), B, T, spk_t.data_ptr() if spk_t is not None else None,
                                                sid, emb.data_ptr(), self._stream()), "embed")
        return emb

    forward = __call__

    # -- generate (gpt.py:313-569, infer_text=False) ----------------------------------------------
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    @torch.no_grad()
    def _report_saturations(self, h, st, what: str) -> int:
        """fp16 stores of unbounded values saturate instead of overflowing to inf (fp16 engines: SwiGLU outputs, packed residual; fp32 engines:
        the SwiGLU head / tail images of a prompt pass over more than 64 rows, prefill_split.hip, and of decode batches of >= 9 rows).  A non-zero count means the checkpoint drives
        activations past the fp16 range: the result is finite but clipped -- say so."""
        nsat = C.c_int32(0)
        _lib.check(self._lib.ctts_gpt_saturations(h, C.byref(nsat), st), "saturations")
        n = int(nsat.value)
        if n:
            import warnings
            if self.dtype_code == _lib.DTYPE_F16:
                msg = f"use weight_dtype='fp32' for this checkpoint"
            else:
                msg = ("the fp32 engine's prompt passes of more than 64 rows and its decode batches of >= 9 rows keep silu(gate) * up / 16 and the scaled residual rows as fp16 "
                       "head / tail images; THIS call's values were clipped there.  The engine now keeps the exact fp32 kernels (split_decode_rows=0, prefill_split_rows=0) for "
                       "the following calls; construct the GPT with options={'prefill_split_rows': 0, 'split_decode_rows': 0} to keep them from the first call on for this checkpoint")
                for opt in ("split_decode_rows", "prefill_split_rows"):
                    try:
                        if self.get_option(opt) > 0:
                            self.set_option(opt, 0)
                    except Exception:      # noqa: BLE001 (the warning below still tells the caller)
                        pass
            warnings.warn(f"hip GPT (weight_dtype {'fp16' if self.dtype_code == _lib.DTYPE_F16 else 'fp32'}): {n} fp16 stores saturated or were NaN "
                          f"during this {what}; {msg}", RuntimeWarning)
        return n

    def generate(self, emb: torch.Tensor, inputs_ids: torch.Tensor, temperature: torch.Tensor,
                 eos_token: Union[int, torch.Tensor], attention_mask: Optional[torch.Tensor] = None, max_new_token=2048,
                 min_new_token=0, logits_warpers=[], logits_processors=[], infer_text=False, return_attn=False,
                 return_hidden=False, stream=False, show_tqdm=True, ensure_non_empty=True, stream_batch=24,
                 context=None, noise="auto", seed: Optional[int] = None, max_restarts: int = 64, utt_ids=None, max_new_tokens_per_row=None):
        """`noise`: "torch" draws q = empty(B*4,V).exponential_() per step from torch's CPU generator -- the very numbers
        torch.multinomial consumes in the reference, so TorchSeedContext(seed) reproduces the CPU path's tokens (costs
        ~21 ns of host time per element: hidden behind the GPU up to batch ~8, 3x the step time at batch 32); "device"
        uses the on-device Philox generator keyed by `seed` (None: one draw from torch's CPU generator, so manual_seed
        still makes the call reproducible); "auto" (default) = "torch" for the batches the reference itself can run
        (<= 4 sequences, pipeline:391-397) as long as a step draws at most 16k numbers (code mode), "device" otherwise
        (larger batches, the 21178-wide refine-text pass); or an array [n_draws, B*4, V].
        `utt_ids` (device noise): one global utterance id per sequence (default 0..B-1) -- the device noise stream of a sequence is keyed by
        (seed, its utterance id, codebook, its own step and regenerate attempt), not by its batch row, so an utterance samples the same
        noise in whatever slice / batch position / rank it is served.  `max_new_tokens_per_row`: per-sequence token limits (<= max_new_token)."""
        if return_attn:
            raise _lib.HipBackendError("return_attn=True is unsupported (the reference's eager attention path is broken, SURVEY F2)")
        if not self._finalized:
            raise _lib.HipBackendError("weights not loaded")
        if self._busy_token.owner is not None:
            # engine state (batch, step counters, noise staging ring) and the KV cache -- shared with LoRA-merged sibling engines -- belong
            # to ONE generate() at a time
            raise _lib.HipBackendError("GPT.generate is already running on this engine (or on an engine sharing its KV cache): exhaust or close "
                                       "the previous generator first")
        self._busy_token.owner = self
        try:
            yield from self._generate(emb, inputs_ids, temperature, eos_token, attention_mask, max_new_token, min_new_token, logits_warpers,
                                      logits_processors, infer_text, return_hidden, stream, ensure_non_empty, stream_batch, context, noise, seed,
                                      max_restarts, utt_ids, max_new_tokens_per_row)
        finally:
            self._busy_token.owner = None

    def _generate(self, emb, inputs_ids, temperature, eos_token, attention_mask, max_new_token, min_new_token, logits_warpers, logits_processors,
                  infer_text, return_hidden, stream, ensure_non_empty, stream_batch, context, noise, seed, max_restarts, utt_ids=None,
                  row_limits=None):
        context = context or Context()
        lib, h = self._lib, self._h
        B, T = int(inputs_ids.shape[0]), int(inputs_ids.shape[1])
        V, H, NVQ = (self.num_text_tokens if infer_text else self.num_audio_tokens), self.model_dim, self.num_vq
        rows_per_seq = 1 if infer_text else NVQ          # multinomial rows per sequence: [B, V_text] vs [B*4, 626] (gpt.py:444-467)
        dev = self.device
        max_new_token = int(max_new_token)
        sc = sampler_cfg_from_objects(temperature, int(eos_token), max_new_token, min_new_token, logits_warpers, logits_processors, NVQ,
                                      infer_text=infer_text)
        if isinstance(noise, str) and noise == "auto":
            # host draws cost ~21 ns per element and step: 10k elements (4 sequences x 4 x 626) hide behind the GPU step, one
            # 21178-wide refine-text row already does not (measured: 502 vs 423 us/step at batch 1, 1.2 vs 0.5 ms at batch 4)
            noise = "torch" if (B <= 4 and B * rows_per_seq * V <= 16384) else "device"
        if seed is None:
            seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if (isinstance(noise, str) and noise == "device") else 0
        mask = torch.ones(B, T, dtype=torch.int32, device=dev) if attention_mask is None else attention_mask.to(dev).to(torch.int32).contiguous()
        emb = emb.to(dev, dtype=torch.float32).contiguous()
        # uninitialised: only [b, :end_idx[b]] is ever read, and every one of those rows was written by the step that produced it (the
        # default max_new_token = 2048 made this a 201 MB zero-fill per call at batch 32)
        ids = torch.empty(B, max_new_token, NVQ, dtype=torch.int32, device=dev)
        hid = torch.empty(B, max_new_token, H, dtype=torch.float32, device=dev) if return_hidden else None
        finish = torch.zeros(B, dtype=torch.int32, device=dev)
        end_idx = torch.zeros(B, dtype=torch.int32, device=dev)
        n_draws = max_new_token + max_restarts
        qbuf = None
        rng_states = []
        host_q = None
        if isinstance(noise, str) and noise == "torch":
            qbuf = torch.empty(n_draws, B * rows_per_seq, V, dtype=torch.float32, device=dev)
        elif isinstance(noise, str) and noise == "device":
            qbuf = None
        else:
            host_q = torch.as_tensor(noise, dtype=torch.float32)
            n_draws = int(host_q.shape[0])
            qbuf = host_q.to(dev).contiguous()
        uid_arr = lim_arr = None
        if utt_ids is not None:
            uid_arr = np.ascontiguousarray([int(u) for u in utt_ids], dtype=np.uint64)
            assert uid_arr.size == B, "utt_ids: one id per sequence"
        if row_limits is not None:
            lim_arr = np.ascontiguousarray([int(u) for u in row_limits], dtype=np.int32)
            assert lim_arr.size == B, "max_new_tokens_per_row: one limit per sequence"
        io = _lib.GenIO(ids=ids.data_ptr(), hiddens=hid.data_ptr() if hid is not None else None, finish=finish.data_ptr(),
                        end_idx=end_idx.data_ptr(), noise=qbuf.data_ptr() if qbuf is not None else None, n_draws=n_draws,
                        seed=int(seed), utt_ids=uid_arr.ctypes.data if uid_arr is not None else None,
                        row_limits=lim_arr.ctypes.data if lim_arr is not None else None)
        drawn = 0
        stage_cap = max(int(self.chunk_steps), int(stream_batch) if stream else 1, 1)
        stage = self._staging(B * rows_per_seq, V, stage_cap) if (isinstance(noise, str) and noise == "torch") else None

        def draw_to(n):
            nonlocal drawn
            if not (isinstance(noise, str) and noise == "torch"):
                return
            n = min(n, n_draws)
            if n <= drawn:
                return
            while drawn < n:
                # persistent pinned staging ring (3 slots): exponential_ straight into pinned memory, stream-ordered async
                # upload that lands after the chunk in flight and before the steps that read it; a slot is reused once the
                # copy that last read it has completed
                k = min(n - drawn, stage_cap)
                slot = self._stage_next % 3
                self._stage_next += 1
                buf, ev = stage[slot]
                if ev is not None:
                    ev.synchronize()
                for i in range(k):
                    rng_states.append(torch.random.get_rng_state())
                    buf[i].exponential_(1)
                qbuf[drawn:drawn + k].copy_(buf[:k], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(dev))
                stage[slot][1] = ev
                drawn += k

        import time as _t
        tm = self.host_timing = {}
        t_last = [_t.perf_counter()]

        def tick(name):
            now = _t.perf_counter()
            tm[name] = tm.get(name, 0.0) + (now - t_last[0]) * 1e3
            t_last[0] = now

        with torch.cuda.device(dev):
            st = self._stream()
            tick("setup")
            _lib.check(lib.ctts_gpt_begin(h, B, T, mask.data_ptr(), C.byref(sc), C.byref(io), st), "begin")
            _lib.check(lib.ctts_gpt_prefill(h, emb.data_ptr(), st), "prefill")
            tick("begin+prefill")
            steps, alld = C.c_int32(0), C.c_int32(0)
            used_draws = 0
            # step 0 (+ ensure_non_empty regenerate, gpt.py:496-525)
            while True:
                draw_to(used_draws + 1)
                _lib.check(lib.ctts_gpt_sample(h, st), "sample")
                used_draws += 1
                _lib.check(lib.ctts_gpt_progress(h, C.byref(steps), C.byref(alld), st), "progress")
                if bool(finish.any().item()):
                    if ensure_non_empty and used_draws < max_restarts:
                        _lib.check(lib.ctts_gpt_restart(h, st), "restart")
                        continue
                    self._restore_rng(rng_states, used_draws)
                    return                                   # gpt.py:525 bare return
                break
            tick("step0")
            chunk = int(stream_batch) if stream else self.chunk_steps
            if stream:
                # streaming: synchronous polling; like the reference (gpt.py:531-543) a partial result is yielded whenever the number
                # of generated tokens is a multiple of stream_batch -- also when that step was the last one, so the final yield
                # below can repeat it, exactly as the reference does
                while not alld.value and steps.value < max_new_token and not context.get():
                    n = min(chunk - steps.value % chunk, max_new_token - steps.value)
                    draw_to(used_draws + n)
                    _lib.check(lib.ctts_gpt_decode(h, n, 1 if self.use_graph else 0, st), "decode")
                    prev = steps.value
                    _lib.check(lib.ctts_gpt_progress(h, C.byref(steps), C.byref(alld), st), "progress")
                    used_draws += steps.value - prev
                    if steps.value == prev and not alld.value:
                        raise _lib.HipBackendError("decode made no progress (device state inconsistent)")
                    if steps.value % chunk == 0 and steps.value > prev:
                        yield self._outputs(ids, hid, end_idx, infer_text)
            else:
                # one chunk stays in flight while the previous chunk's progress words are inspected (no GPU bubble at the
                # poll); steps launched after every sequence finished exit at their first instruction on the device
                pins = [torch.zeros(4, dtype=torch.int32).pin_memory() for _ in range(2)]
                evs = [torch.cuda.Event() for _ in range(2)]
                restarts = used_draws - steps.value            # draws spent on ensure_non_empty regenerations
                launched, n_chunks, pending = steps.value, 0, []
                # finished-row compaction (no counterpart in the reference, where finished rows keep computing until the slowest one ends,
                # gpt.py:527-546): the per-row finish flags travel with the progress words; rows known to have finished are dropped from
                # the decode batch at the next chunk boundary (ctts_gpt_compact), down to the next size of compact_size()
                compacting = self.compact and (not infer_text) and B >= 8
                if compacting:
                    chunk = max(4, min(chunk, self.compact_chunk))
                rowpins = [torch.zeros(2 * B, dtype=torch.int32).pin_memory() for _ in range(2)] if compacting else None
                row_seq = list(range(B))                       # current decode row -> sequence
                layouts = [None, None]
                done_seq = set()
                self.compactions = []
                while True:
                    while len(pending) < 2 and launched < max_new_token and not context.get():
                        n = min(chunk, max_new_token - launched)
                        tick("loop_other")
                        draw_to(restarts + launched + n)
                        tick("draw")
                        _lib.check(lib.ctts_gpt_decode(h, n, 1 if self.use_graph else 0, st), "decode")
                        tick("decode_launch")
                        launched += n
                        slot = n_chunks % 2
                        n_chunks += 1
                        _lib.check(lib.ctts_gpt_progress_enqueue(h, pins[slot].data_ptr(), st), "progress_enqueue")
                        if compacting:
                            _lib.check(lib.ctts_gpt_rows_enqueue(h, rowpins[slot].data_ptr(), st), "rows_enqueue")
                            layouts[slot] = list(row_seq)
                        evs[slot].record(torch.cuda.current_stream(dev))
                        pending.append(slot)
                    if not pending:
                        break
                    slot = pending.pop(0)
                    tick("loop_other")
                    evs[slot].synchronize()
                    tick("event_wait")
                    if int(pins[slot][2]) or int(pins[slot][0]) >= max_new_token or context.get():
                        break
                    if compacting:
                        flags = rowpins[slot][:2 * len(layouts[slot])].view(-1, 2)[:, 0].tolist()
                        done_seq.update(sq for sq, f in zip(layouts[slot], flags) if f)
                        live = [r for r, sq in enumerate(row_seq) if sq not in done_seq]
                        target = compact_size(len(live))
                        if live and target < len(row_seq):
                            fill = [r for r, sq in enumerate(row_seq) if sq in done_seq][:target - len(live)]
                            keep = np.ascontiguousarray(sorted(live + fill), dtype=np.int32)
                            _lib.check(lib.ctts_gpt_compact(h, keep.ctypes.data_as(C.c_void_p), int(keep.size), st), "compact")
                            row_seq = [row_seq[r] for r in keep.tolist()]
                            self.compactions.append((launched, len(row_seq)))
                prev = steps.value
                _lib.check(lib.ctts_gpt_progress(h, C.byref(steps), C.byref(alld), st), "progress")
                used_draws += steps.value - prev
                tick("final_progress")
            self._restore_rng(rng_states, used_draws)
            self.saturations = self._report_saturations(h, st, "generate()")
            out = self._outputs(ids, hid, end_idx, infer_text)
            tick("outputs")
            yield out

    # -- continuous batching: more utterances than decode rows (no counterpart in the reference) ------------------------------------
    @torch.no_grad()
    def generate_many(self, emb: torch.Tensor, inputs_ids: torch.Tensor, temperature: torch.Tensor, eos_token: Union[int, torch.Tensor],
                      attention_mask: Optional[torch.Tensor] = None, max_new_token=2048, min_new_token=0, logits_warpers=[],
                      logits_processors=[], return_hidden=False, ensure_non_empty=True, context=None, seed: Optional[int] = None,
                      max_restarts: int = 64, utt_ids=None, max_new_tokens_per_row=None, rows: Optional[int] = None, admit_min: Optional[int] = None,
                      on_done=None, infer_text: bool = False, adapter_slots=None) -> GenerationOutputs:
        """N utterances (left-padded prompts emb[N,T,H], like generate()) through `rows` <= max_batch decode rows: whenever utterances
        finish, queued ones take over their rows (ctts_gpt_admit) instead of the whole slice waiting for its slowest row as the reference's
        slices of 4 do (pipeline:391-397, gpt.py:527-546); once the queue is empty finished rows are compacted away (ctts_gpt_compact).
        Device noise only: an utterance's noise stream is keyed by (seed, its utterance id, its own step, its regenerate attempt), its step
        counter / token limit / outputs are its own, so it produces what generate() produces for it in any slice.  ensure_non_empty
        (gpt.py:496-525) acts per utterance: one whose first token is EOS is admitted again with its next attempt, up to `max_restarts`.
        `on_done(list_of_indices)` is called (on the host, while decoding continues) as utterances complete.
        `adapter_slots` = the resident adapter slot (load_adapter) of every utterance or -1 / None: per-utterance LoRA under row re-use (an admitted
        utterance brings its own adapter, ctts_gpt_admit_adapters).
        Returns one GenerationOutputs for all N utterances, in input order."""
        gen = self.generate_many_iter(emb, inputs_ids, temperature, eos_token, attention_mask=attention_mask, max_new_token=max_new_token,
                                      min_new_token=min_new_token, logits_warpers=logits_warpers, logits_processors=logits_processors,
                                      return_hidden=return_hidden, ensure_non_empty=ensure_non_empty, context=context, seed=seed, max_restarts=max_restarts,
                                      utt_ids=utt_ids, max_new_tokens_per_row=max_new_tokens_per_row, rows=rows, admit_min=admit_min, infer_text=infer_text,
                                      adapter_slots=adapter_slots)
        try:
            while True:
                ev = next(gen)
                if on_done is not None:
                    on_done([u for u, _, _ in ev])
        except StopIteration as stop:
            return stop.value

    @torch.no_grad()
    def generate_many_iter(self, emb, inputs_ids, temperature, eos_token, attention_mask=None, max_new_token=2048, min_new_token=0, logits_warpers=[],
                           logits_processors=[], return_hidden=False, ensure_non_empty=True, context=None, seed=None, max_restarts: int = 64,
                           utt_ids=None, max_new_tokens_per_row=None, rows=None, admit_min=None, infer_text: bool = False, progress: bool = False,
                           adapter_slots=None):
        """generate_many as a generator: yields [(utterance index, ids [n,4] long, hiddens [n,768] or None)] for the utterances that completed
        since the last yield -- while the rest keeps decoding (what was yielded is final: its rows were written before the report that showed
        the utterance finished) -- and returns (StopIteration.value) the GenerationOutputs of all N utterances.
        `infer_text=True`: the refine-text pass (gpt.py infer_text: the 21178-way text head, one id per step) through the same row re-use; ids are [n].
        `progress=True` (streaming): between the completion lists the generator also yields ("progress", [(utterance index, tokens so far n, ids[:n],
        hiddens[:n] or None)]) for the utterances still decoding -- the first n tokens of an utterance are final once reported."""
        if not self._finalized:
            raise _lib.HipBackendError("weights not loaded")
        if self._busy_token.owner is not None:
            raise _lib.HipBackendError("GPT.generate is already running on this engine (or on an engine sharing its KV cache)")
        self._busy_token.owner = self
        try:
            return (yield from self._generate_many(emb, inputs_ids, temperature, eos_token, attention_mask, int(max_new_token), min_new_token, logits_warpers,
                                                   logits_processors, return_hidden, ensure_non_empty, context or Context(), seed, max_restarts, utt_ids,
                                                   max_new_tokens_per_row, rows, admit_min, bool(infer_text), bool(progress), adapter_slots))
        finally:
            if adapter_slots is not None:
                self.set_row_adapters(None)
            self._busy_token.owner = None

    def _generate_many(self, emb, inputs_ids, temperature, eos_token, attention_mask, max_new_token, min_new_token, logits_warpers, logits_processors,
                       return_hidden, ensure_non_empty, context, seed, max_restarts, utt_ids, row_limits, rows, admit_min, infer_text=False, progress=False,
                       adapter_slots=None):
        lib, h, dev = self._lib, self._h, self.device
        N, T = int(inputs_ids.shape[0]), int(inputs_ids.shape[1])
        H, NVQ = self.model_dim, self.num_vq
        R = min(N, int(rows) if rows else self.max_batch, self.max_batch)
        sc = sampler_cfg_from_objects(temperature, int(eos_token), max_new_token, min_new_token, logits_warpers, logits_processors, NVQ, infer_text=infer_text)
        if seed is None:
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        mask = torch.ones(N, T, dtype=torch.int32, device=dev) if attention_mask is None else attention_mask.to(dev).to(torch.int32).contiguous()
        emb = emb.to(dev, dtype=torch.float32).contiguous()
        lens = mask.sum(1).cpu().tolist()
        if max(lens) + max_new_token > self.max_seq:
            # (every utterance is admitted with its own prompt, trimmed of padding: the longest one decides)
            raise _lib.HipBackendError(f"generate_many: prompt of {max(lens)} tokens + max_new_token={max_new_token} exceed max_seq_len={self.max_seq}")
        uids = [int(u) for u in utt_ids] if utt_ids is not None else list(range(N))
        lims = [min(max(int(v), 1), max_new_token) for v in row_limits] if row_limits is not None else [max_new_token] * N
        assert len(uids) == N and len(lims) == N
        slots = None
        if adapter_slots is not None:
            slots = [-1 if a is None or int(a) < 0 else int(a) for a in adapter_slots]
            if len(slots) != N or infer_text:
                raise _lib.HipBackendError(f"adapter_slots: {len(slots)} entries for {N} utterances (code mode only)")
        ids = torch.empty(N, max_new_token, NVQ, dtype=torch.int32, device=dev)
        hid = torch.empty(N, max_new_token, H, dtype=torch.float32, device=dev) if return_hidden else None
        finish = torch.zeros(N, dtype=torch.int32, device=dev)
        end_idx = torch.zeros(N, dtype=torch.int32, device=dev)
        admit_min = max(1, int(admit_min) if admit_min else R // 8)
        chunk = max(4, min(self.chunk_steps, self.compact_chunk))
        self.compactions, self.admissions = [], []

        def prompts_of(idx):
            """left-padded prompts of the utterances `idx`, trimmed to the longest of them"""
            Ta = max(1, max(lens[u] for u in idx))
            ii = torch.as_tensor(idx, dtype=torch.long, device=dev)
            return Ta, emb.index_select(0, ii)[:, T - Ta:].contiguous(), mask.index_select(0, ii)[:, T - Ta:].contiguous()

        with torch.cuda.device(dev):
            st = self._stream()
            # Order of service: longest expected utterance first (LPT list scheduling) -- the request ends when its LAST row does, and a long utterance admitted late
            # decodes alone on a 32-row engine (the 256-utterance request: 2905 steps launched for 2518 ideal ones in arrival order).  The expectation is the row's own
            # token limit where the caller gave one, else the prompt length (longer text, longer audio).  Results do not depend on the order: every output is indexed
            # by utterance and the device noise is keyed by the utterance id.  Default: arrival order ("fifo"): a streaming caller wants its first utterance first.
            if getattr(self, "schedule", "fifo") == "longest_first":      # (ChatTTSPlusPipeline's throughput mode orders the request itself; direct callers opt in)
                order = sorted(range(N), key=(lambda u: (-lims[u], u)) if row_limits is not None else (lambda u: (-lens[u], u)))
            else:
                order = list(range(N))
            first = order[:R]
            Ta, emb_a, mask_a = prompts_of(first)
            uid_arr = np.ascontiguousarray([uids[u] for u in first], dtype=np.uint64)
            lim_arr = np.ascontiguousarray([lims[u] for u in first], dtype=np.int32)
            io = _lib.GenIO(ids=ids.data_ptr(), hiddens=hid.data_ptr() if hid is not None else None, finish=finish.data_ptr(),
                            end_idx=end_idx.data_ptr(), noise=None, n_draws=max_new_token, seed=int(seed), utt_ids=uid_arr.ctypes.data,
                            row_limits=lim_arr.ctypes.data)
            if slots is not None:
                self.set_row_adapters([slots[u] for u in first])
            _lib.check(lib.ctts_gpt_begin(h, R, Ta, mask_a.data_ptr(), C.byref(sc), C.byref(io), st), "begin")
            _lib.check(lib.ctts_gpt_prefill(h, emb_a.data_ptr(), st), "prefill")
            _lib.check(lib.ctts_gpt_sample(h, st), "sample")
            queue = [(u, 0) for u in order[R:]]                    # (utterance, regenerate attempt)
            book = RowBook()
            for r in range(R):
                book.seat(r, first[r])
            n_done, since_free = 0, 0
            pins = [torch.zeros(2 * R, dtype=torch.int32).pin_memory() for _ in range(2)]
            evs = [torch.cuda.Event() for _ in range(2)]
            layouts = [None, None]
            pending, n_chunks, launched = [], 0, 1
            while n_done < N and not context.get():
                while len(pending) < 2:
                    _lib.check(lib.ctts_gpt_decode(h, chunk, 1 if self.use_graph else 0, st), "decode")
                    launched += chunk
                    slot = n_chunks % 2
                    n_chunks += 1
                    _lib.check(lib.ctts_gpt_rows_enqueue(h, pins[slot].data_ptr(), st), "rows_enqueue")
                    layouts[slot] = book.layout()
                    evs[slot].record(torch.cuda.current_stream(dev))
                    pending.append(slot)
                slot = pending.pop(0)
                evs[slot].synchronize()
                lay = layouts[slot]
                states = pins[slot][:2 * len(lay)].view(-1, 2).tolist()
                if progress:
                    live = [(book.tickets[tk][0], int(end)) for tk, (fin, end) in zip(lay, states) if tk is not None and tk in book.tickets and not fin and end > 0]
                    if live:
                        yield ("progress", [(u, n, ids[u, :n, 0].to(torch.long) if infer_text else ids[u, :n].to(torch.long), hid[u, :n] if hid is not None else None)
                                            for u, n in live])
                finished_now, again = book.report(lay, states, ensure_non_empty, max_restarts)
                queue = again + queue                              # first token was EOS (gpt.py:496-525): next noise attempt, ahead of the queue
                n_done += len(finished_now)
                if finished_now:
                    yield [(u, ids[u, :n, 0].to(torch.long) if infer_text else ids[u, :n].to(torch.long), hid[u, :n] if hid is not None else None) for u, n in finished_now]
                free = book.free_rows()
                since_free = since_free + 1 if free else 0
                if queue and free and (len(free) >= min(admit_min, len(queue)) or since_free >= 4 or len(free) == len(book.row_tk)):
                    k = min(len(free), len(queue))
                    take, queue = queue[:k], queue[k:]
                    idx = [u for u, _ in take]
                    Ta, emb_a, mask_a = prompts_of(idx)
                    rows_arr = np.ascontiguousarray(free[:k], dtype=np.int32)
                    uid_arr = np.ascontiguousarray([uids[u] for u in idx], dtype=np.uint64)
                    lim_arr = np.ascontiguousarray([lims[u] for u in idx], dtype=np.int32)
                    out_arr = np.ascontiguousarray(idx, dtype=np.int32)
                    att_arr = np.ascontiguousarray([a for _, a in take], dtype=np.int32)
                    if slots is not None:
                        sl_arr = np.ascontiguousarray([slots[u] for u in idx], dtype=np.int32)
                        _lib.check(lib.ctts_gpt_admit_adapters(h, k, rows_arr.ctypes.data_as(C.c_void_p), sl_arr.ctypes.data_as(C.c_void_p), st), "admit_adapters")
                    _lib.check(lib.ctts_gpt_admit(h, k, rows_arr.ctypes.data_as(C.c_void_p), Ta, mask_a.data_ptr(), emb_a.data_ptr(),
                                                  uid_arr.ctypes.data_as(C.c_void_p), lim_arr.ctypes.data_as(C.c_void_p),
                                                  out_arr.ctypes.data_as(C.c_void_p), att_arr.ctypes.data_as(C.c_void_p), st), "admit")
                    for r, (u, a) in zip(free[:k], take):
                        book.seat(r, u, a)
                    self.admissions.append((launched, k))
                    since_free = 0
                elif not queue and self.compact and len(book.row_tk) >= 2:
                    live = book.live_rows()
                    target = compact_size(len(live))
                    if live and target < len(book.row_tk):
                        keep = sorted(live + book.free_rows()[:target - len(live)])
                        karr = np.ascontiguousarray(keep, dtype=np.int32)
                        _lib.check(lib.ctts_gpt_compact(h, karr.ctypes.data_as(C.c_void_p), int(karr.size), st), "compact")
                        book.compact(keep)
                        self.compactions.append((launched, len(book.row_tk)))
            torch.cuda.current_stream(dev).synchronize()
            # the device give-up word (a persistent launch's bounded wait ran out, or a projection tile never got its low-rank term): every later launch of the
            # request was a no-op and the sampler kept drawing from stale rows -- an error, not audio (ctts_gpt_progress raises it, as in generate())
            steps, alld = C.c_int32(0), C.c_int32(0)
            _lib.check(lib.ctts_gpt_progress(h, C.byref(steps), C.byref(alld), st), "progress")
            self.saturations = self._report_saturations(h, st, "generate_many()")
            return self._outputs(ids, hid, end_idx, infer_text)

    def _staging(self, rows: int, V: int, cap: int):
        """Pinned [cap, rows, V] staging slots for the torch-generator noise, kept across calls (pinning is expensive)."""
        key = (rows, V, cap)
        if getattr(self, "_stage_key", None) != key:
            self._stage_key = key
            self._stage_bufs = [[torch.empty(cap, rows, V, dtype=torch.float32).pin_memory(), None] for _ in range(3)]
            self._stage_next = 0
        return self._stage_bufs

    @staticmethod
    def _restore_rng(states, used):
        """Leave torch's CPU generator where the reference would have left it: after `used` multinomial draws."""
        if states and used < len(states):
            torch.random.set_rng_state(states[used])

    def _outputs(self, ids, hid, end_idx, infer_text=False) -> GenerationOutputs:
        n = end_idx.cpu().tolist()
        out_ids = [ids[b, :n[b]].to(torch.long) for b in range(len(n))]                # gpt.py:295-297
        if infer_text:
            out_ids = [i[:, 0] for i in out_ids]                                        # gpt.py:298-299
        out_h = [hid[b, :n[b]] for b in range(len(n))] if hid is not None else []     # gpt.py:301-305
        return GenerationOutputs(ids=out_ids, attentions=[], hiddens=out_h)

    # -- test hooks ------------------------------------------------------------------------------
    def last_logits(self, B: int) -> torch.Tensor:
        out = torch.empty(B, self.num_vq, self.num_audio_tokens, dtype=torch.float32, device=self.device)
        _lib.check(self._lib.ctts_gpt_logits(self._h, out.data_ptr(), self._stream()), "logits")
        return out

    def step_bytes(self, B: int, mean_ctx: float) -> float:
        return float(self._lib.ctts_gpt_step_bytes(self._h, B, float(mean_ctx)))
