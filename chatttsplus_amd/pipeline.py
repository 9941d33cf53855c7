"""ChatTTSPlusPipeline with `infer_type: "hip"` -- the drop-in surface of the reference's
chattts_plus/pipelines/chattts_plus_pipeline.py for the hot path:

    pipe = ChatTTSPlusPipeline(cfg, device=torch.device("cuda"))
    for wavs in pipe.infer(text, params_infer_code=InferCodeParams(...), skip_refine_text=True, ...): ...

What is kept: constructor kwargs, YAML layout (MODELS.<key>.{name,infer_type,kwargs}), `infer()` /
`_infer()` / `_infer_code()` / `_decode_to_wavs()` / speaker helpers, the generator-of-wav-lists result,
InferCodeParams / RefineTextParams, and the CPU text front-end in front of the path (text_frontend.py: sentence splitting,
number spelling, short-sentence merging, the Normalizer -- replaceable callables, defaults as in the reference).  The Chinese
number reader (zh_normalization) and nemo_text_processing are optional plug-ins.  There is no CPU fallback for the models.
"""
from __future__ import annotations

import dataclasses
import json
import logging
import os
from collections import OrderedDict
from dataclasses import dataclass
from typing import Callable, List, Optional, Union

import numpy as np
import torch

from . import _lib, codec, hip_models, text_frontend


@dataclass(repr=False, eq=False)
class RefineTextParams:                      # reference commons/utils.py:12-22
    prompt: str = ""
    top_P: float = 0.7
    top_K: int = 20
    temperature: float = 0.7
    repetition_penalty: float = 1.0
    max_new_token: int = 384
    min_new_token: int = 0
    show_tqdm: bool = True
    ensure_non_empty: bool = True


@dataclass(repr=False, eq=False)
class InferCodeParams(RefineTextParams):     # reference commons/utils.py:25-36
    prompt: str = "[speed_5]"
    spk_emb: Optional[str] = None
    spk_smp: Optional[str] = None
    txt_smp: Optional[str] = None
    temperature: float = 0.3
    repetition_penalty: float = 1.05
    max_new_token: int = 2048
    stream_batch: int = 24
    stream_speed: int = 12000
    pass_first_n_batches: int = 2


class TorchSeedContext:                        # reference commons/utils.py:48-58: seed torch's CPU generator for one request, restore it afterwards
    def __init__(self, seed):
        self.seed, self.state = seed, None

    def __enter__(self):
        self.state = torch.random.get_rng_state()
        torch.manual_seed(self.seed)

    def __exit__(self, exc_type, exc, tb):
        torch.random.set_rng_state(self.state)


class _TopP:                                  # scalar carriers with the attribute names the HF warpers expose
    def __init__(self, top_p, min_tokens_to_keep):
        self.top_p, self.min_tokens_to_keep = float(top_p), int(min_tokens_to_keep)


class _TopK:
    def __init__(self, top_k, min_tokens_to_keep):
        self.top_k, self.min_tokens_to_keep = max(int(top_k), int(min_tokens_to_keep)), int(min_tokens_to_keep)


class _RepPenalty:
    def __init__(self, penalty, max_input_ids, past_window):
        if not isinstance(penalty, float) or not (penalty > 0):
            raise ValueError(f"`penalty` has to be a strictly positive float, but is {penalty}")   # processors.py:9-12
        self.penalty, self.max_input_ids, self.past_window = penalty, int(max_input_ids), int(past_window)


def gen_logits(num_code: int, top_P=0.7, top_K=20, repetition_penalty=1.0):
    """models/processors.py:37-57 -- same construction, scalar carriers instead of HF objects (the hip GPT
    also accepts the real transformers / reference objects: it only reads their attributes)."""
    warpers, processors = [], []
    if top_P is not None:
        warpers.append(_TopP(top_P, 3))
    if top_K is not None:
        warpers.append(_TopK(top_K, 3))
    if repetition_penalty is not None and repetition_penalty != 1:
        processors.append(_RepPenalty(repetition_penalty, num_code, 16))
    return warpers, processors


def _get(cfg, key, default=None):
    try:
        return cfg[key]
    except Exception:
        return getattr(cfg, key, default)


def load_config(path: str) -> dict:
    import yaml
    with open(path) as f:
        return yaml.safe_load(f)


def load_lora_adapter(path: str):
    """peft adapter directory -> [(layer, target, A[r,in], B[out,r], scale)] (pipeline:420-432; train config
    configs/train/train_voice_clone_lora.yaml:72-80: r=8, alpha=16 on q/k/v/o).  Reads adapter_config.json +
    adapter_model.safetensors directly; peft is not needed."""
    from safetensors.numpy import load_file
    with open(os.path.join(path, "adapter_config.json")) as f:
        ac = json.load(f)
    scale = float(ac["lora_alpha"]) / float(ac["r"])
    sd = load_file(os.path.join(path, "adapter_model.safetensors"))
    out = []
    for k, A in sd.items():
        if ".lora_A" not in k:
            continue
        parts = k.split(".")
        li = parts.index("layers")
        layer, target = int(parts[li + 1]), parts[li + 3]
        Bk = k.replace("lora_A", "lora_B")
        out.append((layer, target, np.asarray(A, dtype=np.float32), np.asarray(sd[Bk], dtype=np.float32), scale))
    return out


def merge_short_sentences(pieces: List[str], min_len: int = 30) -> List[str]:
    """The merge step of the reference's text optimisation (pipeline:353-377), after whatever splitter produced `pieces`: sentences
    shorter than `min_len` characters are chained with " [uv_break] " until the chain exceeds `min_len`; a long sentence absorbs the
    chain in front of it; a left-over chain becomes its own utterance if long enough (or if nothing else exists), else it is
    appended to the last utterance.  CPU text work in front of the hot path -- kept because it decides the batch the path sees."""
    out: List[str] = []
    chain = ""
    for piece in pieces:
        if len(piece) < min_len:
            chain += f"{piece} [uv_break] "
            if len(chain) > min_len:
                out.append(chain)
                chain = ""
        else:
            out.append(chain + piece)
            chain = ""
    if len(chain) > min_len or not out:
        out.append(chain)
    elif chain:
        out[-1] += f" [uv_break] {chain}"
    return out


class ChatTTSPlusPipeline:
    def __init__(self, cfg, **kwargs):
        self.logger = logging.getLogger(self.__class__.__name__)
        self.cfg = cfg
        self.device = torch.device(kwargs.get("device", "cuda"))
        self.dtype = torch.float32                 # activations / outputs are fp32; weights per `weight_dtype`
        # CPU text front-end in front of the path (pipeline:147-155,353-388): `normalizer(text, do_text_normalization,
        # do_homophone_replacement, lang)` and `text_splitter(lines) -> sentences`; both replaceable, `text_splitter=None` keeps the
        # input lines as they are.  Defaults: text_frontend.Normalizer over <checkpoint_dir>/homophones_map.json and text_frontend.split_text
        self.normalizer: Optional[Callable] = kwargs.get("normalizer")
        self.text_splitter: Optional[Callable] = kwargs.get("text_splitter", text_frontend.split_text)
        self.load_lora = False
        self._lora_models = OrderedDict()          # lora_path -> merged sibling engine, LRU-bounded (`lora_cache`, default 1)
        self._lora_cache = max(1, int(kwargs.get("lora_cache", 1)))
        self.load_models(**kwargs)

    @classmethod
    def from_components(cls, gpt, synth, tokenizer, device, normalizer=None, text_splitter=None):
        """A pipeline around engines that are already loaded (bench.py's sharded-request leg, services that build their engines themselves):
        the same infer() / infer_sharded() surface without the checkpoint-directory pass of load_models.  No zero-shot encoder, no use_decoder=False."""
        self = object.__new__(cls)
        self.logger = logging.getLogger(cls.__name__)
        self.cfg = None
        self.device = torch.device(device)
        self.dtype = torch.float32
        self.normalizer = normalizer if normalizer is not None else text_frontend.Normalizer(None)
        self.text_splitter = text_splitter
        self.load_lora = False
        self._lora_models = OrderedDict()
        self._lora_cache = 1
        self.models_dict = dict(gpt=gpt, tokenizer=tokenizer)
        self.synth = synth
        self.std = self.mean = None
        self.infer_type = "hip"
        return self

    # -- loading (pipeline:53-155) -----------------------------------------------------------------
    def load_models(self, **kwargs):
        self.models_dict = {}
        models = _get(self.cfg, "MODELS")
        coef = kwargs.get("coef", None)
        if coef is None:
            coef = codec.coef_to_string(torch.rand(100).numpy())            # pipeline:56-59
        self.dave_coef = coef
        ckpt_dir = kwargs.get("checkpoint_dir") or os.environ.get("CHATTTS_PLUS_CHECKPOINT_DIR", "checkpoints")
        self.infer_type = None
        synth = None
        for model_name in models:
            m = models[model_name]
            kw = dict(_get(m, "kwargs") or {})
            mp = kw.get("model_path")
            if mp and not os.path.isabs(mp):
                kw["model_path"] = os.path.join(ckpt_dir, mp.replace("checkpoints/", ""))   # pipeline:70-71
            itype = _get(m, "infer_type")
            if model_name == "tokenizer":
                tok = kwargs.get("tokenizer")
                if tok is None:
                    from .tokenizer import Tokenizer
                    tok = Tokenizer(**kw)
                self.models_dict[model_name] = tok
                continue
            if model_name == "dvae_encode":                 # zero-shot speaker prompt (pipeline:279-284): optional checkpoint
                if itype == "hip" and kw.get("model_path") and os.path.exists(kw["model_path"]):
                    # the same checkpoint's decoder + quantiser serve use_decoder=False (pipeline:292): built on first use (_codes_synth)
                    self._full_ckpt = dict(path=kw["model_path"], decoder_config=dict(kw.get("decoder_config") or {}), vq_config=dict(kw.get("vq_config") or {}))
                    path = kw.pop("model_path")
                    enc = hip_models.DVAEEncoder(device=str(self.device), **kw)
                    sd = dict(torch.load(path, weights_only=True, mmap=True))
                    if "coef" not in sd:
                        sd["coef"] = torch.from_numpy(codec.coef_from_string(coef)).view(1, -1, 1)
                    self.models_dict[model_name] = enc.load_state_dict(sd)
                continue
            if itype != "hip":
                raise _lib.HipBackendError(f"model {model_name}: infer_type={itype!r}; this pipeline serves infer_type 'hip' only")
            self.infer_type = self.infer_type or itype
            if model_name in ("dvae_decode", "vocos") and synth is None:
                dk = dict(_get(models["dvae_decode"], "kwargs"))
                vk = dict(_get(models["vocos"], "kwargs"))
                dcfg = dict(dk["decoder_config"]); dcfg["n_mels"] = 100
                vcfg = dict(vk["backbone_config"]); vcfg.update(vk["head_config"])
                synth = hip_models.Synth(dcfg, vcfg, max_frames=int(kwargs.get("max_frames", 2 * 2048 + 64)), device=self.device,
                                         max_batch=int(kwargs.get("vocoder_batch", 32)))
                self.synth = synth
            if model_name == "vocos":                       # pipeline:93-111
                model_ = hip_models.Vocos(synth)
                model_.load_state_dict(torch.load(kw["model_path"], weights_only=True, mmap=True))
                self._vocos_ckpt = dict(path=kw["model_path"], cfg=vcfg)
            elif model_name == "dvae_decode":
                kw["coef"] = coef
                path = kw.pop("model_path")
                model_ = hip_models.DVAE(synth=synth, **kw)
                sd = dict(torch.load(path, weights_only=True, mmap=True))
                if "coef" not in sd:                        # the buffer is persistent: a checkpoint value wins (dvae.py:222,241-247)
                    sd["coef"] = torch.from_numpy(codec.coef_from_string(coef)).view(1, -1, 1)
                model_.load_state_dict(sd)
            else:
                kw.setdefault("device", str(self.device))
                model_ = getattr(hip_models, _get(m, "name"))(**kw)          # pipeline:113-129 dispatch
            self.models_dict[model_name] = model_.eval().to(self.device)
        spk_stat_path = os.path.join(ckpt_dir, "asset/spk_stat.pt")
        if os.path.exists(spk_stat_path):                   # pipeline:131-145
            spk_stat = torch.load(spk_stat_path, weights_only=True, mmap=True).to(self.device, dtype=self.dtype)
            self.std, self.mean = spk_stat.chunk(2)
        else:
            self.std = self.mean = None
        if self.normalizer is None:                         # pipeline:147-155 (the reference downloads the map when it is missing)
            map_path = os.path.join(ckpt_dir, "homophones_map.json")
            if not os.path.exists(map_path):
                self.logger.warning("%s not found: homophone replacement is off", map_path)
                map_path = None
            self.normalizer = text_frontend.Normalizer(map_path)

    # -- speakers (pipeline:306-331) ---------------------------------------------------------------
    @torch.inference_mode()
    def sample_audio_speaker(self, wav) -> str:
        """pipeline:279-284: 24 kHz mono waveform -> base16384 audio-prompt string (`spk_smp`)."""
        enc = self.models_dict.get("dvae_encode")
        if enc is None:
            raise _lib.HipBackendError("zero-shot speaker needs the dvae_encode checkpoint (DVAE_full.pt) configured with infer_type 'hip'")
        wav = torch.as_tensor(wav, dtype=torch.float32)
        codes = enc(wav.view(1, -1).to(self.device), "encode")[0]
        return codec.encode_prompt(codes.cpu())

    def sample_random_speaker(self) -> str:
        return self._encode_spk_emb(self._sample_random_speaker())

    @staticmethod
    def _encode_spk_emb(spk_emb: torch.Tensor) -> str:
        return codec.encode_spk_emb(spk_emb)

    def _sample_random_speaker(self) -> torch.Tensor:
        if self.std is None:
            raise _lib.HipBackendError("spk_stat.pt not loaded: pass a speaker embedding explicitly")
        dim = self.std.shape[-1]
        return torch.randn(dim, device=self.std.device, dtype=self.std.dtype).mul_(self.std).add_(self.mean)

    # -- hot path callers --------------------------------------------------------------------------
    @torch.no_grad()
    def _infer_code(self, text, stream: bool, return_hidden: bool, params: InferCodeParams, gpt=None, **gen_kwargs):
        """pipeline:157-235 -- same argument plumbing; the GPT object is the hip backend.  `gen_kwargs` (noise, seed, utt_ids) ride through to
        GPT.generate: the device noise stream of an utterance is keyed by the request seed and its global utterance id."""
        gpt = gpt or self.models_dict["gpt"]
        tok = self.models_dict["tokenizer"]
        if not isinstance(text, list):
            text = [text]
        assert len(text), "text should not be empty"
        temperature = params.temperature if isinstance(params.temperature, list) else [params.temperature] * gpt.num_vq
        text = [t.replace("[Stts]", "").replace("[spk_emb]", "").replace("[empty_spk]", "").strip() for t in text]
        if params.prompt:
            text = [params.prompt + i for i in text]
        txt_smp = "" if params.txt_smp is None else params.txt_smp
        tag = "[spk_emb]" if params.spk_emb is not None else "[empty_spk]"
        text = [f"[Stts]{tag}{txt_smp}{i}[Ptts]" for i in text]
        input_ids, attention_mask, text_mask = tok.encode(text, gpt.num_vq, prompt_str=params.spk_smp, device=self.device)
        emb = gpt(input_ids, text_mask, spk_emb=params.spk_emb, spk_emb_ids=tok.spk_emb_ids)     # get_emb + apply_spk_emb, one launch
        num_code = int(gpt.emb_code[0].num_embeddings - 1)
        warpers, processors = gen_logits(num_code=num_code, top_P=params.top_P, top_K=params.top_K,
                                         repetition_penalty=params.repetition_penalty)
        if gen_kwargs.pop("continuous", False):
            # more utterances than decode rows: queued utterances take over rows as they free up (GPT.generate_many_iter: a generator of
            # completion events whose return value is the GenerationOutputs of all utterances)
            gen_kwargs.pop("noise", None)
            return gpt.generate_many_iter(emb, input_ids, temperature=torch.tensor(temperature), eos_token=num_code, attention_mask=attention_mask,
                                     max_new_token=params.max_new_token, min_new_token=params.min_new_token, logits_warpers=warpers,
                                     logits_processors=processors, return_hidden=return_hidden, ensure_non_empty=params.ensure_non_empty, **gen_kwargs)
        return gpt.generate(emb, input_ids, temperature=torch.tensor(temperature), eos_token=num_code, attention_mask=attention_mask,
                            max_new_token=params.max_new_token, min_new_token=params.min_new_token, logits_warpers=warpers,
                            logits_processors=processors, infer_text=False, return_hidden=return_hidden, stream=stream,
                            show_tqdm=params.show_tqdm, ensure_non_empty=params.ensure_non_empty, stream_batch=params.stream_batch, **gen_kwargs)

    @torch.no_grad()
    def _refine_text(self, text, params: RefineTextParams, continuous_rows: int = 0, seed=None, utt_ids=None):
        """pipeline:237-277: "[Sbreak]{text}[Pbreak]{prompt}" -> GPT.generate(infer_text=True) on the 21178-way text head.
        `continuous_rows` > 0 (no counterpart in the reference, which refines slice by slice): all sentences of the request through that many decode
        rows with row re-use (GPT.generate_many, device noise keyed by utterance id: a sentence's refined text does not depend on its neighbours)."""
        gpt, tok = self.models_dict["gpt"], self.models_dict["tokenizer"]
        text = [f"[Sbreak]{i}[Pbreak]{params.prompt}" for i in text]
        input_ids, attention_mask, text_mask = tok.encode(text, gpt.num_vq, device=self.device)
        warpers, processors = gen_logits(num_code=tok.len, top_P=params.top_P, top_K=params.top_K, repetition_penalty=params.repetition_penalty)
        emb = gpt(input_ids, text_mask)
        if continuous_rows > 0:
            return gpt.generate_many(emb, input_ids, temperature=torch.tensor([params.temperature]), eos_token=tok.eos_token, attention_mask=attention_mask,
                                     max_new_token=params.max_new_token, min_new_token=params.min_new_token, logits_warpers=warpers, logits_processors=processors,
                                     ensure_non_empty=params.ensure_non_empty, seed=seed, utt_ids=utt_ids, rows=continuous_rows, infer_text=True)
        return next(gpt.generate(emb, input_ids, temperature=torch.tensor([params.temperature]), eos_token=tok.eos_token,
                                 attention_mask=attention_mask, max_new_token=params.max_new_token, min_new_token=params.min_new_token,
                                 logits_warpers=warpers, logits_processors=processors, infer_text=True, stream=False,
                                 show_tqdm=params.show_tqdm, ensure_non_empty=params.ensure_non_empty))

    def _codes_synth(self):
        """The "decode codes" model of use_decoder=False (pipeline:292: decoder = models_dict["dvae_encode"], i.e. the DVAE_full checkpoint
        run in decode mode on the generated ids, dvae.py:272-291): its decoder stack + the quantiser's project_out + a second copy of the
        Vocos weights in one native handle, created on first use."""
        if getattr(self, "synth_codes", None) is not None:
            return self.synth_codes
        full, voc = getattr(self, "_full_ckpt", None), getattr(self, "_vocos_ckpt", None)
        if full is None or voc is None:
            raise _lib.HipBackendError("use_decoder=False needs the dvae_encode checkpoint (DVAE_full.pt, infer_type 'hip') and vocos configured")
        dcfg = dict(full["decoder_config"]); dcfg["n_mels"] = 100
        sc = hip_models.Synth(dcfg, voc["cfg"], max_frames=int(self.synth.cfg.max_frames), device=self.device, max_batch=int(self.synth.max_batch),
                              vq_cfg=full["vq_config"])
        sd = dict(torch.load(full["path"], weights_only=True, mmap=True))
        if "coef" not in sd:
            sd["coef"] = torch.from_numpy(codec.coef_from_string(self.dave_coef)).view(1, -1, 1)
        sc.load("dvae.", sd)
        sc.load("vocos.", torch.load(voc["path"], weights_only=True, mmap=True))
        self.synth_codes = sc
        return sc

    @torch.inference_mode()
    def _decode_to_wavs(self, result_list, use_decoder: bool = True):
        """pipeline:286-305: per utterance hidden[n,768] -> DVAE -> mel[1,100,2n] -> Vocos -> wav[256(2n-1)]; with use_decoder=False the
        inputs are the generated ids [n,4] and the decoder is the DVAE_full model (pipeline:292,435-439)."""
        if not use_decoder:
            return self._codes_synth().decode_batch(list(result_list))
        if len(result_list) >= 1 and getattr(self, "synth", None) is not None:
            return self.synth.decode_batch(list(result_list))           # one launch sequence for the batch (ctts_synth_batch)
        wavs = []
        decoder, vocos = self.models_dict["dvae_decode"], self.models_dict["vocos"]
        for h in result_list:
            if h.shape[0] == 0:
                wavs.append(torch.zeros(0, device=self.device))
                continue
            mel = decoder(h.permute(1, 0)[None])
            wavs.append(vocos.decode(mel)[0])
        return wavs

    def _adapter_slots(self, gpt, paths) -> List[int]:
        """Slot per utterance for `paths` (adapter directory or None each); adapters are loaded into the engine's resident slots on first
        use and evicted least-recently-used first (at most _lib.MAX_ADAPTERS distinct adapters per slice)."""
        if not hasattr(self, "_slot_of_path"):
            self._slot_of_path = OrderedDict()
        need = [p for p in dict.fromkeys(paths) if p]
        if len(need) > _lib.MAX_ADAPTERS:
            raise _lib.HipBackendError(f"{len(need)} distinct adapters in one slice; the engine holds {_lib.MAX_ADAPTERS}: lower slice_size")
        for p in need:
            if p in self._slot_of_path:
                self._slot_of_path.move_to_end(p)
                continue
            if len(self._slot_of_path) >= _lib.MAX_ADAPTERS:
                victim = next(q for q in self._slot_of_path if q not in need)
                slot = self._slot_of_path.pop(victim)
            else:
                slot = next(i for i in range(_lib.MAX_ADAPTERS) if i not in self._slot_of_path.values())
            gpt.load_adapter(slot, load_lora_adapter(p))
            self._slot_of_path[p] = slot
        return [(-1 if not p else self._slot_of_path[p]) for p in paths]

    def _gpt_for_lora(self, lora_path: Optional[str]):
        """pipeline:420-434,465-470: the reference merges the adapter into a copy of the Llama for the call and restores
        `gpt_org` afterwards, i.e. it holds ONE merged model at a time.  Here a merged sibling engine (GPT.with_lora: its own
        packed weights, the base engine's KV cache shared) is cached per adapter path in a small LRU (`lora_cache`, default 1);
        evicted engines are destroyed, so a service cycling through adapters does not grow HBM."""
        if not lora_path:
            return self.models_dict["gpt"]
        if lora_path in self._lora_models:
            self._lora_models.move_to_end(lora_path)
            return self._lora_models[lora_path]
        while len(self._lora_models) >= self._lora_cache:
            victim = next(iter(self._lora_models))
            if self._lora_models[victim].busy:
                # a partially consumed infer(stream=True) generator still owns the KV cache its siblings share and, possibly, this
                # engine's native handle: destroying it now would be a use-after-free when that generator resumes
                raise _lib.HipBackendError(f"cannot load adapter {lora_path!r}: a generator of an earlier infer() call is still live "
                                           f"(adapter {victim!r} would have to be evicted); exhaust or close it first")
            _, old = self._lora_models.popitem(last=False)
            old.close()
        base = self.models_dict["gpt"]
        self._lora_models[lora_path] = base.with_lora(load_lora_adapter(lora_path))
        return self._lora_models[lora_path]

    def _infer(self, text_in, stream=False, lang=None, skip_refine_text=False, refine_text_only=False, use_decoder=True,
               do_text_normalization=True, do_text_optimization=True, do_homophone_replacement=True,
               params_refine_text=RefineTextParams(), params_infer_code=InferCodeParams(), **kwargs):
        if not isinstance(text_in, list):
            text_in = [text_in]
        if do_text_optimization and self.text_splitter is not None:
            # pipeline:353-377: split on newlines, hand the lines to the (pluggable) sentence splitter, merge short sentences
            lines = [t.strip() for text_ in text_in for t in text_.split("\n") if t.strip()]
            text_in = merge_short_sentences(self.text_splitter(lines))
        text_in = [self.normalizer(t, do_text_normalization, do_homophone_replacement, lang) for t in text_in]
        slice_size = int(kwargs.get("slice_size", self.models_dict["gpt"].max_batch))     # reference: 4 (pipeline:391)
        gpt = self._gpt_for_lora(kwargs.get("lora_path"))
        # per-utterance adapters (the reference can only merge ONE adapter for a whole call): `lora_paths` = one adapter
        # directory (or None) per input text; adapters stay resident in up to 8 slots of the base engine, each row selects its own
        lora_paths = kwargs.get("lora_paths")
        if lora_paths is not None:
            if kwargs.get("lora_path"):
                raise _lib.HipBackendError("lora_path (one merged adapter) and lora_paths (one adapter per utterance) are exclusive")
            if len(lora_paths) != len(text_in):
                raise _lib.HipBackendError(f"lora_paths: {len(lora_paths)} entries for {len(text_in)} utterances (after text splitting)")
        tok = self.models_dict["tokenizer"]
        # Device noise is keyed by (request seed, global utterance id): every slice of the request uses the SAME seed and each utterance its
        # own id, so the result does not depend on slice_size or on which rank serves the utterance (infer_sharded).  `noise="auto"` keeps the
        # reference-compatible torch-generator noise for slices of <= 4 utterances.
        noise_mode = kwargs.get("noise", "auto")
        utt_ids = kwargs.get("utt_ids")
        if utt_ids is None:
            utt_ids = list(range(len(text_in)))
        elif len(utt_ids) != len(text_in):
            raise _lib.HipBackendError(f"utt_ids: {len(utt_ids)} entries for {len(text_in)} utterances (after text splitting)")
        noise_seed = kwargs.get("noise_seed")      # None: drawn from torch's CPU generator when the first slice that uses device noise starts
        # optional per-utterance token limits (<= params_infer_code.max_new_token): ctts_gen_io.row_limits
        utt_limits = kwargs.get("max_new_tokens_per_utterance")
        if utt_limits is not None and len(utt_limits) != len(text_in):
            raise _lib.HipBackendError(f"max_new_tokens_per_utterance: {len(utt_limits)} entries for {len(text_in)} utterances (after text splitting)")
        # `continuous=True` (no counterpart in the reference): the request's utterances are NOT cut into slices that each wait for their slowest
        # row (pipeline:391-397); slice_size decode rows are kept busy -- queued utterances take over the rows of finished ones
        # (GPT.generate_many_iter, ctts_gpt_admit).  Device noise keyed by utterance id: every utterance gets the waveform the sliced path gives it.
        # Lists of waveforms are yielded in input order as prefixes of the request complete.  Not for caller-supplied noise.
        if kwargs.get("continuous") and len(text_in) > slice_size:
            if noise_mode not in ("auto", "device"):
                raise _lib.HipBackendError("continuous=True works with device noise")
            adapter_slots = None
            if lora_paths is not None:
                # per-utterance adapters under row re-use: an admitted utterance brings its own adapter (ctts_gpt_admit_adapters).  All adapters of the request
                # must be resident at once (the engine holds _lib.MAX_ADAPTERS)
                if len({p for p in lora_paths if p}) > _lib.MAX_ADAPTERS:
                    raise _lib.HipBackendError(f"continuous=True: {len({p for p in lora_paths if p})} distinct adapters in the request; the engine holds "
                                               f"{_lib.MAX_ADAPTERS} (use slices: continuous=False)")
                adapter_slots = self._adapter_slots(gpt, lora_paths)
            if noise_seed is None:
                noise_seed = int(torch.randint(0, 2 ** 62, (1,)).item())
            texts_all = list(text_in)
            if not skip_refine_text:
                # pipeline:399-411 for the whole request at once: the refine-text pass keeps `slice_size` rows busy too (its noise is keyed by utterance id on
                # stream 4, so a sentence is refined to the same text whatever shares the batch with it)
                refined = self._refine_text(texts_all, params_refine_text, continuous_rows=slice_size, seed=noise_seed, utt_ids=utt_ids)
                texts_all = tok.decode([i[i.less(tok.break_0_ids)] for i in refined.ids])
                if refine_text_only:
                    yield texts_all
                    return
            texts_all = [t if t.strip().endswith("[uv_break]") else t + " [uv_break]" for t in texts_all]   # pipeline:414-416
            # continuous=True: utterances are admitted in input order and their waveforms are yielded IN ORDER as soon as a prefix of the request
            # is complete (the first list as early as possible, later ones in groups of >= 8 so that the vocoder runs batched) while the rest keeps
            # decoding.  continuous="throughput": longest texts first (longest-processing-time order: the last rows to finish are short
            # utterances), one list at the end.  An utterance's result does not depend on the order -- its noise is keyed by its id.
            ordered = kwargs.get("continuous") != "throughput"
            n_all = len(texts_all)
            # (round 6: by the utterance's own token limit where the caller gave one -- the 256-utterance request of bench.py: 2625 decode steps instead of 2905 for
            #  2518 ideal ones, 30.9 k -> 35.8 k useful tokens/s; `throughput_order = "input"` keeps arrival order)
            if ordered or getattr(self, "throughput_order", "longest_first") != "longest_first":
                order = list(range(n_all))
            elif utt_limits is not None:
                order = sorted(range(n_all), key=lambda i: (-int(utt_limits[i]), -len(texts_all[i]), i))
            else:
                order = sorted(range(n_all), key=lambda i: (-len(texts_all[i]), i))
            pic = params_infer_code
            if torch.is_tensor(pic.spk_emb) and pic.spk_emb.dim() == 2 and pic.spk_emb.shape[0] == n_all:      # one speaker row per utterance
                pic = dataclasses.replace(pic, spk_emb=pic.spk_emb[torch.as_tensor(order, device=pic.spk_emb.device)])
            events = self._infer_code([texts_all[i] for i in order], False, use_decoder, pic, gpt=gpt, continuous=True, seed=noise_seed,
                                      utt_ids=[utt_ids[i] for i in order], rows=slice_size,
                                      max_new_tokens_per_row=[utt_limits[i] for i in order] if utt_limits is not None else None, progress=bool(stream),
                                      **({"adapter_slots": [adapter_slots[i] for i in order]} if adapter_slots is not None else {}))
            if stream:
                # stream=True with row re-use (no counterpart in the reference, whose stream branch serves one slice, pipeline:440-463): every yield is a list of
                # (utterance index, sample window) -- the next [emitted, b) samples of that utterance's prefix waveform, vocoded from the tokens inside the window's
                # receptive field only (Synth.decode_window) -- as soon as `stream_batch` new tokens of it exist; an utterance's last window comes with its completion
                syn = self.synth if use_decoder else self._codes_synth()
                emitted, last_n = {}, {}

                def windows(items, final):
                    out = []
                    for k, n, ids_k, hid_k in items:
                        u = order[k]
                        src = hid_k if use_decoder else ids_k
                        total = 256 * (2 * int(n) - 1) if n > 0 else 0
                        s0 = emitted.get(u, 0)
                        if total > s0 and (final or n - last_n.get(u, 0) >= pic.stream_batch):
                            b = total if final else min(s0 + pic.stream_speed, total)
                            out.append((u, syn.decode_window([src], [s0], [b])[0]))
                            emitted[u], last_n[u] = b, int(n)
                    return out

                for ev in events:
                    if isinstance(ev, tuple) and ev[0] == "progress":
                        got = windows(ev[1], False)
                    else:
                        got = windows([(k, int(ids_k.shape[0]), ids_k, hid_k) for k, ids_k, hid_k in ev], True)
                    if got:
                        yield got
                return
            ready, next_i, first = {}, 0, True
            ids_sink = kwargs.get("_ids_sink")
            # continuous="throughput" + overlap_vocoder=True (round 5; OPT-IN since round 6): the vocoder does not wait for the last utterance.  Finished utterances are
            # vocoded in batches of `vocoder_chunk` on a SIDE stream while the decode rows keep stepping on the caller's stream.  Measured: no gain (256 ragged utterances
            # 2682 vs 2665-2687 ms: the vocoder's GEMMs fill the chip and the decode chain's small kernels queue behind them), and a <= 5-row tail's persistent launch
            # -- which needs all 256 workgroups resident -- then shares the chip with vocoder kernels.  Its waits are bounded and the soak test
            # (tests/test_gpu_pipeline.py::test_vocoder_overlap_soak) drives exactly that overlap, but a default that buys nothing stays off.  One list at the end, in input order.
            overlap = (not ordered) and self.device.type == "cuda" and bool(kwargs.get("overlap_vocoder", False))
            voc_chunk = max(1, int(kwargs.get("vocoder_chunk", 32)))
            side = torch.cuda.Stream(device=self.device) if overlap else None
            t_req = torch.cuda.Event(enable_timing=True) if overlap else None
            t_first = None
            if overlap:
                t_req.record(torch.cuda.current_stream(self.device))
            done_wavs, pending = {}, []

            def vocode(batch_idx):
                nonlocal t_first
                items = [ready.pop(i) for i in batch_idx]
                go = torch.cuda.Event()
                go.record(torch.cuda.current_stream(self.device))      # every token of these utterances was written before this point of the caller's stream
                side.wait_event(go)
                with torch.cuda.stream(side):
                    for it in items:
                        it.record_stream(side)                           # (views of the generate call's buffers: keep the allocator off them until the side stream is done)
                    wavs = self._decode_to_wavs(items, use_decoder)
                    if t_first is None:
                        t_first = torch.cuda.Event(enable_timing=True)
                        t_first.record(side)
                for i, w in zip(batch_idx, wavs):
                    done_wavs[i] = w

            for ev in events:
                for k, ids_k, hid_k in ev:
                    ready[order[k]] = hid_k if use_decoder else ids_k
                    if ids_sink is not None:
                        ids_sink.append((utt_ids[order[k]], ids_k))
                    pending.append(order[k])
                if not ordered:
                    while overlap and len(pending) >= voc_chunk:
                        vocode(pending[:voc_chunk])
                        pending = pending[voc_chunk:]
                    continue
                run = 0
                while next_i + run in ready:
                    run += 1
                if run and (first or run >= 8 or next_i + run == n_all):
                    yield self._decode_to_wavs([ready.pop(next_i + j) for j in range(run)], use_decoder)
                    next_i += run
                    first = False
            if overlap:
                if pending:
                    vocode(pending)
                torch.cuda.current_stream(self.device).wait_stream(side)      # the caller's stream sees the finished waveforms
                side.synchronize()
                self.last_first_audio_ms = t_req.elapsed_time(t_first) if t_first is not None else None
                rest = sorted(done_wavs)
                if rest:
                    yield [done_wavs[i] for i in rest]
                return
            rest = [i for i in range(next_i, n_all) if i in ready]      # (everything, in throughput mode without overlap; nothing unless the run was interrupted, otherwise)
            if rest:
                yield self._decode_to_wavs([ready[i] for i in rest], use_decoder)
            return
        for ii in range(0, len(text_in), slice_size):
            text = list(text_in[ii:ii + slice_size])
            if not skip_refine_text:                                                   # pipeline:399-411
                refined = self._refine_text(text, params_refine_text)
                text_tokens = [i[i.less(tok.break_0_ids)] for i in refined.ids]
                text = tok.decode(text_tokens)
                if refine_text_only:
                    yield text
            if refine_text_only:
                continue
            text = [t if t.strip().endswith("[uv_break]") else t + " [uv_break]" for t in text]   # pipeline:414-416
            length, pass_batch_count, last = 0, 0, None
            if lora_paths is not None:
                gpt.set_row_adapters(self._adapter_slots(gpt, lora_paths[ii:ii + slice_size]))
            gen_kw = {}
            if noise_mode != "auto" or len(text) > 4:      # device (or caller-chosen) noise; slices of <= 4 keep the reference-compatible default
                if noise_seed is None and (noise_mode in ("auto", "device")):
                    noise_seed = int(torch.randint(0, 2 ** 62, (1,)).item())       # ONE draw per request (torch.manual_seed reproduces it)
                gen_kw = dict(noise=("device" if noise_mode == "auto" else noise_mode), seed=noise_seed, utt_ids=utt_ids[ii:ii + slice_size])
            if utt_limits is not None:
                gen_kw["max_new_tokens_per_row"] = list(utt_limits[ii:ii + slice_size])
            results = self._infer_code(text, stream, use_decoder, params_infer_code, gpt=gpt, **gen_kw)
            try:
                for result in results:
                    if not stream:
                        if kwargs.get("_ids_sink") is not None:
                            kwargs["_ids_sink"].extend(zip(utt_ids[ii:ii + slice_size], result.ids))
                        yield self._decode_to_wavs(result.hiddens if use_decoder else result.ids, use_decoder)      # pipeline:435-439
                        continue
                    # The reference's stream branch vocodes the whole prefix for every chunk and indexes a python list with
                    # .shape.  Here every yield is the [length, b) sample window of that same prefix waveform,
                    # vocoded from the tokens inside the window's receptive field only (Synth.decode_window), zero padded
                    # like pad_sequence would pad the shorter utterances.
                    last = result.hiddens if use_decoder else result.ids
                    pass_batch_count += 1
                    if pass_batch_count <= params_infer_code.pass_first_n_batches:
                        continue
                    total = max((256 * (2 * int(h.shape[0]) - 1) if h.shape[0] > 0 else 0) for h in last)
                    b = min(length + params_infer_code.stream_speed, total)
                    if b > length:
                        yield self._window(last, length, b, use_decoder)
                        length = b
            finally:
                if lora_paths is not None:          # the generator stays lazy (streaming works with per-utterance adapters); the row table is
                    gpt.set_row_adapters(None)      # reset when the slice is exhausted, closed or fails
            if stream and last is not None:
                total = max((256 * (2 * int(h.shape[0]) - 1) if h.shape[0] > 0 else 0) for h in last)
                if total > length:
                    yield self._window(last, length, total, use_decoder)

    def _window(self, hiddens, s0: int, s1: int, use_decoder: bool = True) -> torch.Tensor:
        """[B, s1-s0] samples s0..s1 of the padded batch of prefix waveforms."""
        syn = self.synth if use_decoder else self._codes_synth()
        parts = syn.decode_window(list(hiddens), [s0] * len(hiddens), [s1] * len(hiddens))
        out = torch.zeros(len(parts), s1 - s0, device=self.device)
        for u, w in enumerate(parts):
            out[u, :w.shape[0]] = w
        return out

    @torch.no_grad()
    def infer(self, text, stream=False, lang=None, skip_refine_text=False, refine_text_only=False, use_decoder=True,
              do_text_normalization=True, do_text_optimization=True, do_homophone_replacement=True,
              params_refine_text=RefineTextParams(), params_infer_code=InferCodeParams(), **kwargs):
        """pipeline:472-579.  Speaker resolution: `speaker_emb_path` (.pt holding a base16384 str or a tensor),
        else params_infer_code.spk_emb as given (the reference overwrites it with a random speaker whenever no path is passed,
        pipeline:547-556), else a random speaker from spk_stat.  The params object is copied first: the default argument is one shared
        instance, and the reference's in-place writes make a zero-shot prompt or a sampled speaker stick to every later default call."""
        params_infer_code = dataclasses.replace(params_infer_code)
        if kwargs.get("speaker_audio_path"):                                      # zero shot (pipeline:486-499)
            from . import audio
            p = kwargs["speaker_audio_path"]
            assert os.path.exists(p), f"speaker_audio_path {p} not exists!"
            wav, sr = audio.load_audio(p)
            wav = torch.mean(audio.resample(wav, sr, 24000), 0)
            params_infer_code.spk_smp = self.sample_audio_speaker(wav)
            params_infer_code.txt_smp = kwargs.get("speaker_audio_text", "")
            params_infer_code.spk_emb = None
        elif kwargs.get("speaker_emb_path"):
            p = kwargs["speaker_emb_path"]
            assert os.path.exists(p), f"speaker_emb_path {p} not exists!"
            obj = torch.load(p, weights_only=True, map_location="cpu")
            if isinstance(obj, dict):
                obj = next(iter(obj.values()))
            params_infer_code.spk_emb = obj if isinstance(obj, str) else codec.speaker_to_vector(obj)
        elif params_infer_code.spk_emb is None:
            params_infer_code.spk_emb = self.sample_random_speaker()
        return self._infer(text, stream, lang, skip_refine_text, refine_text_only, use_decoder, do_text_normalization,
                           do_text_optimization, do_homophone_replacement, params_refine_text, params_infer_code, **kwargs)

    # -- multi-GPU: utterance sharding (BASELINE configs[3]: batch 256 over 8 GPUs) --------------------------------
    @torch.no_grad()
    def infer_sharded(self, texts: List[str], speaker_index: Optional[List[int]] = None, speaker_table: Optional[torch.Tensor] = None,
                      skip_refine_text: bool = True, params_refine_text=RefineTextParams(), params_infer_code=InferCodeParams(), **kwargs):
        """One request over all ranks of the initialised torch.distributed group (one process per GPU; world 1 works too).
        Every rank calls this with the same `texts`; utterances are independent (the reference runs them as sequential slices
        of 4 with no cross-slice state, pipeline:391-397), so they are split by `dist.partition` (length-balanced snake) and
        each rank runs its own utterances through the ordinary `_infer` (slices of `max_batch`) + `_decode_to_wavs`.  The only
        exchange: ONE broadcast of the speaker table [n_spk, 768] from rank 0 (RCCL over xGMI) and one all-reduce of the
        generated lengths (disjoint supports).  `speaker_index[i]` selects utterance i's row; without a table every utterance
        uses params_infer_code.spk_emb.
        Returns (indices of this rank's utterances, their waveforms in that order, generated token counts of ALL utterances).
        Note SURVEY F8: the reference's repetition penalty skips rows >= 625 of the flattened [B*4] batch; a rank never holds
        more than max_batch <= 128 sequences (512 rows) per call, so the quirk cannot trigger on any rank."""
        from . import dist as cdist
        if not isinstance(texts, list):
            texts = [texts]
        params = dataclasses.replace(params_infer_code)
        n_spk, dim = 1, self.models_dict["gpt"].model_dim
        if speaker_table is not None or speaker_index is not None:
            if speaker_index is None or len(speaker_index) != len(texts):
                raise _lib.HipBackendError("infer_sharded: speaker_index needs one entry per utterance")
            if speaker_table is None and (not torch.distributed.is_initialized() or torch.distributed.get_rank() == 0):
                raise _lib.HipBackendError("infer_sharded: speaker_index given without speaker_table (rank 0 holds the table that is broadcast)")
            n_spk = int(max(speaker_index)) + 1
            if speaker_table is not None:
                speaker_table = torch.as_tensor(speaker_table, dtype=torch.float32).reshape(-1, dim)
                assert speaker_table.shape[0] >= n_spk
                speaker_table = speaker_table[:n_spk]
        else:
            speaker_index = [0] * len(texts)
            if params.spk_emb is None:
                speaker_table = self._sample_random_speaker().view(1, dim) if (not torch.distributed.is_initialized() or torch.distributed.get_rank() == 0) else None
            else:
                speaker_table = codec.speaker_to_vector(params.spk_emb).view(1, dim) if isinstance(params.spk_emb, str) else torch.as_tensor(params.spk_emb, dtype=torch.float32).view(1, dim)
        slice_size = int(kwargs.pop("slice_size", self.models_dict["gpt"].max_batch))
        # the request's noise seed: the same on every rank; each utterance's device noise stream is keyed by (seed, its global index), so N ranks
        # produce what one rank would.  Without `noise_seed` rank 0 draws it from torch's CPU generator -- like infer() does, so torch.manual_seed /
        # TorchSeedContext reproduce a request and two requests differ -- and broadcasts it with the speaker table's process group.
        noise_seed = kwargs.pop("noise_seed", None)
        if noise_seed is None:
            noise_seed = cdist.broadcast_seed(int(torch.randint(0, 2 ** 62, (1,)).item()), self.device)
        noise_seed = int(noise_seed)
        kwargs.pop("noise", None); kwargs.pop("utt_ids", None)
        wavs_local: List[torch.Tensor] = []
        limits_all = kwargs.pop("max_new_tokens_per_utterance", None)          # per GLOBAL utterance; each slice gets its own entries
        if limits_all is not None and len(limits_all) != len(texts):
            raise _lib.HipBackendError(f"max_new_tokens_per_utterance: {len(limits_all)} entries for {len(texts)} utterances")
        lora_all = kwargs.pop("lora_paths", None)                              # one adapter directory (or None) per GLOBAL utterance; each slice gets its own entries
        if lora_all is not None and len(lora_all) != len(texts):
            raise _lib.HipBackendError(f"lora_paths: {len(lora_all)} entries for {len(texts)} utterances")
        ids_out = kwargs.pop("ids_out", None)                                  # optional list: receives this rank's generated ids, in `mine` order
        sink = [] if ids_out is not None else None

        continuous = bool(kwargs.pop("continuous", False))      # each rank keeps slice_size decode rows busy over ALL its utterances (infer(continuous=True))

        def run_local(indices, rows):
            lens = []
            step = len(indices) if continuous else slice_size
            for ii in range(0, len(indices), max(step, 1)):
                sl = indices[ii:ii + step]
                p = dataclasses.replace(params, spk_emb=rows[ii:ii + len(sl)])
                kw_sl = dict(kwargs)
                if limits_all is not None:
                    kw_sl["max_new_tokens_per_utterance"] = [int(limits_all[i]) for i in sl]
                if lora_all is not None:
                    kw_sl["lora_paths"] = [lora_all[i] for i in sl]
                if sink is not None:
                    kw_sl["_ids_sink"] = sink
                for wavs in self._infer([texts[i] for i in sl], False, None, skip_refine_text, False, True, True, False, True,
                                        params_refine_text, p, slice_size=(slice_size if continuous else len(sl)), utt_ids=list(sl), noise="device",
                                        noise_seed=noise_seed, continuous=("throughput" if continuous else False), **kw_sl):
                    wavs_local.extend(wavs)
                    lens.extend([(int(w.shape[0]) // 256 + 1) // 2 if w.shape[0] else 0 for w in wavs])
            return lens

        mine, all_lens = cdist.sharded_generate([len(t) for t in texts], speaker_index, speaker_table, n_spk, dim, self.device, run_local)
        if ids_out is not None:
            by_utt = dict(sink)
            ids_out.extend(by_utt[i] for i in mine)
        return mine, wavs_local, all_lens

