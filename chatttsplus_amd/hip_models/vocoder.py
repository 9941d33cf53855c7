"""hip_models.DVAE / hip_models.Vocos -- drop-ins for chattts_plus.models.DVAE (decode branch,
reference chattts_plus/models/dvae.py:203-291) and vocos.Vocos.decode (third-party, constructed at
pipelines/chattts_plus_pipeline.py:93-111, called at :303) running in libctts_hip.so.

Both classes share one native handle per process-wide (dvae, vocos) pair: the C ABI keeps the two in one
object (ctts_voc) because they share workspaces; `Synth` owns it, `DVAE` / `Vocos` are thin views with
the reference's call surface."""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import numpy as np
import torch

from .. import _lib


def _np(v) -> np.ndarray:
    a = v.detach().cpu().float().numpy() if isinstance(v, torch.Tensor) else np.asarray(v, dtype=np.float32)
    return np.ascontiguousarray(a, dtype=np.float32)


class Synth:
    """Owner of the native DVAE-decoder + Vocos handle."""

    def __init__(self, dvae_cfg: dict, vocos_cfg: dict, max_frames: int = 4096, device="cuda", max_batch: int = 1, vq_cfg: Optional[dict] = None):
        """`vq_cfg` (the YAML's dvae_encode.vq_config: dim, levels, G, R) makes this the "decode codes" model of use_decoder=False
        (pipeline:292): dvae_cfg is then DVAE_full's decoder_config and the input of decode_batch / dvae_decode_codes are code ids."""
        self.device = torch.device(device)
        self.max_batch = int(max_batch)
        self._lib = _lib.load()
        if not torch.cuda.is_available():
            raise _lib.HipBackendError("infer_type='hip' needs a visible MI355X; no CPU fallback")
        self.cfg = _lib.VocCfg(dvae_idim=int(dvae_cfg.get("idim", 384)), dvae_hidden=int(dvae_cfg.get("hidden", 512)),
                               dvae_bn=int(dvae_cfg.get("bn_dim", 128)), dvae_layers=int(dvae_cfg.get("n_layer", 12)),
                               n_mels=int(dvae_cfg.get("n_mels", 100)), vocos_dim=int(vocos_cfg.get("dim", 512)),
                               vocos_inter=int(vocos_cfg.get("intermediate_dim", 1536)), vocos_layers=int(vocos_cfg.get("num_layers", 8)),
                               n_fft=int(vocos_cfg.get("n_fft", 1024)), hop=int(vocos_cfg.get("hop_length", 256)), max_frames=int(max_frames),
                               max_batch=int(max_batch))
        self.vq = None
        if vq_cfg is not None:
            lv = [int(x) for x in vq_cfg.get("levels", (5, 5, 5, 5))]
            if len(lv) != 4:
                raise _lib.HipBackendError("hip DVAE: the quantiser restatement covers 4-dimensional FSQ codes (levels: [5, 5, 5, 5])")
            self.vq = dict(G=int(vq_cfg.get("G", 2)), R=int(vq_cfg.get("R", 2)), levels=lv)
            self.cfg.vq_groups, self.cfg.vq_residuals = self.vq["G"], self.vq["R"]
            for i, l in enumerate(lv):
                self.cfg.vq_levels[i] = l
        self._h = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self._lib.ctts_voc_create(C.byref(self.cfg), C.byref(self._h)), "ctts_voc_create")
        self._loaded = set()
        self._finalized = False

    def __del__(self):
        try:
            if getattr(self, "_h", None) and self._h.value:
                self._lib.ctts_voc_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass

    def load(self, prefix: str, sd: Dict[str, np.ndarray]):
        for k, v in sd.items():
            if prefix == "vocos." and k.startswith("feature_extractor."):
                continue                                     # mel extractor: encode path only
            if prefix == "dvae." and not (k.startswith("decoder.") or k in ("out_conv.weight", "coef") or
                                          (self.vq is not None and ".project_out." in k and k.startswith("vq_layer.quantizer.rvqs."))):
                continue                                     # encoder / downsample_conv / project_in: the zero-shot encode path (hip_models.DVAEEncoder)
            a = _np(v)
            _lib.check(self._lib.ctts_voc_set_weight(self._h, (prefix + k).encode(), a.ctypes.data_as(C.c_void_p), a.size), f"set_weight({prefix + k})")
        self._loaded.add(prefix)
        if {"dvae.", "vocos."} <= self._loaded and not self._finalized:
            with torch.cuda.device(self.device):
                _lib.check(self._lib.ctts_voc_finalize(self._h), "ctts_voc_finalize")
            self._finalized = True

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def dvae_decode(self, hidden: torch.Tensor) -> torch.Tensor:
        """hidden [n,768] fp32 device -> mel [100, 2n]."""
        if not self._finalized:
            raise _lib.HipBackendError("DVAE / Vocos weights not loaded")
        hidden = hidden.to(self.device, dtype=torch.float32).contiguous()
        n = int(hidden.shape[0])
        mel = torch.empty(self.cfg.n_mels, 2 * n, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self._lib.ctts_dvae_decode(self._h, hidden.data_ptr(), n, mel.data_ptr(), self._stream()), "dvae_decode")
        return mel

    def dvae_decode_codes(self, ids: torch.Tensor) -> torch.Tensor:
        """ids [n, G*R] (GPT.generate's ids rows) -> mel [100, 2n]: DVAE_full's decode branch (dvae.py:272-291 with vq_layer)."""
        if not self._finalized or self.vq is None:
            raise _lib.HipBackendError("decode-codes model not loaded (Synth(vq_cfg=...) + DVAE_full / Vocos weights)")
        ids = ids.to(self.device, dtype=torch.int32).contiguous()
        n = int(ids.shape[0])
        assert ids.dim() == 2 and ids.shape[1] == self.vq["G"] * self.vq["R"]
        mel = torch.empty(self.cfg.n_mels, 2 * n, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self._lib.ctts_dvae_decode_codes(self._h, ids.data_ptr(), n, mel.data_ptr(), self._stream()), "dvae_decode_codes")
        return mel

    def vocos_decode(self, mel: torch.Tensor) -> torch.Tensor:
        """mel [100,F] fp32 device -> wav [256 (F-1)]."""
        if not self._finalized:
            raise _lib.HipBackendError("DVAE / Vocos weights not loaded")
        mel = mel.to(self.device, dtype=torch.float32).contiguous()
        F = int(mel.shape[1])
        wav = torch.empty(self.cfg.hop * (F - 1), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self._lib.ctts_vocos_decode(self._h, mel.data_ptr(), F, wav.data_ptr(), self._stream()), "vocos_decode")
        return wav


    def decode_batch(self, hiddens):
        """[hidden [n_b,768]] -> [wav [256 (2 n_b - 1)]]: the whole of _decode_to_wavs (pipeline:286-305) for a batch in ONE
        launch sequence (ctts_synth_batch, grid.z = utterance) instead of the reference's per-utterance loop."""
        if not self._finalized:
            raise _lib.HipBackendError("DVAE / Vocos weights not loaded")
        outs = [None] * len(hiddens)
        todo = [i for i, h in enumerate(hiddens) if h.shape[0] > 0]
        for i, h in enumerate(hiddens):
            if h.shape[0] == 0:
                outs[i] = torch.zeros(0, device=self.device)
        with torch.cuda.device(self.device):
            for c0 in range(0, len(todo), self.max_batch):
                idx = todo[c0:c0 + self.max_batch]
                # the decode-codes model (use_decoder=False) takes the generated ids [n, 4] instead of hidden rows [n, 768]
                hs = [hiddens[i].to(self.device, dtype=torch.int32 if self.vq is not None else torch.float32).contiguous() for i in idx]
                ns = [int(h.shape[0]) for h in hs]
                wavs = [torch.empty(self.cfg.hop * (2 * n - 1), dtype=torch.float32, device=self.device) for n in ns]
                hp = (C.c_void_p * len(idx))(*[h.data_ptr() for h in hs])
                wp = (C.c_void_p * len(idx))(*[w.data_ptr() for w in wavs])
                nt = (C.c_int32 * len(idx))(*ns)
                fn = self._lib.ctts_synth_batch_codes if self.vq is not None else self._lib.ctts_synth_batch
                _lib.check(fn(self._h, hp, nt, len(idx), wp, self._stream()), "synth_batch")
                for i, w in zip(idx, wavs):
                    outs[i] = w
        return outs


    # ---- streaming: vocode only what a stream step emits ------------------------------------------------
    # Left/right context (in mel frames) beyond which an output sample no longer depends on the input: DVAE decoder conv_in
    # 2 x k3 (+-2), 12 blocks x k7 dilation 2 (+-72), out_conv k3 (+-1); Vocos embed k7 (+-3), 8 blocks x k7 (+-24); four
    # overlapping ISTFT frames (+-3).  LayerNorms are per frame.
    HALO_FRAMES = 75 + 27 + 3

    def decode_window(self, hiddens, starts, stops):
        """Samples [starts[u], stops[u]) of the waveform `decode_batch` would return for hiddens[u] (the prefix generated so
        far), computed from the tokens inside the receptive field of that window only.  Bit-identical to slicing the
        full-prefix waveform: every output element runs the same K-order arithmetic whatever rows surround it.  The
        reference's stream branch re-vocodes the whole prefix for every chunk it yields (pipeline:436-461)."""
        subs, offs, meta = [], [], []
        for h, s0, s1 in zip(hiddens, starts, stops):
            a, b, off, s0, s1 = window_token_range(int(h.shape[0]), s0, s1, self.cfg.hop, self.HALO_FRAMES)
            subs.append(h[a:b]); offs.append(off); meta.append((s0, s1))
        wavs = self.decode_batch(subs)
        return [w[s0 - o:s1 - o] if s1 > s0 else w[:0] for w, o, (s0, s1) in zip(wavs, offs, meta)]


def window_token_range(n_tokens: int, s0: int, s1: int, hop: int = 256, halo_frames: int = 105):
    """Tokens [a, b) whose vocoding contains samples [s0, s1) of the n_tokens-long prefix waveform exactly, the sample offset of
    that sub-waveform, and the clamped window.  A sample s depends on ISTFT frames s/hop - 1 .. s/hop + 2, each of which depends
    on `halo_frames - 3` mel frames either side (conv receptive fields); a token is two frames.  A right edge that would fall
    inside the halo of the true end is moved to the true end (the zero padding there is part of the reference's result)."""
    halo = halo_frames + 1
    total = hop * (2 * n_tokens - 1) if n_tokens > 0 else 0
    s0, s1 = max(0, min(int(s0), total)), max(0, min(int(s1), total))
    if s1 <= s0:
        return 0, 0, 0, 0, 0
    f0 = max(0, s0 // hop - halo)
    f1 = min(2 * n_tokens, (s1 + hop - 1) // hop + 1 + halo)
    a, b = f0 // 2, (f1 + 1) // 2
    if 2 * n_tokens - 2 * b < 2 * halo:
        b = n_tokens
    return a, b, 2 * a * hop, s0, s1


class DVAE:
    """Call surface of models.DVAE for the decode branch: DVAE(decoder_config, dim, coef, model_path)(inp[1,768,n]) -> mel[1,100,2n]."""

    def __new__(cls, decoder_config: Optional[dict] = None, encoder_config: Optional[dict] = None, vq_config: Optional[dict] = None, **kwargs):
        # the reference builds both `dvae_encode` and `dvae_decode` from the class name "DVAE" (configs/infer/chattts_plus.yaml:7-48):
        # a config with an encoder / quantiser is the zero-shot encode model
        if encoder_config is not None or vq_config is not None:
            from .encoder import DVAEEncoder
            kwargs.pop("synth", None)
            return DVAEEncoder(decoder_config=decoder_config, encoder_config=encoder_config, vq_config=vq_config, **kwargs)
        return super().__new__(cls)

    def __init__(self, decoder_config: dict, encoder_config: Optional[dict] = None, vq_config: Optional[dict] = None, dim=384,
                 coef=None, synth: Optional[Synth] = None, **kwargs):
        self.decoder_config = dict(decoder_config)
        self.dim = dim
        self.synth = synth
        self.model_path = kwargs.get("model_path")

    def eval(self):
        return self

    def to(self, *a, **k):
        return self

    def load_state_dict(self, sd, strict=True):
        self.synth.load("dvae.", sd)
        return self

    def from_pretrained(self, path):
        return self.load_state_dict(torch.load(path, weights_only=True, mmap=True))

    @torch.inference_mode()
    def __call__(self, inp: torch.Tensor, mode="decode") -> torch.Tensor:
        if mode != "decode":
            raise _lib.HipBackendError("hip DVAE: only mode='decode'")
        assert inp.dim() == 3 and inp.shape[0] == 1, "reference calls the decoder per utterance (pipeline:298-300)"
        hidden = inp[0].permute(1, 0)                      # [n,768]; the pipeline passed hiddens.permute(1,0)[None]
        return self.synth.dvae_decode(hidden)[None]


class Vocos:
    """Call surface the pipeline uses: .decode(mel[1,100,F]) -> wav[1,S], .parameters(), .load_state_dict, .eval(), .to()."""

    def __init__(self, synth: Synth):
        self.synth = synth
        self._p = torch.nn.Parameter(torch.zeros(1, device=synth.device), requires_grad=False)

    def parameters(self):
        return iter([self._p])

    def eval(self):
        return self

    def to(self, *a, **k):
        return self

    def load_state_dict(self, sd, strict=True):
        self.synth.load("vocos.", sd)
        return self

    @torch.inference_mode()
    def decode(self, mel: torch.Tensor) -> torch.Tensor:
        assert mel.dim() == 3 and mel.shape[0] == 1
        return self.synth.vocos_decode(mel[0])[None]
