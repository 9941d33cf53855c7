"""hip_models.DVAEEncoder -- the encode branch of chattts_plus.models.DVAE (reference chattts_plus/models/dvae.py:263-270,
constructed from configs/infer/chattts_plus.yaml `dvae_encode`) running in libctts_hip.so: waveform -> audio-prompt codes
for zero-shot speaker cloning (pipelines/chattts_plus_pipeline.py:279-284,486-499).

Two ingredients are third-party in the reference and absent offline (parity unpinned, see oracle/ref_cpu.py): torchaudio's
MelSpectrogram (the window and filterbank below follow its documented defaults) and vector_quantize_pytorch 1.17.8's
GroupedResidualFSQ (`pre_bound` selects the residual-initialisation detail that differs between its releases)."""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, Optional

import numpy as np
import torch

from .. import _lib
from .vocoder import _np


def hann_window(n_fft: int = 1024) -> np.ndarray:
    """torch.hann_window(n_fft, periodic=True) -- MelSpectrogram's default window."""
    return torch.hann_window(n_fft, periodic=True).numpy().astype(np.float32)


def melscale_fbanks(n_freqs: int = 513, n_mels: int = 100, sample_rate: int = 24000) -> np.ndarray:
    """Triangular filterbank [n_freqs, n_mels] of MelScale(f_min=0, f_max=sr/2, norm=None, mel_scale="htk") (dvae.py:185-192)."""
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m_max = 2595.0 * math.log10(1.0 + (sample_rate / 2.0) / 700.0)
    m_pts = torch.linspace(0.0, m_max, n_mels + 2)
    f_pts = 700.0 * (10 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    fb = torch.clamp(torch.min((-1.0 * slopes[:, :-2]) / f_diff[:-1], slopes[:, 2:] / f_diff[1:]), min=0.0)
    return fb.numpy().astype(np.float32)


class DVAEEncoder:
    """DVAE(decoder_config, encoder_config, vq_config, dim, coef, model_path)(wav[1, n], "encode") -> codes [1, 4, T]."""

    def __init__(self, decoder_config: Optional[dict] = None, encoder_config: Optional[dict] = None, vq_config: Optional[dict] = None, dim=512,
                 coef=None, device="cuda", max_seconds: float = 60.0, pre_bound: bool = True, **kwargs):
        enc = dict(encoder_config or dict(idim=512, odim=1024, hidden=256, n_layer=12, bn_dim=128))
        vq = dict(vq_config or dict(dim=1024, levels=(5, 5, 5, 5), G=2, R=2))
        if tuple(vq.get("levels", (5, 5, 5, 5))) != (5, 5, 5, 5):
            raise _lib.HipBackendError("hip DVAE encoder: GFSQ levels must be [5, 5, 5, 5]")
        self.device = torch.device(device)
        self.model_path = kwargs.get("model_path")
        self._lib = _lib.load()
        if not torch.cuda.is_available():
            raise _lib.HipBackendError("infer_type='hip' needs a visible MI355X; no CPU fallback")
        self.G, self.R = int(vq.get("G", 2)), int(vq.get("R", 2))
        self.cfg = _lib.EncCfg(n_mels=100, dim=int(dim), enc_hidden=int(enc.get("hidden", 256)), enc_bn=int(enc.get("bn_dim", 128)),
                               enc_layers=int(enc.get("n_layer", 12)), enc_odim=int(enc.get("odim", 1024)), vq_groups=self.G, vq_residuals=self.R,
                               n_fft=1024, hop=256, max_samples=int(max_seconds * 24000), pre_bound=1 if pre_bound else 0)
        self._h = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self._lib.ctts_enc_create(C.byref(self.cfg), C.byref(self._h)), "ctts_enc_create")
        self._finalized = False
        self._coef = None if coef is None else coef

    def __del__(self):
        try:
            if getattr(self, "_h", None) and self._h.value:
                self._lib.ctts_enc_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass

    def eval(self):
        return self

    def to(self, *a, **k):
        return self

    def _set(self, name: str, arr: np.ndarray):
        a = np.ascontiguousarray(arr, dtype=np.float32)
        _lib.check(self._lib.ctts_enc_set_weight(self._h, name.encode(), a.ctypes.data_as(C.c_void_p), a.size), f"set_weight({name})")

    def load_state_dict(self, sd: Dict[str, np.ndarray], strict=True):
        for k, v in sd.items():
            if k.startswith(("downsample_conv.", "encoder.")) or k == "coef" or (k.startswith("vq_layer.quantizer.rvqs.") and ".project_in." in k):
                self._set(k, _np(v))                          # decoder.* / out_conv / project_out / mel buffers: not on the encode path
        self._set("mel.window", hann_window(self.cfg.n_fft))
        self._set("mel.fb", melscale_fbanks(self.cfg.n_fft // 2 + 1, self.cfg.n_mels, 24000))
        with torch.cuda.device(self.device):
            _lib.check(self._lib.ctts_enc_finalize(self._h), "ctts_enc_finalize")
        self._finalized = True
        return self

    def from_pretrained(self, path):
        return self.load_state_dict(torch.load(path, weights_only=True, mmap=True))

    def n_codes(self, n_samples: int) -> int:
        return ((1 + n_samples // 256) - 2) // 2 + 1

    @torch.inference_mode()
    def encode(self, wav: torch.Tensor, return_debug: bool = False):
        """wav [n] fp32 @ 24 kHz -> codes int32 [G*R, T] on the device (+ (log-mel / coef [100, F], features [T, 1024]))."""
        if not self._finalized:
            raise _lib.HipBackendError("DVAE encoder weights not loaded")
        wav = wav.to(self.device, dtype=torch.float32).contiguous().view(-1)
        n = int(wav.shape[0])
        T = self.n_codes(n)
        ids = torch.empty(self.G * self.R, T, dtype=torch.int32, device=self.device)
        mel = torch.empty(self.cfg.n_mels, 1 + n // 256, dtype=torch.float32, device=self.device) if return_debug else None
        feat = torch.empty(T, self.cfg.enc_odim, dtype=torch.float32, device=self.device) if return_debug else None
        with torch.cuda.device(self.device):
            st = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
            _lib.check(self._lib.ctts_dvae_encode(self._h, wav.data_ptr(), n, ids.data_ptr(), mel.data_ptr() if mel is not None else None,
                                                  feat.data_ptr() if feat is not None else None, st), "dvae_encode")
        return (ids, mel, feat) if return_debug else ids

    @torch.inference_mode()
    def __call__(self, inp: torch.Tensor, mode="encode") -> torch.Tensor:
        if mode != "encode":
            raise _lib.HipBackendError("hip DVAEEncoder: only mode='encode' (decode is served by hip_models.DVAE)")
        assert inp.dim() == 2 and inp.shape[0] == 1, "reference calls dvae_encode(wav[None], 'encode') (pipeline:283-284)"
        return self.encode(inp[0])[None]
