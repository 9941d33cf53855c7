"""Host-side audio I/O for the zero-shot path (pipelines/chattts_plus_pipeline.py:493-496): read a clip, mix down to mono,
bring it to 24 kHz.  The reference uses torchaudio.load + torchaudio.functional.resample; torchaudio is not available here,
so the reader is scipy's and the resampler restates torchaudio's documented default algorithm (windowed-sinc polyphase
filter, hann window, lowpass_filter_width=6, rolloff=0.99) -- CPU pre-processing outside the hot path, parity unpinned."""
from __future__ import annotations

import math
from typing import Tuple

import numpy as np
import torch


def load_audio(path: str) -> Tuple[torch.Tensor, int]:
    """[channels, n] float32 in [-1, 1] and the sample rate (torchaudio.load semantics for PCM wav files)."""
    from scipy.io import wavfile
    sr, data = wavfile.read(path)
    a = np.asarray(data)
    if a.ndim == 1:
        a = a[:, None]
    if a.dtype == np.int16:
        x = a.astype(np.float32) / 32768.0
    elif a.dtype == np.int32:
        x = a.astype(np.float32) / 2147483648.0
    elif a.dtype == np.uint8:
        x = (a.astype(np.float32) - 128.0) / 128.0
    else:
        x = a.astype(np.float32)
    return torch.from_numpy(np.ascontiguousarray(x.T)), int(sr)


def resample(wav: torch.Tensor, orig_freq: int, new_freq: int, lowpass_filter_width: int = 6, rolloff: float = 0.99) -> torch.Tensor:
    """torchaudio.functional.resample(..., resampling_method="sinc_interp_hann") on [..., n]."""
    if orig_freq == new_freq:
        return wav
    g = math.gcd(int(orig_freq), int(new_freq))
    orig, new = int(orig_freq) // g, int(new_freq) // g
    base = min(orig, new) * rolloff
    width = math.ceil(lowpass_filter_width * orig / base)
    idx = torch.arange(-width, width + orig, dtype=torch.float64)[None, None] / orig
    t = torch.arange(0, -new, -1, dtype=torch.float64)[:, None, None] / new + idx
    t = (t * base).clamp_(-lowpass_filter_width, lowpass_filter_width)
    window = torch.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t = t * math.pi
    scale = base / orig
    kernels = torch.where(t == 0, torch.tensor(1.0, dtype=torch.float64), t.sin() / t) * window * scale
    kernels = kernels.to(torch.float32)                                         # [new, 1, 2 width + orig]
    shape = wav.shape
    x = wav.reshape(-1, shape[-1]).float()
    x = torch.nn.functional.pad(x, (width, width + orig))
    y = torch.nn.functional.conv1d(x[:, None], kernels, stride=orig)            # [b, new, frames]
    y = y.transpose(1, 2).reshape(x.shape[0], -1)
    target = math.ceil(new * shape[-1] / orig)
    return y[..., :target].reshape(shape[:-1] + (target,))
