"""Speaker / prompt wire formats of the reference (host side, boundary code).

The bundled speaker files (assets/speakers/*.pt, demos/.../speaker_pt/*.pt) hold a *str*: base16384 text of
raw-LZMA2-compressed fp16[768] (reference chattts_plus/models/tokenizer.py:139-148,211-222).
`pybase16384` is not installed in this image, so the codec is restated here (14-bit big-endian groups,
7 bytes -> 4 code points offset 0x4E00, tail marker 0x3D00 + remainder) and checked against the real
speaker files in tests/test_codec.py.
"""
from __future__ import annotations

import lzma
from typing import Union

import numpy as np
import torch

_FILTERS = [{"id": lzma.FILTER_LZMA2, "preset": 9 | lzma.PRESET_EXTREME}]


def b14_encode(data: bytes) -> str:
    n = len(data)
    out = []
    full = n // 7
    for g in range(full):
        v = int.from_bytes(data[7 * g: 7 * g + 7], "big")          # 56 bits -> 4 x 14 bits
        out += [0x4E00 + ((v >> s) & 0x3FFF) for s in (42, 28, 14, 0)]
    r = n % 7
    if r:
        tail = data[7 * full:] + b"\x00" * (7 - r)
        v = int.from_bytes(tail, "big")
        units = (8 * r + 13) // 14
        out += [0x4E00 + ((v >> s) & 0x3FFF) for s in (42, 28, 14, 0)][:units]
        out.append(0x3D00 + r)
    return "".join(chr(c) for c in out)


def b14_decode(s: str) -> bytes:
    codes = [ord(c) for c in s]
    r = 0
    if codes and (codes[-1] & 0xFF00) == 0x3D00:
        r = codes[-1] & 0xFF
        codes = codes[:-1]
    out = bytearray()
    n_full = len(codes) // 4 if r == 0 else (len(codes) - (8 * r + 13) // 14) // 4
    for g in range(n_full):
        v = 0
        for c in codes[4 * g: 4 * g + 4]:
            v = (v << 14) | ((c - 0x4E00) & 0x3FFF)
        out += v.to_bytes(7, "big")
    if r:
        rest = codes[4 * n_full:]
        v = 0
        for c in rest:
            v = (v << 14) | ((c - 0x4E00) & 0x3FFF)
        v <<= 14 * (4 - len(rest))
        out += v.to_bytes(7, "big")[:r]
    return bytes(out)


def decode_spk_emb(s: str) -> np.ndarray:
    """Tokenizer._decode_spk_emb (tokenizer.py:139-148) -> fp16[768]."""
    return np.frombuffer(lzma.decompress(b14_decode(s), format=lzma.FORMAT_RAW, filters=_FILTERS), dtype=np.float16).copy()


def encode_spk_emb(spk: Union[np.ndarray, torch.Tensor]) -> str:
    """Tokenizer._encode_spk_emb / pipeline._encode_spk_emb (tokenizer.py:211-222; pipeline:310-321)."""
    arr = spk.detach().to(dtype=torch.float16, device="cpu").numpy() if isinstance(spk, torch.Tensor) else np.asarray(spk, dtype=np.float16)
    return b14_encode(lzma.compress(arr.tobytes(), format=lzma.FORMAT_RAW, filters=_FILTERS))


def decode_prompt(prompt: str) -> torch.Tensor:
    """Tokenizer._decode_prompt (tokenizer.py:180-193): uint16 [num_vq, n] audio-prompt codes."""
    dec = b14_decode(prompt)
    shp = np.frombuffer(dec[:4], dtype="<u2")
    p = np.frombuffer(lzma.decompress(dec[4:], format=lzma.FORMAT_RAW, filters=_FILTERS), dtype="<u2").copy()
    return torch.from_numpy(p.astype(np.int64)).view(*[int(x) for x in shp])


def encode_prompt(prompt: torch.Tensor) -> str:
    """Tokenizer._encode_prompt (tokenizer.py:195-209)."""
    arr = prompt.detach().cpu().numpy().astype("<u2")
    assert arr.ndim == 2
    return b14_encode(np.array(arr.shape, dtype="<u2").tobytes() + lzma.compress(arr.tobytes(), format=lzma.FORMAT_RAW, filters=_FILTERS))


def coef_from_string(coef: str) -> np.ndarray:
    """DVAE coef string (dvae.py:217-220; pipeline:56-66): base16384 of fp32[100]."""
    return np.frombuffer(b14_decode(coef), dtype=np.float32).copy()


def coef_to_string(coef: np.ndarray) -> str:
    return b14_encode(np.asarray(coef, dtype=np.float32).tobytes())


def speaker_to_vector(spk_emb: Union[str, np.ndarray, torch.Tensor]) -> torch.Tensor:
    """Canonical speaker form: a [768] fp32 vector (str -> decoded fp16; [1,768] tensors are flattened --
    the reference mishandles that shape)."""
    if isinstance(spk_emb, str):
        v = torch.from_numpy(decode_spk_emb(spk_emb).astype(np.float32))
    else:
        v = torch.as_tensor(spk_emb).detach().float().cpu()
    return v.reshape(-1)


def apply_spk_emb(emb: torch.Tensor, spk_emb, input_ids: torch.Tensor, spk_emb_ids: int) -> torch.Tensor:
    """Tokenizer.apply_spk_emb (tokenizer.py:150-178): rows whose first id is [spk_emb] <- F.normalize(spk, p=2, dim=0, eps=1e-12)."""
    n = torch.nn.functional.normalize(speaker_to_vector(spk_emb), p=2.0, dim=0, eps=1e-12).to(emb.device, dtype=emb.dtype)
    cond = input_ids.to(emb.device).narrow(-1, 0, 1).eq(spk_emb_ids).expand(emb.shape)
    return torch.where(cond, n.expand(emb.shape), emb)
