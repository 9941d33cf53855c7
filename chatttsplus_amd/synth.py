"""Seeded synthetic checkpoints and noise for the ChatTTS hot path.

No ChatTTS checkpoint exists offline, so every test, the bench
and the golden fixtures use weights regenerated from a counter-based PRNG owned
by this repo (numpy Philox keyed by (seed, tensor name)).  Both the build
container (where the goldens are minted against the imported reference) and the
GPU box regenerate bit-identical tensors.

Key names / shapes follow the reference state dicts:
  GPT   -- chattts_plus/models/gpt.py:41-77 (SURVEY.md section 3.1, 196 keys)
  DVAE  -- chattts_plus/models/dvae.py:130-159,203-239 (decode-only: no encoder / vq)
  Vocos -- third-party vocos 0.1.0: backbone.*, head.out.*
"""
from __future__ import annotations

import zlib
from typing import Dict

import numpy as np

# Real configuration (configs/infer/chattts_plus.yaml:66-82 in the reference).
GPT_REAL = dict(hidden_size=768, intermediate_size=3072, num_attention_heads=12, num_hidden_layers=20,
                num_audio_tokens=626, num_text_tokens=21178, num_vq=4)
# Tiny configuration used for exhaustive oracle <-> reference goldens (head_dim must stay 64).
GPT_TINY = dict(hidden_size=128, intermediate_size=256, num_attention_heads=2, num_hidden_layers=2,
                num_audio_tokens=626, num_text_tokens=512, num_vq=4)
DVAE_REAL = dict(idim=384, odim=384, hidden=512, n_layer=12, bn_dim=128, dim=384, n_mels=100)
VOCOS_REAL = dict(input_channels=100, dim=512, intermediate_dim=1536, num_layers=8, n_fft=1024, hop_length=256)


def _rng(seed: int, name: str) -> np.random.Generator:
    key = (int(seed) & 0xFFFFFFFF) << 32 | (zlib.crc32(name.encode()) & 0xFFFFFFFF)
    return np.random.Generator(np.random.Philox(key=key))


def _normal(seed, name, shape, std):
    return (_rng(seed, name).standard_normal(size=shape, dtype=np.float32) * np.float32(std)).astype(np.float32)


def _uniform(seed, name, shape, lo, hi):
    return (_rng(seed, name).random(size=shape, dtype=np.float32) * np.float32(hi - lo) + np.float32(lo)).astype(np.float32)


def gpt_state_dict(cfg: dict = GPT_REAL, seed: int = 1234) -> Dict[str, np.ndarray]:
    """Random GPT checkpoint with the reference's key names (gpt.py:41-77; llama.py:678-749)."""
    H, I, L = cfg["hidden_size"], cfg["intermediate_size"], cfg["num_hidden_layers"]
    V, VT, NVQ = cfg["num_audio_tokens"], cfg["num_text_tokens"], cfg["num_vq"]
    sd: Dict[str, np.ndarray] = {}
    for l in range(L):
        p = f"gpt.layers.{l}."
        for n in ("q_proj", "k_proj", "v_proj", "o_proj"):
            sd[p + f"self_attn.{n}.weight"] = _normal(seed, p + n, (H, H), 0.02)
        sd[p + "mlp.gate_proj.weight"] = _normal(seed, p + "gate", (I, H), 0.02)
        sd[p + "mlp.up_proj.weight"] = _normal(seed, p + "up", (I, H), 0.02)
        sd[p + "mlp.down_proj.weight"] = _normal(seed, p + "down", (H, I), 0.02)
        # norms ~ 1 +- 10% so that a dropped norm weight is caught by parity tests
        sd[p + "input_layernorm.weight"] = 1.0 + _normal(seed, p + "ln1", (H,), 0.1)
        sd[p + "post_attention_layernorm.weight"] = 1.0 + _normal(seed, p + "ln2", (H,), 0.1)
    sd["gpt.norm.weight"] = 1.0 + _normal(seed, "gpt.norm", (H,), 0.1)
    for i in range(NVQ):
        sd[f"emb_code.{i}.weight"] = _normal(seed, f"emb_code.{i}", (V, H), 0.02)
    sd["emb_text.weight"] = _normal(seed, "emb_text", (VT, H), 0.02)
    # weight-normed heads: weight = g * v / ||v||_row  (torch parametrizations.weight_norm, dim=0)
    v = _normal(seed, "head_text.v", (VT, H), 0.02)
    sd["head_text.parametrizations.weight.original1"] = v
    sd["head_text.parametrizations.weight.original0"] = (
        np.linalg.norm(v, axis=1, keepdims=True) * (1.0 + _normal(seed, "head_text.g", (VT, 1), 0.1))).astype(np.float32)
    for i in range(NVQ):
        v = _normal(seed, f"head_code.{i}.v", (V, H), 0.02)
        sd[f"head_code.{i}.parametrizations.weight.original1"] = v
        sd[f"head_code.{i}.parametrizations.weight.original0"] = (
            np.linalg.norm(v, axis=1, keepdims=True) * (1.0 + _normal(seed, f"head_code.{i}.g", (V, 1), 0.1))).astype(np.float32)
    return sd


def stress_gpt_state_dict(sd: Dict[str, np.ndarray], proj_scale: float = 8.0, head_scale: float = 8.0) -> Dict[str, np.ndarray]:
    """Stress variant of a synthetic GPT checkpoint: every q/k/v/o/gate/up/down projection scaled by `proj_scale` (attention
    logits x proj_scale^2 -> peaked attention; MLP outputs x proj_scale^3 -> outlier channels in the residual stream) and the
    weight-norm gains of the code/text heads by `head_scale` (sharpened sampling distributions).  N(0, 0.02^2) weights alone
    give near-uniform attention and small activations -- the easy case for fp16 weights / KV."""
    out = {k: v.copy() for k, v in sd.items()}
    for k in out:
        if k.endswith("_proj.weight"):
            out[k] = (out[k] * np.float32(proj_scale)).astype(np.float32)
        elif k.startswith("head_") and k.endswith("original0"):
            out[k] = (out[k] * np.float32(head_scale)).astype(np.float32)
    return out


def _conv(seed, name, cout, cin_per_group, k, bias=True):
    fan_in = cin_per_group * k
    out = {name + ".weight": _normal(seed, name + ".w", (cout, cin_per_group, k), 1.0 / np.sqrt(fan_in))}
    if bias:
        out[name + ".bias"] = _normal(seed, name + ".b", (cout,), 0.05)
    return out


def _linear(seed, name, cout, cin, bias=True, gain=1.0):
    out = {name + ".weight": _normal(seed, name + ".w", (cout, cin), gain / np.sqrt(cin))}
    if bias:
        out[name + ".bias"] = _normal(seed, name + ".b", (cout,), 0.05)
    return out


def _convnext(seed, p, dim, inter, k):
    sd = {}
    sd.update(_conv(seed, p + "dwconv", dim, 1, k))
    sd[p + "norm.weight"] = 1.0 + _normal(seed, p + "norm.w", (dim,), 0.1)
    sd[p + "norm.bias"] = _normal(seed, p + "norm.b", (dim,), 0.05)
    sd.update(_linear(seed, p + "pwconv1", inter, dim))
    sd.update(_linear(seed, p + "pwconv2", dim, inter))
    sd[p + "gamma"] = _uniform(seed, p + "gamma", (dim,), 0.05, 0.3)
    return sd


def dvae_state_dict(cfg: dict = DVAE_REAL, seed: int = 1234) -> Dict[str, np.ndarray]:
    """Decode-only DVAE checkpoint (dvae.py:130-159 DVAEDecoder, :235 out_conv, :222 coef)."""
    sd: Dict[str, np.ndarray] = {}
    sd["coef"] = _uniform(seed, "dvae.coef", (1, cfg["n_mels"], 1), 0.5, 1.5)
    sd.update(_conv(seed, "decoder.conv_in.0", cfg["bn_dim"], cfg["idim"], 3))
    sd.update(_conv(seed, "decoder.conv_in.2", cfg["hidden"], cfg["bn_dim"], 3))
    for i in range(cfg["n_layer"]):
        sd.update(_convnext(seed, f"decoder.decoder_block.{i}.", cfg["hidden"], cfg["hidden"] * 4, 7))
    sd.update(_conv(seed, "decoder.conv_out", cfg["odim"], cfg["hidden"], 1, bias=False))
    sd.update(_conv(seed, "out_conv", cfg["n_mels"], cfg["dim"], 3, bias=False))
    return sd


DVAE_ENC_REAL = dict(dim=512, n_mels=100, enc_idim=512, enc_odim=1024, enc_hidden=256, enc_n_layer=12, enc_bn_dim=128,
                     vq_dim=1024, vq_levels=(5, 5, 5, 5), vq_G=2, vq_R=2)


def dvae_encoder_state_dict(cfg: dict = DVAE_ENC_REAL, seed: int = 1234) -> Dict[str, np.ndarray]:
    """The encode-side tensors of DVAE_full.pt (configs/infer/chattts_plus.yaml dvae_encode; dvae.py:224-231 downsample_conv
    + encoder, :66-81 GFSQ -> vector_quantize_pytorch GroupedResidualFSQ `rvqs.{g}.project_in`): zero-shot speaker path."""
    sd: Dict[str, np.ndarray] = {}
    D = cfg["dim"]
    sd["coef"] = _uniform(seed, "dvae_enc.coef", (1, cfg["n_mels"], 1), 0.5, 1.5)
    sd.update(_conv(seed, "downsample_conv.0", D, cfg["n_mels"], 3))
    sd.update(_conv(seed, "downsample_conv.2", D, D, 4))
    sd.update(_conv(seed, "encoder.conv_in.0", cfg["enc_bn_dim"], cfg["enc_idim"], 3))
    sd.update(_conv(seed, "encoder.conv_in.2", cfg["enc_hidden"], cfg["enc_bn_dim"], 3))
    for i in range(cfg["enc_n_layer"]):
        sd.update(_convnext(seed, f"encoder.decoder_block.{i}.", cfg["enc_hidden"], cfg["enc_hidden"] * 4, 7))
    sd.update(_conv(seed, "encoder.conv_out", cfg["enc_odim"], cfg["enc_hidden"], 1, bias=False))
    per = cfg["vq_dim"] // cfg["vq_G"]
    for g in range(cfg["vq_G"]):
        # project_in spreads the codes over all five levels (tanh(z) * 2.002 with |z| ~ 1)
        sd.update(_linear(seed, f"vq_layer.quantizer.rvqs.{g}.project_in", len(cfg["vq_levels"]), per, gain=1.0))
    return sd


# decode side of DVAE_full.pt (configs/infer/chattts_plus.yaml dvae_encode: decoder_config idim 512 / hidden 256, dim 512) + the quantiser's
# project_out: what use_decoder=False runs (pipeline:292; dvae.py:272-291 with vq_layer, :85-96 GFSQ._embed)
DVAE_FULL_DEC = dict(idim=512, odim=512, hidden=256, n_layer=12, bn_dim=128, dim=512, n_mels=100, vq_levels=(5, 5, 5, 5), vq_G=2, vq_R=2)


def dvae_full_decoder_state_dict(cfg: dict = DVAE_FULL_DEC, seed: int = 1234) -> Dict[str, np.ndarray]:
    """decoder.* / out_conv / coef of a DVAE_full-shaped checkpoint + `vq_layer.quantizer.rvqs.{g}.project_out` (Linear 4 -> dim)."""
    sd = dict(dvae_state_dict(cfg, seed + 77))
    for g in range(cfg["vq_G"]):
        sd.update(_linear(seed, f"vq_layer.quantizer.rvqs.{g}.project_out", cfg["dim"], len(cfg["vq_levels"]), gain=1.0))
    return sd


def speaker_wave(seed: int, n_samples: int) -> np.ndarray:
    """Synthetic 24 kHz speech-like test signal in [-1, 1]: a few drifting harmonics plus noise."""
    t = np.arange(n_samples, dtype=np.float64) / 24000.0
    r = _rng(seed, "speaker_wave")
    f0 = 110.0 + 40.0 * np.sin(2 * np.pi * 0.7 * t + r.random())
    ph = 2 * np.pi * np.cumsum(f0) / 24000.0
    x = sum(a * np.sin(k * ph + r.random() * 6.28) for k, a in ((1, 0.5), (2, 0.3), (3, 0.15), (5, 0.08), (9, 0.04)))
    x = x * (0.6 + 0.4 * np.sin(2 * np.pi * 3.1 * t)) + 0.02 * r.standard_normal(n_samples)
    return (0.8 * x / np.abs(x).max()).astype(np.float32)


def vocos_state_dict(cfg: dict = VOCOS_REAL, seed: int = 1234) -> Dict[str, np.ndarray]:
    """Vocos checkpoint (upstream vocos 0.1.0 VocosBackbone + ISTFTHead key names)."""
    sd: Dict[str, np.ndarray] = {}
    D = cfg["dim"]
    sd.update(_conv(seed, "backbone.embed", D, cfg["input_channels"], 7))
    sd["backbone.norm.weight"] = 1.0 + _normal(seed, "backbone.norm.w", (D,), 0.1)
    sd["backbone.norm.bias"] = _normal(seed, "backbone.norm.b", (D,), 0.05)
    for i in range(cfg["num_layers"]):
        sd.update(_convnext(seed, f"backbone.convnext.{i}.", D, cfg["intermediate_dim"], 7))
    sd["backbone.final_layer_norm.weight"] = 1.0 + _normal(seed, "backbone.fln.w", (D,), 0.1)
    sd["backbone.final_layer_norm.bias"] = _normal(seed, "backbone.fln.b", (D,), 0.05)
    # keep log-magnitudes and phases O(1): exp(mag) stays well inside the 1e2 clip most of the time
    sd.update(_linear(seed, "head.out", cfg["n_fft"] + 2, D, gain=0.5))
    n = cfg["n_fft"]
    sd["head.istft.window"] = (0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n, dtype=np.float64) / n)).astype(np.float32)
    return sd


def prompt_ids(batch: int, length: int, num_text_tokens: int, seed: int = 1234, num_vq: int = 4,
               pad_left=None) -> tuple:
    """Synthetic text prompt: ids ~ U[0,V_text) replicated over num_vq (tokenizer.py:127),
    left-padded attention mask (tokenizer.py:96-115).  Returns (input_ids[B,T,4] i64, attention_mask[B,T] i64)."""
    ids = _rng(seed, "prompt").integers(0, num_text_tokens, size=(batch, length), dtype=np.int64)
    mask = np.ones((batch, length), dtype=np.int64)
    if pad_left is not None:
        for b, p in enumerate(pad_left):
            mask[b, :p] = 0
            ids[b, :p] = 0
    return np.repeat(ids[:, :, None], num_vq, axis=2).copy(), mask


def exp_noise(seed: int, step: int, rows: int, vocab: int) -> np.ndarray:
    """Exp(1) noise q[rows, vocab] for the exponential-race sampler (argmax p/q == multinomial)."""
    u = _rng(seed, f"noise.{step}").random(size=(rows, vocab), dtype=np.float64)
    return (-np.log1p(-u)).astype(np.float32).clip(min=np.float32(1e-30))


def speaker_vector(seed: int = 1234, dim: int = 768) -> np.ndarray:
    """Synthetic speaker embedding (real ones are 768 fp16 values, std ~4.8)."""
    return (_normal(seed, "speaker", (dim,), 4.8)).astype(np.float16).astype(np.float32)


# ---- a complete synthetic "checkpoint directory" + toy tokenizer: what the end-to-end tests, the 2-process sharding test and bench.py's
# ---- sharded-request leg build their ChatTTSPlusPipeline from (no ChatTTS checkpoint exists offline)
TOY_VOCAB = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]", "[Stts]", "[Ptts]", "[spk_emb]", "[empty_spk]", "[uv_break]", "[break_0]",
             "[Ebreak]", "[speed_5]", "[Sbreak]", "[Pbreak]", "a", "b", "c", "d"]


def toy_tokenizer(dir_path: str):
    """chatttsplus_amd.tokenizer.Tokenizer around a BertTokenizerFast over TOY_VOCAB (one token per word / special tag)."""
    import os
    from transformers import BertTokenizerFast
    from .tokenizer import Tokenizer
    os.makedirs(dir_path, exist_ok=True)
    vp = os.path.join(dir_path, "vocab.txt")
    with open(vp, "w") as f:
        f.write("\n".join(TOY_VOCAB))
    bt = BertTokenizerFast(vocab_file=vp, do_lower_case=False)
    bt.add_special_tokens({"additional_special_tokens": [v for v in TOY_VOCAB if v.startswith("[") and v not in ("[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]")]})
    return Tokenizer(tokenizer=bt)


def write_checkpoints(dir_path: str, seed: int = 1234, full: bool = True) -> str:
    """<dir>/asset/{GPT,Decoder,Vocos[,DVAE_full]}.pt with the synthetic real-size weights (torch.save of the state dicts); returns dir_path."""
    import os
    import torch
    os.makedirs(os.path.join(dir_path, "asset"), exist_ok=True)
    items = [("GPT.pt", gpt_state_dict(GPT_REAL, seed)), ("Decoder.pt", dvae_state_dict(DVAE_REAL, seed)), ("Vocos.pt", vocos_state_dict(VOCOS_REAL, seed))]
    if full:
        esd = dvae_encoder_state_dict(DVAE_ENC_REAL, seed)
        esd.update(dvae_full_decoder_state_dict(DVAE_FULL_DEC, seed))
        items.append(("DVAE_full.pt", esd))
    for name, sd in items:
        torch.save({k: torch.from_numpy(v) for k, v in sd.items()}, os.path.join(dir_path, "asset", name))
    return dir_path


def toy_texts(n: int, lo: int, hi: int, seed: int = 1234):
    """n texts of U{lo..hi} toy words ("a".."d"): one token per word."""
    r = _rng(seed, "toy_texts")
    lens = r.integers(lo, hi + 1, size=n)
    return [" ".join("abcd"[int(c)] for c in r.integers(0, 4, size=int(k))) for k in lens]
