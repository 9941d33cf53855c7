"""One-shot closure of the pins this repo cannot make offline (SURVEY 8c rows V1 / P3 and the N2 notes).

Run on a box that HAS the third-party packages the reference depends on and / or the real ChatTTS checkpoints:

    python tools/pin_third_party.py [--checkpoints DIR] [--gpu] [--json out.json]

Every section runs only when its package / file is present and reports `pinned`, `FAILED` or `absent`; the exit code is non-zero iff a present
package DISAGREES with this repo's restatement.  What each section settles:

  vocos                   oracle/ref_cpu.vocos_backbone + head against vocos.models.VocosBackbone / vocos.heads.ISTFTHead with the same state dict
                          (pipelines/chattts_plus_pipeline.py:93-111,303) -- the backbone wiring and state-dict keys, the one V1 caveat
  peft                    hip_models.GPT.add_lora's rule W += (alpha / r) B A and pipeline.load_lora_adapter's file layout against a LoraConfig adapter
                          saved by peft and merged by merge_and_unload (pipeline:420-432; configs/train/train_voice_clone_lora.yaml:72-80) -- row P3
  torchaudio              oracle mel_features against torchaudio.transforms.MelSpectrogram as models/dvae.py:184-191 builds it
  vector_quantize_pytorch oracle gfsq_indices / gfsq_latent_from_indices against GroupedResidualFSQ (dvae.py:66-96), BOTH pre_bound settings:
                          says which one the installed release implements (the repo defaults to pre_bound=True)
  zh_normalization        chatttsplus_amd.zh_numbers.read_numbers_zh against TextNormalizer on the number / date / clock forms it claims
  checkpoints             GPT.pt / Decoder.pt / DVAE_full.pt / Vocos.pt / spk_stat.pt / tokenizer.pt: key coverage against what the loaders expect, dtypes,
                          value ranges against the fp16 limits the fast mode and the split weight images assume; with --gpu, loaded through hip_models and
                          one short generate() + vocoder pass (finite output, zero saturation counter in fp32 mode)

tests/test_pin_third_party.py wraps the sections as skip-if-absent tests."""
import argparse
import importlib
import json
import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from chatttsplus_amd import synth  # noqa: E402


def have(mod: str) -> bool:
    try:
        importlib.import_module(mod)
        return True
    except Exception:
        return False


def _maxrel(a: torch.Tensor, b: torch.Tensor) -> float:
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))


def pin_vocos():
    if not have("vocos"):
        return dict(status="absent")
    from vocos.heads import ISTFTHead
    from vocos.models import VocosBackbone
    from oracle import ref_cpu
    cfg = synth.VOCOS_REAL
    sd = synth.vocos_state_dict(cfg, 1234)
    bb = VocosBackbone(input_channels=cfg["input_channels"], dim=cfg["dim"], intermediate_dim=cfg["intermediate_dim"], num_layers=cfg["num_layers"])
    hd = ISTFTHead(dim=cfg["dim"], n_fft=cfg["n_fft"], hop_length=cfg["hop_length"], padding="center")
    tsd = {k: torch.from_numpy(v) for k, v in sd.items()}
    missing_b = bb.load_state_dict({k[len("backbone."):]: v for k, v in tsd.items() if k.startswith("backbone.")}, strict=True)
    missing_h = hd.load_state_dict({k[len("head."):]: v for k, v in tsd.items() if k.startswith("head.")}, strict=True)
    mel = torch.from_numpy(synth._normal(7, "pin.mel", (1, 100, 173), 1.0))
    with torch.no_grad():
        feat_pkg = bb(mel)
        wav_pkg = hd(feat_pkg)[0]
    feat = ref_cpu.vocos_backbone(sd, mel[0])
    wav = ref_cpu.vocos_decode(sd, mel[0])
    e1 = _maxrel(feat.reshape(-1), feat_pkg.reshape(-1) if feat_pkg.numel() == feat.numel() else feat_pkg.transpose(1, 2).reshape(-1))
    e2 = _maxrel(wav.reshape(-1), wav_pkg.reshape(-1))
    ok = e1 <= 1e-4 and e2 <= 1e-4
    return dict(status="pinned" if ok else "FAILED", strict_state_dict=str((missing_b, missing_h)), backbone_rel_err=e1, waveform_rel_err=e2)


def pin_peft():
    if not (have("peft") and have("transformers")):
        return dict(status="absent")
    from peft import LoraConfig, get_peft_model
    from transformers import LlamaConfig, LlamaModel
    from chatttsplus_amd.pipeline import load_lora_adapter
    torch.manual_seed(0)
    cfg = LlamaConfig(hidden_size=768, intermediate_size=3072, num_attention_heads=12, num_hidden_layers=2, vocab_size=32, max_position_embeddings=64)
    base = LlamaModel(cfg)
    w0 = {n: p.detach().clone() for n, p in base.named_parameters()}
    model = get_peft_model(base, LoraConfig(r=8, lora_alpha=16, target_modules=["q_proj", "k_proj", "v_proj", "o_proj"], lora_dropout=0.0))   # train_voice_clone_lora.yaml:72-80
    for n, p in model.named_parameters():
        if "lora_B" in n:
            torch.nn.init.normal_(p, std=0.02)
    with tempfile.TemporaryDirectory() as td:
        model.save_pretrained(td)
        adapters = load_lora_adapter(td)                                    # this repo's reader of peft's on-disk layout (no peft import)
    merged = model.merge_and_unload()
    w1 = dict(merged.named_parameters())
    worst, n = 0.0, 0
    for (layer, target, A, B, scale) in adapters:
        key = f"layers.{layer}.self_attn.{target}.weight"
        want = w1[key].detach()
        got = w0[key] + float(scale) * torch.from_numpy(B) @ torch.from_numpy(A)
        worst = max(worst, _maxrel(got, want)); n += 1
    ok = n == 2 * 4 and worst <= 1e-6
    return dict(status="pinned" if ok else "FAILED", adapters_read=n, merge_rel_err=worst)


def pin_torchaudio():
    if not have("torchaudio"):
        return dict(status="absent")
    import torchaudio
    from oracle import ref_cpu
    mel = torchaudio.transforms.MelSpectrogram(sample_rate=24000, n_fft=1024, hop_length=256, n_mels=100, center=True, power=1)   # models/dvae.py:184-191
    wav = torch.from_numpy(synth.speaker_wave(3, 24000))
    with torch.no_grad():
        want = mel(wav[None])[0]
    got = ref_cpu.mel_features(wav)
    got = got if got.shape == want.shape else got.T
    e = _maxrel(got, want)
    return dict(status="pinned" if e <= 1e-4 else "FAILED", mel_rel_err=e, shape=list(want.shape))


def pin_gfsq():
    if not have("vector_quantize_pytorch"):
        return dict(status="absent")
    from vector_quantize_pytorch import GroupedResidualFSQ
    from oracle import ref_cpu
    q = GroupedResidualFSQ(dim=1024, levels=[5, 5, 5, 5], num_quantizers=2, groups=2).eval()      # dvae.py:66-81
    cfg = synth.DVAE_ENC_REAL
    sd = synth.dvae_encoder_state_dict(cfg, 1234)
    psd = q.state_dict()
    for k in list(psd):                                                  # project_in / project_out of the two groups from the synthetic checkpoint
        src = "vq_layer.quantizer." + k
        if src in sd:
            psd[k] = torch.from_numpy(sd[src])
    q.load_state_dict(psd)
    x = torch.from_numpy(synth._normal(11, "pin.gfsq", (1, 37, 1024), 1.0))
    with torch.no_grad():
        _, ind = q(x)                                                    # [G, B, T, R]
    res = {}
    for pb in (True, False):
        mine = ref_cpu.gfsq_indices(x[0], sd, pre_bound=pb)              # [G * R, T] in the reference's interleave
        want = ind.permute(0, 3, 1, 2).reshape(-1, x.shape[1]) if ind.dim() == 4 else ind
        res[f"pre_bound_{pb}_mismatch"] = float((mine.to(torch.int64) != want.to(torch.int64)).float().mean()) if mine.shape == want.shape else f"shape {tuple(mine.shape)} vs {tuple(want.shape)}"
    good = [k for k, v in res.items() if v == 0.0]
    res["status"] = "pinned" if good else "FAILED"
    res["installed_release_behaves_as"] = good[0] if good else None
    return res


def pin_zh():
    if not have("zh_normalization"):
        return dict(status="absent")
    from zh_normalization import TextNormalizer
    from chatttsplus_amd.zh_numbers import read_numbers_zh
    tn = TextNormalizer()
    cases = ["我有123个苹果", "今天是2024年5月17日", "现在是10:30", "价格是3.14元", "电话13800138000", "增长了50%", "-5度", "第3名", "100年"]
    bad = [(c, "".join(tn.normalize(c)), read_numbers_zh(c)) for c in cases if "".join(tn.normalize(c)) != read_numbers_zh(c)]
    return dict(status="pinned" if not bad else "FAILED", cases=len(cases), differences=bad[:5])


def pin_checkpoints(ckpt_dir, gpu):
    if not ckpt_dir or not os.path.isdir(ckpt_dir):
        return dict(status="absent")
    out = dict(status="pinned")
    asset = os.path.join(ckpt_dir, "asset") if os.path.isdir(os.path.join(ckpt_dir, "asset")) else ckpt_dir
    expect = {"GPT.pt": set(synth.gpt_state_dict(synth.GPT_REAL, 0)), "Decoder.pt": set(synth.dvae_state_dict(synth.DVAE_REAL, 0)),
              "Vocos.pt": set(synth.vocos_state_dict(synth.VOCOS_REAL, 0)),
              "DVAE_full.pt": set(synth.dvae_encoder_state_dict(synth.DVAE_ENC_REAL, 0)) | set(synth.dvae_full_decoder_state_dict(synth.DVAE_FULL_DEC, 0))}
    for name, keys in expect.items():
        p = os.path.join(asset, name)
        if not os.path.exists(p):
            out[name] = "absent"
            continue
        sd = torch.load(p, weights_only=True, map_location="cpu", mmap=True)
        have_k = set(sd)
        amax = max(float(v.abs().max()) for v in sd.values() if torch.is_tensor(v) and v.is_floating_point())
        out[name] = dict(missing_keys=sorted(keys - have_k)[:8], unexpected_keys=sorted(have_k - keys)[:8], dtypes=sorted({str(v.dtype) for v in sd.values() if torch.is_tensor(v)}),
                         abs_max=amax, fits_fp16_weights=amax < 65504.0, fits_split_images=(amax < 1023.0 if name == "GPT.pt" else amax < 255.0))
        if (keys - have_k) or not out[name]["fits_split_images"]:
            out["status"] = "FAILED"
    if gpu and torch.cuda.is_available():
        from chatttsplus_amd.pipeline import ChatTTSPlusPipeline, InferCodeParams, load_config
        cfg = load_config(os.path.join(ROOT, "configs", "infer", "chattts_plus_hip.yaml"))
        cfg["MODELS"]["gpt"]["kwargs"].update(max_batch=4, max_seq_len=768)
        pipe = ChatTTSPlusPipeline(cfg, device="cuda", checkpoint_dir=ckpt_dir)
        torch.manual_seed(2)
        wavs = list(pipe.infer(["四川美食确实以辣闻名，但也有不辣的选择。"], skip_refine_text=True, params_infer_code=InferCodeParams(max_new_token=256, show_tqdm=False)))[0]
        g = pipe.models_dict["gpt"]
        out["gpu_generate"] = dict(samples=[int(w.shape[0]) for w in wavs], finite=all(bool(torch.isfinite(w).all()) for w in wavs), saturations=int(g.saturations),
                                   persistent_rows=g.get_option("persistent_rows"))
        if not out["gpu_generate"]["finite"] or out["gpu_generate"]["saturations"]:
            out["status"] = "FAILED"
    return out


SECTIONS = dict(vocos=pin_vocos, peft=pin_peft, torchaudio=pin_torchaudio, vector_quantize_pytorch=pin_gfsq, zh_normalization=pin_zh)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--checkpoints", default=os.environ.get("CHATTTS_PLUS_CHECKPOINT_DIR"))
    ap.add_argument("--gpu", action="store_true")
    ap.add_argument("--json")
    a = ap.parse_args()
    res = {}
    for name, fn in SECTIONS.items():
        try:
            res[name] = fn()
        except Exception as e:                       # a present package that cannot be driven the way the reference drives it is a finding, not a crash
            res[name] = dict(status="FAILED", error=f"{type(e).__name__}: {e}")
    try:
        res["checkpoints"] = pin_checkpoints(a.checkpoints, a.gpu)
    except Exception as e:
        res["checkpoints"] = dict(status="FAILED", error=f"{type(e).__name__}: {e}")
    for k, v in res.items():
        print(f"{k:24s} {v['status']:8s} {json.dumps({kk: vv for kk, vv in v.items() if kk != 'status'}, default=str)[:300]}")
    if a.json:
        json.dump(res, open(a.json, "w"), indent=1, default=str)
    return 1 if any(v["status"] == "FAILED" for v in res.values()) else 0


if __name__ == "__main__":
    sys.exit(main())
