"""End-to-end drop-in check on the GPU: ChatTTSPlusPipeline.infer() with infer_type "hip" (synthetic checkpoints on
disk, a tiny BertTokenizerFast, a REAL bundled speaker string) against the oracle chain
GPT -> DVAE -> Vocos on the same tokens/speaker/seed.  fp32 parity mode: identical utterance lengths (== identical
token ids) and waveform RMS error <= 1e-3 of the signal RMS (north_star tolerance)."""
import os

import numpy as np
import pytest
import torch

from chatttsplus_amd import codec, synth
from oracle import ref_cpu
from tests.helpers import GOLDEN

pytestmark = pytest.mark.gpu

VOCAB = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]", "[Stts]", "[Ptts]", "[spk_emb]", "[empty_spk]", "[uv_break]", "[break_0]",
         "[Ebreak]", "[speed_5]", "a", "b", "c", "d"]


def _tokenizer(tmp_path):
    from transformers import BertTokenizerFast
    from chatttsplus_amd.tokenizer import Tokenizer
    (tmp_path / "vocab.txt").write_text("\n".join(VOCAB))
    bt = BertTokenizerFast(vocab_file=str(tmp_path / "vocab.txt"), do_lower_case=False)
    bt.add_special_tokens({"additional_special_tokens": [v for v in VOCAB if v.startswith("[") and v not in ("[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]")]})
    return Tokenizer(tokenizer=bt)


def test_pipeline_infer_matches_oracle_chain(tmp_path):
    from chatttsplus_amd.pipeline import ChatTTSPlusPipeline, InferCodeParams, load_config
    cfg = load_config(os.path.join(os.path.dirname(GOLDEN), "..", "configs", "infer", "chattts_plus_hip.yaml"))
    cfg["MODELS"]["gpt"]["kwargs"].update(weight_dtype="fp32", max_batch=4, max_seq_len=256)
    os.makedirs(tmp_path / "asset")
    gsd = synth.gpt_state_dict(synth.GPT_REAL, 1234)
    dsd = synth.dvae_state_dict(synth.DVAE_REAL, 1234)
    vsd = synth.vocos_state_dict(synth.VOCOS_REAL, 1234)
    for name, sd in (("GPT.pt", gsd), ("Decoder.pt", dsd), ("Vocos.pt", vsd)):
        torch.save({k: torch.from_numpy(v) for k, v in sd.items()}, tmp_path / "asset" / name)
    tok = _tokenizer(tmp_path)
    pipe = ChatTTSPlusPipeline(cfg, device="cuda", tokenizer=tok, checkpoint_dir=str(tmp_path))
    spk = torch.load(os.path.join(GOLDEN, "speakers", "2222.pt"), weights_only=True)       # base16384 string
    params = InferCodeParams(prompt="[speed_5]", spk_emb=spk, max_new_token=40, min_new_token=2, show_tqdm=False)
    texts = ["a b c d a b", "c a"]
    torch.manual_seed(11)
    outs = list(pipe.infer(list(texts), skip_refine_text=True, do_text_optimization=False, params_infer_code=params))
    assert len(outs) == 1 and len(outs[0]) == 2
    wavs = [w.cpu().numpy() for w in outs[0]]

    # oracle chain on the same inputs
    wrapped = [f"[Stts][spk_emb][speed_5]{t} [uv_break][Ptts]" for t in texts]
    ids, att, tm = tok.encode(wrapped, 4)
    o = ref_cpu.OracleGPT(gsd, 12)
    emb = o.apply_spk_emb(o.embed(ids, tm), torch.from_numpy(codec.decode_spk_emb(spk)), ids, tok.spk_emb_ids)
    torch.manual_seed(11)
    ref = o.generate(emb, ids, ref_cpu.SamplerParams(min_new_token=2), attention_mask=att, max_new_token=40)
    for b in range(2):
        n = ref.ids[b].shape[0]
        assert wavs[b].shape[0] == 256 * (2 * n - 1), f"utterance {b}: {wavs[b].shape[0]} samples vs {n} reference tokens"
        mel = ref_cpu.dvae_decode(dsd, ref.hiddens[b])
        wav_ref = ref_cpu.vocos_decode(vsd, mel).numpy()
        rms = float(np.sqrt(np.mean((wavs[b] - wav_ref) ** 2))) / float(np.sqrt(np.mean(wav_ref ** 2)))
        assert rms <= 1e-3, f"utterance {b}: waveform rms-rel {rms}"


def test_pipeline_rejects_unserved_paths(tmp_path):
    from chatttsplus_amd import _lib
    from chatttsplus_amd.pipeline import ChatTTSPlusPipeline
    pipe = object.__new__(ChatTTSPlusPipeline)
    pipe.normalizer = lambda t, *a, **k: t
    pipe.text_splitter = None
    with pytest.raises(_lib.HipBackendError):
        next(pipe._infer(["x"], skip_refine_text=False))


def test_lora_merge_matches_oracle():
    """BASELINE config 5: LoRA (r=8, alpha=16 -> scale 2.0) on q/k/v/o merged into the packed weights (pipeline:420-432)."""
    from chatttsplus_amd.hip_models import GPT
    cfg = dict(synth.GPT_REAL); cfg["num_hidden_layers"] = 3
    sd = synth.gpt_state_dict(cfg, 1234)
    rng = np.random.Generator(np.random.Philox(key=31))
    adapters = []
    merged = {k: v.copy() for k, v in sd.items()}
    for l in range(3):
        for t in ("q_proj", "k_proj", "v_proj", "o_proj"):
            A = (rng.standard_normal((8, 768)) * 0.05).astype(np.float32); B = (rng.standard_normal((768, 8)) * 0.05).astype(np.float32)
            adapters.append((l, t, A, B, 2.0))
            merged[f"gpt.layers.{l}.self_attn.{t}.weight"] = (merged[f"gpt.layers.{l}.self_attn.{t}.weight"] + 2.0 * (B @ A)).astype(np.float32)
    llama = dict(hidden_size=768, intermediate_size=3072, num_attention_heads=12, num_hidden_layers=3)
    base = GPT(llama, max_batch=2, max_seq_len=64, weight_dtype="fp32")
    base.load_state_dict(sd)
    g = base.with_lora(adapters)
    ids, mask = synth.prompt_ids(2, 10, cfg["num_text_tokens"], 9, pad_left=[0, 2])
    o = ref_cpu.OracleGPT(merged, 12)
    emb = o.embed(torch.from_numpy(ids), torch.ones(2, 10, dtype=torch.bool))
    lw = [type("P", (), dict(top_p=0.7, min_tokens_to_keep=3))(), type("K", (), dict(top_k=20))()]
    lp = [type("R", (), dict(penalty=1.05, past_window=16, max_input_ids=625))()]
    torch.manual_seed(2)
    ref = o.generate(emb, torch.from_numpy(ids), ref_cpu.SamplerParams(min_new_token=8), attention_mask=torch.from_numpy(mask), max_new_token=8)
    torch.manual_seed(2)
    out = list(g.generate(emb.cuda(), torch.from_numpy(ids), torch.tensor([0.3] * 4), 625, attention_mask=torch.from_numpy(mask), max_new_token=8,
                          min_new_token=8, logits_warpers=lw, logits_processors=lp, return_hidden=True))[-1]
    for b in range(2):
        assert torch.equal(out.ids[b].cpu(), ref.ids[b])
        assert float((out.hiddens[b].cpu() - ref.hiddens[b]).abs().max()) < 1e-4
