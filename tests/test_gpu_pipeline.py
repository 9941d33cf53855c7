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


def _make_pipe(tmp_path, max_batch=8, max_seq_len=256):
    """ChatTTSPlusPipeline on synthetic checkpoints written to tmp_path (the drop-in path: YAML -> infer_type "hip" -> hip_models)."""
    from chatttsplus_amd.pipeline import ChatTTSPlusPipeline, load_config
    cfg = load_config(os.path.join(os.path.dirname(GOLDEN), "..", "configs", "infer", "chattts_plus_hip.yaml"))
    cfg["MODELS"]["gpt"]["kwargs"].update(weight_dtype="fp32", max_batch=max_batch, max_seq_len=max_seq_len)
    os.makedirs(tmp_path / "asset")
    gsd = synth.gpt_state_dict(synth.GPT_REAL, 1234)
    dsd = synth.dvae_state_dict(synth.DVAE_REAL, 1234)
    vsd = synth.vocos_state_dict(synth.VOCOS_REAL, 1234)
    esd = synth.dvae_encoder_state_dict(synth.DVAE_ENC_REAL, 1234)
    esd.update(synth.dvae_full_decoder_state_dict(synth.DVAE_FULL_DEC, 1234))        # DVAE_full.pt = encode side + decode side + quantiser (one coef)
    for name, sd in (("GPT.pt", gsd), ("Decoder.pt", dsd), ("Vocos.pt", vsd), ("DVAE_full.pt", esd)):
        torch.save({k: torch.from_numpy(v) for k, v in sd.items()}, tmp_path / "asset" / name)
    tok = _tokenizer(tmp_path)
    pipe = ChatTTSPlusPipeline(cfg, device="cuda", tokenizer=tok, checkpoint_dir=str(tmp_path))
    return pipe, tok, dict(gpt=gsd, dvae=dsd, vocos=vsd, enc=esd)


def test_vocoder_overlap_soak(tmp_path):
    """continuous="throughput" with overlap_vocoder=True (opt-in): finished utterances are vocoded in batches of `vocoder_chunk` on a SIDE stream while the remaining rows keep
    decoding -- here on 3 / 5 rows, i.e. on PERSISTENT launches that need all 256 workgroups resident while cnx_gemm_kernel batches hold CUs (the serial
    counterpart is _decode_to_wavs, pipeline:435-457).  14 ragged utterances, vocoder_chunk=2 -> >= 6 vocoder batches start while rows are still stepping.  Every waveform
    must equal the overlap-free run, no bounded wait may give up (pl_state.error == 0), and the whole request is repeated (soak)."""
    import ctypes as C
    import json
    from chatttsplus_amd.pipeline import InferCodeParams
    pipe, tok, _ = _make_pipe(tmp_path)
    spk = torch.load(os.path.join(GOLDEN, "speakers", "2222.pt"), weights_only=True)
    texts = ["a b c d a b", "c a", "b", "d d c", "a b", "c c c c a", "b a d", "a", "b b a c d", "c d", "a a a b", "d", "c b a", "b d d a c c"]
    lims = [40, 9, 14, 22, 6, 31, 12, 5, 36, 8, 17, 4, 11, 27]            # ragged per-utterance token limits: rows finish (and are re-used) at different steps
    pv = InferCodeParams(prompt="[speed_5]", spk_emb=spk, max_new_token=40, min_new_token=40, show_tqdm=False)
    gpt = pipe.models_dict["gpt"]
    assert gpt.get_option("persistent_rows") == 8

    def run(rows, **kw):
        res = list(pipe.infer(list(texts), skip_refine_text=True, do_text_optimization=False, params_infer_code=pv, noise="device", noise_seed=31, slice_size=rows,
                              continuous="throughput", max_new_tokens_per_utterance=lims, **kw))
        assert len(res) == 1 and len(res[0]) == len(texts)
        return [w.cpu().numpy() for w in res[0]]

    def pl_error():
        buf = (C.c_uint * 2)()
        n = C.c_size_t(0)
        from chatttsplus_amd import _lib
        _lib.check(gpt._lib.ctts_gpt_debug_read(gpt._h, b"pl_state", buf, 8, C.byref(n), None), "debug_read")
        return int(buf[1])

    loops, worst, t_overlap = 0, 0.0, []
    for rows in (3, 5):
        ref = run(rows)
        assert [w.shape[0] for w in ref] == [256 * (2 * n - 1) for n in lims]
        for it in range(26):
            got = run(rows, overlap_vocoder=True, vocoder_chunk=2)
            loops += 1
            assert pl_error() == 0, f"rows={rows} loop {it}: a persistent launch gave up waiting while the vocoder held CUs"
            for u in range(len(texts)):
                a, b = ref[u], got[u]
                assert a.shape == b.shape, (rows, it, u)
                rel = float(np.sqrt(np.mean((a - b) ** 2))) / float(np.sqrt(np.mean(a ** 2)))
                worst = max(worst, rel)
                assert rel <= 1e-4, (rows, it, u, rel)
            t_overlap.append(getattr(pipe, "last_first_audio_ms", None))
    # a non-positive chunk must not loop forever: clamped to 1
    one = run(3, overlap_vocoder=True, vocoder_chunk=0)
    assert all(a.shape == b.shape for a, b in zip(one, ref))
    out = os.path.join(os.path.dirname(GOLDEN), "..", "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "r06_overlap_soak.json"), "w") as f:
            json.dump(dict(loops=loops, utterances=len(texts), rows=[3, 5], vocoder_chunk=2, worst_rel_rms=worst, give_ups=0,
                           first_vocoder_batch_ms=[t for t in t_overlap if t is not None][:4]), f)


def test_pipeline_infer_matches_oracle_chain(tmp_path):
    from chatttsplus_amd.pipeline import InferCodeParams
    pipe, tok, sds = _make_pipe(tmp_path)
    gsd, dsd, vsd, esd = sds["gpt"], sds["dvae"], sds["vocos"], sds["enc"]
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

    # use_decoder=False (pipeline:292,435-439): the DVAE_full model decodes the generated CODE IDS (GFSQ._embed + its own decoder stack)
    torch.manual_seed(11)
    outs_c = list(pipe.infer(list(texts), skip_refine_text=True, do_text_optimization=False, use_decoder=False, params_infer_code=params))
    assert len(outs_c) == 1 and len(outs_c[0]) == 2
    for b in range(2):
        wav_ref = ref_cpu.vocos_decode(vsd, ref_cpu.dvae_decode_codes(esd, ref.ids[b])).numpy()
        w = outs_c[0][b].cpu().numpy()
        assert w.shape == wav_ref.shape, f"use_decoder=False, utterance {b}: {w.shape} vs {wav_ref.shape}"
        rms = float(np.sqrt(np.mean((w - wav_ref) ** 2))) / float(np.sqrt(np.mean(wav_ref ** 2)))
        assert rms <= 1e-3, f"use_decoder=False, utterance {b}: waveform rms-rel {rms}"
        assert not np.allclose(w, wavs[b], atol=1e-3)                      # really a different decoder

    # stream=True: the yields are consecutive sample windows of the prefix waveform; their total length is the final length
    sp = InferCodeParams(prompt="[speed_5]", spk_emb=spk, max_new_token=40, min_new_token=2, show_tqdm=False, stream_batch=8,
                         stream_speed=3000, pass_first_n_batches=1)
    torch.manual_seed(11)
    chunks = list(pipe.infer(list(texts), stream=True, skip_refine_text=True, do_text_optimization=False, params_infer_code=sp))
    assert len(chunks) >= 2 and all(c.shape[0] == 2 for c in chunks)
    assert sum(int(c.shape[1]) for c in chunks) == max(w.shape[0] for w in wavs)
    tail = chunks[-1].cpu().numpy()
    for b in range(2):                              # the last window is vocoded from the complete utterance: equals the tail of infer()
        n_tail = tail.shape[1] - (max(w.shape[0] for w in wavs) - wavs[b].shape[0])
        if n_tail > 0:
            assert np.array_equal(tail[b, :n_tail], wavs[b][wavs[b].shape[0] - n_tail:])

    # zero shot (pipeline:486-499): a 16 kHz stereo clip -> mono 24 kHz -> DVAE encoder -> audio-prompt codes in front of the text
    from scipy.io import wavfile
    from chatttsplus_amd import audio
    clip = synth.speaker_wave(9, 16000)
    wavfile.write(tmp_path / "spk.wav", 16000, np.stack([clip, 0.5 * clip], 1))
    mono = torch.mean(audio.resample(torch.from_numpy(np.stack([clip, 0.5 * clip], 0)), 16000, 24000), 0)
    smp = pipe.sample_audio_speaker(mono)
    codes = codec.decode_prompt(smp)
    assert codes.shape == (4, ((1 + 24000 // 256) - 2) // 2 + 1) and int(codes.max()) < 625
    ref_codes = ref_cpu.dvae_encode(esd, mono)
    assert float((codes != ref_codes.to(torch.int64)).float().mean()) <= 0.01
    zs = list(pipe.infer(list(texts), skip_refine_text=True, do_text_optimization=False, speaker_audio_path=str(tmp_path / "spk.wav"),
                         speaker_audio_text="a b", params_infer_code=InferCodeParams(max_new_token=12, min_new_token=12, show_tqdm=False)))
    assert len(zs) == 1 and [w.shape[0] for w in zs[0]] == [256 * 23, 256 * 23] and all(bool(torch.isfinite(w).all()) for w in zs[0])

    # infer_sharded with no process group (world 1) on the real engine: per-utterance speaker rows from a table; every utterance must
    # equal the plain infer() run with its own speaker and the same noise key (request seed, utterance id 0)
    table = torch.stack([codec.speaker_to_vector(spk), torch.from_numpy(synth.speaker_vector(5))], 0)
    p1 = InferCodeParams(prompt="[speed_5]", max_new_token=10, min_new_token=10, show_tqdm=False)
    for which in (0, 1):
        torch.manual_seed(5)
        mine, sw, lens = pipe.infer_sharded(["a b c d"], speaker_index=[which], speaker_table=table, params_infer_code=p1, noise_seed=0)
        torch.manual_seed(5)
        plain = list(pipe.infer(["a b c d"], skip_refine_text=True, do_text_optimization=False, noise="device", noise_seed=0,
                                params_infer_code=InferCodeParams(prompt="[speed_5]", spk_emb=table[which], max_new_token=10, min_new_token=10, show_tqdm=False)))[0]
        assert mine == [0] and lens == [10] and torch.equal(sw[0], plain[0]), f"speaker {which}"
    mine, sw, lens = pipe.infer_sharded(["a b", "c d a", "b"], speaker_index=[1, 0, 1], speaker_table=table, params_infer_code=p1)
    assert mine == [0, 1, 2] and lens == [10, 10, 10] and [int(w.shape[0]) for w in sw] == [256 * 19] * 3

    # Partition invariance (what an N-rank infer_sharded needs to equal the 1-rank result): the device noise of an
    # utterance is keyed by (request seed, global utterance id, its own step / attempt), so slices of 2, 4 or all 7 utterances -- different
    # batch rows, different left padding, different kernels (split-K path at <= 4 rows) -- produce the same tokens; waveforms agree to
    # rounding.  The sharded entry point (world 1) gives the same again.
    many = ["a b c d a b", "c a", "b", "d d c", "a b", "c c c c a", "b a d"]
    pv = InferCodeParams(prompt="[speed_5]", spk_emb=spk, max_new_token=24, min_new_token=3, show_tqdm=False)
    runs = {}
    for ss in (2, 4, 8):
        got = []
        for w in pipe.infer(list(many), skip_refine_text=True, do_text_optimization=False, params_infer_code=pv, noise="device", noise_seed=77, slice_size=ss):
            got.extend(w)
        runs[ss] = [w.cpu().numpy() for w in got]
    _, sw, lens = pipe.infer_sharded(list(many), params_infer_code=pv, noise_seed=77, slice_size=3)
    runs["sharded"] = [w.cpu().numpy() for w in sw]
    _, sw, lens = pipe.infer_sharded(list(many), params_infer_code=pv, noise_seed=77, slice_size=3, continuous=True)
    runs["sharded_continuous"] = [w.cpu().numpy() for w in sw]
    # continuous batching (infer(continuous=True): 7 utterances through 2 / 3 decode rows, rows re-used as utterances finish): same waveforms
    for rows in (2, 3):
        res = list(pipe.infer(list(many), skip_refine_text=True, do_text_optimization=False, params_infer_code=pv, noise="device", noise_seed=77,
                              slice_size=rows, continuous=True))
        assert len(res) >= 2 and sum(len(r) for r in res) == len(many)      # waveforms arrive in input order, the first list before the rest is done
        runs[f"continuous{rows}"] = [w.cpu().numpy() for r in res for w in r]
    res = list(pipe.infer(list(many), skip_refine_text=True, do_text_optimization=False, params_infer_code=pv, noise="device", noise_seed=77,
                          slice_size=3, continuous="throughput"))
    assert len(res) == 1 and len(res[0]) == len(many)                       # longest texts first, one list at the end
    runs["continuous_lpt"] = [w.cpu().numpy() for w in res[0]]
    for key in (4, 8, "sharded", "sharded_continuous", "continuous2", "continuous3", "continuous_lpt"):
        for u in range(len(many)):
            a, b2 = runs[2][u], runs[key][u]
            assert a.shape == b2.shape, f"slice {key}, utterance {u}: {a.shape} vs {b2.shape} samples (token count differs)"
            assert float(np.sqrt(np.mean((a - b2) ** 2))) <= 1e-4 * float(np.sqrt(np.mean(a ** 2))), f"slice {key}, utterance {u}"

    # stream=True + continuous (round 4): (utterance, sample window) pairs while rows are re-used; per utterance the windows are consecutive, add up to the
    # utterance's waveform length and the last one (vocoded from the complete utterance) equals the tail of the non-streamed waveform
    sp3 = InferCodeParams(prompt="[speed_5]", spk_emb=spk, max_new_token=24, min_new_token=3, show_tqdm=False, stream_batch=6, stream_speed=3000)
    per = {}
    for chunk in pipe.infer(list(many), stream=True, skip_refine_text=True, do_text_optimization=False, params_infer_code=sp3, noise_seed=77, slice_size=3, continuous=True):
        for u, w in chunk:
            per.setdefault(u, []).append(w.cpu().numpy())
    assert sorted(per) == list(range(len(many)))
    for u, ws in per.items():
        ref = runs[4][u]
        assert sum(w.shape[0] for w in ws) == ref.shape[0], (u, [w.shape[0] for w in ws], ref.shape)
        assert np.allclose(ws[-1], ref[ref.shape[0] - ws[-1].shape[0]:], rtol=0, atol=1e-4 * float(np.abs(ref).max())), u
    assert any(len(ws) > 1 for ws in per.values())

    # default infer() path: refine-text pass first (pipeline:399-411), then code inference on the refined text
    from chatttsplus_amd.pipeline import RefineTextParams
    rp = RefineTextParams(max_new_token=6, show_tqdm=False)
    only = list(pipe.infer(list(texts), skip_refine_text=False, refine_text_only=True, do_text_optimization=False, params_refine_text=rp,
                           params_infer_code=InferCodeParams(spk_emb=spk, max_new_token=8, show_tqdm=False)))
    assert len(only) == 1 and len(only[0]) == 2 and all(isinstance(t, str) for t in only[0])
    full = list(pipe.infer(list(texts), skip_refine_text=False, do_text_optimization=False, params_refine_text=rp,
                           params_infer_code=InferCodeParams(spk_emb=spk, max_new_token=8, min_new_token=8, show_tqdm=False)))
    assert len(full) == 1 and [w.shape[0] for w in full[0]] == [256 * 15, 256 * 15]
    # continuous=True also serves the refine-text pass through row re-use (round 4): 7 sentences on 2 and on 3 decode rows refine to the same texts
    # (noise keyed by utterance id on stream 4), and the synthesis that follows keeps its waveform count
    rp2 = RefineTextParams(max_new_token=9, show_tqdm=False)
    pc = InferCodeParams(spk_emb=spk, max_new_token=6, min_new_token=6, show_tqdm=False)
    ref2 = list(pipe.infer(list(many), refine_text_only=True, do_text_optimization=False, params_refine_text=rp2, params_infer_code=pc, continuous=True, slice_size=2, noise_seed=5))
    ref3 = list(pipe.infer(list(many), refine_text_only=True, do_text_optimization=False, params_refine_text=rp2, params_infer_code=pc, continuous=True, slice_size=3, noise_seed=5))
    assert len(ref2) == 1 and len(ref2[0]) == len(many) and ref2 == ref3, (ref2, ref3)
    both = list(pipe.infer(list(many), do_text_optimization=False, params_refine_text=rp2, params_infer_code=pc, continuous=True, slice_size=3, noise_seed=5))
    assert sum(len(l) for l in both) == len(many) and all(w.shape[0] == 256 * 11 for l in both for w in l)

    # default text optimisation (text_frontend.split_text + short-sentence merge + Normalizer, pipeline:349-388): two short lines become ONE
    # utterance joined by [uv_break] -- identical to handing that utterance over with the optimisation switched off
    p2 = InferCodeParams(prompt="[speed_5]", spk_emb=spk, max_new_token=12, min_new_token=12, show_tqdm=False)
    torch.manual_seed(3)
    merged = list(pipe.infer(["a b c", "d a"], skip_refine_text=True, params_infer_code=p2))
    torch.manual_seed(3)
    direct = list(pipe.infer(["a b c [uv_break] d a [uv_break] "], skip_refine_text=True, do_text_optimization=False, params_infer_code=p2))
    assert len(merged) == 1 and len(merged[0]) == 1 and merged[0][0].shape[0] == 256 * 23 and torch.equal(merged[0][0], direct[0][0])


def test_refine_text_generate_golden_bit_exact():
    """infer_text=True on the HIP path (21178-way head, emb_text re-embed, 1024-thread text sampler) reproduces the token ids
    of the imported reference under the same torch seed (fp32 parity mode), including a ragged early-EOS batch."""
    from chatttsplus_amd.hip_models import GPT
    from chatttsplus_amd.pipeline import gen_logits
    from tests.helpers import load_golden
    for name in ("gpt_real_text_b2", "gpt_real_text_eos"):
        z, meta = load_golden(name)
        sd = synth.gpt_state_dict(synth.GPT_REAL, int(meta["weight_seed"]))
        eos = int(meta["eos"])
        sd["head_text.parametrizations.weight.original0"][eos] *= float(meta["eos_boost"])
        g = GPT(dict(hidden_size=768, intermediate_size=3072, num_attention_heads=12, num_hidden_layers=20), max_batch=4, max_seq_len=128, weight_dtype="fp32")
        g.load_state_dict(sd)
        B, T = int(meta["B"]), int(meta["T"])
        ids, mask = synth.prompt_ids(B, T, 21178, int(meta["prompt_seed"]), pad_left=[int(x) for x in meta["pad_left"]])
        emb = g(torch.from_numpy(ids), torch.ones(B, T, dtype=torch.bool))
        w, p = gen_logits(21178, 0.7, 20, 1.0)
        torch.manual_seed(int(meta["torch_seed"]))
        out = next(g.generate(emb, torch.from_numpy(ids), torch.tensor([0.7]), eos, attention_mask=torch.from_numpy(mask), max_new_token=int(meta["max_new"]),
                              min_new_token=int(meta["min_new"]), logits_warpers=w, logits_processors=p, infer_text=True, noise="torch"))
        assert [int(i.shape[0]) for i in out.ids] == z["lens"].tolist(), name
        for b, n in enumerate(z["lens"]):
            assert out.ids[b].dim() == 1 and np.array_equal(out.ids[b].cpu().numpy(), z["ids"][b, :n].astype(np.int64)), f"{name} row {b}"
        with pytest.raises(Exception):
            w2, p2 = gen_logits(21178, 0.7, 20, 1.2)
            next(g.generate(emb, torch.from_numpy(ids), torch.tensor([0.7]), eos, max_new_token=4, logits_warpers=w2, logits_processors=p2, infer_text=True))
        del g
    # the same pass in the device-noise mode (the pipeline's default for it): fixture minted by the reference's generate with
    # multinomial = argmax(p / q) on stream 4 of the device noise (oracle/make_golden.py golden_refine_text_device_noise)
    z, meta = load_golden("gpt_real_text_device_noise")
    sd = synth.gpt_state_dict(synth.GPT_REAL, int(meta["weight_seed"]))
    eos = int(meta["eos"])
    sd["head_text.parametrizations.weight.original0"][eos] *= float(meta["eos_boost"])
    g = GPT(dict(hidden_size=768, intermediate_size=3072, num_attention_heads=12, num_hidden_layers=20), max_batch=4, max_seq_len=128, weight_dtype="fp32")
    g.load_state_dict(sd)
    B, T = int(meta["B"]), int(meta["T"])
    ids, mask = synth.prompt_ids(B, T, 21178, int(meta["prompt_seed"]), pad_left=[int(x) for x in meta["pad_left"]])
    emb = g(torch.from_numpy(ids), torch.ones(B, T, dtype=torch.bool))
    w, p = gen_logits(21178, 0.7, 20, 1.0)
    out = next(g.generate(emb, torch.from_numpy(ids), torch.tensor([0.7]), eos, attention_mask=torch.from_numpy(mask), max_new_token=int(meta["max_new"]),
                          min_new_token=int(meta["min_new"]), logits_warpers=w, logits_processors=p, infer_text=True, noise="device",
                          seed=int(meta["noise_seed"]), utt_ids=[int(u) for u in meta["utt_ids"]]))
    assert [int(i.shape[0]) for i in out.ids] == z["lens"].tolist()
    for b, n in enumerate(z["lens"]):
        assert np.array_equal(out.ids[b].cpu().numpy(), z["ids"][b, :n].astype(np.int64)), f"device noise, row {b}"
    # ... and through continuous batching (round 4: text rows carry the same per-row state as code rows, ctts_gpt_admit serves infer_text): the three
    # sentences on TWO decode rows -- the third takes over the row of whichever finishes first -- and on one row; every sentence keeps the reference's ids
    for rows in (2, 1):
        many = g.generate_many(emb, torch.from_numpy(ids), torch.tensor([0.7]), eos, attention_mask=torch.from_numpy(mask), max_new_token=int(meta["max_new"]),
                               min_new_token=int(meta["min_new"]), logits_warpers=w, logits_processors=p, seed=int(meta["noise_seed"]),
                               utt_ids=[int(u) for u in meta["utt_ids"]], rows=rows, infer_text=True)
        assert g.admissions, "nothing was admitted"
        assert [int(i.shape[0]) for i in many.ids] == z["lens"].tolist(), rows
        for b, n in enumerate(z["lens"]):
            assert many.ids[b].dim() == 1 and np.array_equal(many.ids[b].cpu().numpy(), z["ids"][b, :n].astype(np.int64)), f"generate_many on {rows} row(s), sentence {b}"
    del g


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


def test_pipeline_lora_path_end_to_end_batch32(tmp_path):
    """BASELINE configs[4]: a peft adapter directory on disk (adapter_config.json + adapter_model.safetensors, r=8, alpha=16 on
    q/k/v/o of every layer -- configs/train/train_voice_clone_lora.yaml:72-80) served through pipeline.infer(lora_path=...) at
    batch 32, against the oracle with algebraically merged weights (peft merge_and_unload: W + (alpha/r) B A, pipeline:420-432;
    peft itself is absent offline: parity unpinned, pinned algebraically).  fp32 parity mode: identical utterance lengths
    (== identical token ids) and waveform RMS error <= 1e-3.  Also: the base engine is untouched afterwards and the merged
    engines are LRU-bounded."""
    import json
    from safetensors.numpy import save_file
    from chatttsplus_amd.pipeline import ChatTTSPlusPipeline, InferCodeParams, load_config
    cfg = load_config(os.path.join(os.path.dirname(GOLDEN), "..", "configs", "infer", "chattts_plus_hip.yaml"))
    cfg["MODELS"]["gpt"]["kwargs"].update(weight_dtype="fp32", max_batch=32, max_seq_len=128)
    os.makedirs(tmp_path / "asset")
    gsd = synth.gpt_state_dict(synth.GPT_REAL, 1234)
    dsd = synth.dvae_state_dict(synth.DVAE_REAL, 1234)
    vsd = synth.vocos_state_dict(synth.VOCOS_REAL, 1234)
    for name, sd in (("GPT.pt", gsd), ("Decoder.pt", dsd), ("Vocos.pt", vsd)):
        torch.save({k: torch.from_numpy(v) for k, v in sd.items()}, tmp_path / "asset" / name)
    # two adapters on disk in peft's layout
    merged = {}
    for ai, seed in enumerate((31, 32)):
        rng = np.random.Generator(np.random.Philox(key=seed))
        d = tmp_path / f"lora{ai}"
        os.makedirs(d)
        tensors = {}
        m = {k: v.copy() for k, v in gsd.items()}
        for l in range(20):
            for t in ("q_proj", "k_proj", "v_proj", "o_proj"):
                A = (rng.standard_normal((8, 768)) * 0.05).astype(np.float32); Bm = (rng.standard_normal((768, 8)) * 0.05).astype(np.float32)
                tensors[f"base_model.model.layers.{l}.self_attn.{t}.lora_A.weight"] = A
                tensors[f"base_model.model.layers.{l}.self_attn.{t}.lora_B.weight"] = Bm
                m[f"gpt.layers.{l}.self_attn.{t}.weight"] = (m[f"gpt.layers.{l}.self_attn.{t}.weight"] + 2.0 * (Bm @ A)).astype(np.float32)
        save_file(tensors, str(d / "adapter_model.safetensors"))
        (d / "adapter_config.json").write_text(json.dumps(dict(r=8, lora_alpha=16, target_modules=["q_proj", "k_proj", "v_proj", "o_proj"], peft_type="LORA")))
        merged[ai] = m
    tok = _tokenizer(tmp_path)
    pipe = ChatTTSPlusPipeline(cfg, device="cuda", tokenizer=tok, checkpoint_dir=str(tmp_path))
    spk = torch.load(os.path.join(GOLDEN, "speakers", "2222.pt"), weights_only=True)
    rng = np.random.Generator(np.random.Philox(key=5))
    texts = [" ".join("abcd"[int(c)] for c in rng.integers(0, 4, size=int(n))) for n in rng.integers(2, 12, size=32)]
    N = 10
    params = InferCodeParams(prompt="[speed_5]", spk_emb=spk, max_new_token=N, min_new_token=N, show_tqdm=False)
    q = torch.from_numpy(np.stack([synth.exp_noise(21, i, 4 * 32, 626) for i in range(N)]))

    def run(lora):
        gpt = pipe._gpt_for_lora(lora)
        g0 = gpt.generate
        gpt.generate = lambda *a, **k: g0(*a, **dict({kk: v for kk, v in k.items() if kk not in ("noise", "seed", "utt_ids")}, noise=q))   # fixed noise rows: comparable with the oracle at any batch size
        try:
            return list(pipe.infer(list(texts), skip_refine_text=True, do_text_optimization=False, params_infer_code=params, lora_path=lora))[0]
        finally:
            gpt.generate = g0

    wrapped = [f"[Stts][spk_emb][speed_5]{t} [uv_break][Ptts]" for t in texts]
    ids, att, tm = tok.encode(wrapped, 4)
    for ai in (0, 1):
        wavs = run(str(tmp_path / f"lora{ai}"))
        o = ref_cpu.OracleGPT(merged[ai], 12)
        emb = o.apply_spk_emb(o.embed(ids, tm), torch.from_numpy(codec.decode_spk_emb(spk)), ids, tok.spk_emb_ids)
        ref = o.generate(emb, ids, ref_cpu.SamplerParams(min_new_token=N), attention_mask=att, max_new_token=N, noise=ref_cpu.ArrayNoise(q))
        for b in (0, 5, 17, 31):
            assert wavs[b].shape[0] == 256 * (2 * ref.ids[b].shape[0] - 1)
            wav_ref = ref_cpu.vocos_decode(vsd, ref_cpu.dvae_decode(dsd, ref.hiddens[b])).numpy()
            w = wavs[b].cpu().numpy()
            rms = float(np.sqrt(np.mean((w - wav_ref) ** 2))) / float(np.sqrt(np.mean(wav_ref ** 2)))
            assert rms <= 1e-3, f"adapter {ai} utterance {b}: waveform rms-rel {rms}"
    assert len(pipe._lora_models) == 1                               # LRU of one: adapter 0's engine was destroyed when adapter 1 came in
    # per-utterance adapters (lora_paths) in slices of 4 vs continuous batching on 4 rows: an admitted utterance brings its own adapter
    # (ctts_gpt_admit_adapters) and gets the waveform the sliced path gives it
    paths = [(str(tmp_path / "lora0"), None, str(tmp_path / "lora1"))[i % 3] for i in range(14)]
    p2 = InferCodeParams(prompt="[speed_5]", spk_emb=spk, max_new_token=20, min_new_token=3, show_tqdm=False)
    kw = dict(skip_refine_text=True, do_text_optimization=False, params_infer_code=p2, lora_paths=paths, slice_size=4, noise="device", noise_seed=9,
              max_new_tokens_per_utterance=[4 + (5 * i) % 17 for i in range(14)])          # ragged ends: rows free up at different steps
    sliced = [w for chunk in pipe.infer(list(texts[:14]), **kw) for w in chunk]
    cont = [w for chunk in pipe.infer(list(texts[:14]), continuous=True, **kw) for w in chunk]
    plain = [w for chunk in pipe.infer(list(texts[:14]), **dict(kw, lora_paths=None)) for w in chunk]
    assert len(sliced) == len(cont) == 14 and len({w.shape[0] for w in sliced}) > 1
    for u in range(14):
        assert sliced[u].shape == cont[u].shape, f"utterance {u}: {cont[u].shape[0]} samples with row re-use, {sliced[u].shape[0]} in slices"
        assert float((sliced[u] - cont[u]).abs().max()) <= 1e-4 * float(sliced[u].abs().max()), u
    assert any(sliced[u].shape != plain[u].shape or float((sliced[u] - plain[u]).abs().max()) > 1e-3 * float(plain[u].abs().max()) for u in range(14) if paths[u]), "the adapters change nothing"
    for u in (1, 4, 7):                                              # utterances without an adapter are the base model's
        assert sliced[u].shape == plain[u].shape and float((sliced[u] - plain[u]).abs().max()) <= 1e-4 * float(plain[u].abs().max())
    # the base engine is untouched: same result as an engine that never saw an adapter
    base = run(None)
    o = ref_cpu.OracleGPT(gsd, 12)
    emb = o.apply_spk_emb(o.embed(ids, tm), torch.from_numpy(codec.decode_spk_emb(spk)), ids, tok.spk_emb_ids)
    ref = o.generate(emb, ids, ref_cpu.SamplerParams(min_new_token=N), attention_mask=att, max_new_token=N, noise=ref_cpu.ArrayNoise(q))
    wav_ref = ref_cpu.vocos_decode(vsd, ref_cpu.dvae_decode(dsd, ref.hiddens[3])).numpy()
    w = base[3].cpu().numpy()
    assert float(np.sqrt(np.mean((w - wav_ref) ** 2))) / float(np.sqrt(np.mean(wav_ref ** 2))) <= 1e-3


@pytest.mark.parametrize("wd", ["fp32", "fp16"])
def test_per_utterance_lora_matches_per_row_merged_oracle(wd):
    """several adapters inside ONE batch (the reference can only merge one adapter for a whole call, pipeline:420-432).
    Rows 0/3 use adapter 0, rows 1/4 adapter 1, rows 2/5 none; every row must equal the oracle run with THAT row's merged weights
    (W + scale * B A on q/k/v/o).  fp32: token ids identical and hiddens <= 1e-4 (the fused W x + s B (A x) differs from the merged
    product only by fp32 rounding); fp16: first hiddens within 2e-3 rel-RMS.  Also: switching the per-row path off restores the base model."""
    from chatttsplus_amd.hip_models import GPT
    cfg = dict(synth.GPT_REAL); cfg["num_hidden_layers"] = 6
    llama = dict(hidden_size=768, intermediate_size=3072, num_attention_heads=12, num_hidden_layers=6)
    sd = synth.gpt_state_dict(cfg, 1234)
    rng = np.random.Generator(np.random.Philox(key=77))
    adapters, merged = [], []
    for ai in range(2):
        ad, m = [], {k: v.copy() for k, v in sd.items()}
        r = 8 if ai == 0 else 4                                            # different ranks
        for l in range(6):
            for t in ("q_proj", "k_proj", "v_proj", "o_proj"):
                A = (rng.standard_normal((r, 768)) * 0.05).astype(np.float32); Bm = (rng.standard_normal((768, r)) * 0.05).astype(np.float32)
                ad.append((l, t, A, Bm, 2.0))
                m[f"gpt.layers.{l}.self_attn.{t}.weight"] = (m[f"gpt.layers.{l}.self_attn.{t}.weight"] + 2.0 * (Bm @ A)).astype(np.float32)
        adapters.append(ad); merged.append(m)
    B, T, N = 6, 14, 8
    pads = [0, 3, 0, 5, 1, 2]
    ids, mask = synth.prompt_ids(B, T, cfg["num_text_tokens"], 19, pad_left=pads)
    q = torch.from_numpy(np.stack([synth.exp_noise(31, i, 4 * B, 626) for i in range(N)]))
    slots = [0, 1, -1, 0, 1, -1]
    lw = [type("P", (), dict(top_p=0.7, min_tokens_to_keep=3))(), type("K", (), dict(top_k=20))()]
    lp = [type("R", (), dict(penalty=1.05, past_window=16, max_input_ids=625))()]
    g = GPT(llama, max_batch=B, max_seq_len=64, weight_dtype=wd)
    g.load_state_dict(sd)
    g.load_adapter(0, adapters[0]); g.load_adapter(1, adapters[1])

    def run(row_slots):
        g.set_row_adapters(row_slots)
        emb = g(torch.from_numpy(ids), torch.ones(B, T, dtype=torch.bool))
        out = list(g.generate(emb, torch.from_numpy(ids), torch.tensor([0.3] * 4), 625, attention_mask=torch.from_numpy(mask), max_new_token=N,
                              min_new_token=N, logits_warpers=lw, logits_processors=lp, return_hidden=True, noise=q))[-1]
        g.set_row_adapters(None)
        return out

    out = run(slots)
    base = run(None)
    refs = {}
    for s_ in (0, 1, -1):
        rows = [b for b in range(B) if slots[b] == s_]
        o = ref_cpu.OracleGPT(merged[s_] if s_ >= 0 else sd, 12)
        emb = o.embed(torch.from_numpy(ids[rows]), torch.ones(len(rows), T, dtype=torch.bool))
        qr = q.view(N, B, 4, 626)[:, rows].reshape(N, 4 * len(rows), 626).contiguous()
        ref = o.generate(emb, torch.from_numpy(ids[rows]), ref_cpu.SamplerParams(min_new_token=N), attention_mask=torch.from_numpy(mask[rows]),
                         max_new_token=N, noise=ref_cpu.ArrayNoise(qr))
        for j, b in enumerate(rows):
            refs[b] = (ref.ids[j], ref.hiddens[j])
    for b in range(B):
        rid, rh = refs[b]
        h0 = out.hiddens[b][0].cpu()
        rel = float((h0 - rh[0]).pow(2).mean().sqrt() / rh[0].pow(2).mean().sqrt())
        if wd == "fp32":
            assert torch.equal(out.ids[b].cpu(), rid), f"row {b} (slot {slots[b]}): token ids differ from the merged-weights oracle"
            assert float((out.hiddens[b].cpu() - rh).abs().max()) <= 1e-4, f"row {b}"
        else:
            assert rel <= 2e-3, f"row {b} (slot {slots[b]}): first hidden rel-RMS {rel}"
    # the adapters matter (rows with an adapter differ from the base model) and switching them off restores the base model exactly
    assert not torch.equal(out.hiddens[0], base.hiddens[0]) and not torch.equal(out.hiddens[1], base.hiddens[1])
    if wd == "fp32":
        for b in (2, 5):
            assert torch.equal(base.ids[b], out.ids[b])
    g.close()


@pytest.mark.parametrize("wd", ["fp32", "fp16"])
def test_per_utterance_lora_inside_the_projection_launches_equals_the_separate_launches(wd):
    """Decode steps evaluate the rows' low-rank terms in worker workgroups of the QKV / o_proj launches (lora_worker.h, tagged-granule hand-off inside
    the launch) instead of two more launches per layer (lora.hip).  Same arithmetic: fp32 token ids identical, hiddens within 2e-5 (the RMSNorm factor is
    reduced in a different order; fp16: first hidden states within 2e-3), at 3 rows, 20 rows (two 16-row chunks) and 40 rows (32-row chunks), with and without graphs."""
    from chatttsplus_amd.hip_models import GPT
    cfg = dict(synth.GPT_REAL); cfg["num_hidden_layers"] = 4
    llama = dict(hidden_size=768, intermediate_size=3072, num_attention_heads=12, num_hidden_layers=4)
    sd = synth.gpt_state_dict(cfg, 1234)
    rng = np.random.Generator(np.random.Philox(key=78))
    g = GPT(llama, max_batch=40, max_seq_len=96, weight_dtype=wd)
    g.load_state_dict(sd)
    for ai in range(3):
        ad = []
        for l in range(4):
            for t in ("q_proj", "k_proj", "v_proj", "o_proj"):
                r = (4, 8, 16)[ai]
                ad.append((l, t, (rng.standard_normal((r, 768)) * 0.05).astype(np.float32), (rng.standard_normal((768, r)) * 0.05).astype(np.float32), 2.0))
        g.load_adapter(ai, ad)
    lw = [type("P", (), dict(top_p=0.7, min_tokens_to_keep=3))(), type("K", (), dict(top_k=20))()]
    lp = [type("R", (), dict(penalty=1.05, past_window=16, max_input_ids=625))()]
    assert g.get_option("lora_fold") == 1
    for B in (3, 20, 40):
        T, N = 12, 24
        ids, mask = synth.prompt_ids(B, T, cfg["num_text_tokens"], 23, pad_left=[(3 * b) % 7 for b in range(B)])
        slots = [(b % 4) - 1 for b in range(B)]                         # -1 (none), 0, 1, 2
        runs = {}
        for fold, graph in ((0, True), (1, True), (1, False)):
            g.set_option("lora_fold", fold)
            g.use_graph = graph
            g.set_row_adapters(slots)
            emb = g(torch.from_numpy(ids), torch.ones(B, T, dtype=torch.bool))
            runs[(fold, graph)] = list(g.generate(emb, torch.from_numpy(ids), torch.tensor([0.3] * 4), 625, attention_mask=torch.from_numpy(mask), max_new_token=N,
                                                  min_new_token=N, logits_warpers=lw, logits_processors=lp, return_hidden=True, noise="device", seed=3))[-1]
            g.set_row_adapters(None)
        g.use_graph = True
        for key in ((1, True), (1, False)):
            for b in range(B):
                if wd == "fp32":
                    assert torch.equal(runs[key].ids[b], runs[(0, True)].ids[b]), f"{wd} B={B} row {b} {key}: ids differ from the separate launches"
                    assert float((runs[key].hiddens[b] - runs[(0, True)].hiddens[b]).abs().max()) <= 2e-5, (wd, B, b, key)
                else:               # fp16 K / V: an ulp in the RMSNorm factor can move a rounding and, steps later, a token -- the first step's hidden states are the statement
                    h0, r0 = runs[key].hiddens[b][0], runs[(0, True)].hiddens[b][0]
                    assert float((h0 - r0).pow(2).mean().sqrt() / r0.pow(2).mean().sqrt()) <= 2e-3, (wd, B, b, key)
            if wd == "fp16":
                assert torch.equal(torch.stack(list(runs[(1, True)].ids)), torch.stack(list(runs[(1, False)].ids))), "graph replay != eager launches"
    g.set_option("lora_fold", 1)
    # row re-use with adapters: 12 utterances through 3 rows, each with its own slot (an admitted utterance brings its adapter: ctts_gpt_admit_adapters).
    # Folded and separate launches see the same admissions: identical tokens.  Against slices: the admitted utterance's LAST prompt token takes the decode
    # kernels instead of the prompt-pass ones (ulp-level differences, with or without adapters), so the statement is the first hidden state within 1e-5 --
    # a wrong or missing adapter moves it by O(1) -- and the first tokens.
    if wd == "fp32":
        NU, T, N = 12, 10, 18
        ids, mask = synth.prompt_ids(NU, T, cfg["num_text_tokens"], 29, pad_left=[(2 * b) % 5 for b in range(NU)])
        slots = [(b % 4) - 1 for b in range(NU)]
        emb = g(torch.from_numpy(ids), torch.ones(NU, T, dtype=torch.bool))
        kw = dict(max_new_token=N, min_new_token=2, logits_warpers=lw, logits_processors=lp, return_hidden=True, seed=11)
        many = {}
        for fold in (1, 0):
            g.set_option("lora_fold", fold)
            many[fold] = g.generate_many(emb, torch.from_numpy(ids), torch.tensor([0.3] * 4), 625, attention_mask=torch.from_numpy(mask), utt_ids=list(range(NU)), rows=3,
                                         adapter_slots=slots, **kw)
            assert g.admissions, "no utterance was admitted into a freed row"
        g.set_option("lora_fold", 1)
        plain = g.generate_many(emb, torch.from_numpy(ids), torch.tensor([0.3] * 4), 625, attention_mask=torch.from_numpy(mask), utt_ids=list(range(NU)), rows=3, **kw)
        for u in range(NU):
            assert torch.equal(many[1].ids[u], many[0].ids[u]) and float((many[1].hiddens[u] - many[0].hiddens[u]).abs().max()) <= 2e-5, u
        for i in range(0, NU, 3):
            sl = slice(i, i + 3)
            g.set_row_adapters(slots[sl])
            ref = list(g.generate(emb[sl].contiguous(), torch.from_numpy(ids[sl]), torch.tensor([0.3] * 4), 625, attention_mask=torch.from_numpy(mask[sl]), noise="device",
                                  utt_ids=list(range(NU))[sl], **kw))[-1]
            g.set_row_adapters(None)
            for j, u in enumerate(range(i, i + 3)):
                assert float((many[1].hiddens[u][0] - ref.hiddens[j][0]).abs().max()) <= 1e-5, f"utterance {u} (slot {slots[u]}): not the adapter a slice gives it"
                assert torch.equal(many[1].ids[u][:4], ref.ids[j][:4]), u
                if slots[u] >= 0:
                    assert float((many[1].hiddens[u][0] - plain.hiddens[u][0]).abs().max()) > 1e-3, f"utterance {u}: its adapter changes nothing"
                else:
                    assert torch.equal(many[1].ids[u], plain.ids[u]), u
    g.close()


def test_row_adapters_survive_a_second_generate_and_compaction():
    """`set_row_adapters` applies to "the following generate() calls".  ctts_gpt_compact clears the engine's live adapter flag once every adapter-carrying
    row has left the batch; ctts_gpt_begin restored the per-row slots only while that flag was still set -- so a second generate() after ONE set_row_adapters() call ran
    without its adapters, silently.  Here the only adapter row finishes first (its limit is short), the batch is compacted, and the same call is repeated: identical
    tokens and hidden states, and different from the adapter-less run."""
    from chatttsplus_amd.hip_models import GPT
    g = GPT(dict(hidden_size=768, intermediate_size=3072, num_attention_heads=12, num_hidden_layers=20), max_batch=8, max_seq_len=128, weight_dtype="fp32")
    try:
        g.load_state_dict(synth.gpt_state_dict(synth.GPT_REAL, 1234))
        rl = np.random.Generator(np.random.Philox(key=5))
        g.load_adapter(0, [(l, t, (rl.standard_normal((8, 768)) * 0.05).astype(np.float32), (rl.standard_normal((768, 8)) * 0.05).astype(np.float32), 2.0)
                           for l in range(20) for t in ("q_proj", "k_proj", "v_proj", "o_proj")])
        B, T, N = 8, 12, 64
        ids, mask = synth.prompt_ids(B, T, 21178, 77)
        emb = g(torch.from_numpy(ids), torch.ones(B, T, dtype=torch.bool))
        lw = [type("P", (), dict(top_p=0.7, min_tokens_to_keep=3))(), type("K", (), dict(top_k=20))()]

        def run():
            return list(g.generate(emb, torch.from_numpy(ids), torch.tensor([0.3] * 4), 625, attention_mask=torch.from_numpy(mask), max_new_token=N, min_new_token=N,
                                   logits_warpers=lw, logits_processors=[], return_hidden=True, noise="device", seed=3, max_new_tokens_per_row=[4] + [N] * 7))[-1]      # (compaction serves batches of >= 8 sequences)

        plain = run()
        g.set_row_adapters([0] + [-1] * 7)                 # ONE call; the adapter row stops after 4 tokens and is compacted away
        first = run()
        assert g.compactions, "the case has no teeth: no row left the batch"
        second = run()
        g.set_row_adapters(None)
        assert float((first.hiddens[0] - plain.hiddens[0]).abs().max()) > 1e-3, "the adapter changes nothing: the case has no teeth"
        for b in range(B):
            assert torch.equal(second.ids[b], first.ids[b]) and torch.equal(second.hiddens[b], first.hiddens[b]), f"row {b}: the second generate() lost the adapters"
        for b in range(1, B):
            assert torch.equal(first.ids[b], plain.ids[b]), f"adapter-less row {b} differs from the plain run"
        third = run()                                       # ... and after set_row_adapters(None) the engine is plain again
        assert torch.equal(third.hiddens[0], plain.hiddens[0])
    finally:
        g.close()


def test_per_utterance_lora_through_the_split_prompt_pass():
    """Round 5: a prompt pass that carries per-utterance adapters used to leave the parity engine's split GEMMs (3-term fp16 products, prefill_split.hip)
    for the decode kernels over all B x T rows -- 11.9 vs 4.8 ms at 32 x 48 tokens.  The split GEMMs' q|k|v and o_proj epilogues now add the rows' low-rank terms
    (the two lora.hip launches per layer; for o_proj from the head / tail images of the attention output).  Against an engine that never uses the split pass
    (`prefill_split_rows` 0): same tokens, first hidden states within the split pass's own tolerance; adapters still matter; adapter-less rows equal the plain run."""
    from chatttsplus_amd.hip_models import GPT
    cfg = dict(synth.GPT_REAL); cfg["num_hidden_layers"] = 6
    llama = dict(hidden_size=768, intermediate_size=3072, num_attention_heads=12, num_hidden_layers=6)
    sd = synth.gpt_state_dict(cfg, 1234)
    rng = np.random.Generator(np.random.Philox(key=79))
    ads = [[(l, t, (rng.standard_normal((r, 768)) * 0.05).astype(np.float32), (rng.standard_normal((768, r)) * 0.05).astype(np.float32), 2.0)
            for l in range(6) for t in ("q_proj", "k_proj", "v_proj", "o_proj")] for r in (8, 16)]
    B, T, N = 10, 48, 6                                                  # 480 prompt rows: the split pass (every pass of more than 64 rows since round 6; 384 before)
    ids, mask = synth.prompt_ids(B, T, cfg["num_text_tokens"], 31, pad_left=[(5 * b) % 11 for b in range(B)])
    slots = [(b % 3) - 1 for b in range(B)]                              # -1 (none), 0, 1
    lw = [type("P", (), dict(top_p=0.7, min_tokens_to_keep=3))(), type("K", (), dict(top_k=20))()]
    outs = {}
    for name, opts in (("split", {}), ("rows", {"prefill_split_rows": 0})):
        g = GPT(llama, max_batch=B, max_seq_len=T + N + 8, weight_dtype="fp32", options=opts)
        try:
            g.load_state_dict(sd)
            assert g.get_option("prefill_split_rows") == (65 if name == "split" else 0)
            for i, ad in enumerate(ads):
                g.load_adapter(i, ad)
            emb = g(torch.from_numpy(ids), torch.ones(B, T, dtype=torch.bool))
            for key, sl in (("lora", slots), ("plain", None)):
                g.set_row_adapters(sl)
                outs[(name, key)] = list(g.generate(emb, torch.from_numpy(ids), torch.tensor([0.3] * 4), 625, attention_mask=torch.from_numpy(mask), max_new_token=N,
                                                    min_new_token=N, logits_warpers=lw, logits_processors=[], return_hidden=True, noise="device", seed=3))[-1]
                g.set_row_adapters(None)
            if name == "split":      # round 6: the same pass with its split GEMMs forced onto the 256-row counter-phased blocks (480 rows would take the 128 x 128 ones): the adapter terms
                base = None          # ride in the shared epilogue, every element accumulates in the same order -> the same bits (K slicing of the down projection off: another order)
                g.set_option("prefill_splitk_rows", 0)
                for shape in (0, -3, -4):
                    g.set_option("prefill_pp_blocks", shape)
                    g.set_row_adapters(slots)
                    pp = list(g.generate(emb, torch.from_numpy(ids), torch.tensor([0.3] * 4), 625, attention_mask=torch.from_numpy(mask), max_new_token=N,
                                         min_new_token=N, logits_warpers=lw, logits_processors=[], return_hidden=True, noise="device", seed=3))[-1]
                    g.set_row_adapters(None)
                    if base is None:
                        base = pp
                        for b in range(B):      # ... and against the default (K sliced four ways at 480 rows): the same tokens, hidden rows one summation order apart
                            assert torch.equal(base.ids[b], outs[("split", "lora")].ids[b]), b
                            assert float((base.hiddens[b] - outs[("split", "lora")].hiddens[b]).abs().max()) <= 2e-5, b
                    for b in range(B):
                        assert torch.equal(pp.ids[b], base.ids[b]) and torch.equal(pp.hiddens[b], base.hiddens[b]), (shape, b)
        finally:
            g.close()
    for b in range(B):
        a, r = outs[("split", "lora")], outs[("rows", "lora")]
        assert torch.equal(a.ids[b], r.ids[b]), f"row {b} (slot {slots[b]}): tokens differ between the split prompt pass and the row kernels"
        assert float((a.hiddens[b][0] - r.hiddens[b][0]).abs().max()) <= 1e-4, (b, float((a.hiddens[b][0] - r.hiddens[b][0]).abs().max()))
        if slots[b] >= 0:
            assert float((a.hiddens[b][0] - outs[("split", "plain")].hiddens[b][0]).abs().max()) > 1e-3, f"row {b}: its adapter changes nothing"
        else:
            assert torch.equal(a.ids[b], outs[("split", "plain")].ids[b]), b
