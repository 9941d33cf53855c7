"""Parity of the *benchmarked* mode (fp16 weights / KV, fp32 accumulate) on the quantity north_star names: the mel and the
waveform.  The oracle (fp32 CPU restatement of the reference) runs free; its token ids are forced into the HIP fp16 engine step
by step, the fp16-mode hiddens go through the HIP DVAE decoder + Vocos, and mel / waveform are compared with the oracle chain
on the oracle's fp32 hiddens.

Tolerances (written below, stated the same way in DESIGN.md section 2):
  * fp32 parity mode: mel / waveform RMS error <= 1e-3 of the signal RMS -- north_star's tolerance (measured ~1e-5);
  * fp16 mode (the reference's own GPU dtype, pipeline:37-41; the benchmarked mode): <= 2e-3 (measured: 1.2e-3 / 1.3e-3 mel / wav at
    batch 1 over 256 tokens, worst row of a batch of 32 over 48 tokens 1.6e-3 / 1.75e-3).  It does NOT meet 1e-3 and no
    16-bit float weight format can on this model: rounding the packed weights to fp16 ALONE gives 1.05e-3 on the hiddens
    (tools/fp16_error_budget.py: weights 1.05e-3, MFMA operands 0.69e-3, KV 0.51e-3, all three 1.27e-3 -- the HIP engine
    measures 1.26e-3); the reference's own GPU path (model.half(): every op output in fp16) emulates to 2.2e-3.

Also here: the stress-weights cases (projections x8, sharpened heads): peaked attention and outlier channels instead of the
near-uniform attention N(0, 0.02^2) weights give."""
import ctypes as C

import numpy as np
import pytest
import torch

from chatttsplus_amd import synth
from oracle import ref_cpu

pytestmark = pytest.mark.gpu

LW = [type("P", (), dict(top_p=0.7, min_tokens_to_keep=3))(), type("K", (), dict(top_k=20))()]
LP = [type("R", (), dict(penalty=1.05, past_window=16, max_input_ids=625))()]
LLAMA = dict(hidden_size=768, intermediate_size=3072, num_attention_heads=12, num_hidden_layers=20)
MEL_WAV_TOL = {"fp32": 1e-3, "fp16": 2e-3}          # north_star: "within 1e-3 RMS on the mel/waveform" -- see the module docstring for fp16


def _rel_rms(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return float(np.sqrt(np.mean((a - b) ** 2)) / np.sqrt(np.mean(b ** 2)))


def teacher_forced_hiddens(g, emb, mask, forced, N):
    """Runs the HIP engine over the oracle's token ids (ctts_gpt_force_ids before every decode step); returns hiddens [B,N,768]."""
    from chatttsplus_amd import _lib
    from chatttsplus_amd.hip_models.gpt import sampler_cfg_from_objects
    lib, h, dev = g._lib, g._h, g.device
    B, T = mask.shape
    sc = sampler_cfg_from_objects(torch.tensor([0.3] * 4), 625, N, N, LW, LP, 4)
    out_ids = torch.zeros(B, N, 4, dtype=torch.int32, device=dev)
    hid = torch.zeros(B, N, 768, dtype=torch.float32, device=dev)
    fin = torch.zeros(B, dtype=torch.int32, device=dev); end = torch.zeros(B, dtype=torch.int32, device=dev)
    io = _lib.GenIO(ids=out_ids.data_ptr(), hiddens=hid.data_ptr(), finish=fin.data_ptr(), end_idx=end.data_ptr(), noise=None, n_draws=0, seed=1)
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    msk = torch.from_numpy(mask).to(dev).to(torch.int32)
    embd = emb.to(dev).contiguous()
    forced = forced.to(dev)
    _lib.check(lib.ctts_gpt_begin(h, B, T, msk.data_ptr(), C.byref(sc), C.byref(io), st), "begin")
    _lib.check(lib.ctts_gpt_prefill(h, embd.data_ptr(), st), "prefill")
    _lib.check(lib.ctts_gpt_sample(h, st), "sample")
    for i in range(1, N):
        f = forced[:, i - 1].contiguous()
        _lib.check(lib.ctts_gpt_force_ids(h, f.data_ptr(), st), "force")
        _lib.check(lib.ctts_gpt_decode(h, 1, 1, st), "decode")
    torch.cuda.synchronize()
    return hid


_cache = {}


def engines(wd, stress=False, max_batch=32, max_seq=400):
    from chatttsplus_amd.hip_models import GPT
    key = (wd, stress, max_batch, max_seq)
    if key not in _cache:
        sd = synth.gpt_state_dict(synth.GPT_REAL, 1234)
        if stress:
            sd = synth.stress_gpt_state_dict(sd)
        g = GPT(LLAMA, max_batch=max_batch, max_seq_len=max_seq, weight_dtype=wd)
        g.load_state_dict(sd)
        _cache[key] = (g, sd)
    return _cache[key]


@pytest.fixture(scope="module")
def vocoder():
    from chatttsplus_amd.hip_models import Synth
    s = Synth(dict(synth.DVAE_REAL), dict(synth.VOCOS_REAL), max_frames=1024, max_batch=32)
    dsd, vsd = synth.dvae_state_dict(synth.DVAE_REAL, 1234), synth.vocos_state_dict(synth.VOCOS_REAL, 1234)
    s.load("dvae.", dsd); s.load("vocos.", vsd)
    return s, dsd, vsd


@pytest.mark.parametrize("wd,B,T,pad,N", [("fp16", 1, 48, None, 256), ("fp16", 32, 24, [i % 17 for i in range(32)], 48),
                                          ("fp32", 1, 48, None, 96), ("fp32", 32, 24, [i % 17 for i in range(32)], 24)])
def test_mel_and_waveform_tolerance_by_mode(vocoder, wd, B, T, pad, N):
    s, dsd, vsd = vocoder
    g, sd = engines(wd)
    ids, mask = synth.prompt_ids(B, T, synth.GPT_REAL["num_text_tokens"], 700 + B, pad_left=pad)
    o = ref_cpu.OracleGPT(sd, 12)
    emb = o.embed(torch.from_numpy(ids), torch.ones(B, T, dtype=torch.bool))
    ref = o.generate(emb, torch.from_numpy(ids), ref_cpu.SamplerParams(min_new_token=N), attention_mask=torch.from_numpy(mask),
                     max_new_token=N, noise=ref_cpu.SeededNoise(9))
    forced = torch.stack(list(ref.ids), 0).to(torch.int32)
    hid = teacher_forced_hiddens(g, emb, mask, forced, N)
    wavs = s.decode_batch([hid[b] for b in range(B)])
    worst = dict(hid=0.0, mel=0.0, wav=0.0)
    for b in (range(B) if B <= 4 else range(0, B, 5)):
        mel_ref = ref_cpu.dvae_decode(dsd, ref.hiddens[b])
        wav_ref = ref_cpu.vocos_decode(vsd, mel_ref).numpy()
        mel = s.dvae_decode(hid[b]).cpu().numpy()
        worst["hid"] = max(worst["hid"], _rel_rms(hid[b].cpu().numpy(), ref.hiddens[b].numpy()))
        worst["mel"] = max(worst["mel"], _rel_rms(mel, mel_ref.numpy()))
        worst["wav"] = max(worst["wav"], _rel_rms(wavs[b].cpu().numpy(), wav_ref))
    print(f"{wd}-mode parity B={B} N={N}: rel-RMS hidden {worst['hid']:.2e} mel {worst['mel']:.2e} wav {worst['wav']:.2e}")
    assert worst["mel"] <= MEL_WAV_TOL[wd] and worst["wav"] <= MEL_WAV_TOL[wd], worst


def test_stress_weights_fp32_ids_bit_exact_and_fp16_hiddens():
    """Projections x8 + sharpened heads (synth.stress_gpt_state_dict): fp32 parity mode still reproduces the oracle's token ids
    under the same torch seed (free running) with hiddens within 1e-4.  fp16 mode: the x64 attention logits and x512 MLP outputs
    amplify every 16-bit rounding -- the rounding model of tools/fp16_error_budget.py predicts 2.2e-2 on these weights (weights
    alone 1.3e-2, MFMA operands 1.1e-2, KV 0.8e-2; the reference's own model.half() path emulates to 3.9e-2) and the HIP engine
    measures 2.1e-2: it behaves as the rounding model says, no kernel loses precision on outliers.  Asserted: <= 3e-2."""
    g32, sd = engines("fp32", stress=True, max_batch=4, max_seq=128)
    B, T, N = 3, 20, 24
    ids, mask = synth.prompt_ids(B, T, synth.GPT_REAL["num_text_tokens"], 91, pad_left=[0, 4, 11])
    o = ref_cpu.OracleGPT(sd, 12)
    emb = o.embed(torch.from_numpy(ids), torch.ones(B, T, dtype=torch.bool))
    torch.manual_seed(21)
    ref = o.generate(emb, torch.from_numpy(ids), ref_cpu.SamplerParams(min_new_token=N), attention_mask=torch.from_numpy(mask), max_new_token=N)
    torch.manual_seed(21)
    out = list(g32.generate(emb.cuda(), torch.from_numpy(ids), torch.tensor([0.3] * 4), 625, attention_mask=torch.from_numpy(mask),
                            max_new_token=N, min_new_token=N, logits_warpers=LW, logits_processors=LP, return_hidden=True, noise="torch"))[-1]
    for b in range(B):
        assert torch.equal(out.ids[b].cpu(), ref.ids[b]), f"stress weights, fp32: row {b} token ids differ"
        r = _rel_rms(out.hiddens[b].cpu().numpy(), ref.hiddens[b].numpy())
        assert r <= 1e-4, f"stress weights, fp32: row {b} hidden rel-RMS {r}"
    g16, _ = engines("fp16", stress=True, max_batch=4, max_seq=128)
    forced = torch.stack(list(ref.ids), 0).to(torch.int32)
    hid = teacher_forced_hiddens(g16, emb, mask, forced, N)
    worst = max(_rel_rms(hid[b].cpu().numpy(), ref.hiddens[b].numpy()) for b in range(B))
    print(f"stress weights fp16 teacher-forced hidden rel-RMS {worst:.2e}")
    assert worst <= 3e-2, worst


@pytest.mark.parametrize("B", [2, 12])
def test_fp16_overflow_saturates_and_is_reported(B):
    """a checkpoint whose SwiGLU outputs leave the fp16 range.  The reference's GPU path (`model.half()`, pipeline:37-41)
    turns such a value into inf and the row into NaN; the fp16 engine SATURATES the store at +-65504 and REPORTS it (ctts_gpt_saturations,
    RuntimeWarning from generate()) -- finite hiddens, valid token ids, no silent NaN.  The fp32 engine is unaffected on its exact kernels (batch 2, and batch 12 after
    it has dropped its head / tail decode kernels by itself).  Batch 2 runs the split-K path, batch 12 the packed residual path."""
    import warnings
    from chatttsplus_amd.hip_models import GPT
    sd = synth.gpt_state_dict(synth.GPT_REAL, 1234)
    for l in (3, 11):
        for k in ("gate_proj", "up_proj"):
            sd[f"gpt.layers.{l}.mlp.{k}.weight"] = sd[f"gpt.layers.{l}.mlp.{k}.weight"] * 600.0
    # the emulated .half() path overflows: act = silu(gate(x)) * up(x) on a unit-RMS row, rounded to fp16
    xh = torch.from_numpy(np.random.Generator(np.random.Philox(key=5)).standard_normal((4, 768)).astype(np.float32))
    xh = xh / xh.pow(2).mean(-1, keepdim=True).sqrt()
    act = torch.nn.functional.silu(xh @ torch.from_numpy(sd["gpt.layers.3.mlp.gate_proj.weight"]).t()) * (xh @ torch.from_numpy(sd["gpt.layers.3.mlp.up_proj.weight"]).t())
    assert bool(torch.isinf(act.half()).any()), "test premise: the fp16 rounding of this checkpoint's SwiGLU output overflows"
    ids, mask = synth.prompt_ids(B, 10, 21178, 91)
    out = {}
    for wd in ("fp16", "fp32"):
        g = GPT(LLAMA, max_batch=B, max_seq_len=64, weight_dtype=wd)
        g.load_state_dict(sd)
        emb = g(torch.from_numpy(ids), torch.ones(B, 10, dtype=torch.bool))
        with warnings.catch_warnings(record=True) as rec:
            warnings.simplefilter("always")
            o = list(g.generate(emb, torch.from_numpy(ids), torch.tensor([0.3] * 4), 625, attention_mask=torch.from_numpy(mask), max_new_token=6,
                                min_new_token=6, logits_warpers=LW, logits_processors=LP, return_hidden=True, noise="device", seed=3))[-1]
        out[wd] = (o, g.saturations, [str(w.message) for w in rec if issubclass(w.category, RuntimeWarning)])
        g.close()
    o16, nsat16, warn16 = out["fp16"]
    o32, nsat32, warn32 = out["fp32"]
    assert nsat16 > 0 and any("saturated" in w for w in warn16), (nsat16, warn16)
    if B <= 8:
        assert nsat32 == 0 and not warn32
    else:
        # round 6: from 9 rows on the fp32 engine multiplies on the fp16 pipes with head / tail operands (silu(gate) * up / 16 as an fp16 pair): THIS checkpoint leaves that
        # range too.  The engine must say so and fall back to its exact fp32 decode kernels by itself: the second call below is clean and equals an engine that never used them.
        assert nsat32 > 0 and any("split_decode_rows" in w for w in warn32), (nsat32, warn32)
        g = GPT(LLAMA, max_batch=B, max_seq_len=64, weight_dtype="fp32")
        gx = GPT(LLAMA, max_batch=B, max_seq_len=64, weight_dtype="fp32", options={"split_decode_rows": 0, "prefill_split_rows": 0})
        try:
            res = []
            for eng, calls in ((g, 2), (gx, 1)):
                eng.load_state_dict(sd)
                emb = eng(torch.from_numpy(ids), torch.ones(B, 10, dtype=torch.bool))
                for _ in range(calls):
                    with warnings.catch_warnings(record=True) as rec:
                        warnings.simplefilter("always")
                        o = list(eng.generate(emb, torch.from_numpy(ids), torch.tensor([0.3] * 4), 625, attention_mask=torch.from_numpy(mask), max_new_token=6,
                                              min_new_token=6, logits_warpers=LW, logits_processors=LP, return_hidden=True, noise="device", seed=3))[-1]
                    res.append((o, eng.saturations))
            assert res[0][1] > 0 and res[1][1] == 0 and res[2][1] == 0, [r[1] for r in res]
            assert g.get_option("split_decode_rows") == 0 and g.get_option("prefill_split_rows") == 0      # (the 120-row prompt pass runs on the images too since round 6's 65-row threshold)
            for b in range(B):
                assert torch.equal(res[1][0].ids[b], res[2][0].ids[b]) and torch.equal(res[1][0].hiddens[b], res[2][0].hiddens[b]), b
            o32 = res[1][0]
        finally:
            g.close(); gx.close()
    for o in (o16, o32):
        assert all(bool(torch.isfinite(h).all()) for h in o.hiddens), "NaN / inf reached the hidden states"
        assert all(int(i.min()) >= 0 and int(i.max()) < 626 and i.shape[0] == 6 for i in o.ids)
