"""The oracle (oracle/ref_cpu.py) against the golden vectors minted from the imported reference
(oracle/make_golden.py).  CPU only.  Bit-exact for token ids; <=2e-5 abs on fp32 hiddens / mels."""
import os

import numpy as np
import pytest
import torch

from chatttsplus_amd import synth
from oracle import ref_cpu
from tests.helpers import GOLDEN, gen_case_inputs, load_golden


def _run_oracle(name, cfg):
    z, meta = load_golden(name)
    sd, ids, mask, spk = gen_case_inputs(meta, cfg)
    o = ref_cpu.OracleGPT(sd, cfg["num_attention_heads"])
    emb = o.embed(torch.from_numpy(ids), torch.ones(ids.shape[:2], dtype=torch.bool))
    if spk is not None:
        emb = o.apply_spk_emb(emb, torch.from_numpy(spk), torch.from_numpy(ids), int(meta["spk_id"]))
    temp = float(meta["temperature"]) if "temperature" in meta else 0.3
    sp = ref_cpu.SamplerParams(temperature=[temp] * 4, min_new_token=int(meta["min_new"]))
    if "temperatures" in meta:                      # gpt_real_params: one temperature per codebook, non-default top-p / top-k / penalty
        sp = ref_cpu.SamplerParams(temperature=[float(t) for t in meta["temperatures"]], top_p=float(meta["top_p"]), top_k=int(meta["top_k"]),
                                   repetition_penalty=float(meta["rep"]), min_new_token=int(meta["min_new"]))
    torch.manual_seed(int(meta["torch_seed"]))
    out = o.generate(emb, torch.from_numpy(ids), sp, attention_mask=torch.from_numpy(mask), max_new_token=int(meta["max_new"]))
    return z, meta, emb, out


@pytest.mark.parametrize("name,cfg", [
    ("gpt_tiny_b2_pad", synth.GPT_TINY),
    ("gpt_tiny_regen", synth.GPT_TINY),
    ("gpt_real_b1", synth.GPT_REAL),
    ("gpt_real_b2_pad", synth.GPT_REAL),
    ("gpt_real_greedy", synth.GPT_REAL),
    ("gpt_real_b4_ragged", synth.GPT_REAL),
    ("gpt_real_regen", synth.GPT_REAL),
    ("gpt_real_params", synth.GPT_REAL),
    ("gpt_real_long", synth.GPT_REAL),
])
def test_generate_matches_reference(name, cfg):
    z, meta, emb, out = _run_oracle(name, cfg)
    lens = z["lens"]
    assert [int(i.shape[0]) for i in out.ids] == lens.tolist()
    np.testing.assert_allclose(emb[:, -1].numpy(), z["emb_last"], atol=0, rtol=0)
    np.testing.assert_allclose(emb[:, 1].numpy(), z["emb_row1"], atol=1e-7, rtol=0)
    for b, n in enumerate(lens):
        assert np.array_equal(out.ids[b].numpy(), z["ids"][b, :n].astype(np.int64)), f"row {b} token ids differ"
        err = np.abs(out.hiddens[b].numpy() - z["hiddens"][b, :n]).max()
        assert err <= 2e-5, f"row {b} hidden err {err}"


def test_sampler_cases_match_reference():
    z = np.load(__import__("os").path.join(__import__("tests.helpers", fromlist=["GOLDEN"]).GOLDEN, "sampler_cases.npz"))
    for c in range(int(z["n"])):
        temp, top_p, top_k, rep, step, min_new = z[f"c{c}_params"]
        rows = z[f"c{c}_logits"].shape[0]
        sp = ref_cpu.SamplerParams(temperature=[temp] * 4, top_p=float(top_p), top_k=int(top_k), repetition_penalty=float(rep),
                                   min_new_token=int(min_new))
        idx = ref_cpu.sample_step(torch.from_numpy(z[f"c{c}_logits"]), torch.from_numpy(z[f"c{c}_history"]),
                                  torch.from_numpy(z[f"c{c}_q"]), int(step), sp,
                                  torch.full((rows, 1), float(temp), dtype=torch.float32))
        assert np.array_equal(idx.numpy(), z[f"c{c}_idx"].astype(np.int64)), f"case {c}"


def test_dvae_decode_matches_reference():
    z = np.load(__import__("os").path.join(__import__("tests.helpers", fromlist=["GOLDEN"]).GOLDEN, "dvae_real.npz"))
    sd = synth.dvae_state_dict(synth.DVAE_REAL, int(z["weight_seed"]))
    hid = np.random.Generator(np.random.Philox(key=int(z["hidden_seed"]))).standard_normal((int(z["n"]), 768)).astype(np.float32)
    mel = ref_cpu.dvae_decode(sd, torch.from_numpy(hid)).numpy()
    assert mel.shape == z["mel"].shape
    assert np.abs(mel - z["mel"]).max() <= 2e-5


def test_dvae_decode_matches_reference_at_edge_lengths():
    """1, 5 and 333 tokens (reference-minted mels): the lengths tests/test_gpu_vocoder.py checks the HIP kernels against this oracle."""
    z = np.load(os.path.join(GOLDEN, "dvae_real_lengths.npz"))
    sd = synth.dvae_state_dict(synth.DVAE_REAL, int(z["weight_seed"]))
    for n, seed in zip(z["lengths"], z["hidden_seeds"]):
        hid = np.random.Generator(np.random.Philox(key=int(seed))).standard_normal((int(n), 768)).astype(np.float32)
        mel = ref_cpu.dvae_decode(sd, torch.from_numpy(hid)).numpy()
        want = z[f"mel_{int(n)}"]
        assert mel.shape == want.shape == (100, 2 * int(n))
        assert np.abs(mel - want).max() <= 2e-5, int(n)


def test_vocos_istft_self_consistency():
    """Vocos is parity-unpinned (third-party, absent): check the restatement's ISTFT against the direct definition."""
    sd = {k: torch.from_numpy(v) for k, v in synth.vocos_state_dict(synth.VOCOS_REAL, 1234).items()}
    mel = torch.from_numpy(np.random.Generator(np.random.Philox(key=9)).standard_normal((100, 24)).astype(np.float32))
    wav = ref_cpu.vocos_decode(sd, mel)
    assert wav.shape[0] == 256 * (24 - 1)
    re, im = ref_cpu.vocos_head_spec(sd, ref_cpu.vocos_backbone(sd, mel))
    wav2 = ref_cpu.istft_direct(re, im, sd["head.istft.window"])
    scale = wav.abs().max().item()
    assert (wav - wav2).abs().max().item() <= 1e-4 * max(scale, 1.0)


@pytest.mark.parametrize("name", ["gpt_real_text_b2", "gpt_real_text_eos"])
def test_refine_text_generate_matches_reference(name):
    """infer_text=True pass (21178-way head_text, single temperature, emb_text re-embed) against the imported reference."""
    z, meta = load_golden(name)
    cfg = synth.GPT_REAL
    sd = synth.gpt_state_dict(cfg, int(meta["weight_seed"]))
    eos = int(meta["eos"])
    sd["head_text.parametrizations.weight.original0"][eos] *= float(meta["eos_boost"])
    B, T = int(meta["B"]), int(meta["T"])
    ids, mask = synth.prompt_ids(B, T, cfg["num_text_tokens"], int(meta["prompt_seed"]), pad_left=[int(x) for x in meta["pad_left"]])
    o = ref_cpu.OracleGPT(sd, cfg["num_attention_heads"])
    emb = o.embed(torch.from_numpy(ids), torch.ones(B, T, dtype=torch.bool))
    torch.manual_seed(int(meta["torch_seed"]))
    out = o.generate_text(emb, torch.from_numpy(ids), 0.7, eos, attention_mask=torch.from_numpy(mask), max_new_token=int(meta["max_new"]),
                          min_new_token=int(meta["min_new"]))
    assert [int(i.shape[0]) for i in out.ids] == z["lens"].tolist()
    for b, n in enumerate(z["lens"]):
        assert np.array_equal(out.ids[b].numpy(), z["ids"][b, :n].astype(np.int64))


def test_dvae_encode_oracle_golden():
    """Zero-shot encode branch: the oracle's conv stack equals the reference's own downsample_conv + encoder modules (pinned);
    mel extractor / GFSQ are restated third-party code (unpinned) frozen in the same fixture."""
    import torch
    from chatttsplus_amd import synth
    from oracle import ref_cpu
    z = np.load(os.path.join(GOLDEN, "dvae_encode_real.npz"))
    sd = synth.dvae_encoder_state_dict(synth.DVAE_ENC_REAL, int(z["weight_seed"]))
    wav = torch.from_numpy(synth.speaker_wave(int(z["wave_seed"]), int(z["n_samples"])))
    mel = ref_cpu.mel_features(wav)
    assert np.abs(mel.numpy() - z["mel"]).max() <= 1e-5
    feat = ref_cpu.dvae_encoder_features(sd, torch.from_numpy(z["mel"])).numpy()
    assert np.abs(feat - z["feat"]).max() <= 1e-4
    f = torch.from_numpy(z["feat"]).transpose(0, 1)
    sdt = {k: torch.from_numpy(v) for k, v in sd.items()}
    assert np.array_equal(ref_cpu.gfsq_indices(f, sdt, pre_bound=True).numpy(), z["ids_pre_bound"])
    assert np.array_equal(ref_cpu.gfsq_indices(f, sdt, pre_bound=False).numpy(), z["ids"])
    assert np.array_equal(ref_cpu.dvae_encode(sd, wav).numpy(), z["ids_pre_bound"])


def test_mel_features_against_direct_dft():
    """The restated torchaudio MelSpectrogram (unpinned) against an independent float64 evaluation: reflect padding, periodic hann
    window, |rfft|, HTK triangular filterbank, log(clip)."""
    from chatttsplus_amd import synth
    wav = synth.speaker_wave(2, 5000).astype(np.float64)
    n_fft, hop = 1024, 256
    xp = np.pad(wav, n_fft // 2, mode="reflect")
    win = 0.5 - 0.5 * np.cos(2 * np.pi * np.arange(n_fft) / n_fft)
    F = 1 + len(wav) // hop
    spec = np.stack([np.abs(np.fft.rfft(xp[f * hop:f * hop + n_fft] * win)) for f in range(F)], 1)      # [513, F]
    hz = np.linspace(0, 12000, 513)
    mel_pts = np.linspace(0.0, 2595.0 * np.log10(1.0 + 12000.0 / 700.0), 102)
    f_pts = 700.0 * (10 ** (mel_pts / 2595.0) - 1.0)
    fb = np.zeros((513, 100))
    for m in range(100):
        lo, ce, hi = f_pts[m], f_pts[m + 1], f_pts[m + 2]
        fb[:, m] = np.maximum(0.0, np.minimum((hz - lo) / (ce - lo), (hi - hz) / (hi - ce)))
    ref = np.log(np.clip(fb.T @ spec, 1e-5, None))
    got = ref_cpu.mel_features(torch.from_numpy(wav.astype(np.float32))).numpy()
    assert got.shape == ref.shape == (100, F)
    assert np.abs(got - ref).max() <= 2e-3


def test_gfsq_indices_are_nearest_codebook_entries():
    """The restated GroupedResidualFSQ index arithmetic (unpinned) against the definition of the implicit FSQ codebook: index i
    names the grid point ((i // 5^j) % 5 - 2) / 2 per dimension j, and the chosen entry is the nearest one to bound(z)."""
    from chatttsplus_amd import synth
    sd = {k: torch.from_numpy(v) for k, v in synth.dvae_encoder_state_dict(synth.DVAE_ENC_REAL, 1234).items()}
    rng = np.random.Generator(np.random.Philox(key=8))
    x = torch.from_numpy((rng.standard_normal((50, 1024)) * 1.5).astype(np.float32))
    lv = torch.tensor([5.0] * 4)
    grid = torch.stack([((torch.arange(625) // (5 ** j)) % 5 - 2) / 2.0 for j in range(4)], 1)          # implicit codebook [625, 4]
    for pre_bound in (True, False):
        ids = ref_cpu.gfsq_indices(x, sd, pre_bound=pre_bound)
        assert ids.shape == (4, 50) and int(ids.min()) >= 0 and int(ids.max()) < 625
        for g in range(2):
            z = torch.nn.functional.linear(x[:, g * 512:(g + 1) * 512], sd[f"vq_layer.quantizer.rvqs.{g}.project_in.weight"],
                                           sd[f"vq_layer.quantizer.rvqs.{g}.project_in.bias"])
            res = ref_cpu.fsq_bound(z, lv) if pre_bound else z
            for r in range(2):
                scale = 4.0 ** (-r)
                target = ref_cpu.fsq_bound(res / scale, lv) / 2.0                                        # bounded value in code units
                nearest = torch.cdist(target, grid).argmin(1)
                assert torch.equal(nearest.to(torch.int32), ids[g * 2 + r]), (pre_bound, g, r)
                res = res - grid[nearest] * scale


@pytest.mark.parametrize("F", [2, 9, 60])
def test_vocos_second_independent_restatement_agrees(F):
    """Vocos is third-party and absent offline (parity unpinned).  What can be done offline: a second restatement written separately
    from the upstream module list (tests/vocos_independent.py, float64 numpy) must agree with the oracle's."""
    from tests.vocos_independent import vocos_decode_f64
    vsd = synth.vocos_state_dict(synth.VOCOS_REAL, 1234)
    mel = np.random.Generator(np.random.Philox(key=900 + F)).standard_normal((100, F)).astype(np.float32)
    a = ref_cpu.vocos_decode(vsd, torch.from_numpy(mel)).numpy().astype(np.float64)
    b = vocos_decode_f64(vsd, mel)
    assert a.shape == b.shape == (256 * (F - 1),)
    rel = float(np.sqrt(np.mean((a - b) ** 2)) / np.sqrt(np.mean(b ** 2)))
    assert rel <= 2e-5, rel


@pytest.mark.parametrize("pre_bound", [True, False])
def test_gfsq_indices_round_trip_to_the_quantised_latent(pre_bound):
    """The property the reference relies on (GFSQ._embed / get_output_from_indices, dvae.py:85-96): the indices alone rebuild the
    quantised latent.  Both `pre_bound` variants of the (unpinned) residual-FSQ restatement satisfy it exactly, the indices are
    valid base-5 numbers (< 625), and the quantiser's reconstruction error stays inside one last-level step."""
    sd = synth.dvae_encoder_state_dict(synth.DVAE_ENC_REAL, 1234)
    x = torch.from_numpy(np.random.Generator(np.random.Philox(key=17)).standard_normal((300, 1024)).astype(np.float32)) * 1.5
    ids, lat = ref_cpu.gfsq_quantize(x, {k: torch.from_numpy(v) for k, v in sd.items()}, pre_bound=pre_bound)
    assert ids.shape == (4, 300) and int(ids.min()) >= 0 and int(ids.max()) < 625
    back = ref_cpu.gfsq_latent_from_indices(ids)
    assert torch.equal(back, lat)                                   # decode(indices) == quantised latent, exactly
    # every level of every dimension is used somewhere (the synthetic project_in spreads the codes): a wrong basis would not
    for r in range(4):
        digits = torch.stack([(ids[r].to(torch.int64) // 5 ** d) % 5 for d in range(4)], 1)
        assert digits.min() == 0 and digits.max() == 4
    # reconstruction: |target - latent| <= half a step of the last quantiser (0.25 * 0.5 / 2) wherever the first level did not saturate
    for g in range(2):
        w, b = torch.from_numpy(sd[f"vq_layer.quantizer.rvqs.{g}.project_in.weight"]), torch.from_numpy(sd[f"vq_layer.quantizer.rvqs.{g}.project_in.bias"])
        z = torch.nn.functional.linear(x[:, g * 512:(g + 1) * 512], w, b)
        target = ref_cpu.fsq_bound(z, torch.tensor([5., 5., 5., 5.])) if pre_bound else z
        inner = target.abs() < 0.9
        assert float((target - lat[g])[inner].abs().max()) <= 0.2


def test_generate_matches_reference_at_batch_32():
    """gpt_real_b32: 32 sequences with 23 different left paddings, minted from the reference's own GPT.generate -- the oracle is pinned at the
    batch size the HIP path is measured at (BASELINE configs[2]); ids of every row, hiddens of four rows."""
    z, meta = load_golden("gpt_real_b32")
    sd, ids, mask, _ = gen_case_inputs(meta, synth.GPT_REAL)
    o = ref_cpu.OracleGPT(sd, 12)
    emb = o.embed(torch.from_numpy(ids), torch.ones(ids.shape[:2], dtype=torch.bool))
    np.testing.assert_allclose(emb[:, -1].numpy(), z["emb_last"], atol=0, rtol=0)
    torch.manual_seed(int(meta["torch_seed"]))
    out = o.generate(emb, torch.from_numpy(ids), ref_cpu.SamplerParams(min_new_token=int(meta["min_new"])), attention_mask=torch.from_numpy(mask),
                     max_new_token=int(meta["max_new"]))
    assert [int(i.shape[0]) for i in out.ids] == z["lens"].tolist()
    assert np.array_equal(np.stack([i.numpy() for i in out.ids]), z["ids"].astype(np.int64))
    for k, r in enumerate(int(x) for x in meta["hidden_rows"]):
        assert np.abs(out.hiddens[r].numpy() - z["hiddens"][k]).max() <= 2e-5, r


def test_generate_matches_reference_at_batch_32_ragged_finish():
    """gpt_real_b32_ragged: BASELINE configs[2] free-running through the reference's own GPT.generate with boosted EOS rows -- 32 sequences, 23
    left paddings, rows ending at 2 .. 96 tokens (gpt.py:483-494,527-546: finish / end_idx bookkeeping while finished rows keep computing)."""
    z, meta = load_golden("gpt_real_b32_ragged")
    assert len(set(z["lens"].tolist())) >= 12 and z["lens"].max() >= 64
    sd, ids, mask, _ = gen_case_inputs(meta, synth.GPT_REAL)              # (applies the EOS boost of the fixture)
    o = ref_cpu.OracleGPT(sd, 12)
    emb = o.embed(torch.from_numpy(ids), torch.ones(ids.shape[:2], dtype=torch.bool))
    np.testing.assert_allclose(emb[:, -1].numpy(), z["emb_last"], atol=0, rtol=0)
    torch.manual_seed(int(meta["torch_seed"]))
    out = o.generate(emb, torch.from_numpy(ids), ref_cpu.SamplerParams(min_new_token=int(meta["min_new"])), attention_mask=torch.from_numpy(mask),
                     max_new_token=int(meta["max_new"]))
    assert [int(i.shape[0]) for i in out.ids] == z["lens"].tolist()
    for b, n in enumerate(z["lens"]):
        assert np.array_equal(out.ids[b].numpy(), z["ids"][b, :n].astype(np.int64)), b
    for k, r in enumerate(int(x) for x in meta["hidden_rows"]):
        n = int(z["lens"][r])
        assert np.abs(out.hiddens[r].numpy() - z["hiddens"][k, :n]).max() <= 2e-5, r


def test_dvae_full_decode_codes_matches_reference_decoder_stack():
    """use_decoder=False (pipeline:292): the oracle's ids -> GFSQ._embed -> decoder chain against the mel the REFERENCE's DVAE module produced
    from the same latent (dvae_full_decode_real.npz); the embed itself (third-party quantiser) is checked against its defining property:
    the latent of a code id is the implicit FSQ codebook entry."""
    z = np.load(os.path.join(GOLDEN, "dvae_full_decode_real.npz"))
    sd = synth.dvae_full_decoder_state_dict(synth.DVAE_FULL_DEC, int(z["weight_seed"]))
    for n in (int(x) for x in z["lengths"]):
        ids = torch.from_numpy(z[f"ids_{n}"].astype(np.int64))
        mel = ref_cpu.dvae_decode_codes(sd, ids).numpy()
        assert np.abs(mel - z[f"mel_{n}"]).max() <= 2e-5, n
    # implicit codebook: id = sum_d level_d * 5^d  ->  code_d = (level_d - 2) / 2; second residual level scaled by 1/4
    ids = torch.tensor([[0, 0, 0, 0], [624, 0, 312, 0], [1, 5, 25, 125]])
    lat = ref_cpu.gfsq_latent_from_indices(ids.t().contiguous())
    assert torch.equal(lat[0][0], torch.tensor([-1.25, -1.25, -1.25, -1.25]))          # id 0: all levels 0 -> -1, + (-1) / 4
    assert torch.equal(lat[0][1], torch.tensor([0.75, 0.75, 0.75, 0.75]))              # 624 -> +1 on every dim, second id 0 -> -1 / 4
    assert torch.equal(lat[1][2], torch.tensor([-1.0, -1.0, -0.5, -1.0]) + torch.tensor([-1.0, -1.0, -1.0, -0.5]) / 4)


def test_device_noise_mode_matches_reference():
    """gpt_real_device_noise: the reference's own GPT.generate served the reference's way (slices of 4) with torch.multinomial replaced by its
    definition argmax(p / q) on the DEVICE noise stream (oracle/device_noise.py keyed by request seed, utterance id, codebook, the utterance's
    own step) -- the oracle fed the same stream reproduces every utterance, ragged EOS endings included.  The GPU tests hold noise="device"
    (sliced, and continuous batching) to this fixture."""
    from oracle.device_noise import exp_noise
    z, meta = load_golden("gpt_real_device_noise")
    sd, ids, mask, _ = gen_case_inputs(meta, synth.GPT_REAL)
    seed, uids, N = int(meta["noise_seed"]), [int(u) for u in meta["utt_ids"]], int(meta["max_new"])
    assert len(set(z["lens"].tolist())) >= 4 and z["lens"].min() < 16
    o = ref_cpu.OracleGPT(sd, 12)
    got_ids, got_h = [], []
    for s0 in range(0, len(uids), int(meta["slice_size"])):
        sl = slice(s0, s0 + int(meta["slice_size"]))
        us = uids[sl]
        q = np.stack([np.stack([exp_noise(seed, us[r // 4], r % 4, step, 0, 626) for r in range(4 * len(us))]) for step in range(N)])
        emb = o.embed(torch.from_numpy(ids[sl]), torch.ones(ids[sl].shape[:2], dtype=torch.bool))
        out = o.generate(emb, torch.from_numpy(ids[sl]), ref_cpu.SamplerParams(min_new_token=int(meta["min_new"])), attention_mask=torch.from_numpy(mask[sl]),
                         max_new_token=N, noise=ref_cpu.ArrayNoise(q))
        got_ids += out.ids
        got_h += out.hiddens
    assert [int(i.shape[0]) for i in got_ids] == z["lens"].tolist()
    for b, n in enumerate(z["lens"]):
        assert np.array_equal(got_ids[b].numpy(), z["ids"][b, :n].astype(np.int64)), b
    for k, r in enumerate(int(x) for x in meta["hidden_rows"]):
        n = int(z["lens"][r])
        assert np.abs(got_h[r].numpy() - z["hiddens"][k, :n]).max() <= 2e-5, r


def test_refine_text_device_noise_mode_matches_reference():
    """gpt_real_text_device_noise: the refine-text pass of the reference with multinomial = argmax(p / q) on stream 4 of the device noise."""
    from oracle.device_noise import exp_noise
    z, meta = load_golden("gpt_real_text_device_noise")
    cfg = synth.GPT_REAL
    sd = synth.gpt_state_dict(cfg, int(meta["weight_seed"]))
    eos = int(meta["eos"])
    sd["head_text.parametrizations.weight.original0"][eos] *= float(meta["eos_boost"])
    B, T, N = int(meta["B"]), int(meta["T"]), int(meta["max_new"])
    seed, uids = int(meta["noise_seed"]), [int(u) for u in meta["utt_ids"]]
    ids, mask = synth.prompt_ids(B, T, cfg["num_text_tokens"], int(meta["prompt_seed"]), pad_left=[int(x) for x in meta["pad_left"]])
    o = ref_cpu.OracleGPT(sd, cfg["num_attention_heads"])
    emb = o.embed(torch.from_numpy(ids), torch.ones(B, T, dtype=torch.bool))
    q = np.stack([np.stack([exp_noise(seed, uids[b], 4, step, 0, 21178) for b in range(B)]) for step in range(N)])
    out = o.generate_text(emb, torch.from_numpy(ids), 0.7, eos, attention_mask=torch.from_numpy(mask), max_new_token=N, min_new_token=int(meta["min_new"]),
                          noise=ref_cpu.ArrayNoise(q))
    assert [int(i.shape[0]) for i in out.ids] == z["lens"].tolist() and len(set(z["lens"].tolist())) == 3
    for b, n in enumerate(z["lens"]):
        assert np.array_equal(out.ids[b].numpy(), z["ids"][b, :n].astype(np.int64))
