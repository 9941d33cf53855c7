"""CPU tests of the host side: C-ABI library loads and exports every declared symbol, header/binding agree,
sampler-config plumbing, RoPE table, tokenizer batching, no-GPU failure is loud (no CPU fallback)."""
import os
import re

import numpy as np
import pytest
import torch

from chatttsplus_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_header_symbol():
    lib = _lib.load()
    hdr = open(os.path.join(ROOT, "include", "ctts_hip.h")).read()
    declared = set(re.findall(r"\b(ctts_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"ctts_gpt", "ctts_voc"}
    bound = {n for n, _, _ in _lib.SYMBOLS}
    assert declared == bound, f"header vs binding mismatch: {declared ^ bound}"
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.ctts_version() >= 1


def test_struct_layouts_match_header_sizes():
    import ctypes as C
    assert C.sizeof(_lib.GptCfg) == 9 * 4
    assert C.sizeof(_lib.SamplerCfg) == 4 * 4 + 4 + 3 * 4 + 17 * 4 + 6 * 4
    assert C.sizeof(_lib.VocCfg) == 12 * 4 + 2 * 4 + 4 * 4
    assert C.sizeof(_lib.GenIO) == 5 * 8 + 8 + 8 + 2 * 8  # 5 pointers, int32 (+pad), uint64, 2 pointers


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_no_gpu_fails_loudly():
    from chatttsplus_amd.hip_models import GPT
    with pytest.raises(_lib.HipBackendError):
        GPT(dict(hidden_size=768, intermediate_size=3072, num_attention_heads=12, num_hidden_layers=1))


def test_sampler_cfg_from_reference_style_objects():
    from chatttsplus_amd.hip_models.gpt import sampler_cfg_from_objects
    from chatttsplus_amd.pipeline import gen_logits
    w, p = gen_logits(625, 0.7, 20, 1.05)
    sc = sampler_cfg_from_objects(torch.tensor([0.3, 0.4, 0.5, 0.6]), 625, 2048, 3, w, p, 4)
    assert [round(sc.temperature[i], 6) for i in range(4)] == [0.3, 0.4, 0.5, 0.6]
    assert sc.top_p_threshold == float(np.float32(1 - 0.7)) and sc.top_k == 20 and sc.min_tokens_to_keep == 3
    assert sc.use_penalty == 1 and sc.past_window == 16 and sc.max_input_ids == 625
    tab = torch.pow(1.05, torch.arange(17))
    assert all(sc.penalty_table[i] == float(tab[i]) for i in range(17))
    # HF objects carry the same attribute names
    from transformers.generation import TopKLogitsWarper, TopPLogitsWarper
    sc2 = sampler_cfg_from_objects(torch.tensor([0.3]), 625, 10, 0, [TopPLogitsWarper(0.7, min_tokens_to_keep=3), TopKLogitsWarper(1, min_tokens_to_keep=3)], [], 4)
    assert sc2.top_k == 3 and sc2.use_penalty == 0 and sc2.top_p_threshold == sc.top_p_threshold
    with pytest.raises(_lib.HipBackendError):
        sampler_cfg_from_objects(torch.tensor([0.3]), 625, 10, 0, [object()], [], 4)


def test_rope_table_matches_oracle():
    from chatttsplus_amd.hip_models.gpt import rope_table
    from oracle import ref_cpu
    o = object.__new__(ref_cpu.OracleGPT)
    o.inv_freq = 1.0 / (10000.0 ** (torch.arange(0, 64, 2, dtype=torch.int64).float() / 64))
    cos, sin = ref_cpu.OracleGPT._rope(o, torch.arange(300)[None])
    tab = rope_table(300)
    assert np.array_equal(tab[:, :32], cos[0, :, :32].numpy()) and np.array_equal(tab[:, 32:], sin[0, :, :32].numpy())


def test_tokenizer_left_padding_and_prompt(tmp_path):
    from transformers import BertTokenizerFast
    from chatttsplus_amd import codec
    from chatttsplus_amd.tokenizer import Tokenizer
    vocab = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]", "[Stts]", "[Ptts]", "[spk_emb]", "[empty_spk]", "[uv_break]", "[break_0]",
             "[Ebreak]", "[speed_5]", "a", "b", "c", "d"]
    (tmp_path / "vocab.txt").write_text("\n".join(vocab))
    bt = BertTokenizerFast(vocab_file=str(tmp_path / "vocab.txt"), do_lower_case=False)
    bt.add_special_tokens({"additional_special_tokens": [v for v in vocab if v.startswith("[") and v not in ("[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]")]})
    tok = Tokenizer(tokenizer=bt)
    assert tok.spk_emb_ids == 7 and tok.eos_token == 11 and tok.break_0_ids == 10
    ids, att, tm = tok.encode(["[Stts][spk_emb]a b c[Ptts]", "[Stts][spk_emb]a[Ptts]"], 4)
    assert ids.shape == (2, 6, 4) and att.tolist() == [[1] * 6, [0, 0, 1, 1, 1, 1]]
    assert (ids[..., 0] == ids[..., 3]).all() and ids[1, 2, 0] == 5 and ids[1, 0, 0] == 0
    prompt = torch.randint(0, 626, (4, 5))
    ids2, att2, tm2 = tok.encode(["[Stts][spk_emb]a[Ptts]"], 4, prompt_str=codec.encode_prompt(prompt))
    assert ids2.shape == (1, 9, 4) and torch.equal(ids2[0, 4:], prompt.t()) and tm2[0].tolist() == [True] * 4 + [False] * 5 and att2.all()


def test_load_lora_adapter_reads_peft_files(tmp_path):
    import json
    from safetensors.numpy import save_file
    from chatttsplus_amd.pipeline import load_lora_adapter
    (tmp_path / "adapter_config.json").write_text(json.dumps(dict(r=8, lora_alpha=16, target_modules=["q_proj", "v_proj"])))
    A = np.random.randn(8, 768).astype(np.float32); B = np.random.randn(768, 8).astype(np.float32)
    save_file({"base_model.model.layers.3.self_attn.q_proj.lora_A.weight": A, "base_model.model.layers.3.self_attn.q_proj.lora_B.weight": B},
              str(tmp_path / "adapter_model.safetensors"))
    ad = load_lora_adapter(str(tmp_path))
    assert len(ad) == 1 and ad[0][0] == 3 and ad[0][1] == "q_proj" and ad[0][4] == 2.0 and np.array_equal(ad[0][2], A)


def test_audio_resample_and_loader(tmp_path):
    """Zero-shot host pre-processing (pipeline:493-496): wav reader + windowed-sinc resampler."""
    import numpy as np
    import torch
    from scipy.io import wavfile
    from chatttsplus_amd import audio
    sr = 16000
    t = np.arange(sr, dtype=np.float64) / sr
    x = 0.5 * np.sin(2 * np.pi * 440.0 * t)
    wavfile.write(tmp_path / "a.wav", sr, np.stack([x, x], 1).astype(np.float32))
    wav, got_sr = audio.load_audio(str(tmp_path / "a.wav"))
    assert got_sr == sr and wav.shape == (2, sr)
    y = audio.resample(wav, sr, 24000)
    assert y.shape == (2, 24000)
    ref = 0.5 * np.sin(2 * np.pi * 440.0 * np.arange(24000) / 24000.0)
    assert np.abs(y[0].numpy()[200:-200] - ref[200:-200]).max() < 2e-3      # a 440 Hz tone survives 16 -> 24 kHz
    assert audio.resample(wav, 24000, 24000) is wav
    wavfile.write(tmp_path / "b.wav", sr, (x * 32767).astype(np.int16))
    w16, _ = audio.load_audio(str(tmp_path / "b.wav"))
    assert w16.shape == (1, sr) and abs(float(w16.abs().max()) - 0.5) < 1e-3


def test_window_token_range_covers_the_receptive_field():
    """Streaming: the token range chosen for a sample window contains every mel frame the window can depend on
    (modelled with a symmetric dependency of halo-3 frames per ISTFT frame and 4 overlapping ISTFT frames), keeps real context on
    both sides unless the range touches the utterance edge, and never exceeds the prefix."""
    import chatttsplus_amd.hip_models.vocoder as voc        # pure python: importing does not touch the native library
    hop, halo = 256, 105
    rng = np.random.Generator(np.random.Philox(key=3))
    for _ in range(2000):
        n = int(rng.integers(1, 900))
        total = hop * (2 * n - 1)
        s0 = int(rng.integers(0, total + 50)); s1 = int(rng.integers(s0, total + 400))
        a, b, off, c0, c1 = voc.window_token_range(n, s0, s1, hop, halo)
        if c1 <= c0:
            assert (a, b) == (0, 0)
            continue
        assert 0 <= a < b <= n and off == 2 * a * hop and 0 <= c0 < c1 <= total
        need_lo = max(0, c0 // hop - 1 - (halo - 3))                    # first / last mel frame the window depends on
        need_hi = min(2 * n - 1, (c1 - 1) // hop + 2 + (halo - 3))
        assert 2 * a <= need_lo and need_hi <= 2 * b - 1
        # context rule: a truncated edge lies at least a halo away from every needed output frame
        if a > 0:
            assert c0 // hop - 1 - 2 * a >= halo - 3
        if b < n:
            assert 2 * b - 1 - ((c1 - 1) // hop + 2) >= halo - 3
        assert c1 - off <= hop * (2 * (b - a) - 1)                      # the window ends inside the sub-waveform


def test_dvae_class_dispatches_to_the_encoder_for_encode_configs():
    """`name: "DVAE"` builds both dvae_decode and dvae_encode in the reference's YAML: a config with encoder / vq goes to the
    zero-shot encoder class (which, like every hip model, refuses to run without a GPU)."""
    from chatttsplus_amd import _lib, hip_models
    with pytest.raises(_lib.HipBackendError, match="MI355X"):
        hip_models.DVAE(decoder_config=dict(idim=512, odim=512, hidden=256, n_layer=12, bn_dim=128),
                        encoder_config=dict(idim=512, odim=1024, hidden=256, n_layer=12, bn_dim=128),
                        vq_config=dict(dim=1024, levels=[5, 5, 5, 5], G=2, R=2), dim=512)


def test_merge_short_sentences_matches_the_reference_rule():
    """pipeline:353-377: the merge of short sentences behind the (pluggable) splitter -- checked against a direct transcription of
    the rule's observable behaviour on hand-made cases (the reference function cannot be called in isolation: it is inlined in _infer)."""
    from chatttsplus_amd.pipeline import merge_short_sentences as m
    long_a, long_b = "x" * 40, "y" * 31
    assert m([long_a, long_b]) == [long_a, long_b]
    assert m(["hi"]) == ["hi [uv_break] "]                                   # nothing else exists: the chain is emitted as is
    assert m(["hi", long_a]) == ["hi [uv_break] " + long_a]                  # a long sentence absorbs the chain in front of it
    assert m([long_a, "hi"]) == [long_a + " [uv_break] hi [uv_break] "]      # short left-over appended to the last utterance
    s12 = "a" * 12
    out = m([s12, s12, s12, long_a])                                         # the chain exceeds 30 characters after two pieces
    assert out == [f"{s12} [uv_break] {s12} [uv_break] ", f"{s12} [uv_break] " + long_a]
    assert m([]) == [""]


def test_adapter_slot_cache_lru(tmp_path, monkeypatch):
    """pipeline._adapter_slots: per-utterance adapters -> resident engine slots, loaded once, least-recently-used evicted first."""
    import chatttsplus_amd.pipeline as pl
    from chatttsplus_amd import _lib
    loaded = []
    monkeypatch.setattr(pl, "load_lora_adapter", lambda p: [("adapter-of", p)])

    class G:
        def load_adapter(self, slot, adapters):
            loaded.append((slot, adapters[0][1]))
    pipe = object.__new__(pl.ChatTTSPlusPipeline)
    g = G()
    assert pipe._adapter_slots(g, ["a", None, "b", "a"]) == [0, -1, 1, 0]
    assert loaded == [(0, "a"), (1, "b")]
    assert pipe._adapter_slots(g, ["b", "b"]) == [1, 1] and len(loaded) == 2          # cached
    many = [f"p{i}" for i in range(_lib.MAX_ADAPTERS - 1)]
    slots = pipe._adapter_slots(g, many + ["b"])                                        # "a" is the least recently used -> evicted
    assert len(set(slots)) == _lib.MAX_ADAPTERS and "a" not in pipe._slot_of_path and slots[-1] == 1
    import pytest
    with pytest.raises(_lib.HipBackendError):
        pipe._adapter_slots(g, [f"q{i}" for i in range(_lib.MAX_ADAPTERS + 1)])


def test_infer_resolves_the_speaker_on_a_copy_of_the_params(tmp_path):
    """infer() (pipeline:472-579): speaker_emb_path (.pt with a base16384 string or a tensor) > explicit params.spk_emb > random speaker; the
    caller's / the shared default params object is never written to."""
    import torch
    from chatttsplus_amd import codec
    from chatttsplus_amd.pipeline import ChatTTSPlusPipeline, InferCodeParams
    pipe = object.__new__(ChatTTSPlusPipeline)
    pipe.std, pipe.mean = torch.ones(768), torch.zeros(768)
    seen = []
    pipe._infer = lambda text, *a, **k: seen.append(a[9]) or iter(())         # a[9] = params_infer_code
    default = ChatTTSPlusPipeline.infer.__wrapped__.__defaults__[-1] if hasattr(ChatTTSPlusPipeline.infer, "__wrapped__") else None
    pipe.infer("a")
    pipe.infer("a")
    assert isinstance(seen[0].spk_emb, str) and isinstance(seen[1].spk_emb, str) and seen[0].spk_emb != seen[1].spk_emb    # re-sampled per call
    if default is not None:
        assert default.spk_emb is None and default.spk_smp is None                                                     # shared default untouched
    mine = InferCodeParams(spk_emb="given")
    pipe.infer("a", params_infer_code=mine)
    assert seen[2].spk_emb == "given" and seen[2] is not mine
    vec = torch.randn(768)
    torch.save(codec.encode_spk_emb(vec), tmp_path / "s.pt")
    torch.save(vec.view(1, 768), tmp_path / "t.pt")
    pipe.infer("a", params_infer_code=mine, speaker_emb_path=str(tmp_path / "s.pt"))
    pipe.infer("a", params_infer_code=mine, speaker_emb_path=str(tmp_path / "t.pt"))
    assert isinstance(seen[3].spk_emb, str) and torch.equal(torch.as_tensor(seen[4].spk_emb).view(-1), vec) and mine.spk_emb == "given"


def test_torch_seed_context_like_the_reference():
    """commons/utils.py:48-58 -- what demos wrap a request in: the tokens of a request depend on the seed only, the caller's generator state is restored."""
    import torch
    from chatttsplus_amd.pipeline import TorchSeedContext
    torch.manual_seed(123)
    before = torch.random.get_rng_state()
    with TorchSeedContext(7):
        a = torch.rand(3)
    assert torch.equal(torch.random.get_rng_state(), before)
    with TorchSeedContext(7):
        assert torch.equal(torch.rand(3), a)


def test_row_book_late_reports_admission_and_compaction():
    """GPT.generate_many's bookkeeping (hip_models.gpt.RowBook): row reports arrive late and are read through the ticket layout of the moment
    they were enqueued -- a report older than an admission must not finish the row's new occupant, rows renumbered by a compaction are
    still found, a first-token EOS is queued again with the next attempt (gpt.py:496-525) until max_restarts."""
    from chatttsplus_amd.hip_models.gpt import RowBook
    b = RowBook()
    for r in range(4):
        b.seat(r, utt=r)
    lay0 = b.layout()                                   # report A enqueued
    lay1 = b.layout()                                   # report B enqueued (one chunk later, same seating)
    # report A: utterance 1 finished by its limit, utterance 2's FIRST token was EOS
    done, again = b.report(lay0, [(0, 9), (1, 9), (3, 0), (0, 9)], ensure_non_empty=True, max_restarts=3)
    assert done == [(1, 9)] and again == [(2, 1)] and b.free_rows() == [1, 2]
    b.seat(1, utt=7)                                    # admissions into the freed rows (enqueued AFTER report B)
    b.seat(2, utt=2, attempt=1)
    # report B still shows the old occupants of rows 1 and 2 as finished: nobody new may finish through it
    done, again = b.report(lay1, [(0, 17), (1, 9), (3, 0), (1, 17)], True, 3)
    assert done == [(3, 17)] and again == [] and b.free_rows() == [3]
    lay2 = b.layout()
    # compaction: rows 0, 1, 2 kept -> a report enqueued before it is still attributed correctly afterwards
    b.compact([0, 1, 2])
    assert b.live_rows() == [0, 1, 2] and b.free_rows() == []
    done, again = b.report(lay2, [(1, 20), (0, 4), (3, 0), (1, 17)], True, 3)       # utterance 0 done; utterance 2 EOS at step 0 AGAIN
    assert done == [(0, 20)] and again == [(2, 2)]
    assert b.free_rows() == [0, 2]
    b.seat(0, utt=2, attempt=2)
    done, again = b.report(b.layout(), [(3, 0), (0, 9), (0, 0)], True, 3)           # third first-token EOS: max_restarts reached -> gives up (empty result)
    assert done == [(2, 0)] and again == []
    # without ensure_non_empty a first-token EOS simply completes the utterance
    c = RowBook()
    c.seat(0, 5)
    assert c.report(c.layout(), [(3, 0)], False, 64) == ([(5, 0)], [])
    with pytest.raises(AssertionError):
        b.seat(1, utt=9)                                # row 1 is occupied (utterance 7)


def test_warper_list_must_be_top_p_then_top_k():
    """The kernel applies top-p first and top-k second -- the order processors.gen_logits builds (models/processors.py:43-48) and gpt.py:474-475 applies;
    the two do not commute, so any other list is rejected instead of silently re-ordered."""
    from chatttsplus_amd import _lib
    from chatttsplus_amd.hip_models.gpt import sampler_cfg_from_objects
    P = type("P", (), dict(top_p=0.7, min_tokens_to_keep=3))
    K = type("K", (), dict(top_k=20))
    sc = sampler_cfg_from_objects([0.3] * 4, 625, 8, 0, [P(), K()], [], 4)
    assert sc.top_k == 20 and abs(sc.top_p_threshold - 0.3) < 1e-6
    assert sampler_cfg_from_objects([0.3] * 4, 625, 8, 0, [K()], [], 4).top_p_threshold < 0          # top-k alone / top-p alone stay legal
    assert sampler_cfg_from_objects([0.3] * 4, 625, 8, 0, [P()], [], 4).top_k == 0
    with pytest.raises(_lib.HipBackendError, match="top-p, top-k"):
        sampler_cfg_from_objects([0.3] * 4, 625, 8, 0, [K(), P()], [], 4)
    with pytest.raises(_lib.HipBackendError, match="more than one top-k"):
        sampler_cfg_from_objects([0.3] * 4, 625, 8, 0, [P(), K(), K()], [], 4)
    with pytest.raises(_lib.HipBackendError):
        sampler_cfg_from_objects([0.3] * 4, 625, 8, 0, [P(), P()], [], 4)


def test_joint_row_sum_exchange_pattern_of_the_persistent_launch():
    """persist_layer.hip `pl_reduce_store`: a wave reduces P <= 16 partial sums per lane TOGETHER -- at each of wave_sum's four DPP steps (quad xor 1, quad xor 2,
    half-mirror, mirror) a lane keeps one half of its values and hands the other half to its partner.  A numpy model of the 64 lanes checks what the kernel relies on:
    (1) the lane that the step pairs me with holds the same values as I do and keeps the other half (the parities f0 = b0^b2, f1 = b1^b2, f2 = b2^b3, f3 = b3);
    (2) after the four steps a lane holds the 16-lane row sum of value idx = f0 P/2 + f1 P/4 + ..., added in wave_sum's order (bit-identical in fp32);
    (3) the writer lanes cover every (value, row) exactly once."""
    rng = np.random.Generator(np.random.Philox(key=3))
    lanes = np.arange(64)
    b = [(lanes >> i) & 1 for i in range(4)]
    f = [b[0] ^ b[2], b[1] ^ b[2], b[2] ^ b[3], b[3]]
    row, l16 = lanes >> 4, lanes & 15
    partner = [lanes ^ 1, lanes ^ 2, (lanes & ~7) | (7 - (lanes & 7)), (lanes & ~15) | (15 - l16)]

    def wave_sum_rows(x):                       # common.h wave_sum up to the row sums: v += dpp(v), four times, fp32
        v = x.astype(np.float32).copy()
        for p in partner:
            v = (v + v[p]).astype(np.float32)
        return v                                # every lane: the sum of its 16-lane row

    for P in (1, 2, 4, 8, 16):
        vals = rng.standard_normal((64, P)).astype(np.float32)          # [lane][value]
        cur = [vals[:, i].copy() for i in range(P)]
        held = [set(range(P)) for _ in range(64)]                        # original indices a lane still holds
        C = P
        for s in range(4):
            if C > 1:
                for ln in range(64):
                    assert held[ln] == held[partner[s][ln]], "the partner must hold the same values"
                    assert f[s][ln] != f[s][partner[s][ln]], "... and keep the other half"
                nxt = []
                for i in range(C // 2):
                    keep = np.where(f[s] == 1, cur[i + C // 2], cur[i])
                    give = np.where(f[s] == 1, cur[i], cur[i + C // 2])
                    nxt.append((keep + give[partner[s]]).astype(np.float32))
                for ln in range(64):
                    order = sorted(held[ln])
                    held[ln] = set(order[C // 2:] if f[s][ln] else order[:C // 2])
                cur, C = nxt, C // 2
            else:
                cur = [(cur[0] + cur[0][partner[s]]).astype(np.float32)]
        # the value a lane ends with, by the kernel's formula
        idx = np.zeros(64, dtype=np.int64)
        writer = np.ones(64, dtype=bool)
        c = P
        for s in range(4):
            if c > 1:
                idx += f[s] * (c // 2)
                c //= 2
            else:
                writer &= f[s] == 0
        seen = set()
        for ln in range(64):
            assert held[ln] == {int(idx[ln])}
            want = wave_sum_rows(vals[:, idx[ln]])[ln]
            assert cur[0][ln] == want, "not bit-identical to wave_sum's row sum"
            if writer[ln]:
                key = (int(idx[ln]), int(row[ln]))
                assert key not in seen
                seen.add(key)
        assert seen == {(i, r) for i in range(P) for r in range(4)}
