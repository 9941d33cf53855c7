"""The persistent decode launch (persist_layer.hip: the 20-layer stack of a batch-1 decode step as ONE launch of 256 resident workgroups, activations
handed between workgroups as tagged 8-byte granules) against the launch path it replaces -- same loop of the reference, gpt.py:389-546 over
llama.py:719-749 -- plus the advisor's round-3 finding on the repetition-penalty gate under continuous batching.

The bit-exact-vs-reference statement for the mode rides on the existing goldens: every batch-1 golden of tests/test_gpu_gpt.py and the 512-token
property test now run through the persistent launch (the fp32 engine's default for one decode row)."""
import numpy as np
import pytest
import torch

from chatttsplus_amd import synth

pytestmark = pytest.mark.gpu

LW = [type("P", (), dict(top_p=0.7, min_tokens_to_keep=3))(), type("K", (), dict(top_k=20))()]
LP = [type("R", (), dict(penalty=1.05, past_window=16, max_input_ids=625))()]
LLAMA = dict(hidden_size=768, intermediate_size=3072, num_attention_heads=12, num_hidden_layers=20)


@pytest.fixture(scope="module")
def gpt():
    from chatttsplus_amd.hip_models import GPT
    g = GPT(LLAMA, max_batch=8, max_seq_len=2100, weight_dtype="fp32")
    g.load_state_dict(synth.gpt_state_dict(synth.GPT_REAL, 1234))
    yield g
    g.close()


def _gen(g, B, P, N, pad=None, lp=LP, seed=7):
    ids, mask = synth.prompt_ids(B, P, 21178, 4321, pad_left=pad)
    emb = g(torch.from_numpy(ids), torch.ones(B, P, dtype=torch.bool))
    res = list(g.generate(emb, torch.from_numpy(ids), torch.tensor([0.3] * 4), 625, attention_mask=torch.from_numpy(mask), max_new_token=N, min_new_token=N,
                          logits_warpers=LW, logits_processors=lp, return_hidden=True, noise="device", seed=seed))[-1]
    return res.ids, res.hiddens


def test_persistent_launch_is_the_default_and_matches_the_launch_path(gpt):
    g = gpt
    assert g.get_option("persistent_rows") == 8, "fp32 engines serve up to eight decode rows through the persistent launch by default (5: one attention item per workgroup; 6..8: two)"
    # contexts beyond 512 keys split every (row, head) over 2..5 attention workgroups (2 at 600, 3 at 1000, 5 at 1900; two rows: at most 2)
    # (round 5: a key share's tail streams 4 steps per round trip, so 3-4 rows stay on the persistent launch up to 1400 keys -- (4, 1200) and (3, 1300) below)
    cases = [(1, 48, 96, None), (1, 600, 24, None), (1, 1000, 40, None), (1, 1900, 24, None), (2, 700, 16, [0, 150]), (4, 1200, 16, [0, 30, 7, 300]), (3, 1300, 16, None),
             (2, 40, 32, [0, 9]), (3, 33, 24, [0, 5, 17]), (4, 48, 24, [3, 0, 11, 20]), (5, 40, 24, [0, 4, 9, 2, 13]), (5, 1300, 12, None)]
    for (B, P, N, pad) in cases:
        g.set_option("persistent_rows", 0)
        ref_ids, ref_h = _gen(g, B, P, N, pad)
        variants = [dict(persistent_rows=5), dict(persistent_rows=5, persistent_layers_per_launch=1), dict(persistent_rows=5, persistent_schedule=2, persistent_poll=1),
                    dict(persistent_rows=5, persistent_schedule=1, persistent_poll=0), dict(persistent_rows=5, persistent_schedule=3, persistent_delay=0, persistent_delay_act=0, persistent_delay_x=0)]
        for v in variants:
            for k, val in v.items():
                g.set_option(k, val)
            ids, hid = _gen(g, B, P, N, pad)
            for b in range(B):
                assert torch.equal(ids[b], ref_ids[b]), f"B={B} P={P} {v}: row {b} tokens differ from the launch path"
                assert float((hid[b] - ref_h[b]).abs().max()) <= 5e-5, (B, P, v, b)
            g.set_option("persistent_layers_per_launch", 0)
            g.set_option("persistent_schedule", 3)
            g.set_option("persistent_poll", 0)
            g.set_option("persistent_delay", 12); g.set_option("persistent_delay_act", 14); g.set_option("persistent_delay_x", 15)
    g.set_option("persistent_rows", 8)


def test_persistent_launch_serves_six_to_eight_rows_with_two_attention_items_per_workgroup(gpt):
    """Round 6: 12 R (row, head) items exceed the 64 attention workgroups from 6 rows on; the launch then gives every attention workgroup TWO items (4 compute waves + 1
    edge wave each, 192 keys per item requested before the query exists, the rest streamed behind it).  Same loop of the reference (gpt.py:389-546 over llama.py:719-749):
    token ids identical to the launch chain, hidden rows within the persistent launch's tolerance, at contexts inside and beyond the prefetched 192 keys."""
    g = gpt
    cases = [(6, 40, 24, [0, 4, 9, 2, 13, 1]), (7, 48, 24, None), (8, 33, 32, [0, 5, 17, 3, 0, 9, 30, 2]), (8, 250, 16, None), (6, 420, 12, [0, 100, 7, 0, 300, 1]), (7, 650, 8, None)]
    try:
        for (B, P, N, pad) in cases:
            g.set_option("persistent_rows", 0)
            ref_ids, ref_h = _gen(g, B, P, N, pad)
            g.set_option("persistent_rows", 8)
            assert g.get_option("persistent_rows") == 8
            ids, hid = _gen(g, B, P, N, pad)
            for b in range(B):
                assert torch.equal(ids[b], ref_ids[b]), f"B={B} P={P}: row {b} tokens differ from the launch path"
                assert float((hid[b] - ref_h[b]).abs().max()) <= 5e-5, (B, P, b)
        # bitwise reproducible, graph == eager
        a_ids, a_h = _gen(g, 8, 48, 40)
        b_ids, b_h = _gen(g, 8, 48, 40)
        g.use_graph = False
        try:
            c_ids, c_h = _gen(g, 8, 48, 40)
        finally:
            g.use_graph = True
        for b in range(8):
            assert torch.equal(a_ids[b], b_ids[b]) and torch.equal(a_h[b], b_h[b])
            assert torch.equal(a_ids[b], c_ids[b]) and torch.equal(a_h[b], c_h[b])
    finally:
        g.set_option("persistent_rows", 8)


def test_persistent_launch_replay_is_bitwise_reproducible_and_graph_equals_eager(gpt):
    g = gpt
    g.set_option("persistent_rows", 1)
    a_ids, a_h = _gen(g, 1, 48, 128)
    b_ids, b_h = _gen(g, 1, 48, 128)
    g.use_graph = False
    try:
        c_ids, c_h = _gen(g, 1, 48, 128)
    finally:
        g.use_graph = True
    assert torch.equal(a_ids[0], b_ids[0]) and torch.equal(a_h[0], b_h[0]), "two replays differ (fixed reduction orders, no atomics on the data path)"
    assert torch.equal(a_ids[0], c_ids[0]) and torch.equal(a_h[0], c_h[0]), "hipGraph replay != eager launches"
    g.set_option("persistent_rows", 8)


def test_repetition_penalty_reaches_every_utterance_of_a_long_queue(gpt):
    """the F8 gate (processors.py:23-27: the penalty skips rows >= 625 of the flattened [B * 4] batch the REFERENCE runs) was applied to the
    utterance's index in the caller's output arrays; under continuous batching that index grows without bound, so from utterance 157 on the penalty
    was silently off.  170 utterances through 8 rows with a strong penalty: every utterance must equal what a slice gives it."""
    g = gpt
    NU, T, N = 170, 6, 14
    strong = [type("R", (), dict(penalty=2.0, past_window=16, max_input_ids=625))()]
    ids, mask = synth.prompt_ids(NU, T, 21178, 91)
    emb = g(torch.from_numpy(ids), torch.ones(NU, T, dtype=torch.bool))
    uids = list(range(NU))
    kw = dict(max_new_token=N, min_new_token=N, logits_warpers=LW, return_hidden=False, seed=5)

    def sliced(lp, lo, hi):
        out = []
        for i in range(lo, hi, 8):
            sl = slice(i, min(i + 8, hi))
            out += list(g.generate(emb[sl].contiguous(), torch.from_numpy(ids[sl]), torch.tensor([0.3] * 4), 625, attention_mask=torch.from_numpy(mask[sl]),
                                   noise="device", utt_ids=uids[sl], logits_processors=lp, **kw))[-1].ids
        return out

    many = g.generate_many(emb, torch.from_numpy(ids), torch.tensor([0.3] * 4), 625, attention_mask=torch.from_numpy(mask), utt_ids=uids, rows=8,
                           logits_processors=strong, **kw)
    want = sliced(strong, 144, NU)
    plain = sliced([], 144, NU)
    assert any(not torch.equal(a, b) for a, b in zip(want[13:], plain[13:])), "the penalty changes nothing on these utterances: the case has no teeth"
    for u in range(144, NU):
        assert torch.equal(many.ids[u], want[u - 144]), f"utterance {u}: continuous batching sampled without the repetition penalty"


def test_out_of_range_ids_raise_like_nn_embedding(gpt):
    """gpt.py:125-149: nn.Embedding raises IndexError on an id outside its table; the gather kernel does not check, so GPT.__call__ does."""
    g = gpt
    ids, _ = synth.prompt_ids(2, 6, 21178, 3)
    tm = torch.ones(2, 6, dtype=torch.bool)
    g(torch.from_numpy(ids), tm)                                         # fine
    bad = ids.copy(); bad[1, 2, :] = 21178                               # one past the text table
    with pytest.raises(IndexError):
        g(torch.from_numpy(bad), tm)
    bad = ids.copy(); bad[0, 0, :] = -1
    with pytest.raises(IndexError):
        g(torch.from_numpy(bad), tm)
    code = ids.copy(); code[:, 4:, :] = 625                              # code rows (text_mask 0): ids < 626 per codebook
    tm2 = tm.clone(); tm2[:, 4:] = False
    g(torch.from_numpy(code), tm2)
    code[0, 5, 3] = 626
    with pytest.raises(IndexError):
        g(torch.from_numpy(code), tm2)


def test_a_withheld_hand_off_ends_the_step_with_an_error_instead_of_hanging(gpt):
    """Every wait of the persistent launch is bounded: with one workgroup withholding its columns of (x + attention) in layer 7 (test hook), the waiting
    waves give up after ~0.3 s, the device error word turns the rest of the launch and every later launch into no-ops, and the host gets a
    HipBackendError naming an edge (whichever starved consumer ran out of passes first) -- then the engine serves the next request normally."""
    import time
    from chatttsplus_amd import _lib
    g = gpt
    g.set_option("persistent_rows", 8)
    ref_ids, _ = _gen(g, 1, 24, 12)
    g.set_option("persistent_fault", 8)
    t0 = time.perf_counter()
    with pytest.raises(_lib.HipBackendError, match="gave up waiting on edge [2-5]"):
        _gen(g, 1, 24, 12)
    assert time.perf_counter() - t0 < 20.0, "the give-up took too long: a wait is not bounded"
    g.set_option("persistent_fault", 0)
    ids, _ = _gen(g, 1, 24, 12)
    assert torch.equal(ids[0], ref_ids[0]), "the engine did not recover after the reported give-up"


def test_two_engines_of_one_process_take_turns_with_their_persistent_launches():
    """VERDICT r4 / ADVICE r4: the persistent mode's lock is per process, so two engines of ONE process (a base engine and a second pipeline / a LoRA sibling) both own
    it; driven from two threads on two streams their launches -- each needs all 256 workgroups resident -- could starve each other until the 0.3 s give-up.  Decode
    calls that launch persistent kernels now take turns (an event chain between the streams, gpt_engine.hip PersistTurn): both threads finish without a give-up and
    with the tokens a solo run gives them."""
    import threading
    from chatttsplus_amd.hip_models import GPT
    sd = synth.gpt_state_dict(synth.GPT_REAL, 1234)
    gs = [GPT(LLAMA, max_batch=2, max_seq_len=400, weight_dtype="fp32") for _ in range(2)]
    try:
        for g in gs:
            g.load_state_dict(sd)
            assert g.get_option("persistent_rows") == 8
        solo = [_gen(gs[i], 1 + i, 40, 96, seed=11 + i) for i in range(2)]
        out, err = [None, None], [None, None]

        def work(i):
            try:
                with torch.cuda.stream(torch.cuda.Stream()):
                    for _ in range(3):
                        out[i] = _gen(gs[i], 1 + i, 40, 96, seed=11 + i)
            except BaseException as e:          # noqa: BLE001 (reported below)
                err[i] = e

        ts = [threading.Thread(target=work, args=(i,)) for i in range(2)]
        for t in ts:
            t.start()
        for t in ts:
            t.join(timeout=300)
        assert err == [None, None], err
        for i in range(2):
            for r in range(1 + i):
                assert torch.equal(out[i][0][r], solo[i][0][r]) and torch.equal(out[i][1][r], solo[i][1][r]), f"engine {i} row {r}: concurrent run differs from the solo run"
    finally:
        for g in gs:
            g.close()


def test_persistent_weight_images_are_built_on_first_need(gpt):
    """the persistent launch keeps its own per-workgroup image of the layer weights (755 MB for 20 layers).  It is built by the first decode call of <= 4
    rows, not at load time: an engine that only serves larger batches never allocates it."""
    from chatttsplus_amd.hip_models import GPT
    g = GPT(LLAMA, max_batch=12, max_seq_len=200, weight_dtype="fp32")
    try:
        g.load_state_dict(synth.gpt_state_dict(synth.GPT_REAL, 1234))
        assert g.get_option("persistent_rows") == 8
        torch.cuda.synchronize()
        free0 = torch.cuda.mem_get_info()[0]
        _gen(g, 12, 24, 8)                                   # the launch chain: no image
        torch.cuda.synchronize()
        free1 = torch.cuda.mem_get_info()[0]
        ids_a, _ = _gen(g, 1, 24, 8)                         # first <= 4-row decode: 20 x 37.75 MB
        torch.cuda.synchronize()
        free2 = torch.cuda.mem_get_info()[0]
        assert free0 - free1 < 300 * 2 ** 20, "the batch-12 request allocated the persistent weight images"
        assert free1 - free2 > 600 * 2 ** 20, "the batch-1 request did not build the persistent weight images"
        ref_ids, _ = _gen(gpt, 1, 24, 8)
        assert torch.equal(ids_a[0], ref_ids[0])
    finally:
        g.close()


def test_fp16_engines_use_the_persistent_launch_too():
    """Round 5: the fast mode had stayed on the launch chain, so at batch 1 it was SLOWER than the parity mode (0.373 vs 0.284 ms/step).  The persistent
    launch now also takes a half-precision weight image and a half K / V cache (activations, granules and accumulation stay fp32): default for <= 5 rows of an fp16
    engine (6..8 rows stay on its launch chain, which is faster there).  Against the fp16 launch chain (which rounds the MFMA operands to fp16): the same first tokens, hidden states within the fast mode's tolerance while the
    tokens agree; replays bitwise identical, hipGraph == eager."""
    from chatttsplus_amd.hip_models import GPT
    g = GPT(LLAMA, max_batch=4, max_seq_len=300, weight_dtype="fp16")
    try:
        g.load_state_dict(synth.gpt_state_dict(synth.GPT_REAL, 1234))
        assert g.get_option("persistent_rows") == 5
        for B in (1, 3):
            g.set_option("persistent_rows", 0)
            c_ids, c_h = _gen(g, B, 40, 32, [0, 5, 9][:B])
            g.set_option("persistent_rows", 5)
            p_ids, p_h = _gen(g, B, 40, 32, [0, 5, 9][:B])
            q_ids, q_h = _gen(g, B, 40, 32, [0, 5, 9][:B])
            g.use_graph = False
            try:
                e_ids, e_h = _gen(g, B, 40, 32, [0, 5, 9][:B])
            finally:
                g.use_graph = True
            for b in range(B):
                assert torch.equal(p_ids[b], q_ids[b]) and torch.equal(p_h[b], q_h[b]), "two replays differ"
                assert torch.equal(p_ids[b], e_ids[b]) and torch.equal(p_h[b], e_h[b]), "hipGraph replay != eager launches"
                same = int((p_ids[b] == c_ids[b]).all(-1).to(torch.int32).cumprod(0).sum())
                assert same >= 3, f"B={B} row {b}: only {same} leading tokens agree with the fp16 launch chain"
                rel = float((p_h[b][:same] - c_h[b][:same]).pow(2).mean().sqrt() / c_h[b][:same].pow(2).mean().sqrt())
                assert rel <= 2e-3, f"B={B} row {b}: hidden states {rel} away from the fp16 launch chain over the {same} agreeing steps"
    finally:
        g.close()


@pytest.mark.parametrize("wd", ["fp32", "fp16"])
def test_per_utterance_adapters_ride_inside_the_persistent_launch(wd):
    """Round 6: rows that carry a LoRA adapter (pipeline:420-432 per row instead of per batch) stay on the persistent launch -- waves 6 / 7 of
    the GEMV workgroups compute u = A h, the edge lanes add scale * B u to their q / k / v / o_proj rows (persist_layer.hip LORA).  Against the launch chain's worker
    workgroups (lora_worker.h; itself held to the per-row merged-weights oracle by tests/test_gpu_pipeline.py): fp32 token ids identical, hiddens <= 5e-5;
    fp16: first hidden states within 2e-3.  1..8 rows, rows with different adapters / ranks / none, with and without graphs."""
    import ctypes as C
    from chatttsplus_amd import _lib
    from chatttsplus_amd.hip_models import GPT
    cfg = dict(synth.GPT_REAL); cfg["num_hidden_layers"] = 6
    llama = dict(LLAMA, num_hidden_layers=6)
    sd = synth.gpt_state_dict(cfg, 1234)
    rng = np.random.Generator(np.random.Philox(key=79))
    g = GPT(llama, max_batch=8, max_seq_len=96, weight_dtype=wd)
    g.load_state_dict(sd)
    for ai, r in enumerate((8, 16, 3)):
        ad = []
        for l in range(6):
            for t in ("q_proj", "k_proj", "v_proj", "o_proj"):
                ad.append((l, t, (rng.standard_normal((r, 768)) * 0.05).astype(np.float32), (rng.standard_normal((768, r)) * 0.05).astype(np.float32), 2.0 - 0.5 * ai))
        g.load_adapter(ai, ad)
    assert g.get_option("persistent_lora") == 1
    g.set_option("persistent_rows", 8)

    def epoch():
        buf = np.zeros(8, dtype=np.uint8); got = C.c_size_t(0)
        _lib.check(g._lib.ctts_gpt_debug_read(g._h, b"pl_state", buf.ctypes.data_as(C.c_void_p), 8, C.byref(got), g._stream()), "debug_read")
        return int(buf.view(np.uint32)[0]), int(buf.view(np.uint32)[1])

    def run(B, slots, on, graphs):
        g.set_option("persistent_lora", on)
        g.use_graph = graphs
        g.set_row_adapters(slots)
        ids, mask = synth.prompt_ids(B, 18, cfg["num_text_tokens"], 23, pad_left=[(3 * b) % 5 for b in range(B)])
        emb = g(torch.from_numpy(ids), torch.ones(B, 18, dtype=torch.bool))
        out = list(g.generate(emb, torch.from_numpy(ids), torch.tensor([0.3] * 4), 625, attention_mask=torch.from_numpy(mask), max_new_token=12, min_new_token=12,
                              logits_warpers=LW, logits_processors=LP, return_hidden=True, noise="device", seed=5))[-1]
        g.set_row_adapters(None)
        return out

    for B, slots in ((1, [1]), (2, [0, -1]), (3, [2, 1, 0]), (5, [0, 1, -1, 2, 1]), (6, [1, -1, 0, 0, 2, -1]), (8, [0, 1, 2, -1, 1, 0, -1, 2])):
        ref = run(B, slots, 0, True)
        e0, _ = epoch()
        for graphs in (True, False):
            out = run(B, slots, 1, graphs)
            e1, err = epoch()
            assert err == 0 and e1 > e0, f"B={B}: the persistent launch did not run (launch counter {e0} -> {e1}, error word {err})"
            e0 = e1
            for b in range(B):
                if wd == "fp32":
                    assert torch.equal(out.ids[b], ref.ids[b]), f"B={B} row {b} (slot {slots[b]}, graphs {graphs}): token ids differ from the launch chain"
                    assert float((out.hiddens[b] - ref.hiddens[b]).abs().max()) <= 5e-5, f"B={B} row {b}"
                else:
                    h0, r0 = out.hiddens[b][0], ref.hiddens[b][0]
                    assert float((h0 - r0).pow(2).mean().sqrt() / r0.pow(2).mean().sqrt()) <= 2e-3, f"B={B} row {b}"
        base = run(B, None, 1, True)
        assert any(slots[b] >= 0 and not torch.equal(base.hiddens[b], ref.hiddens[b]) for b in range(B)), "the adapters change nothing"
    g.close()
