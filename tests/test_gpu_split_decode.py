"""Decode projections of the fp32 (parity) engine on the fp16 matrix pipes (round 6; skinny_gemm.hip dispatch_split, common.h split_t): from
`split_decode_rows` rows on, weights and operands are head / tail fp16 image pairs and a product is three v_mfma_f32_16x16x32_f16 instead of eight
exact-f32 MFMAs -- the prompt pass's arithmetic (prefill_split.hip) applied to the loop of gpt.py:389-546 over llama.py:719-749.

The bit-exact-vs-reference statement rides on the goldens of tests/test_gpu_gpt.py (gpt_real_b32_ragged, the teacher-forced 17 / 32 / 33-row cases run
through this path by default).  Here: the path against the exact-f32 kernels it replaces, on the same engine."""
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
    g = GPT(LLAMA, max_batch=40, max_seq_len=400, weight_dtype="fp32")
    g.load_state_dict(synth.gpt_state_dict(synth.GPT_REAL, 1234))
    yield g
    g.close()


def _gen(g, B, P, N, pad=None, seed=7, graph=True):
    ids, mask = synth.prompt_ids(B, P, 21178, 4321, pad_left=pad)
    emb = g(torch.from_numpy(ids), torch.ones(B, P, dtype=torch.bool))
    g.use_graph = graph
    try:
        res = list(g.generate(emb, torch.from_numpy(ids), torch.tensor([0.3] * 4), 625, attention_mask=torch.from_numpy(mask), max_new_token=N, min_new_token=N,
                              logits_warpers=LW, logits_processors=LP, return_hidden=True, noise="device", seed=seed))[-1]
    finally:
        g.use_graph = True
    return res.ids, res.hiddens


@pytest.mark.parametrize("B,P,N,pad", [(9, 40, 24, None), (16, 33, 24, [(5 * i) % 30 for i in range(16)]), (17, 48, 24, None), (24, 40, 16, [(7 * i) % 36 for i in range(24)]),
                                       (32, 48, 32, [(3 * i) % 40 for i in range(32)]), (40, 30, 8, None)])
def test_split_decode_matches_the_exact_f32_kernels(gpt, B, P, N, pad):
    g = gpt
    assert g.get_option("split_decode_rows") == 9, "fp32 engines run decode batches of >= 9 rows on the head / tail images by default"
    try:
        g.set_option("split_decode_rows", 0)
        ref_ids, ref_h = _gen(g, B, P, N, pad)
        g.set_option("split_decode_rows", 9)
        ids, hid = _gen(g, B, P, N, pad)
    finally:
        g.set_option("split_decode_rows", 9)
    worst = 0.0
    for b in range(B):
        assert torch.equal(ids[b], ref_ids[b]), f"B={B}: row {b} tokens differ from the exact-f32 kernels"
        worst = max(worst, float((hid[b] - ref_h[b]).abs().max()))
    # hidden rows are O(1) (the final RMSNorm's output times its weight); the oracle tolerance of tests/test_gpu_gpt.py is 2e-5 abs
    assert worst <= 2e-5, (B, worst)


@pytest.mark.parametrize("opts", [dict(split_nbg2_rows=99), dict(weight_prefetch_kb=48), dict(weight_prefetch_kb=0)])
def test_split_decode_launch_shapes_agree(gpt, opts):
    """The other launch shapes of the same arithmetic -- 16-row chunks instead of 32-row blocks (layer 0's exact-f32 q|k|v kernel then splits K over 8 waves instead of
    16: another summation order, hence the tolerance), prefetch workgroups on / off (bitwise the same: they only warm L2) -- sample the same tokens."""
    g = gpt
    base = {k: g.get_option(k) for k in opts}
    ref_ids, ref_h = _gen(g, 24, 40, 16, [(7 * i) % 36 for i in range(24)])
    try:
        for k, v in opts.items():
            g.set_option(k, v)
        ids, hid = _gen(g, 24, 40, 16, [(7 * i) % 36 for i in range(24)])
    finally:
        for k, v in base.items():
            g.set_option(k, v)
    for b in range(24):
        assert torch.equal(ids[b], ref_ids[b]), (opts, b)
        # (2e-5 = the oracle tolerance used above; measured 0.8e-5 with 4-wave attention blocks at 24 rows, 1.01e-5 with the 8-wave blocks of round 6)
        assert float((hid[b] - ref_h[b]).abs().max()) <= (2e-5 if "split_nbg2_rows" in opts else 0.0), (opts, b)


def test_split_decode_graph_equals_eager_and_replays_bitwise(gpt):
    g = gpt
    a_ids, a_h = _gen(g, 32, 48, 24)
    b_ids, b_h = _gen(g, 32, 48, 24)
    c_ids, c_h = _gen(g, 32, 48, 24, graph=False)
    for b in range(32):
        assert torch.equal(a_ids[b], b_ids[b]) and torch.equal(a_h[b], b_h[b]), "two replays differ (fixed reduction orders, no atomics on the data path)"
        assert torch.equal(a_ids[b], c_ids[b]) and torch.equal(a_h[b], c_h[b]), "hipGraph replay != eager launches"


def test_split_decode_rows_are_independent_of_the_batch_they_ride_in(gpt):
    """An utterance's tokens must not depend on its neighbours (the reference decodes rows independently, llama.py:719-749): rows 0..8 of a 32-row batch == the same
    rows decoded as a 9-row batch (both on the split kernels, different 16-row chunk populations)."""
    g = gpt
    ids32, _ = _gen(g, 32, 48, 24)
    ids9, _ = _gen(g, 9, 48, 24)
    for b in range(9):
        assert torch.equal(ids32[b], ids9[b]), b


@pytest.mark.parametrize("B,P,pad", [(24, 40, [(7 * i) % 36 for i in range(24)]), (7, 300, [(41 * i) % 200 for i in range(7)]), (40, 390, None)])
def test_prompt_pass_block_shapes_agree_bitwise(gpt, B, P, pad):
    """The prompt pass's split GEMMs (llama.py:619-621,666,737-739 over all prompt rows) have four block shapes: 64 x 64 (the shortest passes), 128 x 128 (two blocks per CU
    on a 2-stage LDS ring, or one per CU on a 4-stage ring when the grid does not fill the chip) and, for long passes,
    256 rows x 256 or 192 features with two counter-phased wave groups (prefill_split_gemm_pp_kernel<EPI, 4 | 3>; option `prefill_pp_blocks`: 0 = never, -4 / -3 =
    always that shape, > 0 = by round count).  Every output element accumulates its k-tiles in the same order in all of them, so tokens AND hidden states must agree
    bit for bit -- at row counts that are no multiple of 256 (960, 2100: partial last blocks) and at the largest pass this engine holds (15600 rows)."""
    g = gpt
    base, sk, r4 = g.get_option("prefill_pp_blocks"), g.get_option("prefill_splitk_rows"), g.get_option("prefill_ring4_blocks")
    assert base > 0, "long prompt passes may take the counter-phased kernel by default"
    assert r4 == 256, "launches of at most one 128 x 128 block per CU keep three k-tile stages in flight (4-stage LDS ring) by default"
    try:
        g.set_option("prefill_splitk_rows", 0)                  # (short passes slice the down projection's K: another summation order, its own test below)
        g.set_option("prefill_pp_blocks", 0)
        g.set_option("prefill_ring4_blocks", 0)
        sm = g.get_option("prefill_small_blocks")
        assert sm == 192, "the shortest passes run on 64 x 64 blocks by default"
        g.set_option("prefill_small_blocks", 0)
        ref_ids, ref_h = _gen(g, B, P, 3, pad)
        outs = {}
        g.set_option("prefill_small_blocks", 1 << 20)           # every 128 x 128 launch as four times as many 64 x 64 blocks
        outs["64x64"] = _gen(g, B, P, 3, pad)
        g.set_option("prefill_small_blocks", sm)
        g.set_option("prefill_ring4_blocks", 1 << 20)           # every 128 x 128 launch on the 4-stage ring
        outs["ring4"] = _gen(g, B, P, 3, pad)
        g.set_option("prefill_ring4_blocks", r4)
        for shape in (-4, -3, base):
            g.set_option("prefill_pp_blocks", shape)
            outs[shape] = _gen(g, B, P, 3, pad)
    finally:
        g.set_option("prefill_pp_blocks", base)
        g.set_option("prefill_splitk_rows", sk)
        g.set_option("prefill_ring4_blocks", r4)
    for shape, (ids, hid) in outs.items():
        for b in range(B):
            assert torch.equal(ids[b], ref_ids[b]), (shape, B, P, b)
            assert torch.equal(hid[b], ref_h[b]), (shape, B, P, b, float((hid[b] - ref_h[b]).abs().max()))


@pytest.mark.parametrize("B,P,pad", [(10, 48, [(5 * b) % 11 for b in range(10)]), (24, 40, [(7 * i) % 36 for i in range(24)]), (32, 64, None)])
def test_short_prompt_passes_slice_the_down_projection(gpt, B, P, pad):
    """Prompt passes of <= `prefill_splitk_rows` rows (default 2048) slice the down projection's K = 3072 four ways over grid.z and add the slices in order
    (prefill_split.hip sp_launch, resid_combine_kernel; llama.py:739): 6 blocks per 128 rows otherwise walk 96 k-tiles each on a chip of 256 CUs.  Another summation
    order than the unsliced kernel's: the same tokens, hidden rows within the oracle tolerance of tests/test_gpu_gpt.py; twice the same bits (no atomics)."""
    g = gpt
    sk = g.get_option("prefill_splitk_rows")
    assert sk == 2048
    try:
        g.set_option("prefill_splitk_rows", 0)
        ref_ids, ref_h = _gen(g, B, P, 4, pad)
        g.set_option("prefill_splitk_rows", sk)
        ids, hid = _gen(g, B, P, 4, pad)
        ids2, hid2 = _gen(g, B, P, 4, pad)
    finally:
        g.set_option("prefill_splitk_rows", sk)
    worst = 0.0
    for b in range(B):
        assert torch.equal(ids[b], ref_ids[b]), (B, P, b)
        assert torch.equal(ids[b], ids2[b]) and torch.equal(hid[b], hid2[b]), (B, P, b)
        worst = max(worst, float((hid[b] - ref_h[b]).abs().max()))
    assert 0.0 < worst <= 2e-5, (B, P, worst)                    # (> 0: the sliced path did run)


def test_an_engine_without_the_images_refuses_the_option():
    from chatttsplus_amd.hip_models import GPT
    g = GPT(LLAMA, max_batch=12, max_seq_len=30, weight_dtype="fp32", options={"split_decode_rows": 0, "prefill_split_rows": 0})
    g.load_state_dict(synth.gpt_state_dict(synth.GPT_REAL, 1234))
    try:
        assert g.get_option("split_decode_rows") == 0
        with pytest.raises(Exception, match="head / tail"):
            g.set_option("split_decode_rows", 9)
        ids, _ = _gen(g, 10, 12, 6)          # still decodes, on the exact-f32 kernels
        assert len(ids) == 10
    finally:
        g.close()


@pytest.mark.parametrize("B", [22, 32, 40])
def test_decode_attention_block_shapes_agree(gpt, B):
    """Round 6: the decode attention keeps 8-wave blocks up to 42 rows on fp32 engines ("attn_wide_blocks" = 512; until then 4-wave blocks from one block per CU on:
    round 5's 20 -> 22-row step).  Both shapes evaluate llama.py:653-661 with different splits of the key range over the waves: same tokens, hidden rows within the
    oracle tolerance."""
    g = gpt
    assert g.get_option("attn_wide_blocks") == 512
    pad = [(5 * i) % 28 for i in range(B)]
    ref_ids, ref_h = _gen(g, B, 40, 16, pad)
    try:
        g.set_option("attn_wide_blocks", 0)                 # 0 = 256: 4-wave blocks at these row counts
        ids, hid = _gen(g, B, 40, 16, pad)
    finally:
        g.set_option("attn_wide_blocks", 512)
    for b in range(B):
        assert torch.equal(ids[b], ref_ids[b]), (B, b)
        assert float((hid[b] - ref_h[b]).abs().max()) <= 2e-5, (B, b)
