"""The N>1 path on CPU: world_size-2 gloo processes exercise partition + speaker broadcast + length gather
(chatttsplus_amd/dist.py) with a stub per-rank synthesis function."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from chatttsplus_amd import dist as cdist


def test_partition_balanced_and_complete():
    lengths = [5, 100, 7, 64, 64, 3, 90, 12, 33]
    for world in (1, 2, 4, 8):
        shards = cdist.partition(lengths, world)
        assert sorted(i for s in shards for i in s) == list(range(len(lengths)))
        loads = [sum(lengths[i] for i in s) for s in shards]
        assert max(len(s) for s in shards) - min(len(s) for s in shards) <= 1
        if world == 2:
            assert abs(loads[0] - loads[1]) <= max(lengths)
    assert cdist.partition([], 4) == [[], [], [], []]


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lengths = [10, 40, 20, 30, 25, 5, 60]
        spk_idx = [0, 1, 2, 0, 1, 2, 0]
        table = torch.arange(3 * 8, dtype=torch.float32).view(3, 8) if rank == 0 else None

        def run_local(indices, rows):
            # stub synthesis: generated length = prompt length + first speaker-row element (checks the broadcast)
            return [lengths[i] + int(rows[j, 0].item()) for j, i in enumerate(indices)]

        mine, full = cdist.sharded_generate(lengths, spk_idx, table, 3, 8, torch.device("cpu"), run_local)
        q.put((rank, mine, full))
    finally:
        dist.destroy_process_group()


def test_sharded_generate_world2_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    lengths = [10, 40, 20, 30, 25, 5, 60]
    spk_first = [0, 8, 16, 0, 8, 16, 0]
    expect = [l + s for l, s in zip(lengths, spk_first)]
    seen = []
    for rank, mine, full in res:
        assert full == expect
        seen += mine
    assert sorted(seen) == list(range(len(lengths)))


# ---- ChatTTSPlusPipeline.infer_sharded: the real _infer/_infer_code/_decode_to_wavs wiring with a fake engine -------------------
VOCAB = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]", "[Stts]", "[Ptts]", "[spk_emb]", "[empty_spk]", "[uv_break]", "[break_0]",
         "[Ebreak]", "[speed_5]", "a", "b", "c", "d"]


class _FakeGPT:
    """Stands in for hip_models.GPT on CPU: same call surface; the generated length of an utterance is a pure function of its
    prompt and of its OWN speaker row, so a wrong shard / wrong speaker row / wrong order shows up in the lengths."""
    num_vq, model_dim, max_batch = 4, 8, 3

    def __init__(self):
        self.emb_code = [type("E", (), dict(num_embeddings=626))() for _ in range(4)]
        self.calls = []

    def __call__(self, input_ids, text_mask, spk_emb=None, spk_emb_ids=None):
        B, T = input_ids.shape[:2]
        spk = torch.as_tensor(spk_emb, dtype=torch.float32).reshape(-1, self.model_dim).expand(B, -1)
        emb = torch.zeros(B, T, self.model_dim)
        emb[:, 0, 0] = text_mask.sum(1).float()           # valid prompt tokens
        emb[:, 0, 1] = spk[:, 0]                          # first element of this utterance's speaker row
        return emb

    def generate(self, emb, inputs_ids, temperature, eos_token, attention_mask=None, max_new_token=2048, return_hidden=False, **kw):
        B = emb.shape[0]
        self.calls.append(B)
        assert B <= self.max_batch
        # what keys each utterance's device noise stream: (noise mode, request seed, global utterance id) -- recorded per utterance
        assert kw.get("noise") == "device" and len(kw["utt_ids"]) == B
        self.noise_keys = getattr(self, "noise_keys", []) + [(int(kw["seed"]), int(u)) for u in kw["utt_ids"]]
        n = [int(emb[b, 0, 0].item()) + int(emb[b, 0, 1].item()) for b in range(B)]
        yield type("O", (), dict(ids=[torch.zeros(k, 4, dtype=torch.long) for k in n], attentions=[],
                                 hiddens=[torch.full((k, 768), float(k)) for k in n]))


    def generate_many_iter(self, emb, inputs_ids, temperature, eos_token, attention_mask=None, max_new_token=2048, return_hidden=False, **kw):
        """continuous batching stand-in: utterances complete shortest first, two per event (so never in input order)"""
        N = emb.shape[0]
        assert len(kw["utt_ids"]) == N and kw.get("rows", 0) <= self.max_batch and "noise" not in kw
        self.many_calls = getattr(self, "many_calls", []) + [(N, int(kw["rows"]))]
        self.noise_keys = getattr(self, "noise_keys", []) + [(int(kw["seed"]), int(u)) for u in kw["utt_ids"]]
        n = [int(emb[b, 0, 0].item()) + int(emb[b, 0, 1].item()) for b in range(N)]
        lim = kw.get("max_new_tokens_per_row")
        if lim is not None:
            n = [min(a, int(b)) for a, b in zip(n, lim)]
        ids = [torch.zeros(k, 4, dtype=torch.long) for k in n]
        hid = [torch.full((k, 768), float(k)) for k in n]
        order = sorted(range(N), key=lambda b: (n[b], b))
        if kw.get("progress"):                       # streaming: half-way reports of every utterance that is long enough, like the engine's chunk reports
            half = [(b, n[b] // 2, ids[b][:n[b] // 2], hid[b][:n[b] // 2] if return_hidden else None) for b in range(N) if n[b] // 2 >= 2]
            if half:
                yield ("progress", half)
        for i in range(0, N, 2):
            yield [(b, ids[b], hid[b] if return_hidden else None) for b in order[i:i + 2]]
        return type("O", (), dict(ids=ids, attentions=[], hiddens=hid if return_hidden else []))


class _FakeSynth:
    def decode_batch(self, hiddens):
        return [torch.full((256 * (2 * h.shape[0] - 1),), float(h.shape[0])) if h.shape[0] else torch.zeros(0) for h in hiddens]


def _fake_pipeline(tmpdir):
    from transformers import BertTokenizerFast
    from chatttsplus_amd.pipeline import ChatTTSPlusPipeline
    from chatttsplus_amd.tokenizer import Tokenizer
    vf = os.path.join(tmpdir, "vocab.txt")
    with open(vf, "w") as f:
        f.write("\n".join(VOCAB))
    bt = BertTokenizerFast(vocab_file=vf, do_lower_case=False)
    bt.add_special_tokens({"additional_special_tokens": [v for v in VOCAB if v.startswith("[") and v not in ("[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]")]})
    pipe = object.__new__(ChatTTSPlusPipeline)
    pipe.device = torch.device("cpu")
    pipe.normalizer = lambda t, *a, **k: t
    pipe.text_splitter = None
    pipe.models_dict = dict(gpt=_FakeGPT(), tokenizer=Tokenizer(tokenizer=bt))
    pipe.synth = _FakeSynth()
    pipe._lora_models, pipe._lora_cache = {}, 1
    pipe.std = pipe.mean = None
    return pipe


TEXTS = ["a b c d a b c", "a", "b c", "d d d d d d d d d", "a b", "c c c", "b", "a b c d", "d a", "c b a d c b a"]
SPK_IDX = [0, 1, 2, 1, 0, 2, 2, 1, 0, 1]


def _expected_lengths(pipe):
    tok = pipe.models_dict["tokenizer"]
    out = []
    for t, s in zip(TEXTS, SPK_IDX):
        ids, att, tm = tok.encode([f"[Stts][spk_emb][speed_5]{t} [uv_break][Ptts]"], 4)
        out.append(int(tm.sum()) + 3 * s + 1)             # speaker table row s = [3 s + 1, ...]
    return out


def _pipe_worker(rank, world, port, q, tmpdir, continuous=False, seed=4242):
    from chatttsplus_amd.pipeline import InferCodeParams
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pipe = _fake_pipeline(tmpdir)
        table = (torch.arange(3, dtype=torch.float32)[:, None] * 3 + 1).expand(3, 8).contiguous() if rank == 0 else None
        kw = {} if seed is None else dict(noise_seed=seed)
        torch.manual_seed(100 + rank)                     # (seed=None: the ranks' own generators disagree; rank 0's draw must win)
        mine, wavs, all_lens = pipe.infer_sharded(list(TEXTS), speaker_index=SPK_IDX, speaker_table=table,
                                                  params_infer_code=InferCodeParams(show_tqdm=False), continuous=continuous, **kw)
        gpt = pipe.models_dict["gpt"]
        q.put((rank, mine, [int(w.shape[0]) for w in wavs], all_lens, gpt.calls if not continuous else getattr(gpt, "many_calls", []), gpt.noise_keys))
    finally:
        dist.destroy_process_group()


def test_pipeline_infer_sharded_world2_gloo(tmp_path):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pipe_worker, args=(r, world, port, q, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    expect = _expected_lengths(_fake_pipeline(str(tmp_path)))
    seen = []
    for rank, mine, wav_samples, all_lens, calls, noise_keys in res:
        # partition invariance of the sampling noise: utterance i is keyed by (request seed, i) on whatever rank / slice serves it --
        # exactly the keys a single rank uses (test_pipeline_infer_sharded_single_process)
        assert noise_keys == [(4242, i) for i in mine], (rank, noise_keys, mine)
        assert all_lens == expect, (rank, all_lens, expect)                       # every rank sees every utterance's length
        assert wav_samples == [256 * (2 * expect[i] - 1) for i in mine]           # local waveforms, in the order of `mine`
        assert sum(calls) == len(mine) and max(calls) <= 3                        # sliced at max_batch
        seen += mine
    assert sorted(seen) == list(range(len(TEXTS)))


def test_pipeline_infer_sharded_seed_is_rank0s_draw_world2_gloo(tmp_path):
    """Without noise_seed the request's seed is drawn from torch's CPU generator on rank 0 and broadcast (a constant default made every
    sharded request sample the same noise and ignored torch.manual_seed): both ranks key their utterances with rank 0's draw."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pipe_worker, args=(r, world, port, q, str(tmp_path), False, None)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    torch.manual_seed(100)
    want = int(torch.randint(0, 2 ** 62, (1,)).item())
    for rank, mine, wav_samples, all_lens, calls, noise_keys in res:
        assert noise_keys == [(want, i) for i in mine], (rank, noise_keys[:2], want)


def test_pipeline_infer_sharded_continuous_world2_gloo(tmp_path):
    """infer_sharded(continuous=True): every rank keeps its decode rows busy over ALL its utterances (one generate_many call per rank when it holds
    more utterances than rows); same lengths, same per-utterance noise keys, waveforms in the order of the rank's utterances."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pipe_worker, args=(r, world, port, q, str(tmp_path), True)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    expect = _expected_lengths(_fake_pipeline(str(tmp_path)))
    seen = []
    for rank, mine, wav_samples, all_lens, many_calls, noise_keys in res:
        assert sorted(noise_keys) == [(4242, i) for i in sorted(mine)], (rank, noise_keys, mine)
        assert all_lens == expect
        assert wav_samples == [256 * (2 * expect[i] - 1) for i in mine]
        assert many_calls == [(len(mine), 3)]                                     # one continuous call over the rank's 5 utterances on 3 rows
        seen += mine
    assert sorted(seen) == list(range(len(TEXTS)))


def test_pipeline_infer_sharded_single_process(tmp_path):
    """No process group: world 1, same entry point."""
    from chatttsplus_amd.pipeline import InferCodeParams
    pipe = _fake_pipeline(str(tmp_path))
    table = (torch.arange(3, dtype=torch.float32)[:, None] * 3 + 1).expand(3, 8).contiguous()
    mine, wavs, all_lens = pipe.infer_sharded(list(TEXTS), speaker_index=SPK_IDX, speaker_table=table, params_infer_code=InferCodeParams(show_tqdm=False),
                                              noise_seed=4242)
    assert mine == list(range(len(TEXTS))) and all_lens == _expected_lengths(pipe)
    assert pipe.models_dict["gpt"].noise_keys == [(4242, i) for i in range(len(TEXTS))]          # same keys as the 2-rank run gives each utterance
    with pytest.raises(Exception, match="speaker_table"):                                          # index without a table: a clear error, not an AttributeError
        pipe.infer_sharded(list(TEXTS), speaker_index=SPK_IDX, speaker_table=None)
    assert [int(w.shape[0]) for w in wavs] == [256 * (2 * n - 1) for n in all_lens]
    # per-utterance adapters of a sharded request: `lora_paths` names one adapter (or None) per GLOBAL utterance, every slice gets its own entries
    import chatttsplus_amd.pipeline as pl
    gpt = pipe.models_dict["gpt"]
    loaded, row_tables = [], []
    gpt.load_adapter = lambda slot, ad: loaded.append((slot, ad))
    gpt.set_row_adapters = lambda slots: row_tables.append(None if slots is None else list(slots))
    orig = pl.load_lora_adapter
    pl.load_lora_adapter = lambda path: f"adapter:{path}"
    try:
        paths = [("A", None, "B")[i % 3] for i in range(len(TEXTS))]
        mine2, wavs2, lens2 = pipe.infer_sharded(list(TEXTS), speaker_index=SPK_IDX, speaker_table=table, params_infer_code=InferCodeParams(show_tqdm=False),
                                                 noise_seed=4242, lora_paths=paths)
        with pytest.raises(Exception, match="lora_paths"):
            pipe.infer_sharded(list(TEXTS), speaker_index=SPK_IDX, speaker_table=table, lora_paths=paths[:3])
    finally:
        pl.load_lora_adapter = orig
    assert lens2 == all_lens and sorted(ad for _, ad in loaded) == ["adapter:A", "adapter:B"]
    slot_of = {ad.split(":")[1]: s_ for s_, ad in loaded}
    given = [t for t in row_tables if t is not None]
    assert [x for t in given for x in t] == [(-1 if paths[i] is None else slot_of[paths[i]]) for i in mine2]       # slices of 3 rows, in this rank's order


def test_pipeline_continuous_yields_in_input_order(tmp_path):
    """infer(continuous=...) host logic with the fake engine: utterances complete shortest first, the waveform lists still come out in INPUT
    order (first list as soon as utterance 0 is there, later ones in runs of >= 8 or at the end), every utterance keeps its own speaker row,
    noise key and token limit; continuous="throughput" hands the engine the longest texts first and yields one list; requests that fit the
    decode rows take the ordinary sliced path."""
    from chatttsplus_amd.pipeline import InferCodeParams
    from chatttsplus_amd import _lib
    pipe = _fake_pipeline(str(tmp_path))
    gpt = pipe.models_dict["gpt"]
    table = (torch.arange(3, dtype=torch.float32)[:, None] * 3 + 1).expand(3, 8).contiguous()
    rows = table[torch.tensor(SPK_IDX)]                                   # one speaker row per utterance
    want = _expected_lengths(pipe)
    p = InferCodeParams(show_tqdm=False, spk_emb=rows, prompt="[speed_5]")
    for mode in (True, "throughput"):
        gpt.noise_keys, gpt.many_calls = [], []
        lists = list(pipe._infer(list(TEXTS), False, None, True, False, True, True, False, True, params_infer_code=p, slice_size=3,
                                 utt_ids=list(range(100, 110)), noise_seed=9, continuous=mode))
        flat = [w for l in lists for w in l]
        assert [(int(w.shape[0]) // 256 + 1) // 2 for w in flat] == want, mode                       # input order, own speaker row
        assert gpt.many_calls == [(10, 3)]
        assert sorted(gpt.noise_keys) == [(9, 100 + i) for i in range(10)]
        if mode is True:
            assert len(lists) >= 2 and len(lists[0]) >= 1 and all(len(l) >= 8 or l is lists[0] or l is lists[-1] for l in lists)
        else:
            assert len(lists) == 1
            assert [u for _, u in gpt.noise_keys] == [100 + i for i in sorted(range(10), key=lambda i: -len(TEXTS[i] + " [uv_break]"))]   # longest first
    # per-utterance limits travel with their utterance (both modes reorder nothing the caller sees)
    lim = [2, 50, 1, 50, 3, 50, 50, 4, 50, 5]
    lists = list(pipe._infer(list(TEXTS), False, None, True, False, True, True, False, True, params_infer_code=p, slice_size=3, noise_seed=9,
                             continuous="throughput", max_new_tokens_per_utterance=lim))
    assert [(int(w.shape[0]) // 256 + 1) // 2 for w in lists[0]] == [min(a, b) for a, b in zip(want, lim)]
    # round 6: where the caller gives token limits, throughput mode serves the LONGEST LIMIT first (the request ends with its last row: a long utterance admitted late decodes
    # alone), ties by text length; `throughput_order = "input"` keeps arrival order
    gpt.noise_keys = []
    list(pipe._infer(list(TEXTS), False, None, True, False, True, True, False, True, params_infer_code=p, slice_size=3, noise_seed=9, utt_ids=list(range(100, 110)),
                     continuous="throughput", max_new_tokens_per_utterance=lim))
    assert [u for _, u in gpt.noise_keys] == [100 + i for i in sorted(range(10), key=lambda i: (-lim[i], -len(TEXTS[i] + " [uv_break]"), i))]
    pipe.throughput_order = "input"
    gpt.noise_keys = []
    list(pipe._infer(list(TEXTS), False, None, True, False, True, True, False, True, params_infer_code=p, slice_size=3, noise_seed=9, utt_ids=list(range(100, 110)),
                     continuous="throughput", max_new_tokens_per_utterance=lim))
    assert [u for _, u in gpt.noise_keys] == [100 + i for i in range(10)]
    del pipe.throughput_order
    # fits the decode rows -> ordinary path (generate), and the combinations that cannot work are refused
    gpt.calls, gpt.many_calls = [], []
    list(pipe._infer(list(TEXTS[:3]), False, None, True, False, True, True, False, True, params_infer_code=InferCodeParams(show_tqdm=False, spk_emb=rows[:3]),
                     slice_size=3, noise="device", noise_seed=9, continuous=True))
    assert gpt.calls == [3] and gpt.many_calls == []
    # stream=True + continuous (round 4): (utterance index, sample window) pairs; the windows of an utterance are consecutive and add up to its waveform
    class _WinSynth(_FakeSynth):
        def decode_window(self, hiddens, s0, s1):
            assert len(hiddens) == 1 and 0 <= s0[0] < s1[0] <= 256 * (2 * hiddens[0].shape[0] - 1)
            return [torch.arange(s0[0], s1[0], dtype=torch.float32)]
    pipe.synth = _WinSynth()
    ps = InferCodeParams(show_tqdm=False, spk_emb=rows, prompt="[speed_5]", stream_batch=2, stream_speed=100000)
    got = list(pipe._infer(list(TEXTS), True, None, True, False, True, True, False, True, params_infer_code=ps, slice_size=3, noise_seed=9, continuous=True))
    per = {}
    for chunk in got:
        for u, w in chunk:
            per.setdefault(u, []).append(w)
    assert sorted(per) == list(range(10))
    for u, ws in per.items():
        full = torch.cat(ws)
        assert full.shape[0] == 256 * (2 * want[u] - 1) and torch.equal(full, torch.arange(full.shape[0], dtype=torch.float32)), u     # consecutive, gap-free
    assert any(len(ws) > 1 for ws in per.values()), "no utterance was streamed in more than one window"
    with pytest.raises(_lib.HipBackendError):
        list(pipe._infer(list(TEXTS), False, None, True, False, True, True, False, True, params_infer_code=p, slice_size=3, continuous=True, noise="torch"))


def test_pipeline_continuous_with_per_utterance_adapters_host_logic(tmp_path, monkeypatch):
    """infer(continuous=True, lora_paths=[...]) host logic with the fake engine: every adapter of the request is loaded into its own resident slot once, the
    engine gets one slot (or -1) per utterance IN THE ORDER it gets the utterances (continuous="throughput" reorders them), and a request with more
    distinct adapters than the engine holds is refused with a message that names the way out."""
    from chatttsplus_amd.pipeline import InferCodeParams
    import chatttsplus_amd.pipeline as pl
    from chatttsplus_amd import _lib
    pipe = _fake_pipeline(str(tmp_path))
    gpt = pipe.models_dict["gpt"]
    loaded = []
    gpt.load_adapter = lambda slot, ad: loaded.append((slot, ad))
    monkeypatch.setattr(pl, "load_lora_adapter", lambda path: f"adapter:{path}")
    seen = {}
    many0 = gpt.generate_many_iter

    def many(*a, **kw):
        seen["slots"], seen["utt_ids"] = list(kw.pop("adapter_slots")), list(kw["utt_ids"])
        return many0(*a, **kw)
    gpt.generate_many_iter = many
    table = (torch.arange(3, dtype=torch.float32)[:, None] * 3 + 1).expand(3, 8).contiguous()
    p = InferCodeParams(show_tqdm=False, spk_emb=table[torch.tensor(SPK_IDX)], prompt="[speed_5]")
    paths = [("A", None, "B", "A", None)[i % 5] for i in range(10)]
    for mode in (True, "throughput"):
        loaded.clear(); pipe.__dict__.pop("_slot_of_path", None)
        lists = list(pipe._infer(list(TEXTS), False, None, True, False, True, True, False, True, params_infer_code=p, slice_size=3, utt_ids=list(range(100, 110)),
                                 noise_seed=9, continuous=mode, lora_paths=paths))
        assert sum(len(l) for l in lists) == 10
        assert sorted(ad for _, ad in loaded) == ["adapter:A", "adapter:B"] and len({s for s, _ in loaded}) == 2       # each adapter once, own slot
        slot_of = {ad.split(":")[1]: s for s, ad in loaded}
        assert seen["slots"] == [(-1 if paths[u - 100] is None else slot_of[paths[u - 100]]) for u in seen["utt_ids"]], mode
    too_many = [f"P{i}" for i in range(_lib.MAX_ADAPTERS + 1)] + [None] * (10 - _lib.MAX_ADAPTERS - 1)
    with pytest.raises(_lib.HipBackendError, match="continuous=False"):
        list(pipe._infer(list(TEXTS), False, None, True, False, True, True, False, True, params_infer_code=p, slice_size=3, noise_seed=9, continuous=True,
                         lora_paths=too_many))
