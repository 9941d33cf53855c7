"""bench.py -- decode tokens/s of the ChatTTS code-token decoder hot path on MI355X.

A "step" is one pass of the hot path over one batch: 20 decoder layers + 4 heads + sampler for B sequences,
i.e. B generated tokens (token = 4 code indices = 21.33 ms of audio).  Default workload =
BASELINE.json configs[1]: batch 1, 48-token synthetic prompt, top-p 0.7 / top-k 20 / T 0.3 / rep 1.05,
a 512-token generation (min_new = max_new: EOS masked).  Default mode = the PARITY mode (`--dtype fp32`: fp32 weights + KV,
exact-f32 MFMA): the mode in which token ids are bit-exact against the reference CPU path and mel / waveform stay within
north_star's 1e-3 -- the mode the YAML ships.  `--dtype fp16` = the fast mode (fp16 weights + KV, fp32 accumulate: the reference's own
GPU dtype, pipeline:37-41; mel / waveform 1.2-1.75e-3); its figures ride in `extra` of the default line.  Inputs are resident in HBM
when the timed region starts.

The timed window of K steps is placed in the MIDDLE of the 512-token generation whatever K is (the steps before it run
untimed), so the measured context length -- and with it the KV bytes per step -- is that of the whole generation
(mean context = prompt + 256 +- a few), not that of its first K steps.

    python bench.py --gpus 1 --steps 512 --warmup 16
    python bench.py --gpus 8 ...                 # no env needed: re-executes itself through torch.distributed.run
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

N > 1: utterances are independent -> every rank decodes its own batch (weak scaling); the only collective on the
path is one RCCL broadcast of the speaker-embedding table before the timed region.  The JSON line also carries an `extra`
block (same process, after the headline leg): batch 32 per GPU (configs[2]; aggregated over the ranks = configs[3] at N=8),
batch 32 with mixed-length left-padded prompts, a 512-token prompt (north_star's "synthetic 512-token prompts"), and batch 32
through a LoRA-merged engine (configs[4]).  `--no-extras` skips them.
"""
import argparse
import ctypes as C
import json
import os
import socket
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X spec (6290 GB/s measured float4 copy, MI355X_MICROARCH.md)
GEN_TOKENS = 512               # the generation the timed window is centred in (BASELINE configs[1])
LLAMA = dict(hidden_size=768, intermediate_size=3072, num_attention_heads=12, num_hidden_layers=20)


def cpu_baseline(prompt_len: int, sample_steps: int):
    """The oracle (CPU restatement of the reference, kind="port") timed on this host, rank 0 only."""
    from chatttsplus_amd import synth
    from oracle import ref_cpu
    # small-matrix GEMV workload: 16 threads is near the best the oracle gets on a many-core host (one thread per
    # logical CPU is ~10x slower); the thread count used is what `cores` reports
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    cfg = synth.GPT_REAL
    sd = synth.gpt_state_dict(cfg, 1234)
    ids, mask = synth.prompt_ids(1, prompt_len, cfg["num_text_tokens"], 1234)
    o = ref_cpu.OracleGPT(sd, cfg["num_attention_heads"])
    emb = o.embed(torch.from_numpy(ids), torch.ones(1, prompt_len, dtype=torch.bool))
    times = {}
    for n in (4, 4 + sample_steps):
        sp = ref_cpu.SamplerParams(min_new_token=n)
        t0 = time.perf_counter()
        o.generate(emb, torch.from_numpy(ids), sp, attention_mask=torch.from_numpy(mask), max_new_token=n,
                   noise=ref_cpu.SeededNoise(1234))
        times[n] = time.perf_counter() - t0
    dt = max(times[4 + sample_steps] - times[4], 1e-9)
    return dict(value=round(sample_steps / dt, 3), unit="tokens/s", cores=torch.get_num_threads(), kind="port",
                sample=f"oracle/ref_cpu.OracleGPT (torch fp32) batch 1, prompt {prompt_len}, {sample_steps} decode steps "
                       f"after a 4-step run is subtracted, same synthetic weights; host has {os.cpu_count()} logical CPUs")


class _stdout_to_stderr:
    """RCCL prints a version banner on STDOUT when it creates a communicator; the contract is ONE JSON line on stdout.  File-descriptor level
    redirection (the banner comes from C code) around the group's creation and its first collective."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        sys.stdout.flush()
        try:
            C.CDLL(None).fflush(None)            # the banner sits in the C library's stdio buffer (fully buffered when stdout is a pipe or a file): out with it, to stderr
        except Exception:
            pass
        os.dup2(self.saved, 1)
        os.close(self.saved)
        return False


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def self_spawn(args):
    """`python bench.py --gpus N` without a launcher: replace this process by torch.distributed.run with N ranks on this node."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "16")
    sys.stdout.flush()
    os.execvpe(cmd[0], cmd, env)



def _request_256(n_utt, rank_count_hint=None):
    """The sharded request of BASELINE configs[3]: n_utt utterances, prompt lengths P ~ U{16..96} tokens (toy words + the 5 tags the pipeline wraps a
    text in), target lengths U{128..512} enforced per utterance, speaker i % 4.  Identical on every rank (seeded)."""
    from chatttsplus_amd import synth
    texts = synth.toy_texts(n_utt, 11, 91, seed=256)
    rng = np.random.Generator(np.random.Philox(key=2560))
    limits = [int(x) for x in rng.integers(128, 513, size=n_utt)]
    return texts, limits, [i % 4 for i in range(n_utt)]


def _digest(values):
    import hashlib
    return hashlib.sha1(np.ascontiguousarray(np.asarray(values, dtype=np.int64)).tobytes()).hexdigest()[:16]


def sharded_request_leg(pipe, dev, rank, world, n_utt=256, rows=32, reps=2, max_new=512):
    """ONE batched-synthesis request over all ranks through ChatTTSPlusPipeline.infer_sharded(continuous=True) -- partition, speaker-table and seed
    broadcast from rank 0, per-rank continuous batching on `rows` decode rows, DVAE decoder + Vocos, length all-reduce: what `--gpus N` means for
    north_star's "batched synthesis" (the reference's counterpart is the sequential slice loop pipeline:391-397).  Host-inclusive wall clock,
    MAX over ranks; rep 0 captures the decode graphs, the best of the later reps is reported.  `ids_digest` hashes a checksum of every
    utterance's token ids (all-reduced over the ranks): identical for N = 1, 2, 4, 8 iff every utterance got the same tokens; `utterance_checksums_mod_65521`
    lets a reader count HOW MANY did (see the comment at the return statement: a different schedule can flip a tied draw in 1-2 of 256 utterances)."""
    import torch.distributed as dist
    from chatttsplus_amd import synth
    from chatttsplus_amd.pipeline import InferCodeParams
    texts, limits, spk_index = _request_256(n_utt)
    limits = [min(l, max_new) for l in limits]
    table = torch.from_numpy(np.stack([synth.speaker_vector(1234 + i) for i in range(4)])) if rank == 0 else None
    params = InferCodeParams(prompt="[speed_5]", max_new_token=max_new, min_new_token=max_new, show_tqdm=False)   # EOS masked: every row runs to its own limit
    cpu = dev.type == "cpu"
    grouped = dist.is_available() and dist.is_initialized()          # (--force-pg: a one-rank group runs every rendezvous too)

    def sync():
        if not cpu:
            torch.cuda.synchronize(dev)

    best = None
    walls = []
    for rep in range(reps + 1):
        ids = []
        sync()
        if grouped:
            dist.barrier()
        t0 = time.perf_counter()
        mine, wavs, lens = pipe.infer_sharded(list(texts), speaker_index=spk_index, speaker_table=table, params_infer_code=params, noise_seed=4242,
                                              slice_size=rows, continuous=True, max_new_tokens_per_utterance=limits, ids_out=ids)
        sync()
        t_mine = time.perf_counter() - t0
        if grouped:
            dist.barrier()
        dt = time.perf_counter() - t0
        if lens != limits:
            raise SystemExit(f"sharded request invalid: generated lengths {lens[:6]}.. != targets {limits[:6]}..")
        if [int(w.shape[0]) for w in wavs] != [256 * (2 * limits[i] - 1) for i in mine]:
            raise SystemExit("sharded request invalid: waveform lengths do not match the token counts")
        tdev = torch.device("cpu") if cpu else dev
        tmax = torch.tensor([dt], dtype=torch.float64, device=tdev)
        mine_tokens = float(sum(limits[i] for i in mine))
        per = torch.zeros(world, 2, dtype=torch.float64, device=tdev)
        per[rank, 0] = mine_tokens; per[rank, 1] = t_mine
        chk = torch.zeros(n_utt, dtype=torch.int64, device=tdev)
        for u, t in zip(mine, ids):
            tt = t.to(torch.int64).cpu()
            w = (torch.arange(tt.numel(), dtype=torch.int64) % 8191 + 1).view(tt.shape)
            chk[u] = int((tt * w).sum())
        if grouped:
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dist.all_reduce(per, op=dist.ReduceOp.SUM)
            dist.all_reduce(chk, op=dist.ReduceOp.SUM)
        if rep == 0:
            continue
        wall = float(tmax.item())
        walls.append(wall)
        if best is None or wall < best["wall_s"]:
            tok = per[:, 0].cpu().tolist()
            best = dict(first_audio_ms=getattr(pipe, "last_first_audio_ms", None), wall_s=wall, per_rank_useful_tokens=[int(x) for x in tok], per_rank_busy_s=[round(x, 4) for x in per[:, 1].cpu().tolist()],
                        ids_digest=_digest(chk.cpu().numpy()), utt_sums=[int(x) % 65521 for x in chk.cpu().tolist()])
    total = float(sum(limits))
    audio_s = sum(256 * (2 * n - 1) for n in limits) / 24000.0
    tok = best["per_rank_useful_tokens"]
    return {"utterances": n_utt, "decode_rows_per_gpu": rows, "prompt_lengths": "U{16..96} tokens", "target_lengths": "U{128..512} tokens", "speakers": 4,
            "path": "ChatTTSPlusPipeline.infer_sharded(continuous=True): partition + speaker/seed broadcast + GPT (continuous batching) + DVAE decoder + Vocos + length all-reduce",
            "first_vocoder_batch_done_ms_rank0": (round(best["first_audio_ms"], 1) if best.get("first_audio_ms") is not None else None),
            "wall_ms": round(best["wall_s"] * 1e3, 2), "wall_ms_reps": [round(w * 1e3, 2) for w in walls], "useful_tokens": int(total), "useful_tokens_per_s": round(total / best["wall_s"], 1),
            "audio_seconds": round(audio_s, 2), "rtf_audio_s_per_wall_s": round(audio_s / best["wall_s"], 2),
            "per_rank_useful_tokens": tok, "per_rank_busy_s": best["per_rank_busy_s"],
            "load_imbalance_max_over_mean": round(max(tok) / (sum(tok) / len(tok)), 4) if sum(tok) else None,
            "lengths_digest": _digest(limits), "ids_digest": best["ids_digest"],
            # per-utterance checksums: across world sizes (or row counts, or service orders) count the utterances that kept their tokens.  For a FIXED schedule a request
            # reproduces bit for bit; across schedules an utterance passes through different kernels (persistent launch / 16-row groups / 32-row blocks, prompt passes of
            # different heights) whose hidden rows agree to 2e-5, and a draw whose two best candidates tie within that flips: measured 1-2 of 256 utterances
            # (profiles/r06_order_digest_probe.jsonl), so `ids_digest` is an all-or-nothing check of the SAME schedule only
            "utterance_checksums_mod_65521": best["utt_sums"]}


class _DryGPT:
    """CPU stand-in for hip_models.GPT in `--dry-run`: same call surface as the pipeline uses; an utterance's ids are a pure function of its global
    id and length, so `ids_digest` must not depend on the world size."""
    num_vq, model_dim, max_batch = 4, 768, 32

    def __init__(self):
        self.emb_code = [type("E", (), dict(num_embeddings=626))() for _ in range(4)]

    def __call__(self, input_ids, text_mask, spk_emb=None, spk_emb_ids=None):
        return torch.zeros(input_ids.shape[0], input_ids.shape[1], 8)

    def _ids(self, u, n):
        return ((torch.arange(n * 4, dtype=torch.int64) * 31 + 7 * int(u)) % 626).view(n, 4).to(torch.int32)

    def generate(self, emb, inputs_ids, temperature, eos_token, max_new_token=2048, return_hidden=False, **kw):
        lim = kw.get("max_new_tokens_per_row") or [max_new_token] * emb.shape[0]
        ids = [self._ids(u, int(n)) for u, n in zip(kw["utt_ids"], lim)]
        yield type("O", (), dict(ids=ids, attentions=[], hiddens=[torch.zeros(i.shape[0], 768) for i in ids]))

    def generate_many_iter(self, emb, inputs_ids, temperature, eos_token, max_new_token=2048, return_hidden=False, **kw):
        lim = kw.get("max_new_tokens_per_row") or [max_new_token] * emb.shape[0]
        ids = [self._ids(u, int(n)) for u, n in zip(kw["utt_ids"], lim)]
        time.sleep(0.002 * len(ids))
        for b in range(len(ids)):
            yield [(b, ids[b], torch.zeros(ids[b].shape[0], 768) if return_hidden else None)]
        return type("O", (), dict(ids=ids, attentions=[], hiddens=[]))


class _DrySynth:
    def decode_batch(self, hiddens):
        return [torch.zeros(256 * (2 * h.shape[0] - 1)) if h.shape[0] else torch.zeros(0) for h in hiddens]

def dry_run(args, world, rank):
    """Same rendezvous sequence as the real run -- speaker broadcast; per leg: barrier, timed region, barrier, MAX all-reduce, per-rank gather; the legs
    in the real order (headline, batch 32, the sharded request, 512-token prompts); group destroyed; CPU baseline on rank 0; ONE JSON line from rank 0 --
    on the gloo backend with a sleep instead of the decode loop: validates the N > 1 control flow on CPU."""
    import torch.distributed as dist
    from chatttsplus_amd import synth
    from chatttsplus_amd.dist import broadcast_speakers
    grouped = world > 1 or args.force_pg
    if grouped:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            os.environ.setdefault("MASTER_PORT", str(_free_port()))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        assert dist.get_world_size() == args.gpus
    table = torch.from_numpy(synth.speaker_vector(1234))[None] if rank == 0 else None
    spk = broadcast_speakers(table, 1, 768, torch.device("cpu"))
    assert abs(float(spk.norm()) - float(torch.from_numpy(synth.speaker_vector(1234)).norm())) < 1e-4

    def leg(batch, steps, nap):
        if grouped:
            dist.barrier()
        t0 = time.perf_counter()
        time.sleep(nap * (1 + rank))             # ranks finish at different times: the MAX must win
        mine = time.perf_counter() - t0
        if grouped:
            dist.barrier()
        dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
        per = [torch.tensor([mine], dtype=torch.float64)]
        if grouped:
            dist.all_reduce(dt, op=dist.ReduceOp.MAX)
            per = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
            dist.all_gather(per, torch.tensor([mine], dtype=torch.float64))
        return float(dt), [float(p) for p in per]

    dt, per = leg(args.batch, args.steps, 0.05)
    extra = {}
    d32, p32 = leg(32, args.steps, 0.01)
    extra["batch32"] = {"tokens_per_s": round(32 * args.steps * world / d32, 1), "per_rank_tokens_per_s": [round(32 * args.steps / p, 1) for p in p32],
                        "rtf_end_to_end": round(32 * args.steps * world * (512 / 24000.0) / d32, 2)}
    # the sharded-request leg on the same control flow: real ChatTTSPlusPipeline.infer_sharded host code (partition, speaker + seed broadcast, per-rank
    # slices / continuous batching, length all-reduce) around CPU stand-ins for the engines
    import tempfile
    from chatttsplus_amd.pipeline import ChatTTSPlusPipeline
    with tempfile.TemporaryDirectory() as td:
        pipe = ChatTTSPlusPipeline.from_components(_DryGPT(), _DrySynth(), synth.toy_tokenizer(td), "cpu")
        extra["sharded_request"] = sharded_request_leg(pipe, torch.device("cpu"), rank, world, n_utt=48, rows=8, reps=1, max_new=160)
    d5, p5 = leg(1, args.steps, 0.01)
    extra["prompt512_batch1"] = {"tokens_per_s": round(args.steps * world / d5, 1), "per_rank_tokens_per_s": [round(args.steps / p, 1) for p in p5]}
    pg = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "forced_at_world_1": bool(args.force_pg and world == 1)} if grouped else None
    if grouped:
        dist.destroy_process_group()
    if rank == 0:
        # rank 0 alone, after the last rendezvous -- as in the real run (a tiny sample here: the dry run checks the control flow, not the number)
        cpu = cpu_baseline(8, 8) if args.cpu_steps > 0 else None
        print(json.dumps({"metric": "decode tokens/s", "value": round(args.batch * args.steps * world / dt, 2), "unit": "tokens/s",
                          "extra": extra, "cpu_baseline": cpu, "process_group": pg,
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 5),
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
                          "world_size": world, "per_rank_tokens_per_s": [round(args.batch * args.steps / p, 2) for p in per],
                          "config": {"workload": "DRY RUN (no GPU work)"}}))


class Leg:
    """One timed decode leg on an engine: begin -> prompt pass -> first sample -> untimed steps up to the window -> K timed steps."""

    def __init__(self, g, dev, rank, world, grouped=None):
        self.g, self.dev, self.rank, self.world = g, dev, rank, world
        self.grouped = (world > 1) if grouped is None else bool(grouped)      # a process group exists (--force-pg: also at world 1): every rendezvous of the N > 1 path runs

    def run(self, B, P, K, W, pad_left=None, spk=None, gen_tokens=GEN_TOKENS, use_graph=1, keep_hidden=False):
        import torch.distributed as dist
        from chatttsplus_amd import _lib, synth
        from chatttsplus_amd.hip_models.gpt import sampler_cfg_from_objects
        g, dev = self.g, self.dev
        lib, h = g._lib, g._h
        cfg = synth.GPT_REAL
        s0 = W + max(0, (gen_tokens - K) // 2)               # generated tokens before the timed window (step 0 = first sample)
        max_new = s0 + K
        ids, mask = synth.prompt_ids(B, P, cfg["num_text_tokens"], 1234 + self.rank, pad_left=pad_left)
        spk_id = 21143
        for b in range(B):                                            # "[Stts][spk_emb]..." (pipeline:187-194): slot 1 of the unpadded text
            ids[b, (pad_left[b] if pad_left is not None else 0) + 1, :] = spk_id
        ids_t = torch.from_numpy(ids).to(dev)
        rows = spk[torch.arange(B, device=spk.device) % spk.shape[0]] if (spk is not None and spk.dim() == 2) else spk
        emb = g(ids_t, torch.ones(B, P, dtype=torch.bool, device=dev), spk_emb=rows, spk_emb_ids=spk_id)   # get_emb + apply_spk_emb (one HIP launch)
        lw = [type("P", (), dict(top_p=0.7, min_tokens_to_keep=3))(), type("K", (), dict(top_k=20))()]
        lp = [type("R", (), dict(penalty=1.05, past_window=16, max_input_ids=625))()]
        sc = sampler_cfg_from_objects(torch.tensor([0.3] * 4), 625, max_new, max_new, lw, lp, 4)
        out_ids = torch.zeros(B, max_new, 4, dtype=torch.int32, device=dev)
        hid = torch.zeros(B, max_new, 768, dtype=torch.float32, device=dev) if keep_hidden else None
        fin = torch.zeros(B, dtype=torch.int32, device=dev)
        end = torch.zeros(B, dtype=torch.int32, device=dev)
        io = _lib.GenIO(ids=out_ids.data_ptr(), hiddens=hid.data_ptr() if hid is not None else None, finish=fin.data_ptr(), end_idx=end.data_ptr(),
                        noise=None, n_draws=0, seed=1234 + self.rank)             # on-device Philox noise
        msk = torch.from_numpy(mask).to(dev).to(torch.int32)
        st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(lib.ctts_gpt_begin(h, B, P, msk.data_ptr(), C.byref(sc), C.byref(io), st), "begin")
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        _lib.check(lib.ctts_gpt_prefill(h, emb.data_ptr(), st), "prefill")
        _lib.check(lib.ctts_gpt_sample(h, st), "sample")
        torch.cuda.synchronize(dev)
        prefill_ms = (time.perf_counter() - t0) * 1e3
        # untimed: warm-up (graph capture) + the steps that bring the context to the window's start
        left = s0 - 1
        while left > 0:
            n = min(left, 64)
            _lib.check(lib.ctts_gpt_decode(h, n, use_graph, st), "decode (untimed)")
            left -= n
        torch.cuda.synchronize(dev)
        if self.grouped:
            dist.barrier()
        torch.cuda.synchronize(dev)
        # timed: exactly K steps; the same region is bracketed by HIP events on the launch stream
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record(torch.cuda.current_stream(dev))
        _lib.check(lib.ctts_gpt_decode(h, K, use_graph, st), "decode")
        ev1.record(torch.cuda.current_stream(dev))
        torch.cuda.synchronize(dev)
        mine = time.perf_counter() - t0
        if self.grouped:
            dist.barrier()
        torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t0
        ev_ms = ev0.elapsed_time(ev1)
        steps_done, alld = C.c_int32(0), C.c_int32(0)
        _lib.check(lib.ctts_gpt_progress(h, C.byref(steps_done), C.byref(alld), st), "progress")
        expect = max(s0, 1) + K
        if steps_done.value != expect or int(end.min().item()) != expect:
            raise SystemExit(f"bench invalid: {steps_done.value} steps executed, expected {expect} (end_idx min {int(end.min().item())})")
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        per = [torch.tensor([mine], device=dev, dtype=torch.float64)]
        if self.grouped:
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            per = [torch.zeros(1, device=dev, dtype=torch.float64) for _ in range(self.world)]
            dist.all_gather(per, torch.tensor([mine], device=dev, dtype=torch.float64))
        valid = (mask.sum(1)).astype(np.float64)                                   # attended prompt tokens per sequence
        mean_ctx = float(valid.mean()) + s0 + K / 2.0
        return dict(B=B, P=P, K=K, s0=s0, dt=float(tmax.item()), ev_ms=ev_ms, prefill_ms=prefill_ms, hid=hid, expect=expect, mean_ctx=mean_ctx,
                    per_rank_s=[float(p.item()) for p in per], step_bytes=g.step_bytes(B, mean_ctx))


def run_reps(leg, reps, *a, **kw):
    """The same leg `reps` times (each a fresh begin -> prompt pass -> steps up to the window -> K timed steps); returns the MEDIAN rep (by HIP-event
    time) with the spread of all reps attached: one rep cannot tell a slow box or a slow allocation from a slow kernel (the driver's
    single-rep `batch32_lora_merged` line sat 19 % above `batch32` with nothing to tell why)."""
    rs = [leg.run(*a, **kw) for _ in range(max(1, reps))]
    order = sorted(range(len(rs)), key=lambda i: rs[i]["ev_ms"])
    r = rs[order[len(order) // 2]]
    ms = sorted(x["dt"] / x["K"] * 1e3 for x in rs)
    r["spread_ms_per_step"] = {"reps": len(rs), "min": round(ms[0], 5), "median": round(ms[len(ms) // 2], 5), "max": round(ms[-1], 5)}
    pf = sorted(x["prefill_ms"] for x in rs)                 # the prompt pass likewise: the median rep's (the first pass of a shape pays first-touch costs), spread attached
    r["prefill_ms"] = pf[len(pf) // 2]
    r["spread_prefill_ms"] = {"reps": len(rs), "min": round(pf[0], 3), "median": round(pf[len(pf) // 2], 3), "max": round(pf[-1], 3)}
    return r


def ragged_leg(g, dev, spk, rank, B=32, seed=2024):
    """SURVEY 8d C3 as written: batch 32, prompt lengths P_b ~ U{16..96} (left padded), TARGET lengths n_b ~ U{128..512} enforced per row
    (ctts_gen_io.row_limits: the row counts as finished after n_b tokens, like a sampled EOS).  Host-inclusive wall clock of GPT.generate
    (prompt pass + decode loop), device noise.  `useful` tokens/s = sum(n_b) / wall -- what a caller gets; `padded` = B * max(n_b) / wall --
    what the kernels compute when finished rows stay in the batch (the reference's semantics, gpt.py:527-546).  Measured with finished-row
    compaction off and on (rows dropped from the decode batch at 32-step chunk boundaries, ctts_gpt_compact)."""
    from chatttsplus_amd import synth
    cfg = synth.GPT_REAL
    rng = np.random.Generator(np.random.Philox(key=seed + rank))
    plen = rng.integers(16, 97, size=B); plen[0] = 96
    nb = rng.integers(128, 513, size=B); nb[1] = 512
    P, N = 96, 512
    pad = [int(P - p) for p in plen]
    ids, mask = synth.prompt_ids(B, P, cfg["num_text_tokens"], 1234 + rank, pad_left=pad)
    spk_id = 21143
    for b in range(B):
        ids[b, pad[b] + 1, :] = spk_id
    ids_t = torch.from_numpy(ids).to(dev)
    rows = spk[torch.arange(B, device=spk.device) % spk.shape[0]]
    emb = g(ids_t, torch.ones(B, P, dtype=torch.bool, device=dev), spk_emb=rows, spk_emb_ids=spk_id)
    lw = [type("P", (), dict(top_p=0.7, min_tokens_to_keep=3))(), type("K", (), dict(top_k=20))()]
    lp = [type("R", (), dict(penalty=1.05, past_window=16, max_input_ids=625))()]
    res = {}
    keep = g.compact
    try:
        for mode in ("off", "on"):
            g.compact = (mode == "on")
            dt = 1e9
            for rep in range(3):                                   # rep 0 captures the decode graphs of the (batch size, key split) pairs the run visits; where the host
                torch.cuda.synchronize(dev)                        # learns of a finished row depends on timing, so a later rep can still meet a new pair: best of two
                t0 = time.perf_counter()
                out = list(g.generate(emb, ids_t, torch.tensor([0.3] * 4), 625, attention_mask=torch.from_numpy(mask), max_new_token=N, min_new_token=N,
                                      logits_warpers=lw, logits_processors=lp, return_hidden=False, noise="device", seed=7,
                                      max_new_tokens_per_row=[int(x) for x in nb]))[-1]
                torch.cuda.synchronize(dev)
                if rep:
                    dt = min(dt, time.perf_counter() - t0)
            lens = [int(i.shape[0]) for i in out.ids]
            if lens != [int(x) for x in nb]:
                raise SystemExit(f"ragged leg invalid: generated lengths {lens[:6]}.. != targets {nb[:6].tolist()}..")
            res[mode] = dict(wall_ms=round(dt * 1e3, 2), useful_tokens_per_s=round(float(nb.sum()) / dt, 1),
                             padded_tokens_per_s=round(B * float(nb.max()) / dt, 1), batch_sizes=[c[1] for c in getattr(g, "compactions", [])])
    finally:
        g.compact = keep
    return {"batch": B, "prompt_lengths": "U{16..96} left-padded to 96", "target_lengths": "U{128..512}", "useful_tokens": int(nb.sum()),
            "padded_tokens": int(B * nb.max()), "no_compaction": res["off"], "compaction": res["on"],
            "useful_speedup": round(res["on"]["useful_tokens_per_s"] / res["off"]["useful_tokens_per_s"], 3)}


def queue_leg(g, dev, spk, rank, NU=128, rows=32, seed=4048, admit_min=None, modes=("slices", "slices_compaction", "continuous")):
    """A request of 128 utterances (a long text split into sentences) on 32 decode rows: prompt lengths U{16..96}, target lengths U{128..512}
    enforced per row.  Host-inclusive wall clock of the GPT part, device noise, three ways of serving it: the reference's way -- slices, each
    run to its slowest row (pipeline:391-397 with slice_size = 32 instead of 4); slices with finished-row compaction; continuous batching
    (GPT.generate_many / ctts_gpt_admit: queued utterances take over rows as they free up).  Same tokens for every utterance in all three
    (tests/test_gpu_properties.py); `useful` tokens/s = sum(n_b) / wall."""
    from chatttsplus_amd import synth
    cfg = synth.GPT_REAL
    rng = np.random.Generator(np.random.Philox(key=seed + rank))
    plen = rng.integers(16, 97, size=NU); plen[0] = 96
    nb = [int(x) for x in rng.integers(128, 513, size=NU)]
    P, N = 96, 512
    pad = [int(P - p) for p in plen]
    ids, mask = synth.prompt_ids(NU, P, cfg["num_text_tokens"], 99 + rank, pad_left=pad)
    spk_id = 21143
    for b in range(NU):
        ids[b, pad[b] + 1, :] = spk_id
    ids_t = torch.from_numpy(ids).to(dev)
    srows = spk[torch.arange(NU, device=spk.device) % spk.shape[0]]
    emb = g(ids_t, torch.ones(NU, P, dtype=torch.bool, device=dev), spk_emb=srows, spk_emb_ids=spk_id)
    mask_t = torch.from_numpy(mask)
    lw = [type("P", (), dict(top_p=0.7, min_tokens_to_keep=3))(), type("K", (), dict(top_k=20))()]
    lp = [type("R", (), dict(penalty=1.05, past_window=16, max_input_ids=625))()]
    kw = dict(max_new_token=N, min_new_token=N, logits_warpers=lw, logits_processors=lp, return_hidden=False, seed=7)
    uids = list(range(NU))

    def sliced():
        lens = []
        for i in range(0, NU, rows):
            sl = slice(i, i + rows)
            out = list(g.generate(emb[sl].contiguous(), ids_t[sl], torch.tensor([0.3] * 4), 625, attention_mask=mask_t[sl], noise="device", utt_ids=uids[sl],
                                  max_new_tokens_per_row=nb[sl], **kw))[-1]
            lens += [int(i.shape[0]) for i in out.ids]
        return lens

    def continuous():
        out = g.generate_many(emb, ids_t, torch.tensor([0.3] * 4), 625, attention_mask=mask_t, utt_ids=uids, max_new_tokens_per_row=nb, rows=rows, admit_min=admit_min, **kw)
        return [int(i.shape[0]) for i in out.ids]

    res = {}
    keep = g.compact
    try:
        for name, fn, comp in (("slices", sliced, False), ("slices_compaction", sliced, True), ("continuous", continuous, True)):
            if name not in modes:
                continue
            g.compact = comp
            dt = 1e9
            for rep in range(3):                                   # rep 0 captures the decode graphs the run visits; best of the two timed reps (see ragged_leg)
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                lens = fn()
                torch.cuda.synchronize(dev)
                if rep:
                    dt = min(dt, time.perf_counter() - t0)
            if lens != nb:
                raise SystemExit(f"queue leg ({name}) invalid: generated lengths {lens[:6]}.. != targets {nb[:6]}..")
            res[name] = dict(wall_ms=round(dt * 1e3, 2), useful_tokens_per_s=round(float(sum(nb)) / dt, 1))
        if "continuous" in res:
            res["continuous"]["admissions"] = len(getattr(g, "admissions", []))
    finally:
        g.compact = keep
    out = {"utterances": NU, "decode_rows": rows, "prompt_lengths": "U{16..96}", "target_lengths": "U{128..512}", "useful_tokens": int(sum(nb)), **res}
    if "slices" in res:
        base = res["slices"]["useful_tokens_per_s"]
        out["speedup_vs_slices"] = {k: round(res[k]["useful_tokens_per_s"] / base, 3) for k in ("slices_compaction", "continuous") if k in res}
    return out


def summarize(r, world):
    step_ms = r["ev_ms"] / r["K"]
    ach = r["step_bytes"] / (step_ms * 1e-3) / 1e9
    out = _summary(r, world, step_ms, ach)
    if "spread_ms_per_step" in r:
        out["spread_ms_per_step"] = r["spread_ms_per_step"]
    if "spread_prefill_ms" in r:
        out["spread_prefill_plus_first_sample_ms"] = r["spread_prefill_ms"]
    return out


def _summary(r, world, step_ms, ach):
    return {"tokens_per_s": round(r["B"] * r["K"] * world / r["dt"], 1), "ms_per_step": round(r["dt"] / r["K"] * 1e3, 5),
            "step_ms_hip_events": round(step_ms, 5), "batch_per_gpu": r["B"], "prompt_len": r["P"], "steps": r["K"],
            "mean_context": round(r["mean_ctx"], 1), "algorithmic_bytes_per_step": int(r["step_bytes"]),
            "achieved_GBps": round(ach, 1), "frac_of_8TBps": round(ach / HBM_PEAK_GBS, 4), "prefill_plus_first_sample_ms": round(r["prefill_ms"], 3)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=512)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--prompt", type=int, default=48)
    ap.add_argument("--dtype", default="fp32", choices=["fp16", "fp32"], help="fp32 = parity mode (default, what the YAML ships); fp16 = fast mode")
    ap.add_argument("--cpu-steps", type=int, default=192, help="decode steps of the CPU baseline sample (0 = skip)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--gen-tokens", type=int, default=GEN_TOKENS, help="length of the generation the timed window is centred in (0: the window starts right after the warm-up; used by the short profiler passes)")
    ap.add_argument("--no-extras", action="store_true", help="skip the extra legs (batch 32, mixed prompts, 512-token prompt, LoRA)")
    ap.add_argument("--extra-steps", type=int, default=128, help="timed steps of each extra leg")
    ap.add_argument("--option", action="append", default=[], metavar="NAME=VALUE", help="engine option (ctts_gpt_set_option) for A/B runs, e.g. --option persistent_rows=0")
    ap.add_argument("--request-utterances", type=int, default=256, help="utterances of the sharded-request leg (BASELINE configs[3]: 256)")
    ap.add_argument("--extra-reps", type=int, default=3, help="repetitions of every extra leg (median reported, min / max beside it)")
    ap.add_argument("--force-pg", action="store_true", help="create the RCCL process group also at --gpus 1 and take the N > 1 code path end to end "
                    "(barriers, MAX all-reduce, per-rank gather, speaker broadcast, the sharded request with a group): rehearses on a one-GPU box what the 8-GPU run does")
    ap.add_argument("--dry-run", action="store_true", help="CPU/gloo rehearsal of the multi-rank control flow (no HIP work, fake timing); used by tests/test_bench_dryrun.py")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world == 1 and "RANK" not in os.environ:
        if not args.dry_run and torch.cuda.device_count() < args.gpus:
            raise SystemExit(f"--gpus {args.gpus}: only {torch.cuda.device_count()} GPU(s) visible on this node")
        self_spawn(args)                       # does not return
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} rank(s)")
    if args.dry_run:
        return dry_run(args, world, rank)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    grouped = world > 1 or args.force_pg
    if grouped:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            os.environ.setdefault("MASTER_PORT", str(_free_port()))
        with _stdout_to_stderr():
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)   # nccl == RCCL on ROCm
            assert dist.get_world_size() == args.gpus
            dist.barrier()                                                                # the communicator exists (and has said so) before anything is printed

    from chatttsplus_amd import synth
    from chatttsplus_amd.hip_models.gpt import GPT

    B, P, K, W = args.batch, args.prompt, args.steps, args.warmup
    extras = not args.no_extras
    EB, EK = 32, args.extra_steps
    sd = synth.gpt_state_dict(synth.GPT_REAL, 1234)                    # every rank holds a full replica (0.45 GB fp16)
    need_seq = max(P, 512 if extras else 0) + W + max(GEN_TOKENS, args.gen_tokens, K) + 16     # (96: the sharded request's longest prompt)
    g = GPT(LLAMA, max_batch=max(B, EB if extras else 1), max_seq_len=need_seq, weight_dtype=args.dtype, device=str(dev))
    g.load_state_dict(sd)
    for kv in args.option:
        k, v = kv.split("=")
        g.set_option(k, int(v))
    # speaker table (4 distinct speakers) lives on rank 0 and is broadcast over xGMI -- the path's only collective;
    # sequence b of every batch speaks with speaker b % 4
    spk = (torch.from_numpy(np.stack([synth.speaker_vector(1234 + i) for i in range(4)])).to(dev) if rank == 0 else torch.zeros(4, 768, device=dev))
    if grouped:
        dist.broadcast(spk, src=0)
    use_graph = 0 if args.no_graph else 1
    persist_rows = g.get_option("persistent_rows")             # effective value: 1 on fp32 engines that hold the device's persistent-launch lock
    split_rows_dec = g.get_option("split_decode_rows") if args.dtype == "fp32" else 0      # fp32 engines: batches of >= this many rows multiply on the fp16 pipes (head / tail operands)
    leg = Leg(g, dev, rank, world, grouped)
    XR = max(1, args.extra_reps)
    r = leg.run(B, P, K, W, spk=spk, use_graph=use_graph, keep_hidden=True, gen_tokens=args.gen_tokens)
    dt, expect, hid = r["dt"], r["expect"], r["hid"]

    # RTF leg (outside the timed decode region): DVAE decoder + Vocos on the generated hiddens of every local sequence
    voc_ms = None
    try:
        from chatttsplus_amd.hip_models import Synth
        syn = Synth(dict(synth.DVAE_REAL), dict(synth.VOCOS_REAL), max_frames=2 * expect + 64, device=str(dev), max_batch=min(B, 64))   # decode_batch slices
        syn.load("dvae.", synth.dvae_state_dict(synth.DVAE_REAL, 1234))
        syn.load("vocos.", synth.vocos_state_dict(synth.VOCOS_REAL, 1234))
        batch = [hid[b, :expect] for b in range(B)]
        wav = syn.decode_batch(batch)[-1]                                  # warm (same shapes as the timed call)
        torch.cuda.synchronize(dev)
        tv = time.perf_counter()
        wav = syn.decode_batch(batch)[-1]
        torch.cuda.synchronize(dev)
        voc_ms = (time.perf_counter() - tv) * 1e3
        assert wav.shape[0] == 256 * (2 * expect - 1) and bool(torch.isfinite(wav).all())
        del syn
    except Exception as e:          # the decode metric stays valid; report the failure instead of hiding it
        voc_ms = f"failed: {e}"
    del hid
    r["hid"] = None

    extra = {}
    if extras:
        try:
            # configs[2] / configs[3]: batch 32 per GPU, uniform prompts; aggregated over the ranks (MAX time)
            import tempfile
            from chatttsplus_amd.hip_models import Synth
            from chatttsplus_amd.pipeline import ChatTTSPlusPipeline
            syn_r = Synth(dict(synth.DVAE_REAL), dict(synth.VOCOS_REAL), max_frames=2 * 512 + 64, device=str(dev), max_batch=32)
            syn_r.load("dvae.", synth.dvae_state_dict(synth.DVAE_REAL, 1234))
            syn_r.load("vocos.", synth.vocos_state_dict(synth.VOCOS_REAL, 1234))
            e = run_reps(leg, XR, EB, P, EK, W, spk=spk, use_graph=use_graph, keep_hidden=True)
            extra["batch32"] = summarize(e, world)
            extra["batch32"]["per_rank_tokens_per_s"] = [round(EB * EK / p, 1) for p in e["per_rank_s"]]
            # BASELINE's metric asks RTF at batch 32 too: audio seconds of the 32 utterances / (prompt pass + decode steps at the measured rate + one
            # batched DVAE-decoder + Vocos call on their hidden states), rank 0's clock
            try:
                ex32 = e["expect"]
                batch = [e["hid"][b, :ex32] for b in range(EB)]
                syn_r.decode_batch(batch)
                torch.cuda.synchronize(dev)
                tv = time.perf_counter()
                wv = syn_r.decode_batch(batch)
                torch.cuda.synchronize(dev)
                v32 = (time.perf_counter() - tv) * 1e3
                assert len(wv) == EB and all(int(w.shape[0]) == 256 * (2 * ex32 - 1) for w in wv)
                extra["batch32"]["vocoder_ms_for_batch"] = round(v32, 3)
                extra["batch32"]["rtf_end_to_end"] = round(EB * 256 * (2 * ex32 - 1) / 24000.0 / ((e["prefill_ms"] + (e["dt"] / EK) * 1e3 * (ex32 - 1) + v32) / 1e3), 2)
            except Exception as exv:
                extra["batch32"]["rtf_end_to_end"] = f"failed: {exv}"
            e["hid"] = None
            # configs[3]: ONE request of 256 ragged utterances sharded over the ranks through the pipeline's own entry point
            #             (at N = 1 the same request on one GPU: the number the N-GPU lines are compared with)
            with tempfile.TemporaryDirectory() as td:
                pipe = ChatTTSPlusPipeline.from_components(g, syn_r, synth.toy_tokenizer(td), dev)
                extra["sharded_request"] = sharded_request_leg(pipe, dev, rank, world, n_utt=args.request_utterances, rows=EB)
                if world == 1:
                    # the same request in the LATENCY order (`infer(continuous=True)`: arrival order, waveform lists yielded in input order as soon as a prefix of the request is
                    # complete): what a streaming client sees -- time to the first audio, against the throughput mode above whose one list comes at the end
                    try:
                        texts_l, limits_l, spk_l = _request_256(args.request_utterances)
                        table_l = torch.from_numpy(np.stack([synth.speaker_vector(1234 + i) for i in range(4)])).to(dev)
                        from chatttsplus_amd.pipeline import InferCodeParams
                        p_l = InferCodeParams(prompt="[speed_5]", max_new_token=512, min_new_token=512, show_tqdm=False, spk_emb=table_l[torch.tensor(spk_l)])
                        torch.cuda.synchronize(dev)
                        t_l, first_l, n_l = time.perf_counter(), None, 0
                        for lst in pipe.infer(list(texts_l), skip_refine_text=True, do_text_optimization=False, params_infer_code=p_l, slice_size=EB, noise="device", noise_seed=4242,
                                              continuous=True, max_new_tokens_per_utterance=limits_l):
                            if first_l is None:
                                torch.cuda.synchronize(dev)
                                first_l = (time.perf_counter() - t_l) * 1e3
                            n_l += len(lst)
                        torch.cuda.synchronize(dev)
                        w_l = time.perf_counter() - t_l
                        extra["sharded_request"]["latency_order"] = {"first_audio_ms": round(first_l, 1), "wall_ms": round(w_l * 1e3, 2), "utterances": n_l,
                                                                     "useful_tokens_per_s": round(sum(limits_l) / w_l, 1)}
                    except Exception as exl:
                        extra["sharded_request"]["latency_order"] = f"failed: {type(exl).__name__}: {exl}"
                del pipe, syn_r
            # north_star: "decode tokens/s on synthetic 512-token prompts ... at 1/2/4/8 GPUs", batch 1 per GPU
            e = run_reps(leg, XR, 1, 512, EK, W, spk=spk, use_graph=use_graph)
            extra["prompt512_batch1"] = summarize(e, world)
            extra["prompt512_batch1"]["per_rank_tokens_per_s"] = [round(EK / p, 1) for p in e["per_rank_s"]]
            if world > 1:
                # multi-rank runs stop here: the remaining legs characterise one GPU (measured at N = 1) and every leg is a rendezvous --
                # a rank failing inside one of them would leave the others waiting at its barrier
                raise StopIteration
            # BASELINE configs[1] measured WHOLE: one 512-token generation at batch 1, steps 4..512 timed in one piece (the first sample belongs to the
            # prompt pass, three more steps capture the graph) -- the headline above is a K-step window around step 256 of the same generation
            e = run_reps(leg, XR, 1, P, 508, 4, spk=spk, use_graph=use_graph, gen_tokens=0)
            extra["batch1_full512"] = summarize(e, world)
            extra["batch1_full512"]["timed_steps"] = "4..512 of a 512-token generation (mean context %.0f)" % e["mean_ctx"]
            # configs[2] as written: mixed-length utterances, left-padded to the longest prompt (P_b ~ U{16..96}), padded KV
            rng = np.random.Generator(np.random.Philox(key=77 + rank))
            plen = rng.integers(16, 97, size=EB)
            plen[0] = 96
            e = run_reps(leg, XR, EB, 96, EK, W, pad_left=[int(96 - p) for p in plen], spk=spk, use_graph=use_graph)
            extra["batch32_mixed_prompts"] = summarize(e, world)
            extra["batch32_mixed_prompts"]["prompt_lengths"] = "U{16..96} left-padded to 96"
            # SURVEY 8d C3 with ragged TARGET lengths: useful vs padded tokens/s, finished-row compaction off / on
            extra["batch32_ragged_targets"] = ragged_leg(g, dev, spk, rank)
            # the same lengths as a QUEUE: 128 utterances on 32 decode rows -- slices vs slices + compaction vs continuous batching
            extra["queue128_on_32_rows"] = queue_leg(g, dev, spk, rank)
            if persist_rows >= 1:
                # the headline workload on the launch chain (102 dependent launches per step) -- what the persistent launch replaces
                g.set_option("persistent_rows", 0)
                e = run_reps(leg, XR, 1, P, min(K, 256), W, spk=spk, use_graph=use_graph, gen_tokens=args.gen_tokens)
                g.set_option("persistent_rows", persist_rows)
                extra["batch1_launch_chain"] = summarize(e, world)
            e = run_reps(leg, XR, EB, 512, EK, W, spk=spk, use_graph=use_graph, gen_tokens=2 * EK)
            extra["prompt512_batch32"] = summarize(e, world)
            # configs[4]: LoRA (r=8, alpha=16 on q/k/v/o of all layers, train_voice_clone_lora.yaml:72-80) merged into the packed weights
            rl = np.random.Generator(np.random.Philox(key=31))
            adapters = [(l, t, (rl.standard_normal((8, 768)) * 0.02).astype(np.float32), (rl.standard_normal((768, 8)) * 0.02).astype(np.float32), 2.0)
                        for l in range(LLAMA["num_hidden_layers"]) for t in ("q_proj", "k_proj", "v_proj", "o_proj")]
            gl = g.with_lora(adapters)
            e = run_reps(Leg(gl, dev, rank, world, grouped), XR, EB, P, EK, W, spk=spk, use_graph=use_graph)
            extra["batch32_lora_merged"] = summarize(e, world)
            gl.close()
            # the base engine again, right after: tells a drifting box from something the merged sibling does (its kernels and bytes are the base engine's)
            e = run_reps(leg, XR, EB, P, EK, W, spk=spk, use_graph=use_graph)
            extra["batch32_again_after_lora_merged"] = summarize(e, world)
            extra["batch32_lora_merged"]["vs_batch32_same_minute"] = round(extra["batch32_lora_merged"]["ms_per_step"] / extra["batch32_again_after_lora_merged"]["ms_per_step"], 4)
            # per-utterance adapters: 4 different adapters + "none" spread over the 32 sequences of ONE batch, evaluated as
            # W x + scale * B (A x) per row (worker workgroups inside the QKV / o_proj launches, lora_worker.h); the reference can only merge one adapter per call
            for slot in range(4):
                g.load_adapter(slot, [(l, t, (rl.standard_normal((8, 768)) * 0.02).astype(np.float32), (rl.standard_normal((768, 8)) * 0.02).astype(np.float32), 2.0)
                                      for l in range(LLAMA["num_hidden_layers"]) for t in ("q_proj", "k_proj", "v_proj", "o_proj")])
            g.set_row_adapters([(b % 5) - 1 for b in range(EB)])
            e = run_reps(leg, XR, EB, P, EK, W, spk=spk, use_graph=use_graph)
            g.set_row_adapters(None)
            extra["batch32_lora_per_utterance"] = summarize(e, world)
            if args.dtype == "fp16":
                # the parity-proven mode (fp32 weights / KV, exact-f32 MFMA: token ids bit-exact vs the reference CPU path, mel / waveform
                # <= 1e-3 -- DESIGN.md section 2) on the headline workload, same window
                g32 = GPT(LLAMA, max_batch=1, max_seq_len=P + W + max(GEN_TOKENS, K) + 16, weight_dtype="fp32", device=str(dev))
                g32.load_state_dict(sd)
                e = run_reps(Leg(g32, dev, rank, world, grouped), XR, 1, P, min(K, 256), W, spk=spk, use_graph=use_graph)
                extra["parity_mode_fp32_batch1"] = summarize(e, world)
                g32.close()
            else:
                # the fast mode (fp16 weights / KV, fp32 accumulate -- the reference's own GPU dtype, pipeline:37-41): mel / waveform
                # 1.2-1.75e-3 vs the fp32 CPU path (asserted <= 2e-3, tests/test_gpu_fp16_parity.py), token ids not guaranteed; same windows
                g16 = GPT(LLAMA, max_batch=EB, max_seq_len=P + W + max(GEN_TOKENS, K) + 16, weight_dtype="fp16", device=str(dev))
                g16.load_state_dict(sd)
                l16 = Leg(g16, dev, rank, world, grouped)
                extra["fast_mode_fp16_batch1"] = summarize(run_reps(l16, XR, 1, P, min(K, 256), W, spk=spk, use_graph=use_graph), world)
                extra["fast_mode_fp16_batch32"] = summarize(run_reps(l16, XR, EB, P, EK, W, spk=spk, use_graph=use_graph), world)
                extra["fast_mode_fp16_note"] = "mel / waveform rel-RMS 1.2-1.75e-3 vs the fp32 CPU path (north_star asks 1e-3): not the shipped default"
                g16.close()
        except (SystemExit, KeyboardInterrupt):
            raise
        except StopIteration:
            pass
        except Exception as ex:
            if grouped:
                raise                                  # fail loudly rather than desynchronise the ranks
            extra["error"] = f"{type(ex).__name__}: {ex}"

    if rank == 0:
        traffic, traffic_note = None, None
        try:      # HBM bytes/step from the committed rocprofv3 PMC passes (profiles/): same kernels, same batch, same context as the timed window; not live
            tp = [os.path.join(ROOT, "profiles", n) for n in ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json")]
            tp = [q for q in tp if os.path.exists(q)][0]                      # the newest committed PMC passes
            tj = json.load(open(tp))
            ent = tj.get(f"b{B}_{args.dtype}" + ("" if (persist_rows >= B or B > 8) else "_launch_chain"))
            if ent:
                traffic = ent.get("hbm_bytes_per_step")
                traffic_note = (f"PMC FETCH_SIZE (x2, the gfx950 correction of MI355X_MICROARCH.md) + WRITE_SIZE per decode step, separate rocprofv3 --pmc passes of "
                                f"`{ent.get('command')}` (profiles/{os.path.basename(tp)[:8]}*.json): mean context {ent.get('mean_context')} = the timed window's; "
                                f"{ent.get('traffic_over_algorithmic')} x the algorithmic bytes")
        except (OSError, ValueError, KeyError, IndexError):
            pass                                       # (no committed PMC passes for this configuration: traffic stays null)
        step_bytes = r["step_bytes"]
        step_ms = r["ev_ms"] / K
        achieved = step_bytes / (step_ms * 1e-3) / 1e9
        res = {
            "metric": "decode tokens/s", "value": round(B * K * world / dt, 2), "unit": "tokens/s", "n_gpus": world, "steps": K,
            "warmup": W, "ms_per_step": round(dt / K * 1e3, 5), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": round(B * K * world / dt / 110.0, 2),   # 110 token/s (TensorRT fp16, RTX 3060)
            "dtype": "f16" if args.dtype == "fp16" else "f32", "data": "synthetic",
            "world_size": world, "per_rank_tokens_per_s": [round(B * K / p, 2) for p in r["per_rank_s"]],
            "config": {"workload": f"ChatTTS GPT decode, batch {B}/GPU, prompt {P}, steps {r['s0']}..{r['s0'] + K} of a {max(args.gen_tokens, K)}-token generation "
                                   f"(mean context {r['mean_ctx']:.0f}), top-p 0.7 top-k 20 T 0.3 rep 1.05 (BASELINE configs[{1 if B == 1 else 2}]), "
                                   f"random-init weights of the real 20x768 architecture",
                       "batch_per_gpu": B, "prompt_len": P, "untimed_steps_before_window": r["s0"], "weights": args.dtype, "kv_cache": args.dtype,
                       "accumulate": "f32", "hipgraph": bool(use_graph), "parallelism": f"replicas x{world} (utterance sharding)",
                       "decode_path": ("persistent launch (20 layers = 1 launch, persist_layer.hip)" if persist_rows >= B else
                                       "launch chain (5 launches per layer" + (", projections as 3-term fp16 head / tail products" if 0 < split_rows_dec <= B else "") +
                                       (", weight prefetch workgroups" if g.get_option("weight_prefetch_kb") > 0 else "") + ")"),
                       "persistent_rows": persist_rows, "split_decode_rows": split_rows_dec, "weight_prefetch_kb": g.get_option("weight_prefetch_kb")},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "traffic_note": traffic_note,
                         "per": "decode step (one hipGraph replay = 4 steps on the launch chain, 16 on the persistent path, whose step is the layer-stack launch incl. the heads at <= 2 rows + the sampler)",
                         "algorithmic_bytes_per_step": int(step_bytes), "step_ms_hip_events": round(step_ms, 5)},
            "rtf_decode_only": round(B * K * world * (512 / 24000.0) / dt, 2),
            "rtf_end_to_end": (round(B * 256 * (2 * expect - 1) / 24000.0 / ((r["prefill_ms"] + (dt / K) * 1e3 * (expect - 1) + voc_ms) / 1e3), 2)
                               if isinstance(voc_ms, float) else None),
            "vocoder_ms_for_batch": voc_ms if not isinstance(voc_ms, float) else round(voc_ms, 3),
            "rtf_note": f"audio seconds of {B} utterances x {expect} tokens / (prompt pass + {expect - 1} decode steps at the measured rate + DVAE-decoder + Vocos), rank 0",
            "prefill_plus_first_sample_ms": round(r["prefill_ms"], 3),
            "reference_published_tok_s": {"tensorrt_fp16_rtx3060": 110, "pytorch_fp16_rtx3060": 28},
            "extra": extra if extras else None,
        }
        res["process_group"] = ({"backend": dist.get_backend(), "world_size": dist.get_world_size(), "forced_at_world_1": bool(args.force_pg and world == 1)}
                                if grouped else None)
    if grouped:
        dist.destroy_process_group()
    if rank == 0:
        # the CPU baseline runs on rank 0 only, AFTER the last rendezvous (the other ranks are done): an N > 1 line carries it too
        res["cpu_baseline"] = cpu_baseline(P, args.cpu_steps) if args.cpu_steps > 0 else None
        print(json.dumps(res))


if __name__ == "__main__":
    main()
