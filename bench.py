"""bench.py -- decode tokens/s of the ChatTTS code-token decoder hot path on MI355X.

A "step" is one pass of the hot path over one batch: 20 decoder layers + 4 heads + sampler for B sequences,
i.e. B generated tokens (token = 4 code indices = 21.33 ms of audio).  Default workload =
BASELINE.json configs[1]: batch 1, 48-token synthetic prompt, top-p 0.7 / top-k 20 / T 0.3 / rep 1.05,
512 generated tokens (min_new = max_new forces exactly K steps), fp16 weights+KV with fp32 accumulate
(the reference's GPU dtype, pipeline:37-41).  Inputs are resident in HBM when the timed region starts.

    python bench.py --gpus 1 --steps 512 --warmup 16
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

N > 1: utterances are independent -> every rank decodes its own batch (weak scaling); the only
collective on the path is one RCCL broadcast of the speaker-embedding table before the timed region.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X spec (6290 GB/s measured float4 copy, MI355X_MICROARCH.md)


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
    dt = times[4 + sample_steps] - times[4]
    return dict(value=round(sample_steps / dt, 3), unit="tokens/s", cores=torch.get_num_threads(), kind="port",
                sample=f"oracle/ref_cpu.OracleGPT (torch fp32) batch 1, prompt {prompt_len}, {sample_steps} decode steps "
                       f"after a 4-step run is subtracted, same synthetic weights; host has {os.cpu_count()} logical CPUs")


def dry_run(args, world, rank):
    """Same collective sequence as the real run (speaker broadcast, barrier, timed region, barrier, MAX all-reduce, rank-0
    JSON) on the gloo backend with a sleep instead of the decode loop -- validates the N>1 control flow on CPU."""
    import torch.distributed as dist
    from chatttsplus_amd import synth
    from chatttsplus_amd.dist import broadcast_speakers
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    table = torch.from_numpy(synth.speaker_vector(1234))[None] if rank == 0 else None
    spk = broadcast_speakers(table, 1, 768, torch.device("cpu"))
    assert abs(float(spk.norm()) - float(torch.from_numpy(synth.speaker_vector(1234)).norm())) < 1e-4
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    time.sleep(0.05 * (1 + rank))            # ranks finish at different times: the MAX must win
    if world > 1:
        dist.barrier()
    dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    if rank == 0:
        print(json.dumps({"metric": "decode tokens/s", "value": round(args.batch * args.steps * world / float(dt), 2), "unit": "tokens/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(float(dt) / args.steps * 1e3, 5),
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
                          "config": {"workload": "DRY RUN (no GPU work)"}}))
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=512)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--prompt", type=int, default=48)
    ap.add_argument("--dtype", default="fp16", choices=["fp16", "fp32"])
    ap.add_argument("--cpu-steps", type=int, default=192, help="decode steps of the CPU baseline sample (0 = skip)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--dry-run", action="store_true", help="CPU/gloo rehearsal of the multi-rank control flow (no HIP work, fake timing); used by tests/test_bench_dryrun.py")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")
    if args.dry_run:
        return dry_run(args, world, rank)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)   # nccl == RCCL on ROCm

    import ctypes as C
    from chatttsplus_amd import _lib, synth
    from chatttsplus_amd.hip_models.gpt import GPT, sampler_cfg_from_objects

    cfg = synth.GPT_REAL
    B, P, K, W = args.batch, args.prompt, args.steps, args.warmup
    max_new = W + K
    sd = synth.gpt_state_dict(cfg, 1234)                    # every rank holds a full replica (0.45 GB fp16)
    g = GPT(dict(hidden_size=768, intermediate_size=3072, num_attention_heads=12, num_hidden_layers=20), max_batch=max(B, 1),
            max_seq_len=P + max_new + 8, weight_dtype=args.dtype, device=str(dev))
    g.load_state_dict(sd)
    ids, mask = synth.prompt_ids(B, P, cfg["num_text_tokens"], 1234 + rank)
    spk_id = 21143
    ids[:, 1, :] = spk_id                                   # "[Stts][spk_emb]..." layout (pipeline:187-194)
    # speaker table lives on rank 0 and is broadcast over xGMI (the path's only collective)
    spk = torch.from_numpy(synth.speaker_vector(1234)).to(dev) if rank == 0 else torch.zeros(768, device=dev)
    if world > 1:
        dist.broadcast(spk, src=0)
    ids_t = torch.from_numpy(ids).to(dev)
    emb = g(ids_t, torch.ones(B, P, dtype=torch.bool, device=dev), spk_emb=spk, spk_emb_ids=spk_id)   # get_emb + apply_spk_emb (one HIP launch)

    lw = [type("P", (), dict(top_p=0.7, min_tokens_to_keep=3))(), type("K", (), dict(top_k=20))()]
    lp = [type("R", (), dict(penalty=1.05, past_window=16, max_input_ids=625))()]
    sc = sampler_cfg_from_objects(torch.tensor([0.3] * 4), 625, max_new, max_new, lw, lp, 4)
    lib, h = g._lib, g._h
    out_ids = torch.zeros(B, max_new, 4, dtype=torch.int32, device=dev)
    hid = torch.zeros(B, max_new, 768, dtype=torch.float32, device=dev)
    fin = torch.zeros(B, dtype=torch.int32, device=dev)
    end = torch.zeros(B, dtype=torch.int32, device=dev)
    io = _lib.GenIO(ids=out_ids.data_ptr(), hiddens=hid.data_ptr(), finish=fin.data_ptr(), end_idx=end.data_ptr(), noise=None,
                    n_draws=0, seed=1234 + rank)             # on-device Philox noise
    msk = torch.from_numpy(mask).to(dev).to(torch.int32)
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    use_graph = 0 if args.no_graph else 1
    _lib.check(lib.ctts_gpt_begin(h, B, P, msk.data_ptr(), C.byref(sc), C.byref(io), st), "begin")
    t0 = time.perf_counter()
    _lib.check(lib.ctts_gpt_prefill(h, emb.data_ptr(), st), "prefill")
    _lib.check(lib.ctts_gpt_sample(h, st), "sample")
    torch.cuda.synchronize(dev)
    prefill_ms = (time.perf_counter() - t0) * 1e3
    # warmup: W-1 untimed decode steps (the first sample above is step 0), includes graph capture
    _lib.check(lib.ctts_gpt_decode(h, max(W - 1, 1), use_graph, st), "decode warmup")
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    # timed: exactly K steps; the same region is bracketed by HIP events on the launch stream
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(torch.cuda.current_stream(dev))
    _lib.check(lib.ctts_gpt_decode(h, K, use_graph, st), "decode")
    ev1.record(torch.cuda.current_stream(dev))
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    ev_ms = ev0.elapsed_time(ev1)
    steps_done, alld = C.c_int32(0), C.c_int32(0)
    _lib.check(lib.ctts_gpt_progress(h, C.byref(steps_done), C.byref(alld), st), "progress")
    expect = 1 + max(W - 1, 1) + K
    if steps_done.value != expect or int(end.min().item()) != expect:
        raise SystemExit(f"bench invalid: {steps_done.value} steps executed, expected {expect} (end_idx min {int(end.min().item())})")
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
    except Exception as e:          # the decode metric stays valid; report the failure instead of hiding it
        voc_ms = f"failed: {e}"
    tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    if rank == 0:
        traffic = None
        try:      # HBM bytes/step from the committed rocprofv3 PMC passes (profiles/), same kernels and batch; not live
            tr = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))
            traffic = tr.get(f"b{B}_{args.dtype}", {}).get("hbm_bytes_per_step")
        except Exception:
            pass
        mean_ctx = P + W + K / 2.0
        step_bytes = g.step_bytes(B, mean_ctx)
        step_ms = ev_ms / K
        achieved = step_bytes / (step_ms * 1e-3) / 1e9
        res = {
            "metric": "decode tokens/s", "value": round(B * K * world / dt, 2), "unit": "tokens/s", "n_gpus": world, "steps": K,
            "warmup": W, "ms_per_step": round(dt / K * 1e3, 5), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": round(B * K * world / dt / 110.0, 2),   # 110 token/s (TensorRT fp16, RTX 3060)
            "dtype": "f16" if args.dtype == "fp16" else "f32", "data": "synthetic",
            "config": {"workload": f"ChatTTS GPT decode, batch {B}/GPU, prompt {P}, {K} generated tokens, top-p 0.7 top-k 20 T 0.3 rep 1.05 "
                                   f"(BASELINE configs[{1 if B == 1 else 2}]), random-init weights of the real 20x768 architecture",
                       "batch_per_gpu": B, "prompt_len": P, "weights": args.dtype, "kv_cache": args.dtype, "accumulate": "f32",
                       "hipgraph": bool(use_graph), "parallelism": f"replicas x{world} (utterance sharding)"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "traffic_note": "PMC FETCH_SIZE(x2 gfx950 correction)+WRITE_SIZE bytes per step from profiles/r01_pmc_*.json (separate rocprofv3 --pmc passes at --steps 64)",
                         "per": "decode step (one hipGraph replay = 102 kernel launches)",
                         "algorithmic_bytes_per_step": int(step_bytes), "step_ms_hip_events": round(step_ms, 5)},
            "rtf_decode_only": round(B * K * world * (512 / 24000.0) / dt, 2),
            "rtf_end_to_end": (round(B * 256 * (2 * expect - 1) / 24000.0 / ((prefill_ms + (dt / K) * 1e3 * (expect - 1) + voc_ms) / 1e3), 2)
                               if isinstance(voc_ms, float) else None),
            "vocoder_ms_for_batch": voc_ms if not isinstance(voc_ms, float) else round(voc_ms, 3),
            "rtf_note": f"audio seconds of {B} utterances x {expect} tokens / (prompt pass + {expect - 1} decode steps at the measured rate + DVAE-decoder + Vocos), rank 0",
            "prefill_plus_first_sample_ms": round(prefill_ms, 3),
            "reference_published_tok_s": {"tensorrt_fp16_rtx3060": 110, "pytorch_fp16_rtx3060": 28},
        }
        if args.cpu_steps > 0 and world == 1:
            res["cpu_baseline"] = cpu_baseline(P, args.cpu_steps)
        elif world > 1:
            res["cpu_baseline"] = None
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
