"""Copies the summaries tools/final_run_r06.sh left under gpurun_out/fin_r06 into profiles/ under their round-5 names and builds
profiles/r06_pmc_traffic.json from the FETCH_SIZE / WRITE_SIZE passes (taken at the timed window's context).  python tools/collect_profiles_r06.py"""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F = os.path.join(ROOT, "gpurun_out", "fin_r06c" if "c" in sys.argv[1:] else "fin_r06b" if "b" in sys.argv[1:] else "fin_r06")      # `b` / `c`: the later collections (tools/final_run_r06b.sh, r06c.sh) over the first
P = os.path.join(ROOT, "profiles")
pairs = {"bench_b1_fp32.json": "r06_bench_b1_fp32.json", "bench_b1_fp32_steps20.json": "r06_bench_b1_fp32_steps20.json", "bench_b32_fp32.json": "r06_bench_b32_fp32.json",
         "b1_fp32_kernel_stats.csv": "r06_b1_fp32_kernel_stats.csv", "b32_fp32_kernel_stats.csv": "r06_b32_fp32_kernel_stats.csv", "b8_fp32_kernel_stats.csv": "r06_b8_fp32_kernel_stats.csv",
         "step_time_vs_batch_fp32.jsonl": "r06_step_time_vs_batch_fp32.jsonl", "step_time_vs_batch_fp16.jsonl": "r06_step_time_vs_batch_fp16.jsonl",
         "persist_probe.jsonl": "r06_persist_probe.jsonl", "trace_gaps_b1.json": "r06_trace_gaps_b1.json", "ab_persistent_lora.jsonl": "r06_ab_persistent_lora_final.jsonl",
         "request_probe.jsonl": "r06_request_probe_final.jsonl", "prefill_32x512_fp32.log": "r06_prefill_32x512_fp32_ms.log", "long_ctx_probe.jsonl": "r06_ab_persist_long_context.jsonl"}
for src, dst in pairs.items():
    sp = os.path.join(F, src)
    if os.path.exists(sp) and os.path.getsize(sp) > 0:
        shutil.copy(sp, os.path.join(P, dst))
    else:
        print("missing", src) if not ({"b", "c"} & set(sys.argv[1:])) else None
if "c" in sys.argv[1:]:          # third collection: the prompt pass per block shape of its split GEMMs
    for src, dst in (("stats_32x512_pp_0.csv", "r06_prefill_pp_kernel_stats_128x128.csv"), ("stats_32x512_pp_1.csv", "r06_prefill_pp_kernel_stats_policy.csv"),
                     ("prefill_shapes_e2e.log", "r06_prefill_shapes_e2e.log")):
        if os.path.exists(os.path.join(F, src)):
            shutil.copy(os.path.join(F, src), os.path.join(P, dst))
    try:
        mf = json.load(open(os.path.join(F, "pmc_mfma_split.json")))["kernels"]
        ld = json.load(open(os.path.join(F, "pmc_lds_split.json")))
        old = json.load(open(os.path.join(P, "r06_pmc_mfma_split.json")))
        old["prompt_pass_32x512_fp32_counter_phased_gemms"] = {k: {"calls": v["calls"], "avg_us": round(v["avg_us"], 2), "mfma_busy_share_of_simd_cycles_at_2p4GHz": round(v["mfma_busy_share_of_simd_cycles_at_2p4GHz"], 4)}
                                                                for k, v in mf.items() if "prefill" in k or "skinny" in k}
        old["prompt_pass_32x512_fp32_counter_phased_gemms"]["lds_pass"] = ld
        old["note_counter_phased"] = ("third collection (tools/final_run_r06c.sh): the same two counter passes with the split GEMMs on 256-row counter-phased blocks "
                                      "(prefill_split_gemm_pp_kernel<EPI, NT>; default policy)")
        json.dump(old, open(os.path.join(P, "r06_pmc_mfma_split.json"), "w"), indent=1)
    except Exception as e:
        print("no MFMA-busy pass:", e)
fp = os.path.join(F, "bench_b1_fp32_steps20_force_pg.json")
if os.path.exists(fp):          # (RCCL's banner lines, if any, are not part of the evidence)
    lines = [l for l in open(fp) if l.startswith("{")]
    if lines:
        open(os.path.join(P, "r06_bench_b1_fp32_steps20_force_pg.json"), "w").write(lines[-1])
with open(os.path.join(P, "r06_gpu_tests.log"), "w") as f:
    for n in ("pytest_gpu.log", "smoke.log"):
        if os.path.exists(os.path.join(F, n)):
            f.write(open(os.path.join(F, n)).read())
out = {}
CTX = 309                                            # prompt 293, steps 8..24
for t, key, B, cmd in (("b1", "b1_fp32", 1, "bench.py --batch 1 --prompt 293 --steps 16 --warmup 8 --gen-tokens 0"),
                       ("b32", "b32_fp32", 32, "bench.py --batch 32 --prompt 293 --steps 16 --warmup 8 --gen-tokens 0")):
    try:
        fe = json.load(open(os.path.join(F, f"pmc_{t}_FETCH_SIZE.json"))); wr = json.load(open(os.path.join(F, f"pmc_{t}_WRITE_SIZE.json")))
    except Exception as e:
        print("no PMC traffic for", t, e)
        continue
    fr, w = fe["total_per_step"] * 1024, wr["total_per_step"] * 1024
    alg = 4 * (190698240 + B * (CTX + 1) * 30720)
    out[key] = {"fetch_bytes_raw_per_step": fr, "write_bytes_raw_per_step": w, "fetch_bytes_corrected_per_step": 2 * fr, "hbm_bytes_per_step": 2 * fr + w,
                "mean_context": CTX, "algorithmic_bytes_per_step": alg, "traffic_over_algorithmic": round((2 * fr + w) / alg, 3), "command": cmd,
                "kernels_fetch_kb_per_call": {k: round(v["value_per_call"], 1) for k, v in fe["kernels"].items()},
                "note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (KB units x 1024); FETCH_SIZE doubled per MI355X_MICROARCH.md "
                        "(gfx950 counts the 128-B requests of a wide coalesced stream at 64 B); the last 14 decode steps of the run; parity mode (fp32 weights + KV).  "
                        "The persistent launches' polls are served by L2 and do not reach the memory-side counters"}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        shutil.copy(os.path.join(F, f"pmc_{t}_{c}.json"), os.path.join(P, f"r06_pmc_{key}_{c}.json"))
if out:
    try:                                             # (the second collection re-takes batch 32 only)
        old = json.load(open(os.path.join(P, "r06_pmc_traffic.json")))
        out = dict(old, **out)
    except Exception:
        pass
    json.dump(out, open(os.path.join(P, "r06_pmc_traffic.json"), "w"), indent=1)
    print({k: (v["hbm_bytes_per_step"], v["traffic_over_algorithmic"]) for k, v in out.items()})
