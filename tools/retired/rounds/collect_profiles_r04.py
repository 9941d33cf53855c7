"""Copies the summaries tools/final_run_r04.sh left under gpurun_out/fin_r04 into profiles/ under their round-4 names and builds
profiles/r04_pmc_traffic.json from the FETCH_SIZE / WRITE_SIZE passes (taken at the timed window's context).  python tools/collect_profiles_r04.py"""
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F = os.path.join(ROOT, "gpurun_out", "fin_r04")
P = os.path.join(ROOT, "profiles")
pairs = {"bench_b1_fp32.json": "r04_bench_b1_fp32.json", "bench_b1_fp32_steps20.json": "r04_bench_b1_fp32_steps20.json",
         "bench_b1_fp32_launch_chain.json": "r04_bench_b1_fp32_launch_chain.json",
         "b1_fp32_kernel_stats.csv": "r04_b1_fp32_kernel_stats.csv", "b1c_fp32_kernel_stats.csv": "r04_b1_fp32_launch_chain_kernel_stats.csv",
         "b32_fp32_kernel_stats.csv": "r04_b32_fp32_kernel_stats.csv", "step_time_vs_batch_fp32.jsonl": "r04_step_time_vs_batch_fp32.jsonl",
         "step_time_vs_batch_fp16.jsonl": "r04_step_time_vs_batch_fp16.jsonl", "persist_probe.jsonl": "r04_persist_probe_final.jsonl",
         "pmc_b1_sq.json": "r04_pmc_b1_fp32_sq_wait.json"}
for B in (2, 4, 32, 64, 128):
    pairs[f"bench_b{B}_fp32.json"] = f"r04_bench_b{B}_fp32.json"
for B in (1, 32, 64, 128):
    pairs[f"bench_b{B}_fp16.json"] = f"r04_bench_b{B}_fp16.json"
for src, dst in pairs.items():
    sp = os.path.join(F, src)
    if os.path.exists(sp) and os.path.getsize(sp) > 0:
        shutil.copy(sp, os.path.join(P, dst))
    else:
        print("missing", src)
with open(os.path.join(P, "r04_gpu_tests.log"), "w") as f:
    for n in ("pytest_gpu.log", "smoke.log"):
        if os.path.exists(os.path.join(F, n)):
            f.write(open(os.path.join(F, n)).read())
out = {}
CTX = 309                                            # prompt 293, steps 8..24
for t, key, B, cmd in (("b1", "b1_fp32", 1, "bench.py --batch 1 --prompt 293 --steps 16 --warmup 8 --gen-tokens 0"),
                       ("b1c", "b1_fp32_launch_chain", 1, "bench.py --batch 1 --prompt 293 --steps 16 --warmup 8 --gen-tokens 0 --option persistent_rows=0"),
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
                        "The persistent launch's polls are served by L2 and do not reach the memory-side counters"}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        shutil.copy(os.path.join(F, f"pmc_{t}_{c}.json"), os.path.join(P, f"r04_pmc_{key}_{c}.json"))
if out:
    json.dump(out, open(os.path.join(P, "r04_pmc_traffic.json"), "w"), indent=1)
    print({k: (v["hbm_bytes_per_step"], v["traffic_over_algorithmic"]) for k, v in out.items()})
