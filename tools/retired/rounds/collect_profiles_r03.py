"""Copies the summaries tools/final_run_r03.sh left under gpurun_out/fin_r03 into profiles/ under their round-3 names and builds
profiles/r03_pmc_traffic.json from the FETCH_SIZE / WRITE_SIZE passes.  python tools/collect_profiles_r03.py"""
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F = os.path.join(ROOT, "gpurun_out", "fin_r03")
P = os.path.join(ROOT, "profiles")
pairs = {"bench_b1_fp32.json": "r03_bench_b1_fp32.json", "bench_b1_fp32_driver_args.json": "r03_bench_b1_fp32_steps20.json",
         "b1_fp32_kernel_stats.csv": "r03_b1_fp32_kernel_stats.csv", "b32_fp32_kernel_stats.csv": "r03_b32_fp32_kernel_stats.csv",
         "step_time_vs_batch_fp32.jsonl": "r03_step_time_vs_batch_fp32.jsonl", "step_time_vs_batch_fp16.jsonl": "r03_step_time_vs_batch_fp16.jsonl",
         "prefill.log": "r03_prefill_ms.log", "vocoder.log": "r03_vocoder_32x272_ms.log",
         "vocoder_32x272_kernel_stats.csv": "r03_vocoder_32x272_kernel_stats.csv", "prefill_32x512_fp32_kernel_stats.csv": "r03_prefill_32x512_fp32_kernel_stats.csv"}
for B in (32, 64, 128):
    pairs[f"bench_b{B}_fp32.json"] = f"r03_bench_b{B}_fp32.json"
    pairs[f"bench_b{B}_fp16.json"] = f"r03_bench_b{B}_fp16.json"
pairs["bench_b1_fp16.json"] = "r03_bench_b1_fp16.json"
for src, dst in pairs.items():
    sp = os.path.join(F, src)
    if os.path.exists(sp) and os.path.getsize(sp) > 0:
        shutil.copy(sp, os.path.join(P, dst))
    else:
        print("missing", src)
# the queue file keeps the two larger configurations measured separately (512 on 64 rows, 384 on 128 rows): only its first two lines are refreshed
qsrc, qdst = os.path.join(F, "queue.jsonl"), os.path.join(P, "r03_queue128_on_32_rows.jsonl")
if os.path.exists(qsrc):
    new = open(qsrc).read().splitlines()
    old = open(qdst).read().splitlines() if os.path.exists(qdst) else []
    open(qdst, "w").write("\n".join(new + old[2:]) + "\n")
with open(os.path.join(P, "r03_wall_clock.jsonl"), "w") as f:
    for n in ("gen_wall.log", "pipe_wall.log"):
        f.write(open(os.path.join(F, n)).read())
with open(os.path.join(P, "r03_gpu_tests.log"), "w") as f:
    f.write(open(os.path.join(F, "pytest_gpu.log")).read())
    f.write(open(os.path.join(F, "smoke.log")).read())
out = {}
for t in ("b1", "b32"):
    try:
        fe = json.load(open(os.path.join(F, f"pmc_{t}_FETCH_SIZE.json"))); wr = json.load(open(os.path.join(F, f"pmc_{t}_WRITE_SIZE.json")))
    except Exception as e:
        print("no PMC traffic for", t, e)
        continue
    fr, w = fe["total_per_step"] * 1024, wr["total_per_step"] * 1024
    B = 1 if t == "b1" else 32
    alg = 4 * (190698240 + B * (98 + 1) * 30720)
    out[f"{t}_fp32"] = {"fetch_bytes_raw_per_step": fr, "write_bytes_raw_per_step": w, "fetch_bytes_corrected_per_step": 2 * fr, "hbm_bytes_per_step": 2 * fr + w,
                        "algorithmic_bytes_per_step_at_context_98": alg, "traffic_over_algorithmic": round((2 * fr + w) / alg, 3),
                        "note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (KB units x1024); FETCH_SIZE doubled per MI355X_MICROARCH.md "
                                "(gfx950 counts 128-B requests at 64 B for wide coalesced streams); bench.py --steps 64 --warmup 16 --gen-tokens 0 (decode steps 20..80 of the "
                                "generation: mean context ~98 at prompt 48), last 60 decode steps; parity mode (fp32 weights + KV)"}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        shutil.copy(os.path.join(F, f"pmc_{t}_{c}.json"), os.path.join(P, f"r03_pmc_{t}_fp32_{c}.json"))
# the fast mode's traffic was measured in round 2 on the same kernels
try:
    old = json.load(open(os.path.join(P, "r02_pmc_traffic.json")))
    out.update({k: dict(v, note=v["note"] + " (round-2 measurement)") for k, v in old.items()})
except Exception:
    pass
if out:
    json.dump(out, open(os.path.join(P, "r03_pmc_traffic.json"), "w"), indent=1)
    print({k: (v["hbm_bytes_per_step"], v.get("traffic_over_algorithmic")) for k, v in out.items()})
