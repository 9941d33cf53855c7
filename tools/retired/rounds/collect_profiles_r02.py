"""Copies the summaries tools/final_run_r02.sh left under gpurun_out/fin2 into profiles/ under their round-2 names and builds
profiles/r02_pmc_traffic.json from the FETCH_SIZE / WRITE_SIZE passes.  python tools/collect_profiles_r02.py"""
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F = os.path.join(ROOT, "gpurun_out", "fin2")
P = os.path.join(ROOT, "profiles")
pairs = {"bench_b1_fp16.json": "r02_bench_b1_fp16.json", "bench_b1_fp16_driver_args.json": "r02_bench_b1_fp16_steps20.json",
         "bench_b32_fp16.json": "r02_bench_b32_fp16.json", "bench_b64_fp16.json": "r02_bench_b64_fp16.json",
         "bench_b128_fp16.json": "r02_bench_b128_fp16.json", "bench_b1_fp32.json": "r02_bench_b1_fp32.json",
         "bench_b32_fp32.json": "r02_bench_b32_fp32.json", "b1_kernel_stats.csv": "r02_b1_fp16_kernel_stats.csv",
         "b32_kernel_stats.csv": "r02_b32_fp16_kernel_stats.csv", "pre_kernel_stats.csv": "r02_prefill_32x512_kernel_stats.csv",
         "pmc_mfma_prefill.json": "r02_pmc_mfma_prefill.json", "pmc_mfma_vocoder.json": "r02_pmc_mfma_vocoder.json",
         "fp16_token_agreement.json": "r02_fp16_token_agreement.json", "prefill.log": "r02_prefill_ms.log"}
for src, dst in pairs.items():
    sp = os.path.join(F, src)
    if os.path.exists(sp) and os.path.getsize(sp) > 0:
        shutil.copy(sp, os.path.join(P, dst))
    else:
        print("missing", src)
with open(os.path.join(P, "r02_wall_clock.jsonl"), "w") as f:
    for n in ("gen_wall.log", "pipe_wall.log"):
        f.write(open(os.path.join(F, n)).read())
with open(os.path.join(P, "r02_gpu_tests.log"), "w") as f:
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
    out[f"{t}_fp16"] = {"fetch_bytes_raw_per_step": fr, "write_bytes_raw_per_step": w, "fetch_bytes_corrected_per_step": 2 * fr, "hbm_bytes_per_step": 2 * fr + w,
                        "note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (KB units x1024); FETCH_SIZE doubled per MI355X_MICROARCH.md "
                                "(gfx950 counts 128-B requests at 64 B for wide coalesced streams); bench.py --steps 64 --warmup 16 --gen-tokens 0 (decode steps 20..80 of the "
                                "generation: mean context ~98 at prompt 48), last 60 decode steps"}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        shutil.copy(os.path.join(F, f"pmc_{t}_{c}.json"), os.path.join(P, f"r02_pmc_{t}_{c}.json"))
if out:
    json.dump(out, open(os.path.join(P, "r02_pmc_traffic.json"), "w"), indent=1)
for f in ("bench_b1_fp16.json", "bench_b32_fp16.json", "bench_b64_fp16.json", "bench_b128_fp16.json", "bench_b1_fp32.json", "bench_b32_fp32.json"):
    try:
        d = json.load(open(os.path.join(F, f)))
        print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["algorithmic_bytes_per_step"], d.get("rtf_end_to_end"))
        for k, v in (d.get("extra") or {}).items():
            print("    ", k, v if isinstance(v, str) else (v["tokens_per_s"], v["ms_per_step"], v["frac_of_8TBps"], v["mean_context"], v["prefill_plus_first_sample_ms"]))
    except Exception as e:
        print(f, "unreadable", e)
