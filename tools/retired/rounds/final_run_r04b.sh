# Round-4 evidence, second call (after the per-utterance LoRA fold): GPU suite + smoke on the final code, the driver's bench lines, batch 32, the step-time
# curve, the LoRA probe and batch-32 kernel stats.  (PMC traffic and the persistent launch's probes are those of tools/final_run_r04.sh: persist_layer.hip is unchanged.)
set -x
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/fin_r04b
mkdir -p $O
cd $R
timeout 1800 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -3 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 900 python bench.py > $O/bench_b1_fp32.json 2> $O/bench_b1_fp32.err; cut -c1-200 $O/bench_b1_fp32.json
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_b1_fp32_steps20.json 2>/dev/null; cut -c1-200 $O/bench_b1_fp32_steps20.json
timeout 300 python bench.py --batch 32 --steps 256 --cpu-steps 0 --no-extras > $O/bench_b32_fp32.json 2>/dev/null; cut -c1-160 $O/bench_b32_fp32.json
timeout 300 python tools/tb_curve.py fp32 1 2 3 4 5 6 8 10 12 14 16 17 18 20 22 24 26 28 30 32 > $O/step_time_vs_batch_fp32.jsonl 2>/dev/null
timeout 300 python tools/lora_probe.py --modes none,fold,launch > $O/lora_probe.jsonl 2>/dev/null; cat $O/lora_probe.jsonl
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b32 -- python $R/bench.py --batch 32 --steps 64 --warmup 16 --cpu-steps 0 --no-extras > /tmp/prof_b32.log 2>&1
f=$(find /tmp/prof_b32 -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/b32_fp32_kernel_stats.csv
cd $R; ls -la $O
