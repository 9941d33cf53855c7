# Round-6 evidence collection on one MI355X box (one gpurun call): GPU suite + smoke, the bench lines (default, the driver's arguments, --force-pg = the N > 1 path on a
# one-rank RCCL group), step time vs batch (fp32 1..34 rows, fp16), the persistent launch's probe (ids vs the launch chain, step times, phase marks), adapters chain vs persistent,
# the 256-utterance request, the 32 x 512 prompt pass, rocprofv3 kernel stats of batch 1 / batch 32, FETCH / WRITE traffic at the timed window's context (prompt 293 -> mean context 309).
# Small summaries only -> gpurun_out/fin_r06 (tools/collect_profiles_r06.py copies them to profiles/).
set -x
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/fin_r06
mkdir -p $O
cd $R
timeout 1800 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -3 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 900 python bench.py > $O/bench_b1_fp32.json 2> $O/bench_b1_fp32.err; cut -c1-200 $O/bench_b1_fp32.json
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_b1_fp32_steps20.json 2>/dev/null; cut -c1-200 $O/bench_b1_fp32_steps20.json
timeout 600 python bench.py --steps 20 --warmup 5 --force-pg > $O/bench_b1_fp32_steps20_force_pg.json 2>/dev/null; cut -c1-200 $O/bench_b1_fp32_steps20_force_pg.json
timeout 300 python bench.py --batch 32 --steps 256 --cpu-steps 0 --no-extras > $O/bench_b32_fp32.json 2>/dev/null; cut -c1-160 $O/bench_b32_fp32.json
timeout 400 python tools/tb_curve.py fp32 1 2 3 4 5 6 7 8 9 10 12 14 16 17 18 20 22 24 26 28 30 32 33 34 > $O/step_time_vs_batch_fp32.jsonl 2>/dev/null
timeout 300 python tools/tb_curve.py fp16 1 2 3 4 5 6 8 16 24 32 > $O/step_time_vs_batch_fp16.jsonl 2>/dev/null
timeout 400 python tools/persist_probe.py > $O/persist_probe.jsonl 2>/dev/null; grep -c ids_identical $O/persist_probe.jsonl
timeout 600 python tools/ab_options.py fp32 "persistent_lora=0,1" --adapters --batches 1 2 4 5 8 --rounds 3 > $O/ab_persistent_lora.jsonl 2>/dev/null
timeout 400 python tools/request_probe.py > $O/request_probe.jsonl 2>/dev/null; tail -1 $O/request_probe.jsonl | cut -c1-300
timeout 300 python tools/prefill_probe.py 32 512 fp32 > $O/prefill_32x512_fp32.log 2>/dev/null; tail -1 $O/prefill_32x512_fp32.log
timeout 300 python tools/long_ctx_probe.py > $O/long_ctx_probe.jsonl 2>/dev/null
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b1 -- python $R/bench.py --steps 128 --warmup 16 --cpu-steps 0 --no-extras > /tmp/prof_b1.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b32 -- python $R/bench.py --batch 32 --steps 64 --warmup 16 --cpu-steps 0 --no-extras > /tmp/prof_b32.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b8 -- python $R/bench.py --batch 8 --steps 64 --warmup 16 --cpu-steps 0 --no-extras > /tmp/prof_b8.log 2>&1
f=$(find /tmp/prof_b1 -name '*kernel_trace.csv' | head -1); [ -n "$f" ] && python $R/tools/trace_gaps.py $f > $O/trace_gaps_b1.json
for t in b1 b32 b8; do
  f=$(find /tmp/prof_$t -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && head -40 $f > $O/${t}_fp32_kernel_stats.csv
  grep '"metric"' /tmp/prof_$t.log | cut -c1-400 > $O/${t}_prof_bench.json
done
for t in b1 b32; do
  BA="--batch 1"; [ $t = b32 ] && BA="--batch 32"
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_${t}_$c
    timeout 300 rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_${t}_$c -- python $R/bench.py $BA --prompt 293 --steps 16 --warmup 8 --gen-tokens 0 --cpu-steps 0 --no-extras > /tmp/pmc_${t}_$c.log 2>&1
    db=$(find /tmp/pmc_${t}_$c -name '*.db' | head -1)
    [ -n "$db" ] && python $R/tools/rocpd_pmc.py $db $c 14 $O/pmc_${t}_$c.json > /dev/null 2>> $O/pmc_errors.log || { echo "no db for $t $c" >> $O/pmc_errors.log; tail -3 /tmp/pmc_${t}_$c.log >> $O/pmc_errors.log; }
  done
done
cd $R
ls -la $O; cat $O/pmc_errors.log 2>/dev/null | tail -5; head -6 $O/b32_fp32_kernel_stats.csv | cut -c1-200
