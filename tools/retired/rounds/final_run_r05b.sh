# Round-5 evidence, second collection (after the multi-value row sums and the 5-row persistent launch): GPU suite + smoke, the bench lines, step time vs batch (fp32 / fp16),
# the persistent launch's probe, kernel stats and FETCH / WRITE traffic of batch 1 and batch 32.  (The opt-in MFMA stack's evidence is tools/final_run_r05.sh's.)
set -x
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/fin_r05
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -3 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 900 python bench.py > $O/bench_b1_fp32.json 2> $O/bench_b1_fp32.err; cut -c1-200 $O/bench_b1_fp32.json
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_b1_fp32_steps20.json 2>/dev/null; cut -c1-200 $O/bench_b1_fp32_steps20.json
timeout 600 python bench.py --steps 20 --warmup 5 --force-pg > $O/bench_b1_fp32_steps20_force_pg.json 2>/dev/null; cut -c1-200 $O/bench_b1_fp32_steps20_force_pg.json
timeout 300 python bench.py --batch 32 --steps 256 --cpu-steps 0 --no-extras > $O/bench_b32_fp32.json 2>/dev/null; cut -c1-160 $O/bench_b32_fp32.json
timeout 300 python tools/tb_curve.py fp32 1 2 3 4 5 6 8 10 12 14 16 17 18 20 22 24 26 28 30 32 > $O/step_time_vs_batch_fp32.jsonl 2>/dev/null
timeout 300 python tools/tb_curve.py fp16 1 2 3 4 5 6 8 16 24 32 > $O/step_time_vs_batch_fp16.jsonl 2>/dev/null
timeout 300 python tools/persist_probe.py --skip-layer > $O/persist_probe.jsonl 2>/dev/null; grep -c ids_identical $O/persist_probe.jsonl
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b1 -- python $R/bench.py --steps 128 --warmup 16 --cpu-steps 0 --no-extras > /tmp/prof_b1.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b32 -- python $R/bench.py --batch 32 --steps 64 --warmup 16 --cpu-steps 0 --no-extras > /tmp/prof_b32.log 2>&1
f=$(find /tmp/prof_b1 -name '*kernel_trace.csv' | head -1); [ -n "$f" ] && python $R/tools/trace_gaps.py $f > $O/trace_gaps_b1.json
for t in b1 b32; do
  f=$(find /tmp/prof_$t -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/${t}_fp32_kernel_stats.csv
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
ls -la $O | head -40
