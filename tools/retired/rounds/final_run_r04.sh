# Round-4 evidence collection on one MI355X box (one gpurun call): GPU suite + smoke, bench lines (parity default = persistent launch for <= 4 rows;
# launch chain; fast mode), step time vs batch, rocprofv3 kernel stats, FETCH / WRITE traffic AT THE TIMED WINDOW'S CONTEXT (a 293-token prompt puts
# steps 8..24 at mean context 309: the full-length generation overruns rocprofv3's counter collection), the persistent launch's wait counters and its
# per-edge phase marks.  Small summaries only -> gpurun_out/fin_r04 (tools/collect_profiles_r04.py copies them to profiles/).
set -x
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/fin_r04
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -3 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 900 python bench.py > $O/bench_b1_fp32.json 2> $O/bench_b1_fp32.err; cut -c1-200 $O/bench_b1_fp32.json
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_b1_fp32_steps20.json 2>/dev/null; cut -c1-200 $O/bench_b1_fp32_steps20.json
timeout 300 python bench.py --steps 256 --cpu-steps 0 --no-extras --option persistent_rows=0 > $O/bench_b1_fp32_launch_chain.json 2>/dev/null; cut -c1-160 $O/bench_b1_fp32_launch_chain.json
for B in 2 4 32 64 128; do timeout 300 python bench.py --batch $B --steps 256 --cpu-steps 0 --no-extras > $O/bench_b${B}_fp32.json 2>/dev/null; cut -c1-160 $O/bench_b${B}_fp32.json; done
for B in 1 32 64 128; do timeout 300 python bench.py --dtype fp16 --batch $B --steps 256 --cpu-steps 0 --no-extras > $O/bench_b${B}_fp16.json 2>/dev/null; cut -c1-160 $O/bench_b${B}_fp16.json; done
timeout 300 python tools/tb_curve.py fp32 1 2 3 4 5 6 8 10 12 14 16 17 18 20 22 24 26 28 30 32 > $O/step_time_vs_batch_fp32.jsonl 2>/dev/null
timeout 300 python tools/tb_curve.py fp16 1 2 4 5 8 9 12 16 20 24 28 32 > $O/step_time_vs_batch_fp16.jsonl 2>/dev/null
timeout 300 python tools/persist_probe.py > $O/persist_probe.jsonl 2>/dev/null; tail -2 $O/persist_probe.jsonl | cut -c1-600
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b1 -- python $R/bench.py --steps 128 --warmup 16 --cpu-steps 0 --no-extras > /tmp/prof_b1.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b1c -- python $R/bench.py --steps 128 --warmup 16 --cpu-steps 0 --no-extras --option persistent_rows=0 > /tmp/prof_b1c.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b32 -- python $R/bench.py --batch 32 --steps 64 --warmup 16 --cpu-steps 0 --no-extras > /tmp/prof_b32.log 2>&1
for t in b1 b1c b32; do
  f=$(find /tmp/prof_$t -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/${t}_fp32_kernel_stats.csv
  grep '"metric"' /tmp/prof_$t.log | cut -c1-400 > $O/${t}_prof_bench.json
done
for t in b1 b1c b32; do
  BA="--batch 1"; [ $t = b32 ] && BA="--batch 32"; [ $t = b1c ] && BA="--batch 1 --option persistent_rows=0"
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_${t}_$c
    timeout 300 rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_${t}_$c -- python $R/bench.py $BA --prompt 293 --steps 16 --warmup 8 --gen-tokens 0 --cpu-steps 0 --no-extras > /tmp/pmc_${t}_$c.log 2>&1
    db=$(find /tmp/pmc_${t}_$c -name '*.db' | head -1)
    [ -n "$db" ] && python $R/tools/rocpd_pmc.py $db $c 14 $O/pmc_${t}_$c.json > /dev/null 2>> $O/pmc_errors.log || { echo "no db for $t $c" >> $O/pmc_errors.log; tail -3 /tmp/pmc_${t}_$c.log >> $O/pmc_errors.log; }
  done
done
rm -rf /tmp/pmc_sq
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES -d /tmp/pmc_sq -- python $R/bench.py --prompt 293 --steps 16 --warmup 8 --gen-tokens 0 --cpu-steps 0 --no-extras > /tmp/pmc_sq.log 2>&1
db=$(find /tmp/pmc_sq -name '*.db' | head -1); [ -n "$db" ] && python $R/tools/rocpd_counters.py $db $O/pmc_b1_sq.json > /dev/null 2>> $O/pmc_errors.log
cd $R
ls -la $O; cat $O/pmc_errors.log 2>/dev/null | tail -5; head -6 $O/b1_fp32_kernel_stats.csv | cut -c1-200
