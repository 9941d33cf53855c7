# Round 6, second collection (after the decode attention's 8-wave blocks went up to 42 rows): suite + smoke, the bench lines, the fp32 / fp16 step curves, batch-32 kernel stats and traffic.
# -> gpurun_out/fin_r06b (tools/collect_profiles_r06.py b copies them over the first collection's files)
set -x
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/fin_r06b
mkdir -p $O
cd $R
timeout 1800 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -3 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 900 python bench.py > $O/bench_b1_fp32.json 2> $O/bench_b1_fp32.err; cut -c1-200 $O/bench_b1_fp32.json
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_b1_fp32_steps20.json 2>/dev/null; cut -c1-200 $O/bench_b1_fp32_steps20.json
timeout 600 python bench.py --steps 20 --warmup 5 --force-pg > $O/bench_b1_fp32_steps20_force_pg.json 2>/dev/null; cut -c1-200 $O/bench_b1_fp32_steps20_force_pg.json
timeout 300 python bench.py --batch 32 --steps 256 --cpu-steps 0 --no-extras > $O/bench_b32_fp32.json 2>/dev/null; cut -c1-160 $O/bench_b32_fp32.json
timeout 400 python tools/tb_curve.py fp32 1 2 3 4 5 6 7 8 9 10 12 14 16 17 18 20 22 24 26 28 30 32 33 34 40 48 64 > $O/step_time_vs_batch_fp32.jsonl 2>/dev/null
timeout 300 python tools/tb_curve.py fp16 1 2 3 4 5 6 8 16 24 32 48 64 > $O/step_time_vs_batch_fp16.jsonl 2>/dev/null
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b32 -- python $R/bench.py --batch 32 --steps 64 --warmup 16 --cpu-steps 0 --no-extras > /tmp/prof_b32.log 2>&1
f=$(find /tmp/prof_b32 -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && head -40 $f > $O/b32_fp32_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_b32_$c
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_b32_$c -- python $R/bench.py --batch 32 --prompt 293 --steps 16 --warmup 8 --gen-tokens 0 --cpu-steps 0 --no-extras > /tmp/pmc_b32_$c.log 2>&1
  db=$(find /tmp/pmc_b32_$c -name '*.db' | head -1)
  [ -n "$db" ] && python $R/tools/rocpd_pmc.py $db $c 14 $O/pmc_b32_$c.json > /dev/null 2>> $O/pmc_errors.log
done
cd $R; ls $O; head -6 $O/b32_fp32_kernel_stats.csv | cut -c1-200
