# Round evidence collection on one MI355X box: tests, smoke, bench lines, wall clocks, rocprofv3 kernel stats and PMC traffic.
# Writes only small summaries into gpurun_out/fin (databases stay in /tmp).
set -x
export TMPDIR=/tmp
O=gpurun_out/fin
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -3 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 600 python bench.py > $O/bench_b1_fp16.json 2> $O/bench_b1_fp16.err; cut -c1-200 $O/bench_b1_fp16.json
timeout 300 python bench.py --batch 32 --steps 256 --cpu-steps 0 > $O/bench_b32_fp16.json 2>/dev/null; cut -c1-200 $O/bench_b32_fp16.json
timeout 300 python bench.py --dtype fp32 --steps 256 --cpu-steps 0 > $O/bench_b1_fp32.json 2>/dev/null; cut -c1-200 $O/bench_b1_fp32.json
rm -f $O/gen_wall.log $O/pipe_wall.log
for n in torch device; do timeout 200 python tools/gen_wall.py --noise $n 2>&1 | tail -1 >> $O/gen_wall.log; done
timeout 200 python tools/gen_wall.py --batch 32 --steps 256 --noise device 2>&1 | tail -1 >> $O/gen_wall.log
timeout 200 python tools/gen_wall.py --text --steps 128 --noise device 2>&1 | tail -1 >> $O/gen_wall.log
timeout 250 python tools/pipe_wall.py 2>&1 | tail -1 >> $O/pipe_wall.log
timeout 250 python tools/pipe_wall.py --n 32 --tokens 256 2>&1 | tail -1 >> $O/pipe_wall.log
cat $O/gen_wall.log $O/pipe_wall.log
R=$GRAFT_REPO_ROOT
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b1 -- python $R/bench.py --steps 128 --warmup 16 --cpu-steps 0 > /tmp/prof_b1.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b32 -- python $R/bench.py --batch 32 --steps 64 --warmup 16 --cpu-steps 0 > /tmp/prof_b32.log 2>&1
for t in b1 b32; do
  f=$(find /tmp/prof_$t -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $R/$O/${t}_kernel_stats.csv
  grep '"metric"' /tmp/prof_$t.log | cut -c1-400 > $R/$O/${t}_prof_bench.json
done
# HBM traffic: one counter per pass, kernel trace only (no other trace domains)
for t in b1 b32; do
  [ $t = b1 ] && BA="--batch 1" || BA="--batch 32"
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_$t_$c
    timeout 600 rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_${t}_$c -- python $R/bench.py $BA --steps 64 --warmup 16 --cpu-steps 0 > /tmp/pmc_${t}_$c.log 2>&1
    db=$(find /tmp/pmc_${t}_$c -name '*.db' | head -1)
    [ -n "$db" ] && python $R/tools/rocpd_pmc.py $db $c 60 $R/$O/pmc_${t}_$c.json > /dev/null 2>> $R/$O/pmc_errors.log || echo "no db for $t $c" >> $R/$O/pmc_errors.log
  done
done
cd $R
ls -la $O; head -8 $O/b1_kernel_stats.csv
