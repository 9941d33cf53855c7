# Round-3 evidence collection on one MI355X box: tests, smoke, bench lines (parity mode = default, fast mode), step time vs batch size,
# wall clocks, prompt-pass clocks, rocprofv3 kernel stats and the FETCH / WRITE traffic passes of the default (fp32) mode.
# Writes only small summaries into gpurun_out/fin_r03 (databases stay in /tmp).
set -x
export TMPDIR=/tmp
O=gpurun_out/fin_r03
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -3 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 900 python bench.py > $O/bench_b1_fp32.json 2> $O/bench_b1_fp32.err; cut -c1-200 $O/bench_b1_fp32.json
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_b1_fp32_driver_args.json 2>/dev/null; cut -c1-200 $O/bench_b1_fp32_driver_args.json
for B in 32 64 128; do timeout 300 python bench.py --batch $B --steps 256 --cpu-steps 0 --no-extras > $O/bench_b${B}_fp32.json 2>/dev/null; cut -c1-160 $O/bench_b${B}_fp32.json; done
timeout 300 python bench.py --dtype fp16 --steps 512 --cpu-steps 0 --no-extras > $O/bench_b1_fp16.json 2>/dev/null; cut -c1-160 $O/bench_b1_fp16.json
for B in 32 64 128; do timeout 300 python bench.py --dtype fp16 --batch $B --steps 256 --cpu-steps 0 --no-extras > $O/bench_b${B}_fp16.json 2>/dev/null; cut -c1-160 $O/bench_b${B}_fp16.json; done
timeout 300 python tools/tb_curve.py fp32 1 2 3 4 5 6 8 10 12 14 16 17 18 20 22 24 26 28 30 32 > $O/step_time_vs_batch_fp32.jsonl 2>/dev/null
timeout 300 python tools/tb_curve.py fp16 1 2 4 5 8 9 12 16 20 24 28 32 > $O/step_time_vs_batch_fp16.jsonl 2>/dev/null
rm -f $O/gen_wall.log $O/pipe_wall.log $O/prefill.log
for n in torch device; do timeout 200 python tools/gen_wall.py --noise $n 2>&1 | tail -1 >> $O/gen_wall.log; done
timeout 200 python tools/gen_wall.py --batch 32 --steps 256 --noise device 2>&1 | tail -1 >> $O/gen_wall.log
timeout 200 python tools/gen_wall.py --dtype fp16 --noise device 2>&1 | tail -1 >> $O/gen_wall.log
timeout 200 python tools/gen_wall.py --dtype fp16 --batch 32 --steps 256 --noise device 2>&1 | tail -1 >> $O/gen_wall.log
timeout 200 python tools/gen_wall.py --text --steps 128 --noise device 2>&1 | tail -1 >> $O/gen_wall.log
timeout 250 python tools/pipe_wall.py 2>&1 | tail -1 >> $O/pipe_wall.log
timeout 250 python tools/pipe_wall.py --n 32 --tokens 256 2>&1 | tail -1 >> $O/pipe_wall.log
timeout 250 python tools/pipe_wall.py --dtype fp16 2>&1 | tail -1 >> $O/pipe_wall.log
timeout 250 python tools/pipe_wall.py --dtype fp16 --n 32 --tokens 256 2>&1 | tail -1 >> $O/pipe_wall.log
for c in "" "--continuous" "--continuous throughput"; do timeout 250 python tools/pipe_wall.py --n 128 --rows 32 --tokens 512 --ragged --reps 2 $c 2>&1 | tail -1 >> $O/pipe_wall.log; done
for cfg in "1 48" "32 48" "32 96" "32 512"; do for wd in fp32 fp16; do timeout 120 python tools/prefill_probe.py $cfg $wd 2>&1 | grep "prompt pass" | tail -1 | sed "s/$/  ($wd)/" >> $O/prefill.log; done; done
rm -f $O/queue.jsonl $O/vocoder.log
for wd in fp32 fp16; do timeout 250 python tools/queue_probe.py $wd 2>&1 | tail -1 >> $O/queue.jsonl; done
timeout 120 python tools/voc_batch_probe.py 2>&1 | tail -1 >> $O/vocoder.log
cat $O/gen_wall.log $O/pipe_wall.log $O/prefill.log $O/queue.jsonl $O/vocoder.log
R=$GRAFT_REPO_ROOT
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b1 -- python $R/bench.py --steps 128 --warmup 16 --cpu-steps 0 --no-extras > /tmp/prof_b1.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b32 -- python $R/bench.py --batch 32 --steps 64 --warmup 16 --cpu-steps 0 --no-extras > /tmp/prof_b32.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_voc -- python $R/tools/voc_batch_probe.py > /tmp/prof_voc.log 2>&1
f=$(find /tmp/prof_voc -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $R/$O/vocoder_32x272_kernel_stats.csv
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_pre -- python $R/tools/prefill_probe.py 32 512 fp32 > /tmp/prof_pre.log 2>&1
f=$(find /tmp/prof_pre -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $R/$O/prefill_32x512_fp32_kernel_stats.csv
for t in b1 b32; do
  f=$(find /tmp/prof_$t -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $R/$O/${t}_fp32_kernel_stats.csv
  grep '"metric"' /tmp/prof_$t.log | cut -c1-400 > $R/$O/${t}_prof_bench.json
done
# HBM traffic: one counter per pass, kernel trace only (no other trace domains)
for t in b1 b32; do
  [ $t = b1 ] && BA="--batch 1" || BA="--batch 32"
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_${t}_$c
    timeout 600 rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_${t}_$c -- python $R/bench.py $BA --steps 64 --warmup 16 --gen-tokens 0 --cpu-steps 0 --no-extras > /tmp/pmc_${t}_$c.log 2>&1
    db=$(find /tmp/pmc_${t}_$c -name '*.db' | head -1)
    [ -n "$db" ] && python $R/tools/rocpd_pmc.py $db $c 60 $R/$O/pmc_${t}_$c.json > /dev/null 2>> $R/$O/pmc_errors.log || echo "no db for $t $c" >> $R/$O/pmc_errors.log
  done
done
cd $R
ls -la $O; head -8 $O/b1_fp32_kernel_stats.csv | cut -c1-160
