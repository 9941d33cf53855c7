# Round-2 evidence collection on one MI355X box: tests, smoke, bench lines, wall clocks, rocprofv3 kernel stats, PMC traffic, MFMA-busy counters.
# Writes only small summaries into gpurun_out/fin2 (databases stay in /tmp).
set -x
export TMPDIR=/tmp
O=gpurun_out/fin2
mkdir -p $O
timeout 1700 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -3 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 900 python bench.py > $O/bench_b1_fp16.json 2> $O/bench_b1_fp16.err; cut -c1-200 $O/bench_b1_fp16.json
timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench_b1_fp16_driver_args.json 2>/dev/null; cut -c1-200 $O/bench_b1_fp16_driver_args.json
timeout 300 python bench.py --batch 32 --steps 256 --cpu-steps 0 --no-extras > $O/bench_b32_fp16.json 2>/dev/null; cut -c1-200 $O/bench_b32_fp16.json
timeout 300 python bench.py --batch 64 --steps 256 --cpu-steps 0 --no-extras > $O/bench_b64_fp16.json 2>/dev/null; cut -c1-200 $O/bench_b64_fp16.json
timeout 300 python bench.py --batch 128 --steps 256 --cpu-steps 0 --no-extras > $O/bench_b128_fp16.json 2>/dev/null; cut -c1-200 $O/bench_b128_fp16.json
timeout 300 python bench.py --dtype fp32 --steps 256 --cpu-steps 0 --no-extras > $O/bench_b1_fp32.json 2>/dev/null; cut -c1-200 $O/bench_b1_fp32.json
timeout 300 python bench.py --dtype fp32 --batch 32 --steps 128 --cpu-steps 0 --no-extras > $O/bench_b32_fp32.json 2>/dev/null; cut -c1-200 $O/bench_b32_fp32.json
rm -f $O/gen_wall.log $O/pipe_wall.log $O/prefill.log
for n in torch device; do timeout 200 python tools/gen_wall.py --noise $n 2>&1 | tail -1 >> $O/gen_wall.log; done
timeout 200 python tools/gen_wall.py --batch 32 --steps 256 --noise device 2>&1 | tail -1 >> $O/gen_wall.log
timeout 200 python tools/gen_wall.py --text --steps 128 --noise device 2>&1 | tail -1 >> $O/gen_wall.log
timeout 200 python tools/gen_wall.py --text --steps 128 --noise device --batch 8 2>&1 | tail -1 >> $O/gen_wall.log
timeout 250 python tools/pipe_wall.py 2>&1 | tail -1 >> $O/pipe_wall.log
timeout 250 python tools/pipe_wall.py --n 32 --tokens 256 2>&1 | tail -1 >> $O/pipe_wall.log
for cfg in "1 48" "1 512" "32 96" "32 512" "8 2000"; do timeout 120 python tools/prefill_probe.py $cfg 2>&1 | grep "prompt pass" | tail -1 >> $O/prefill.log; done
for cfg in "32 512" "8 2000"; do CTTS_PREFILL_GEMM=0 CTTS_PREFILL_ATTN=0 timeout 120 python tools/prefill_probe.py $cfg 2>&1 | grep "prompt pass" | tail -1 | sed 's/$/  (round-1 kernels: CTTS_PREFILL_GEMM=0 CTTS_PREFILL_ATTN=0)/' >> $O/prefill.log; done
for cfg in "32 96" "4 512" "2 512"; do for th in 2048 1024; do CTTS_PREFILL_GEMM=$th timeout 120 python tools/prefill_probe.py $cfg 2>&1 | grep "prompt pass" | tail -1 | sed "s/$/  (prompt GEMM from $th rows)/" >> $O/prefill.log; done; done
timeout 200 python tools/fp16_agreement.py > $O/fp16_token_agreement.json 2>/dev/null
cat $O/gen_wall.log $O/pipe_wall.log $O/prefill.log
R=$GRAFT_REPO_ROOT
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b1 -- python $R/bench.py --steps 128 --warmup 16 --cpu-steps 0 --no-extras > /tmp/prof_b1.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b32 -- python $R/bench.py --batch 32 --steps 64 --warmup 16 --cpu-steps 0 --no-extras > /tmp/prof_b32.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_pre -- python $R/tools/prefill_probe.py 32 512 > /tmp/prof_pre.log 2>&1
for t in b1 b32 pre; do
  f=$(find /tmp/prof_$t -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $R/$O/${t}_kernel_stats.csv
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
# MFMA-busy counters (north_star): prompt pass and vocoder, SQ counters in their own pass
for t in prefill vocoder; do
  [ $t = prefill ] && CMD="python $R/tools/prefill_probe.py 32 512" || CMD="python $R/tools/voc_probe.py"
  rm -rf /tmp/pmc_mfma_$t
  timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES -d /tmp/pmc_mfma_$t -- $CMD > /tmp/pmc_mfma_$t.log 2>&1
  db=$(find /tmp/pmc_mfma_$t -name '*.db' | head -1)
  [ -n "$db" ] && python $R/tools/rocpd_mfma.py $db $R/$O/pmc_mfma_$t.json 2>> $R/$O/pmc_errors.log || echo "no db for mfma $t" >> $R/$O/pmc_errors.log
done
cd $R
ls -la $O; head -8 $O/b1_kernel_stats.csv
