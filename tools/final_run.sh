set -x
export TMPDIR=/tmp
mkdir -p gpurun_out/fin
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/fin/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/fin/pytest_gpu.log
tail -3 gpurun_out/fin/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > gpurun_out/fin/smoke.log 2>&1; tail -2 gpurun_out/fin/smoke.log
for n in torch device; do timeout 300 python tools/gen_wall.py --noise $n >> gpurun_out/fin/gen_wall.log 2>&1; done
timeout 300 python tools/gen_wall.py --batch 32 --steps 256 --noise torch >> gpurun_out/fin/gen_wall.log 2>&1
cat gpurun_out/fin/gen_wall.log
timeout 600 python bench.py > gpurun_out/fin/bench_b1_fp16.json 2> gpurun_out/fin/bench_b1_fp16.err; cat gpurun_out/fin/bench_b1_fp16.json
timeout 300 python bench.py --batch 32 --steps 256 --cpu-steps 0 > gpurun_out/fin/bench_b32_fp16.json 2>/dev/null; cat gpurun_out/fin/bench_b32_fp16.json
timeout 300 python bench.py --dtype fp32 --steps 256 --cpu-steps 0 > gpurun_out/fin/bench_b1_fp32.json 2>/dev/null; cat gpurun_out/fin/bench_b1_fp32.json
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b1 -- python $GRAFT_REPO_ROOT/bench.py --steps 128 --warmup 16 --cpu-steps 0 > /tmp/prof_b1.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b32 -- python $GRAFT_REPO_ROOT/bench.py --batch 32 --steps 64 --warmup 16 --cpu-steps 0 > /tmp/prof_b32.log 2>&1
cd $GRAFT_REPO_ROOT
for t in b1 b32; do
  f=$(find /tmp/prof_$t -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f gpurun_out/fin/${t}_kernel_stats.csv
  tail -2 /tmp/prof_$t.log > gpurun_out/fin/${t}_prof_bench.json
done
ls -la gpurun_out/fin; head -12 gpurun_out/fin/b1_kernel_stats.csv
