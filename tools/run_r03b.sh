set -x
export TMPDIR=/tmp
O=gpurun_out/r03b
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -40 $O/pytest_gpu.log
timeout 300 python tools/tb_curve.py fp32 > $O/tb_fp32.jsonl 2>$O/tb_fp32.err; cat $O/tb_fp32.jsonl
timeout 300 python tools/tb_curve.py fp16 > $O/tb_fp16.jsonl 2>$O/tb_fp16.err; cat $O/tb_fp16.jsonl
timeout 600 python bench.py --steps 20 --warmup 5 --cpu-steps 0 > $O/bench_default.json 2> $O/bench_default.err; python -c "
import json;d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'],d['roofline']['frac']);print(d['extra'].get('batch32_ragged_targets'));print(d['extra'].get('error'))"
R=$GRAFT_REPO_ROOT
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b1 -- python $R/bench.py --steps 128 --warmup 16 --cpu-steps 0 --no-extras > /tmp/prof_b1.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b32 -- python $R/bench.py --batch 32 --steps 64 --warmup 16 --cpu-steps 0 --no-extras > /tmp/prof_b32.log 2>&1
for t in b1 b32; do
  f=$(find /tmp/prof_$t -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $R/$O/${t}_fp32_kernel_stats.csv
  grep '"metric"' /tmp/prof_$t.log | cut -c1-400 > $R/$O/${t}_prof_bench.json
done
cd $R; head -12 $O/b32_fp32_kernel_stats.csv | cut -c1-200
