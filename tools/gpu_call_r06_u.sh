#!/bin/bash
# round 6, call U: kernel stats of the launch chain at 32 / 17 / 16 rows (split decode kernels)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r06u; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for B in 32 17 16; do
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b$B -- python $GRAFT_REPO_ROOT/bench.py --batch $B --steps 64 --warmup 16 --cpu-steps 0 --no-extras > /tmp/prof_b$B.log 2>&1
  f=$(find /tmp/prof_b$B -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/b${B}_fp32_kernel_stats.csv
  grep '"metric"' /tmp/prof_b$B.log | cut -c1-300 > $O/b${B}_prof_bench.json
  f=$(find /tmp/prof_b$B -name '*kernel_trace.csv' | head -1); [ -n "$f" ] && python $GRAFT_REPO_ROOT/tools/trace_gaps.py $f --main attn_decode_kernel --skip 400 > $O/b${B}_trace_gaps.json 2>/dev/null
done
head -14 $O/b32_fp32_kernel_stats.csv | cut -c1-200; head -12 $O/b17_fp32_kernel_stats.csv | cut -c1-200; head -12 $O/b16_fp32_kernel_stats.csv | cut -c1-200
