#!/bin/bash
# round 6, call B: weight prefetch across launch boundaries -- correctness, interleaved A/B, kernel stats with and without
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06b; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_split_decode.py tests/test_gpu_gpt.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?" > $O/summary.txt
timeout 900 python tools/ab_options.py fp32 "weight_prefetch_kb=0,192,96,48" --batches 6 8 12 16 17 24 32 --rounds 3 > $O/ab_pf_fp32.jsonl 2> $O/ab_pf_fp32.err
timeout 600 python tools/ab_options.py fp16 "weight_prefetch_kb=0,192,96" --batches 6 8 16 32 --rounds 3 > $O/ab_pf_fp16.jsonl 2> $O/ab_pf_fp16.err
for v in 0 192; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_pf$v -- python bench.py --batch 32 --steps 64 --warmup 16 --cpu-steps 0 --no-extras --option weight_prefetch_kb=$v > $O/prof_pf$v.log 2>&1
  f=$(find /tmp/prof_pf$v -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/b32_kernel_stats_pf$v.csv
done
tail -n 3 $O/tests.log; cat $O/summary.txt $O/ab_pf_fp32.jsonl $O/ab_pf_fp16.jsonl; head -12 $O/b32_kernel_stats_pf0.csv | cut -c1-200; head -12 $O/b32_kernel_stats_pf192.csv | cut -c1-200
