#!/bin/bash
# round 6, call A: split decode kernels -- parity tests, then interleaved A/B of the step time
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06a
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_split_decode.py -x -q -m gpu > gpurun_out/r06a/test_split.log 2>&1
echo "split tests rc=$?" >> gpurun_out/r06a/summary.txt
timeout 1200 python -m pytest tests/test_gpu_gpt.py -x -q -m gpu > gpurun_out/r06a/test_gpt.log 2>&1
echo "gpt tests rc=$?" >> gpurun_out/r06a/summary.txt
timeout 600 python tools/ab_options.py fp32 "split_decode_rows=0,9" --batches 9 12 16 17 20 24 32 --rounds 3 > gpurun_out/r06a/ab_split.jsonl 2> gpurun_out/r06a/ab_split.err
timeout 600 python tools/ab_options.py fp32 "nbg2_rows=81,17" --fixed split_decode_rows=9 --batches 17 24 32 --rounds 3 > gpurun_out/r06a/ab_nbg2.jsonl 2> gpurun_out/r06a/ab_nbg2.err
tail -3 gpurun_out/r06a/test_split.log gpurun_out/r06a/test_gpt.log
cat gpurun_out/r06a/summary.txt gpurun_out/r06a/ab_split.jsonl gpurun_out/r06a/ab_nbg2.jsonl
