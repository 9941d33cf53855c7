#!/bin/bash
# round 6, call E: persistent launch at 6-8 rows (two attention items per workgroup): parity, A/B against the chain; RT=2 debug
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06e; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python tools/scratch/rt2_debug.py > $O/rt2_debug.log 2>&1
timeout 900 python -m pytest tests/test_gpu_persistent.py -x -q -m gpu -k "six_to_eight or default" > $O/tests_persist.log 2>&1; echo "persist tests rc=$?" > $O/summary.txt
timeout 900 python -m pytest tests/test_gpu_split_decode.py -x -q -m gpu > $O/tests_split.log 2>&1; echo "split tests rc=$?" >> $O/summary.txt
AB="timeout 600 python tools/ab_options.py"
$AB fp32 "persistent_rows=5,8" --batches 6 7 8 --rounds 3 > $O/ab_p8_fp32.jsonl 2> $O/ab_p8_fp32.err
$AB fp16 "persistent_rows=5,8" --batches 6 7 8 --rounds 3 > $O/ab_p8_fp16.jsonl 2> $O/ab_p8_fp16.err
$AB fp32 "persistent_rows=5,8" --batches 6 8 --rounds 3 --prompt 300 > $O/ab_p8_fp32_p300.jsonl 2> $O/ab_p8_fp32_p300.err
cat $O/rt2_debug.log | tail -8; tail -n 3 $O/tests_persist.log $O/tests_split.log; cat $O/summary.txt $O/ab_*.jsonl
