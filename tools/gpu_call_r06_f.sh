#!/bin/bash
# round 6, call F: 6-8 persistent rows after the spill work (main build and the one-row act sweeps variant), all gpt / split / persistent tests on the current defaults
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06f; mkdir -p $O
export TMPDIR=/tmp
AB="timeout 600 python tools/ab_options.py"
$AB fp32 "persistent_rows=5,8" --batches 6 7 8 --rounds 3 > $O/ab_p8_fp32.jsonl 2> $O/ab_p8_fp32.err
CTTS_HIP_LIB=$PWD/chatttsplus_amd/_lib/libctts_hip_e1.so $AB fp32 "persistent_rows=5,8" --batches 6 7 8 --rounds 3 > $O/ab_p8_fp32_e1.jsonl 2> $O/ab_p8_fp32_e1.err
timeout 1500 python -m pytest tests/test_gpu_split_decode.py tests/test_gpu_gpt.py tests/test_gpu_persistent.py -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?" > $O/summary.txt
$AB fp32 "split_decode_rows=0,9" --batches 9 16 17 24 32 --rounds 3 > $O/ab_split_final.jsonl 2> $O/ab_split_final.err
tail -n 4 $O/tests.log; cat $O/summary.txt $O/ab_*.jsonl
