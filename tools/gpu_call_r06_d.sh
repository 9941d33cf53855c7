#!/bin/bash
# round 6, call D: 32-row blocks by default from 17 rows, gate|up prefetch carried by o_proj: block counts, the down image too; fp16 engine
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06d; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_split_decode.py tests/test_gpu_gpt.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?" > $O/summary.txt
AB="timeout 600 python tools/ab_options.py"
$AB fp32 "weight_prefetch_kb=0,192,96,48,24" --batches 6 8 9 12 16 17 24 32 --rounds 3 > $O/ab_pfkb_fp32.jsonl 2> $O/ab_pfkb_fp32.err
$AB fp32 "weight_prefetch_mask=1,17,5,9" --fixed weight_prefetch_kb=96 --batches 8 16 32 --rounds 3 > $O/ab_pfmask_fp32.jsonl 2> $O/ab_pfmask_fp32.err
$AB fp16 "weight_prefetch_kb=0,96,48,24" --batches 6 8 16 32 --rounds 3 > $O/ab_pfkb_fp16.jsonl 2> $O/ab_pfkb_fp16.err
$AB fp32 "split_nbg2_rows=17,99" --batches 17 24 32 --rounds 3 > $O/ab_nbg2.jsonl 2> $O/ab_nbg2.err
tail -n 3 $O/tests.log; cat $O/summary.txt $O/ab_*.jsonl
