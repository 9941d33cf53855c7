#!/bin/bash
# round 6, call C: launch shapes of the split decode kernels (32-row blocks, two row tiles per workgroup), prefetch mask sweep, vocoder overlap soak
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06c; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_split_decode.py -x -q -m gpu > $O/tests_split.log 2>&1; echo "split tests rc=$?" > $O/summary.txt
timeout 1500 python -m pytest tests/test_gpu_pipeline.py -x -q -m gpu -k "soak or oracle_chain" > $O/tests_pipe.log 2>&1; echo "pipeline tests rc=$?" >> $O/summary.txt
AB="timeout 600 python tools/ab_options.py fp32"
$AB "nbg2_rows=81,17" --fixed weight_prefetch_kb=0 --batches 17 24 32 --rounds 3 > $O/ab_nbg2.jsonl 2> $O/ab_nbg2.err
$AB "split_row_tiles=1,2" --fixed weight_prefetch_kb=0 --batches 9 16 17 32 --rounds 3 > $O/ab_rt.jsonl 2> $O/ab_rt.err
$AB "split_row_tiles=1,2" --fixed weight_prefetch_kb=0,nbg2_rows=17 --batches 17 32 --rounds 3 > $O/ab_rt_nbg2.jsonl 2> $O/ab_rt_nbg2.err
$AB "weight_prefetch_mask=0,1,2,4,8,12,15" --fixed weight_prefetch_kb=48 --batches 12 32 --rounds 3 > $O/ab_pfmask.jsonl 2> $O/ab_pfmask.err
$AB "weight_prefetch_mask=0,4,8,12" --fixed weight_prefetch_kb=24 --batches 8 32 --rounds 3 > $O/ab_pfmask24.jsonl 2> $O/ab_pfmask24.err
tail -n 3 $O/tests_split.log $O/tests_pipe.log; cat $O/summary.txt $O/ab_*.jsonl
