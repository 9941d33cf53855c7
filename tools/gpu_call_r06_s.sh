#!/bin/bash
# round 6, call S: adapters inside the persistent launch with the poll delay by row count: chain vs persistent at 1..8 rows; full persistent + pipeline adapter tests
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06s; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python tools/ab_options.py fp32 "persistent_lora=0,1" --adapters --batches 1 2 3 4 5 6 7 8 --rounds 3 > $O/ab_persistent_lora.jsonl 2> $O/ab_persistent_lora.err
timeout 900 python tools/ab_options.py fp16 "persistent_lora=0,1" --adapters --batches 1 2 4 5 --rounds 3 > $O/ab_persistent_lora_fp16.jsonl 2> $O/ab_persistent_lora_fp16.err
timeout 1200 python -m pytest tests/test_gpu_persistent.py tests/test_gpu_pipeline.py -q -m gpu -x > $O/tests.log 2>&1; echo "tests rc=$?" > $O/summary.txt
cat $O/ab_persistent_lora.jsonl $O/ab_persistent_lora_fp16.jsonl | cut -c1-400; tail -n 4 $O/tests.log; cat $O/summary.txt
