#!/bin/bash
# round 6, call O: per-utterance adapters inside the persistent launch: tests, step time with adapters chain (fold) vs persistent at 1..8 rows
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06o; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_persistent.py -q -m gpu -x -k "adapters" > $O/tests.log 2>&1; echo "tests rc=$?" > $O/summary.txt
timeout 900 python -m pytest tests/test_gpu_pipeline.py -q -m gpu -x -k "lora or adapter" > $O/tests_pipe.log 2>&1; echo "pipe tests rc=$?" >> $O/summary.txt
for B in 1 2 4 5 8; do
  timeout 300 python tools/lora_probe.py --rows $B --tokens 256 --modes none,fold,persist >> $O/lora_probe.jsonl 2>> $O/lora_probe.err
done
timeout 300 python tools/lora_probe.py --rows 1 --tokens 256 --dtype fp16 --modes none,fold,persist >> $O/lora_probe.jsonl 2>> $O/lora_probe.err
tail -n 25 $O/tests.log; tail -n 5 $O/tests_pipe.log; cat $O/summary.txt; cat $O/lora_probe.jsonl; tail -3 $O/lora_probe.err
