#!/bin/bash
# round 6, call R: poll delay of the u granules (adapters inside the persistent launch)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06r; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python tools/ab_options.py fp32 "persistent_delay_lora=0,4,8,12,16,24" --adapters --batches 1 2 4 --rounds 3 > $O/ab_delay_u.jsonl 2> $O/ab_delay_u.err
timeout 900 python tools/ab_options.py fp32 "persistent_lora=0,1" --adapters --batches 1 2 3 4 5 6 8 --rounds 3 > $O/ab_persistent_lora.jsonl 2> $O/ab_persistent_lora.err
cat $O/ab_delay_u.jsonl $O/ab_persistent_lora.jsonl; tail -3 $O/ab_delay_u.err
