#!/bin/bash
# round 6, call W: poll delays re-swept after the 16-byte act granules (batch 1, 2, 4)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06w; mkdir -p $O
export TMPDIR=/tmp
for sw in "persistent_delay_act=6,8,10,12,14,16" "persistent_delay=8,10,12,14" "persistent_delay_x=11,13,15,17" "persistent_delay_att=4,6,8,10,12" "persistent_nap=0,1,2"; do
  timeout 600 python tools/ab_options.py fp32 "$sw" --batches 1 2 4 --rounds 3 --steps 128 >> $O/sweep.jsonl 2>> $O/sweep.err
done
cut -c1-20,60-500 $O/sweep.jsonl
