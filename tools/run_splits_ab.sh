#!/bin/bash
# decode attention key splits at batch 32 / 64 (CTTS_SPLITS diagnostic)
mkdir -p gpurun_out/pfab
L=gpurun_out/pfab/splits.log
: > $L
for cfg in "32 48" "32 512" "64 48"; do
  set -- $cfg
  for S in 1 2; do
    echo "== batch $1 prompt $2 CTTS_SPLITS=$S" >> $L
    CTTS_SPLITS=$S timeout 80 python bench.py --batch $1 --prompt $2 --steps 192 --cpu-steps 0 --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'])" >> $L
  done
done
cat $L
