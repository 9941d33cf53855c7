#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06l; mkdir -p $O
export TMPDIR=/tmp
AB="timeout 600 python tools/ab_options.py"
for lib in r05 cur r05 cur; do
  L=$PWD/chatttsplus_amd/_lib/libctts_hip_r05.so; [ $lib = cur ] && L=$PWD/chatttsplus_amd/_lib/libctts_hip.so
  CTTS_HIP_LIB=$L $AB fp32 "persistent_rows=0" --batches 1 4 8 --rounds 3 >> $O/chain_$lib.jsonl 2>> $O/chain_$lib.err
done
$AB fp32 "split_decode_rows=0,9" --batches 12 32 --rounds 3 > $O/ab_split.jsonl 2> $O/ab_split.err
cat $O/chain_r05.jsonl; echo; cat $O/chain_cur.jsonl; cat $O/ab_split.jsonl
