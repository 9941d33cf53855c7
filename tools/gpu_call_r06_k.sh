#!/bin/bash
# round 6, call K: chain at <= 8 rows after the code-layout fix (vs the round-5 library), where the sharded request's wall clock goes
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06k; mkdir -p $O
export TMPDIR=/tmp
AB="timeout 600 python tools/ab_options.py"
for lib in r05 cur r05 cur; do
  L=$PWD/chatttsplus_amd/_lib/libctts_hip_r05.so; [ $lib = cur ] && L=$PWD/chatttsplus_amd/_lib/libctts_hip.so
  CTTS_HIP_LIB=$L $AB fp32 "persistent_rows=0" --batches 1 4 8 --rounds 3 >> $O/chain_$lib.jsonl 2>> $O/chain_$lib.err
done
timeout 600 python tools/request_probe.py 32 256 > $O/request_probe.jsonl 2> $O/request_probe.err
timeout 600 python tools/request_probe.py 32 256 4 > $O/request_probe_chunk4.jsonl 2> $O/request_probe_chunk4.err
timeout 600 python tools/request_probe.py 32 256 8 2 > $O/request_probe_admit2.jsonl 2> $O/request_probe_admit2.err
$AB fp32 "split_decode_rows=0,9" --batches 9 12 16 17 20 24 28 32 --rounds 3 > $O/ab_split.jsonl 2> $O/ab_split.err
cat $O/chain_r05.jsonl; echo; cat $O/chain_cur.jsonl; cat $O/request_probe*.jsonl; tail -3 $O/request_probe.err; cat $O/ab_split.jsonl
