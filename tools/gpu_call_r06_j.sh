#!/bin/bash
# round 6, call J: the launch chain at <= 8 rows, round-5 library against the current one (same box, same script)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06j; mkdir -p $O
export TMPDIR=/tmp
AB="timeout 600 python tools/ab_options.py"
for lib in r05 cur r05 cur; do
  L=$PWD/chatttsplus_amd/_lib/libctts_hip_r05.so; [ $lib = cur ] && L=$PWD/chatttsplus_amd/_lib/libctts_hip.so
  CTTS_HIP_LIB=$L $AB fp32 "persistent_rows=0" --batches 1 2 4 6 8 --rounds 3 >> $O/chain_$lib.jsonl 2>> $O/chain_$lib.err
done
for lib in r05 cur; do
  L=$PWD/chatttsplus_amd/_lib/libctts_hip_r05.so; [ $lib = cur ] && L=$PWD/chatttsplus_amd/_lib/libctts_hip.so
  CTTS_HIP_LIB=$L timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$lib -- python tools/ab_options.py fp32 "persistent_rows=0" --batches 1 --rounds 2 > $O/prof_$lib.log 2>&1
  f=$(find /tmp/prof_$lib -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/b1_chain_kernel_stats_$lib.csv
done
cat $O/chain_r05.jsonl; echo; cat $O/chain_cur.jsonl; head -9 $O/b1_chain_kernel_stats_r05.csv | cut -c1-220; head -9 $O/b1_chain_kernel_stats_cur.csv | cut -c1-220
