#!/bin/bash
# round 6, call N: the two-item attention workgroups' K / V tail in LDS (PL_TAIL_IT = 4 iterations per wave = 128 keys per item): tests, step time 5..8 rows vs the build without it
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06n; mkdir -p $O
export TMPDIR=/tmp
CTTS_HIP_LIB=$PWD/chatttsplus_amd/_lib/libctts_hip_tail.so timeout 900 python -m pytest tests/test_gpu_persistent.py -q -m gpu -x > $O/tests.log 2>&1; echo "tests rc=$?" > $O/summary.txt
L=$PWD/chatttsplus_amd/_lib
for rep in 1 2; do
  for v in main notail; do
    lib=$L/libctts_hip_tail.so; [ $v != main ] && lib=$L/libctts_hip_$v.so
    CTTS_HIP_LIB=$lib timeout 300 python tools/tb_curve.py fp32 5 6 7 8 >> $O/tb_$v.jsonl 2>> $O/tb_$v.err
  done
done
for v in main notail; do
  lib=$L/libctts_hip_tail.so; [ $v != main ] && lib=$L/libctts_hip_$v.so
  CTTS_HIP_LIB=$lib timeout 600 python tools/ab_options.py fp32 "persistent_rows=5,8" --batches 6 8 --rounds 3 --prompt 300 > $O/ab_p300_$v.jsonl 2> $O/ab_p300_$v.err
  CTTS_HIP_LIB=$lib timeout 600 python tools/ab_options.py fp32 "persistent_rows=5,8" --batches 6 8 --rounds 3 --prompt 150 > $O/ab_p150_$v.jsonl 2> $O/ab_p150_$v.err
done
tail -n 3 $O/tests.log; cat $O/summary.txt
for v in main notail; do echo "== $v"; cut -c1-80 $O/tb_$v.jsonl; cut -c1-300 $O/ab_p300_$v.jsonl $O/ab_p150_$v.jsonl; done
