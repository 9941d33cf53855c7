#!/bin/bash
# round 6, call M: 16-byte act granules (PL_ACT16): tearing microbenchmark, persistent tests, step time 1..8 rows main (act16, 12 per sweep) vs act8 (8-byte granules) vs a16c9
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06m; mkdir -p $O
export TMPDIR=/tmp
timeout 120 tools/mb/tear16 20000 > $O/tear16.jsonl 2> $O/tear16.err
timeout 900 python -m pytest tests/test_gpu_persistent.py tests/test_gpu_gpt.py -q -m gpu -x > $O/tests.log 2>&1; echo "tests rc=$?" > $O/summary.txt
L=$PWD/chatttsplus_amd/_lib
for rep in 1 2; do
  for v in main act8 a16c9; do
    lib=$L/libctts_hip.so; [ $v != main ] && lib=$L/libctts_hip_$v.so
    CTTS_HIP_LIB=$lib timeout 300 python tools/tb_curve.py fp32 1 2 3 4 5 6 7 8 >> $O/tb_$v.jsonl 2>> $O/tb_$v.err
  done
done
CTTS_HIP_LIB=$L/libctts_hip.so timeout 300 python tools/tb_curve.py fp16 1 2 4 5 8 >> $O/tb16_main.jsonl 2>> $O/tb16_main.err
CTTS_HIP_LIB=$L/libctts_hip_act8.so timeout 300 python tools/tb_curve.py fp16 1 2 4 5 8 >> $O/tb16_act8.jsonl 2>> $O/tb16_act8.err
cat $O/tear16.jsonl; tail -n 3 $O/tests.log; cat $O/summary.txt
for v in main act8 a16c9; do echo "== $v"; cut -c1-80 $O/tb_$v.jsonl; done
echo "== fp16 main / act8"; cut -c1-80 $O/tb16_main.jsonl $O/tb16_act8.jsonl
