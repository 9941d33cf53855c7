#!/bin/bash
# round 6, call X: split prompt attention with 8-wave blocks + 16x16x32 P.V: goldens, prompt-pass clocks main vs 4-wave variant, kernel stats
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r06x; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_gpt.py tests/test_gpu_properties.py -q -m gpu -x > $O/tests.log 2>&1; echo "tests rc=$?" > $O/summary.txt
L=$PWD/chatttsplus_amd/_lib
for rep in 1 2; do
  for v in main fs4; do
    lib=$L/libctts_hip.so; [ $v != main ] && lib=$L/libctts_hip_$v.so
    CTTS_HIP_LIB=$lib timeout 300 python tools/prefill_probe.py 32 512 fp32 2>/dev/null | tail -1 >> $O/prefill_32x512_$v.log
    CTTS_HIP_LIB=$lib timeout 300 python tools/prefill_probe.py 8 512 fp32 2>/dev/null | tail -1 >> $O/prefill_8x512_$v.log
  done
done
cd /tmp
for v in main fs4; do
  lib=$L/libctts_hip.so; [ $v != main ] && lib=$L/libctts_hip_$v.so
  CTTS_HIP_LIB=$lib timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_pf_$v -- python $GRAFT_REPO_ROOT/tools/prefill_probe.py 32 512 fp32 > /tmp/prof_pf_$v.log 2>&1
  f=$(find /tmp/prof_pf_$v -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && head -12 $f | cut -c1-220 > $O/prefill_32x512_kernel_stats_$v.csv
done
cd $GRAFT_REPO_ROOT
tail -n 3 $O/tests.log; cat $O/summary.txt; for v in main fs4; do echo "== $v"; cat $O/prefill_32x512_$v.log $O/prefill_8x512_$v.log; cat $O/prefill_32x512_kernel_stats_$v.csv | cut -c1-180 | head -8; done
