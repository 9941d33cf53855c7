mkdir -p gpurun_out
(timeout 600 python -m pytest tests/test_gpu_pipeline.py -m gpu -q -x 2>&1 | tail -15 > gpurun_out/pytest_r2e.log)
run() { # label, env...
  local label=$1; shift
  for B in 32 64; do
    env "$@" timeout 200 python bench.py --steps 128 --batch $B --no-extras --cpu-steps 0 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('B=$B $label', d['value'], d['ms_per_step'], d['roofline']['frac'])
" >> gpurun_out/ab_r2e.log
  done
}
run default X=1
run nbg2 CTTS_NBG2_ROWS=17
run attn4 CTTS_ATTN_WIDE=0
run rtxh2 CTTS_HIP_LIB=$PWD/chatttsplus_amd/_lib/libctts_hip_rtxh2.so
run rtxh2_nbg2 CTTS_HIP_LIB=$PWD/chatttsplus_amd/_lib/libctts_hip_rtxh2.so CTTS_NBG2_ROWS=17
ROOTD=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOTD/gpurun_out/prof_prefill -- python $ROOTD/tools/prefill_probe.py 32 512 > $ROOTD/gpurun_out/prefill_probe.log 2>&1
cd $ROOTD
tail -5 gpurun_out/pytest_r2e.log; cat gpurun_out/ab_r2e.log; grep "prompt pass" gpurun_out/prefill_probe.log
find gpurun_out/prof_prefill -name "*kernel_stats.csv" | head -1 | xargs head -12 | cut -c1-160
