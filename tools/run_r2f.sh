mkdir -p gpurun_out
(timeout 1500 python -m pytest tests/test_gpu_gpt.py tests/test_gpu_properties.py tests/test_gpu_fp16_parity.py tests/test_gpu_vocoder.py -m gpu -q 2>&1 | tail -30 > gpurun_out/pytest_r2f.log)
for cfg in "32 512" "1 512" "4 300" "32 96"; do
  set -- $cfg
  for env in "X=1" "CTTS_PREFILL_GEMM=0" "CTTS_PREFILL_ATTN=0" "CTTS_PREFILL_GEMM=0 CTTS_PREFILL_ATTN=0"; do
    echo "== $env" >> gpurun_out/prefill_ab.log
    env $env timeout 120 python tools/prefill_probe.py $1 $2 2>&1 | grep "prompt pass" | tail -1 >> gpurun_out/prefill_ab.log
  done
done
for B in 8 16; do for W in 1 0; do
  CTTS_ATTN_WIDE=$W timeout 200 python bench.py --steps 128 --batch $B --no-extras --cpu-steps 0 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('B=$B wide=$W', d['value'], d['ms_per_step'])
" >> gpurun_out/prefill_ab.log
done; done
ROOTD=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOTD/gpurun_out/prof_prefill2 -- python $ROOTD/tools/prefill_probe.py 32 512 > /dev/null 2>&1
cd $ROOTD
tail -12 gpurun_out/pytest_r2f.log; cat gpurun_out/prefill_ab.log
find gpurun_out/prof_prefill2 -name "*kernel_stats.csv" | head -1 | xargs head -9 | cut -c1-150
