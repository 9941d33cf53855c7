mkdir -p gpurun_out; rm -f gpurun_out/r2h.log
(timeout 900 python -m pytest tests/test_gpu_gpt.py -m gpu -q 2>&1 | tail -5 > gpurun_out/pytest_r2h.log)
for cfg in "32 512" "8 2000" "16 512"; do
  set -- $cfg
  timeout 120 python tools/prefill_probe.py $1 $2 2>&1 | grep "prompt pass" | tail -1 >> gpurun_out/r2h.log
done
for B in 1 2 4; do for SR in 4 0; do
  CTTS_SPLIT_ROWS=$SR timeout 200 python bench.py --steps 128 --batch $B --no-extras --cpu-steps 0 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('B=$B split_rows=$SR', d['value'], d['ms_per_step'])
" >> gpurun_out/r2h.log
done; done
ROOTD=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOTD/gpurun_out/prof_prefill4 -- python $ROOTD/tools/prefill_probe.py 32 512 > /dev/null 2>&1
cd $ROOTD
tail -3 gpurun_out/pytest_r2h.log; cat gpurun_out/r2h.log
find gpurun_out/prof_prefill4 -name "*kernel_stats.csv" | head -1 | xargs head -8 | cut -c1-150
