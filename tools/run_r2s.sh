mkdir -p gpurun_out; rm -f gpurun_out/r2s.log
(timeout 900 python -m pytest tests/test_gpu_gpt.py tests/test_gpu_fp16_parity.py tests/test_gpu_properties.py -m gpu -q 2>&1 | tail -5 >> gpurun_out/r2s.log)
for B in 1 2 4; do for X in 1 2; do
  CTTS_XH=$X timeout 200 python bench.py --steps 256 --batch $B --no-extras --cpu-steps 0 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('B=$B CTTS_XH=$X', d['value'], d['ms_per_step'])
" >> gpurun_out/r2s.log
done; done
cat gpurun_out/r2s.log
