mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_gpu_sampler.py tests/test_gpu_gpt.py tests/test_gpu_properties.py tests/test_gpu_fp16_parity.py -m gpu -q 2>&1 | tail -30 > gpurun_out/pytest_r2d.log)
for B in 8 16 32 64; do
  for XH in 1 0; do
    CTTS_XH=$XH timeout 200 python bench.py --steps 128 --batch $B --no-extras --cpu-steps 0 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('B=$B XH=$XH', d['value'], d['ms_per_step'], d['roofline']['frac'])
" >> gpurun_out/xh_ab.log
  done
done
tail -8 gpurun_out/pytest_r2d.log; cat gpurun_out/xh_ab.log
