mkdir -p gpurun_out; rm -f gpurun_out/r2v.log
(timeout 900 python -m pytest tests/test_gpu_gpt.py tests/test_gpu_fp16_parity.py tests/test_gpu_properties.py -m gpu -q 2>&1 | tail -5 >> gpurun_out/r2v.log)
for B in 8 32 128; do for X in 0 1; do
  env $( [ $X = 1 ] && echo CTTS_NO_XH_HEADS=1 || echo X=0 ) timeout 200 python bench.py --steps 128 --batch $B --no-extras --cpu-steps 0 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('B=$B no_xh_heads=$X', d['value'], d['ms_per_step'])
" >> gpurun_out/r2v.log
done; done
cat gpurun_out/r2v.log
