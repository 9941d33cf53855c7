mkdir -p gpurun_out; rm -f gpurun_out/r2r.log
(timeout 600 python -m pytest tests/test_gpu_pipeline.py -m gpu -q -k "lora" 2>&1 | tail -4 >> gpurun_out/r2r.log)
timeout 400 python bench.py --steps 64 --warmup 8 --cpu-steps 0 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['ms_per_step']); print({k:(v if isinstance(v,str) else (v['tokens_per_s'], v['ms_per_step'], v['prefill_plus_first_sample_ms'])) for k,v in d['extra'].items()})
" >> gpurun_out/r2r.log
cat gpurun_out/r2r.log
