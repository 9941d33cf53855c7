#!/bin/bash
# folded per-utterance LoRA: bench legs + both LoRA-related test files
cd /root/repo; mkdir -p gpurun_out/r04p
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r04p/bench.log 2>&1
grep '^{' gpurun_out/r04p/bench.log | python -c "
import sys,json; d=json.loads(sys.stdin.read()); e=d['extra']
print(d['value'], d['ms_per_step'])
for k in ('batch32','batch32_lora_merged','batch32_lora_per_utterance'): print(k, e[k].get('ms_per_step'))"
timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_gpt.py -m gpu -x -q > gpurun_out/r04p/tests.log 2>&1; tail -3 gpurun_out/r04p/tests.log
