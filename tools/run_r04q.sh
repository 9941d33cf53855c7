#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r04q
timeout 600 python tools/lora_probe.py --modes none,fold,notake,zeros > gpurun_out/r04q/probe.log 2>&1; cat gpurun_out/r04q/probe.log | tail -4
timeout 900 python -m pytest tests/test_gpu_pipeline.py -m gpu -x -q -k "lora" > gpurun_out/r04q/tests.log 2>&1; tail -3 gpurun_out/r04q/tests.log
