#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r04q
timeout 1200 python -m pytest tests/test_gpu_pipeline.py -m gpu -x -q -k "lora" > gpurun_out/r04q/tests.log 2>&1; tail -8 gpurun_out/r04q/tests.log
