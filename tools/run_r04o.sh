#!/bin/bash
# stream + continuous on the GPU, the give-up test, and a bench sanity line
cd /root/repo; mkdir -p gpurun_out/r04o
timeout 900 python -m pytest tests/test_gpu_persistent.py tests/test_gpu_pipeline.py -m gpu -x -q > gpurun_out/r04o/tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r04o/tests.log
tail -5 gpurun_out/r04o/tests.log


