mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_gpt.py tests/test_gpu_properties.py -m gpu -q 2>&1 | tail -25 > gpurun_out/r2o.log)
timeout 120 python tools/prefill_probe.py 32 512 2>&1 | grep "prompt pass" | tail -1 >> gpurun_out/r2o.log
cat gpurun_out/r2o.log
