set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04f; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_persistent.py tests/test_gpu_gpt.py -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -25 $O/pytest_gpu.log
