mkdir -p gpurun_out; rm -f gpurun_out/prefill_ab2.log
(timeout 1500 python -m pytest tests/test_gpu_gpt.py tests/test_gpu_properties.py tests/test_gpu_fp16_parity.py tests/test_gpu_pipeline.py -m gpu -q 2>&1 | tail -30 > gpurun_out/pytest_r2g.log)
for cfg in "32 512" "1 512" "4 300" "32 96" "8 2000"; do
  set -- $cfg
  for env in "X=1" "CTTS_PREFILL_ATTN=0" "CTTS_PREFILL_GEMM=0" "CTTS_PREFILL_GEMM=0 CTTS_PREFILL_ATTN=0"; do
    echo "== $env" >> gpurun_out/prefill_ab2.log
    env $env timeout 120 python tools/prefill_probe.py $1 $2 2>&1 | grep "prompt pass" | tail -1 >> gpurun_out/prefill_ab2.log
  done
done
ROOTD=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOTD/gpurun_out/prof_prefill3 -- python $ROOTD/tools/prefill_probe.py 32 512 > /dev/null 2>&1
cd $ROOTD
tail -12 gpurun_out/pytest_r2g.log; cat gpurun_out/prefill_ab2.log
find gpurun_out/prof_prefill3 -name "*kernel_stats.csv" | head -1 | xargs head -9 | cut -c1-150
