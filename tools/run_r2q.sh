mkdir -p gpurun_out; rm -f gpurun_out/r2q.log
(timeout 900 python -m pytest tests/test_gpu_gpt.py tests/test_gpu_properties.py -m gpu -q 2>&1 | tail -6 >> gpurun_out/r2q.log)
for cfg in "32 512" "8 2000" "16 512" "1 512"; do
  set -- $cfg
  timeout 120 python tools/prefill_probe.py $1 $2 2>&1 | grep "prompt pass" | tail -1 >> gpurun_out/r2q.log
done
ROOTD=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOTD/gpurun_out/prof_prefill8 -- python $ROOTD/tools/prefill_probe.py 32 512 > /dev/null 2>&1
cd $ROOTD
cat gpurun_out/r2q.log
find gpurun_out/prof_prefill8 -name "*kernel_stats.csv" | head -1 | xargs head -7 | cut -c1-150
