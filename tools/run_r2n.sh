mkdir -p gpurun_out; rm -f gpurun_out/r2n.log
(timeout 900 python -m pytest tests/test_gpu_gpt.py tests/test_gpu_fp16_parity.py -m gpu -q 2>&1 | tail -4 >> gpurun_out/r2n.log)
for cfg in "32 512" "8 2000" "4 512"; do
  set -- $cfg
  timeout 120 python tools/prefill_probe.py $1 $2 2>&1 | grep "prompt pass" | tail -1 >> gpurun_out/r2n.log
done
timeout 300 python bench.py --steps 64 --warmup 8 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['ms_per_step']); print({k:(v if isinstance(v,str) else (v['tokens_per_s'], v['ms_per_step'], v['prefill_plus_first_sample_ms'])) for k,v in d['extra'].items()})
" >> gpurun_out/r2n.log
ROOTD=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOTD/gpurun_out/prof_prefill7 -- python $ROOTD/tools/prefill_probe.py 32 512 > /dev/null 2>&1
cd $ROOTD
cat gpurun_out/r2n.log
find gpurun_out/prof_prefill7 -name "*kernel_stats.csv" | head -1 | xargs head -7 | cut -c1-150
