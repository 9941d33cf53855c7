mkdir -p gpurun_out; O=gpurun_out/fin2; rm -f gpurun_out/r2k.log $O/pmc_errors.log
for B in 32 64; do for UN in 4 8; do
  CTTS_ATTN_UN=$UN timeout 200 python bench.py --steps 128 --batch $B --no-extras --cpu-steps 0 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('B=$B attn UN=$UN', d['value'], d['ms_per_step'])
" >> gpurun_out/r2k.log
done; done
(timeout 600 python -m pytest tests/test_gpu_gpt.py -m gpu -q 2>&1 | tail -3 >> gpurun_out/r2k.log)
R=$PWD
cd /tmp; export TMPDIR=/tmp
for t in b1 b32; do
  [ $t = b1 ] && BA="--batch 1" || BA="--batch 32"
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_${t}_$c
    timeout 600 rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_${t}_$c -- python $R/bench.py $BA --steps 64 --warmup 16 --gen-tokens 0 --cpu-steps 0 --no-extras > /tmp/pmc_${t}_$c.log 2>&1
    echo "rc=$? $t $c" >> $R/gpurun_out/r2k.log
    db=$(find /tmp/pmc_${t}_$c -name '*.db' | head -1)
    [ -n "$db" ] && python $R/tools/rocpd_pmc.py $db $c 60 $R/$O/pmc_${t}_$c.json > /dev/null 2>> $R/$O/pmc_errors.log || { echo "no db for $t $c" >> $R/$O/pmc_errors.log; tail -5 /tmp/pmc_${t}_$c.log >> $R/$O/pmc_errors.log; }
  done
done
cd $R; cat gpurun_out/r2k.log; cat $O/pmc_errors.log; ls $O | grep pmc
