O=gpurun_out/fin2; mkdir -p $O; rm -f $O/pmc_errors.log
R=$PWD
cd /tmp; export TMPDIR=/tmp
for t in b1 b32; do
  [ $t = b1 ] && BA="--batch 1" || BA="--batch 32"
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_${t}_$c
    timeout 600 rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_${t}_$c -- python $R/bench.py $BA --steps 64 --warmup 16 --gen-tokens 0 --cpu-steps 0 --no-extras > /tmp/pmc_${t}_$c.log 2>&1
    db=$(find /tmp/pmc_${t}_$c -name '*.db' | head -1)
    [ -n "$db" ] && python $R/tools/rocpd_pmc.py $db $c 60 $R/$O/pmc_${t}_$c.json > /dev/null 2>> $R/$O/pmc_errors.log || { echo "no db for $t $c" >> $R/$O/pmc_errors.log; tail -5 /tmp/pmc_${t}_$c.log >> $R/$O/pmc_errors.log; }
  done
done
cd $R; cat $O/pmc_errors.log; ls $O | grep pmc
