set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04j; mkdir -p $O
timeout 300 python tools/ab_options.py fp32 "persistent_delay=6,10,14,18,24,32" --batches 1 --rounds 3 > $O/ab.jsonl 2> $O/ab.err
timeout 300 python tools/ab_options.py fp32 "persistent_nap=1,2,4,8" --batches 1 --rounds 3 --fixed persistent_delay=6 >> $O/ab.jsonl 2>> $O/ab.err
timeout 300 python tools/ab_options.py fp32 "persistent_delay=0,6,12,18" --batches 2 4 --rounds 3 >> $O/ab.jsonl 2>> $O/ab.err
cat $O/ab.jsonl; tail -3 $O/ab.err
