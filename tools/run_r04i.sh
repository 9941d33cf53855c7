set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04i; mkdir -p $O
timeout 300 python tools/ab_options.py fp32 "persistent_delay_att=0,6,10,14,18" --batches 1 --rounds 3 > $O/ab.jsonl 2> $O/ab.err
timeout 300 python tools/ab_options.py fp32 "persistent_delay=0,2,4,6" --batches 1 --rounds 3 >> $O/ab.jsonl 2>> $O/ab.err
timeout 300 python tools/ab_options.py fp32 "persistent_delay_att=0,10" --batches 2 4 --rounds 3 >> $O/ab.jsonl 2>> $O/ab.err
cat $O/ab.jsonl; tail -3 $O/ab.err
