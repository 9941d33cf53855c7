set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04h; mkdir -p $O
timeout 300 python tools/ab_options.py fp32 "persistent_pace=0,2,3,4,6,8" --batches 1 --rounds 3 --fixed persistent_rows=4,persistent_schedule=3 > $O/ab.jsonl 2> $O/ab.err
timeout 300 python tools/ab_options.py fp32 "persistent_poll=0,1" --batches 1 2 3 4 --rounds 3 --fixed persistent_rows=4,persistent_schedule=3 >> $O/ab.jsonl 2>> $O/ab.err
timeout 300 python tools/ab_options.py fp32 "persistent_rows=0,4" --batches 4 --rounds 3 --fixed persistent_schedule=3 >> $O/ab.jsonl 2>> $O/ab.err
timeout 300 python tools/ab_options.py fp32 "persistent_pace=2,4,8" --batches 2 4 --rounds 3 --fixed persistent_rows=4,persistent_schedule=3 >> $O/ab.jsonl 2>> $O/ab.err
cat $O/ab.jsonl; tail -3 $O/ab.err
