set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c; mkdir -p $O
timeout 400 python tools/persist_probe.py > $O/persist_probe.jsonl 2> $O/persist_probe.err; echo rc=$?
cut -c1-1200 $O/persist_probe.jsonl; tail -3 $O/persist_probe.err
timeout 300 python tools/ab_options.py fp32 "persistent_schedule=1,2" --batches 1 2 --rounds 3 --fixed persistent_rows=4 > $O/ab.jsonl 2> $O/ab.err
timeout 300 python tools/ab_options.py fp32 "persistent_poll=0,1" --batches 1 2 4 --rounds 3 --fixed persistent_rows=4 >> $O/ab.jsonl 2>> $O/ab.err
cat $O/ab.jsonl; tail -3 $O/ab.err
