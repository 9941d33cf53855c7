set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04k; mkdir -p $O
timeout 300 python tools/ab_options.py fp32 "persistent_delay=8,11,14,17,20" --batches 1 --rounds 3 > $O/ab.jsonl 2> $O/ab.err
timeout 300 python tools/ab_options.py fp32 "persistent_delay_act=8,11,14,17,20,26" --batches 1 --rounds 3 >> $O/ab.jsonl 2>> $O/ab.err
timeout 300 python tools/ab_options.py fp32 "persistent_delay_x=8,11,14,17,20" --batches 1 --rounds 3 >> $O/ab.jsonl 2>> $O/ab.err
timeout 300 python tools/ab_options.py fp32 "persistent_nap_qkv=1,4,16" --batches 1 --rounds 3 >> $O/ab.jsonl 2>> $O/ab.err
timeout 300 python tools/ab_options.py fp32 "persistent_pace=2,3,4,6" --batches 1 --rounds 3 >> $O/ab.jsonl 2>> $O/ab.err
timeout 300 python tools/ab_options.py fp32 "persistent_delay_att=0,10,20,30" --batches 1 --rounds 3 >> $O/ab.jsonl 2>> $O/ab.err
cat $O/ab.jsonl; tail -3 $O/ab.err
