set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04e; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -25 $O/pytest_gpu.log
timeout 300 python tools/ab_options.py fp32 "persistent_rows=0,3" --batches 1 2 3 --rounds 3 > $O/ab_persist_rows.jsonl 2>> $O/ab.err; cat $O/ab_persist_rows.jsonl
timeout 300 python tools/persist_probe.py --skip-layer > $O/persist_probe.jsonl 2> $O/persist_probe.err; tail -3 $O/persist_probe.jsonl | cut -c1-1500
