set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04g; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_persistent.py -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -8 $O/pytest_gpu.log
timeout 300 python tools/ab_options.py fp32 "persistent_schedule=1,3" --batches 1 2 3 --rounds 3 --fixed persistent_rows=3 > $O/ab.jsonl 2> $O/ab.err
cat $O/ab.jsonl; tail -3 $O/ab.err
