set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04m; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_persistent.py -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -12 $O/pytest_gpu.log
for P in 400 600 900 1400 1900; do
timeout 200 python tools/ab_options.py fp32 "persistent_splits=1,0" --batches 1 --rounds 2 --prompt $P --gen-tokens 0 --steps 48 >> $O/ab_ctx.jsonl 2>> $O/ab.err
done
timeout 200 python tools/ab_options.py fp32 "persistent_rows=0,4" --batches 1 --rounds 2 --prompt 1900 --gen-tokens 0 --steps 48 >> $O/ab_ctx.jsonl 2>> $O/ab.err
timeout 200 python tools/ab_options.py fp32 "persistent_splits=1,0" --batches 2 --rounds 2 --prompt 700 --gen-tokens 0 --steps 48 >> $O/ab_ctx.jsonl 2>> $O/ab.err
cat $O/ab_ctx.jsonl; tail -3 $O/ab.err
