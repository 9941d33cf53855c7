set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04l; mkdir -p $O
for P in 200 400 600 900; do
timeout 200 python tools/ab_options.py fp32 "persistent_rows=0,4" --batches 1 --rounds 2 --prompt $P --gen-tokens 0 --steps 48 >> $O/ab_ctx.jsonl 2>> $O/ab.err
done
timeout 200 python tools/ab_options.py fp32 "persistent_rows=0,4" --batches 4 --rounds 2 --prompt 600 --gen-tokens 0 --steps 48 >> $O/ab_ctx.jsonl 2>> $O/ab.err
cat $O/ab_ctx.jsonl; tail -3 $O/ab.err
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -6 $O/pytest_gpu.log
