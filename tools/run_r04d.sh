set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04d; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -25 $O/pytest_gpu.log
timeout 300 python tools/ab_options.py fp32 "down_splitk_rows=0,17" --batches 17 20 24 28 32 --rounds 3 > $O/ab_down_sk.jsonl 2> $O/ab.err
timeout 300 python tools/ab_options.py fp16 "down_splitk_rows=0,9" --batches 9 12 16 24 32 --rounds 3 >> $O/ab_down_sk.jsonl 2>> $O/ab.err
cat $O/ab_down_sk.jsonl; tail -3 $O/ab.err
timeout 300 python tools/ab_options.py fp32 "persistent_rows=0,3" --batches 1 2 3 --rounds 3 > $O/ab_persist_rows.jsonl 2>> $O/ab.err; cat $O/ab_persist_rows.jsonl
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_b1_fp32_steps20.json 2> $O/bench.err; cut -c1-300 $O/bench_b1_fp32_steps20.json; tail -3 $O/bench.err
