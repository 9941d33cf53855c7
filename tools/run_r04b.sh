# Round 4, GPU call: the whole -m gpu suite with the persistent launch as the batch-1 default, smoke, the driver's bench line, and the
# persistent-launch A/Bs (schedule, layers per launch) in bench.py's window.
set -x
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04b
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -25 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_b1_fp32_steps20.json 2> $O/bench.err; cut -c1-400 $O/bench_b1_fp32_steps20.json; tail -3 $O/bench.err
timeout 300 python tools/ab_options.py fp32 "persistent_schedule=0,1" --batches 1 --rounds 3 --fixed persistent_rows=1 > $O/ab_persist.jsonl 2> $O/ab.err
timeout 300 python tools/ab_options.py fp32 "persistent_layers_per_launch=0,1,5" --batches 1 --rounds 3 --fixed persistent_rows=1 >> $O/ab_persist.jsonl 2>> $O/ab.err
timeout 300 python tools/ab_options.py fp32 "persistent_rows=0,1" --batches 1 --rounds 3 >> $O/ab_persist.jsonl 2>> $O/ab.err
cat $O/ab_persist.jsonl; tail -3 $O/ab.err
timeout 300 python tools/persist_probe.py --skip-layer > $O/persist_probe.jsonl 2> $O/persist_probe.err; tail -4 $O/persist_probe.jsonl | cut -c1-1500
