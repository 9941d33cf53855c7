set -x
export TMPDIR=/tmp
O=gpurun_out/r03a
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -30 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; cut -c1-600 $O/bench_default.json; tail -5 $O/bench_default.err
