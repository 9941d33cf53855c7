#!/bin/bash
# round 6, call H: the whole GPU suite on the new defaults + the driver's bench command
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06h; mkdir -p $O
export TMPDIR=/tmp
( time timeout 3000 python -m pytest tests/ -q -m gpu -x > $O/pytest_gpu.log 2>&1 ) 2> $O/pytest_gpu.time; echo "tests rc=$?" > $O/summary.txt
tail -n 5 $O/pytest_gpu.log; cat $O/pytest_gpu.time
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_steps20.json 2> $O/bench_steps20.err ) 2> $O/bench_steps20.time
tail -3 $O/bench_steps20.time; cut -c1-1500 $O/bench_steps20.json; tail -3 $O/bench_steps20.err
