#!/bin/bash
# round 6, call I: the whole GPU suite (no -x)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06i; mkdir -p $O
export TMPDIR=/tmp
( time timeout 3300 python -m pytest tests/ -q -m gpu > $O/pytest_gpu.log 2>&1 ) 2> $O/pytest_gpu.time; echo "tests rc=$?" > $O/summary.txt
tail -n 15 $O/pytest_gpu.log; cat $O/pytest_gpu.time
