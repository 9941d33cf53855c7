# Round 5, call A: first contact of the persistent MFMA decode stack with the hardware + RCCL's first execution.
set -x
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r05_a
mkdir -p $O
cd $R
timeout 420 python tools/pm_probe.py --batches 8,16,32 --steps 16 --marks 32 > $O/pm_probe.jsonl 2> $O/pm_probe.err; echo "probe rc=$?"; cat $O/pm_probe.jsonl; tail -5 $O/pm_probe.err
timeout 420 python -m pytest tests/test_gpu_rccl.py -x -q > $O/rccl.log 2>&1; echo "rccl rc=$?"; tail -15 $O/rccl.log
