# One persistent-MFMA probe call: bash tools/gpu_pm.sh <tag> <pm_probe args...>
set -x
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
T=$1; shift
O=$R/gpurun_out/pm_$T
mkdir -p $O
cd $R
timeout 500 python tools/pm_probe.py "$@" > $O/probe.jsonl 2> $O/probe.err; echo "probe rc=$?"; cat $O/probe.jsonl; tail -3 $O/probe.err
