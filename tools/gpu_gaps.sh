# kernel-to-kernel gaps of the batch-1 decode loop (one gpurun call): rocprofv3 --kernel-trace of a short bench run -> tools/trace_gaps.py
set -x
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/gaps
mkdir -p $O
cd /tmp
for B in ${BATCHES:-1}; do
  rm -rf /tmp/gp_$B
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/gp_$B -- python $R/bench.py --batch $B --steps 128 --warmup 16 --cpu-steps 0 --no-extras $EXTRA > /tmp/gp_$B.log 2>&1
  f=$(find /tmp/gp_$B -name '*kernel_trace.csv' | head -1)
  grep '"metric"' /tmp/gp_$B.log | cut -c1-200
  [ -n "$f" ] && python $R/tools/trace_gaps.py $f --main ${MAIN:-persist_layer_kernel} > $O/gaps_b$B.json
  cat $O/gaps_b$B.json
done
