#!/bin/bash
# Per-kernel durations of the parity prompt pass with the 128 x 128 split GEMM (prefill_pp_blocks=0), the counter-phased kernel forced to 4 / 3 n tiles per wave (-4 / -3)
# and the default policy (1): rocprofv3 --kernel-trace --stats, one run each -> gpurun_out/$1/stats_pp_<v>.csv   (bash tools/prefill_pp_stats.sh OUTDIR [B P] [values])
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/${1:-pp_stats}
B=${2:-32}; P=${3:-512}; V=${4:-"0 -4 -3 1"}
mkdir -p $O
cd /tmp
for v in $V; do
  rm -rf /tmp/pps_$v
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pps_$v -- python $R/tools/prefill_probe.py $B $P fp32 prefill_pp_blocks=$v > /tmp/pps_$v.log 2>&1
  f=$(find /tmp/pps_$v -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && head -12 $f | cut -c1-200 > $O/stats_${B}x${P}_pp_$v.csv || tail -5 /tmp/pps_$v.log > $O/stats_${B}x${P}_pp_$v.csv
  echo "== B=$B P=$P prefill_pp_blocks=$v"; grep "gemm\|attn_prefill" $O/stats_${B}x${P}_pp_$v.csv | cut -d, -f1-4 | cut -c1-150
done
