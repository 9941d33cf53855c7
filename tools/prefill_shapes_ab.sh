#!/bin/bash
# End-to-end prompt pass (begin + prefill + first sample) per block shape of the split GEMMs, interleaved on one engine per (B, P):
# bash tools/prefill_shapes_ab.sh OUTDIR "B P" ...   -> gpurun_out/OUTDIR/e2e.log + a min / median table
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/$1; shift
mkdir -p $O
: > $O/e2e.log
for bp in "$@"; do python $R/tools/prefill_probe.py $bp fp32 prefill_pp_blocks=0 prefill_pp_blocks=-4 prefill_pp_blocks=-3 prefill_pp_blocks=1 2>&1 | grep "prompt pass" >> $O/e2e.log; done
python - $O/e2e.log <<'PY'
import re, sys, collections
d = collections.defaultdict(list)
for l in open(sys.argv[1]):
    m = re.match(r"B=(\d+) P=(\d+) fp32 \{'prefill_pp_blocks': (-?\d+)\}: prompt pass ([\d.]+) ms", l)
    if m: d[(int(m[1]), int(m[2]), int(m[3]))].append(float(m[4]))
for k, v in d.items(): print(k, "min %.3f med %.3f ms" % (min(v), sorted(v)[len(v) // 2]))
PY
