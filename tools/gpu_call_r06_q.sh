#!/bin/bash
# round 6, call Q: adapters inside the persistent launch, second version (masked polls, polls before the barrier, scale on u): tests, marks, step times
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06q; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_persistent.py -q -m gpu -x -k "adapters" > $O/tests.log 2>&1; echo "tests rc=$?" > $O/summary.txt
timeout 400 python tools/persist_probe.py --skip-layer --skip-times --skip-checks --adapters 2> $O/marks_adapters.err | grep phase_marks > $O/marks_adapters.jsonl
for B in 2 4 5 8; do
  timeout 300 python tools/lora_probe.py --rows $B --tokens 256 --modes none,fold,persist >> $O/lora_probe.jsonl 2>> $O/lora_probe.err
done
python - <<'PY'
import json
for l in open("gpurun_out/r06q/marks_adapters.jsonl"):
    j=json.loads(l); print(j["B"], j["edges_us"])
PY
tail -n 5 $O/tests.log; cat $O/summary.txt; cat $O/lora_probe.jsonl
