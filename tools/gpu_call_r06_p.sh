#!/bin/bash
# round 6, call P: phase marks of the persistent launch with / without per-utterance adapters
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06p; mkdir -p $O
export TMPDIR=/tmp
timeout 400 python tools/persist_probe.py --skip-layer --skip-times --skip-checks 2> $O/marks_plain.err | grep phase_marks > $O/marks_plain.jsonl
timeout 400 python tools/persist_probe.py --skip-layer --skip-times --skip-checks --adapters 2> $O/marks_adapters.err | grep phase_marks > $O/marks_adapters.jsonl
python - <<'PY'
import json
for f in ("plain","adapters"):
    for l in open(f"gpurun_out/r06p/marks_{f}.jsonl"):
        j=json.loads(l); print(f, j["B"], j["edges_us"])
PY
tail -2 $O/marks_adapters.err
