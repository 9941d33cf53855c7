#!/bin/bash
# round 6, call T: the 256-utterance request: longest-first vs arrival order; GPU timeline of one request (kernel classes, idle time)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06t; mkdir -p $O
export TMPDIR=/tmp
export CTTS_HIP_LIB=$PWD/chatttsplus_amd/_lib/libctts_hip_snap.so
timeout 600 python tools/request_probe.py > $O/request_probe_lpt.jsonl 2> $O/request_probe_lpt.err
CTTS_SCHEDULE=fifo timeout 600 python tools/request_probe.py > $O/request_probe_fifo.jsonl 2> $O/request_probe_fifo.err
cd /tmp
timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_req -- python $GRAFT_REPO_ROOT/tools/request_probe.py > /tmp/prof_req.log 2>&1
f=$(find /tmp/prof_req -name '*kernel_trace.csv' | head -1); ls -la $f
[ -n "$f" ] && python $GRAFT_REPO_ROOT/tools/request_timeline.py $f > $GRAFT_REPO_ROOT/$O/request_timeline.json 2> $GRAFT_REPO_ROOT/$O/request_timeline.err
cd $GRAFT_REPO_ROOT
cut -c1-400 $O/request_probe_lpt.jsonl $O/request_probe_fifo.jsonl; cat $O/request_timeline.json; tail -3 $O/request_timeline.err /tmp/prof_req.log
