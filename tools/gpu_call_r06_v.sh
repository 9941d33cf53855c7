#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06v; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python tools/two_stream_probe.py 16 2 128 > $O/two_stream.jsonl 2> $O/two_stream.err
timeout 600 python tools/two_stream_probe.py 8 4 128 >> $O/two_stream.jsonl 2>> $O/two_stream.err
timeout 600 python tools/two_stream_probe.py 32 2 128 >> $O/two_stream.jsonl 2>> $O/two_stream.err
cat $O/two_stream.jsonl; tail -5 $O/two_stream.err
