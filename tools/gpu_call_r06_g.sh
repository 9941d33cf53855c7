#!/bin/bash
# round 6, call G: 6-8 persistent rows with buffer-load sweeps; buffer-load sweeps at 1-5 rows (variant build); tests
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06g; mkdir -p $O
export TMPDIR=/tmp
AB="timeout 600 python tools/ab_options.py"
timeout 900 python -m pytest tests/test_gpu_persistent.py -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?" > $O/summary.txt
$AB fp32 "persistent_rows=5,8" --batches 6 7 8 --rounds 3 > $O/ab_p8_fp32.jsonl 2> $O/ab_p8_fp32.err
$AB fp32 "persistent_rows=5,8" --batches 6 8 --rounds 3 --prompt 300 > $O/ab_p8_fp32_p300.jsonl 2> $O/ab_p8_fp32_p300.err
$AB fp16 "persistent_rows=5,8" --batches 6 7 8 --rounds 3 > $O/ab_p8_fp16.jsonl 2> $O/ab_p8_fp16.err
python tools/tb_curve.py fp32 1 2 3 4 5 > $O/tb_main.jsonl 2> $O/tb_main.err
CTTS_HIP_LIB=$PWD/chatttsplus_amd/_lib/libctts_hip_bufall.so python tools/tb_curve.py fp32 1 2 3 4 5 > $O/tb_bufall.jsonl 2> $O/tb_bufall.err
python tools/tb_curve.py fp32 1 2 3 4 5 > $O/tb_main2.jsonl 2> $O/tb_main2.err
CTTS_HIP_LIB=$PWD/chatttsplus_amd/_lib/libctts_hip_bufall.so python tools/tb_curve.py fp32 1 2 3 4 5 > $O/tb_bufall2.jsonl 2> $O/tb_bufall2.err
tail -n 4 $O/tests.log; cat $O/summary.txt $O/ab_*.jsonl $O/tb_*.jsonl
