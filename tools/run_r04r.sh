#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r04r
timeout 900 python -m pytest tests/test_gpu_persistent.py -m gpu -x -q -k "o_proj_launch" > gpurun_out/r04r/tests.log 2>&1; tail -4 gpurun_out/r04r/tests.log
timeout 600 python tools/ab_options.py fp32 attn_oproj_fuse=0,1 --batches 24 32 > gpurun_out/r04r/ab.jsonl 2> gpurun_out/r04r/ab.err
timeout 600 python tools/ab_options.py fp32 attn_oproj_nap=1,4 --fixed attn_oproj_nap0=8 --batches 32 >> gpurun_out/r04r/ab.jsonl 2>> gpurun_out/r04r/ab.err
timeout 600 python tools/ab_options.py fp32 attn_oproj_nap0=0,4,16 --batches 32 >> gpurun_out/r04r/ab.jsonl 2>> gpurun_out/r04r/ab.err
tail -6 gpurun_out/r04r/ab.jsonl | cut -c1-400; tail -3 gpurun_out/r04r/ab.err
