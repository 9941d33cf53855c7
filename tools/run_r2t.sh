O=gpurun_out/fin2; mkdir -p $O
timeout 900 python bench.py > $O/bench_b1_fp16.json 2> $O/bench_b1_fp16.err; cut -c1-200 $O/bench_b1_fp16.json
timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench_b1_fp16_driver_args.json 2>/dev/null; cut -c1-200 $O/bench_b1_fp16_driver_args.json
timeout 300 python bench.py --batch 32 --steps 256 --cpu-steps 0 --no-extras > $O/bench_b32_fp16.json 2>/dev/null; cut -c1-200 $O/bench_b32_fp16.json
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --cpu-steps 0 | cut -c1-120
(timeout 600 python -m pytest tests/test_gpu_gpt.py tests/test_gpu_pipeline.py -m gpu -q 2>&1 | tail -3)
