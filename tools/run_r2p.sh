mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_gpu_pipeline.py -m gpu -q 2>&1 | tail -15 > gpurun_out/r2p.log)
timeout 200 python tools/gen_wall.py --text --steps 128 --noise device 2>&1 | tail -1 >> gpurun_out/r2p.log
timeout 200 python tools/gen_wall.py --text --steps 128 --noise device --batch 8 2>&1 | tail -1 >> gpurun_out/r2p.log
cat gpurun_out/r2p.log
