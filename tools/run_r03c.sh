set -x
export TMPDIR=/tmp
O=gpurun_out/r03c
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -25 $O/pytest_gpu.log
timeout 600 python bench.py --steps 20 --warmup 5 --cpu-steps 0 > $O/bench_default.json 2> $O/bench_default.err; python -c "
import json;d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'],d['roofline']['frac']);print(d['extra'].get('batch32_ragged_targets'));print(d['extra'].get('error'))"
