# Round 6, final tree (after the vocoder's small-batch block shape): the GPU suite, smoke, the bench lines.   -> gpurun_out/fin_r06g
set -x
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/fin_r06g
mkdir -p $O
cd $R
timeout 1800 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -3 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 900 python bench.py > $O/bench_b1_fp32.json 2> $O/bench_b1_fp32.err; cut -c1-200 $O/bench_b1_fp32.json
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_b1_fp32_steps20.json 2>/dev/null; cut -c1-200 $O/bench_b1_fp32_steps20.json
timeout 600 python bench.py --steps 20 --warmup 5 --force-pg > $O/bench_b1_fp32_steps20_force_pg.json 2>/dev/null; cut -c1-200 $O/bench_b1_fp32_steps20_force_pg.json
timeout 300 python bench.py --batch 32 --steps 256 --cpu-steps 0 --no-extras > $O/bench_b32_fp32.json 2>/dev/null; cut -c1-160 $O/bench_b32_fp32.json
