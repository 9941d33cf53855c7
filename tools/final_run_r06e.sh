# Round 6, after the short-pass work (sliced down projection, 64 x 64 blocks, 65-row threshold): smoke, the bench lines, the prompt pass end to end at short and long sizes.
# (The GPU suite on this tree: 142 passed, call r06ax.)   -> gpurun_out/fin_r06e
set -x
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/fin_r06e
mkdir -p $O
cd $R
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 900 python bench.py > $O/bench_b1_fp32.json 2> $O/bench_b1_fp32.err; cut -c1-200 $O/bench_b1_fp32.json
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_b1_fp32_steps20.json 2>/dev/null; cut -c1-200 $O/bench_b1_fp32_steps20.json
timeout 600 python bench.py --steps 20 --warmup 5 --force-pg > $O/bench_b1_fp32_steps20_force_pg.json 2>/dev/null; cut -c1-200 $O/bench_b1_fp32_steps20_force_pg.json
timeout 300 python bench.py --batch 32 --steps 256 --cpu-steps 0 --no-extras > $O/bench_b32_fp32.json 2>/dev/null; cut -c1-160 $O/bench_b32_fp32.json
for bp in "1 96" "2 96" "1 300" "8 56" "32 48" "8 256" "32 128" "16 512" "32 512"; do timeout 120 python tools/prefill_probe.py $bp fp32 2>/dev/null | grep "prompt pass" | cut -c1-48 | tail -2; done > $O/prefill_e2e_final.log
cat $O/prefill_e2e_final.log
