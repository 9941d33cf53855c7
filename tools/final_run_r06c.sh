# Round 6, third collection (after the prompt pass's split GEMMs got their 256-row counter-phased kernel): suite + smoke, the bench lines, the prompt pass per block
# shape (kernel stats, end to end at several sizes), MFMA-busy / LDS counters of the prompt pass.   -> gpurun_out/fin_r06c (tools/collect_profiles_r06.py c)
set -x
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/fin_r06c
mkdir -p $O
cd $R
timeout 1800 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -3 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 900 python bench.py > $O/bench_b1_fp32.json 2> $O/bench_b1_fp32.err; cut -c1-200 $O/bench_b1_fp32.json
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_b1_fp32_steps20.json 2>/dev/null; cut -c1-200 $O/bench_b1_fp32_steps20.json
timeout 600 python bench.py --steps 20 --warmup 5 --force-pg > $O/bench_b1_fp32_steps20_force_pg.json 2>/dev/null; cut -c1-200 $O/bench_b1_fp32_steps20_force_pg.json
timeout 300 python bench.py --batch 32 --steps 256 --cpu-steps 0 --no-extras > $O/bench_b32_fp32.json 2>/dev/null; cut -c1-160 $O/bench_b32_fp32.json
for i in 1 2 3; do timeout 120 python tools/prefill_probe.py 32 512 fp32 2>/dev/null | grep "prompt pass" | cut -c1-50; done > $O/prefill_32x512_fp32.log
bash tools/prefill_pp_stats.sh fin_r06c 32 512 "0 1" > $O/prefill_pp_stats.txt 2>&1
bash tools/prefill_shapes_ab.sh fin_r06c_e2e "32 512" "16 512" "8 512" "32 128" "8 256" "32 48" > $O/prefill_shapes_e2e.txt 2>&1
cp $R/gpurun_out/fin_r06c_e2e/e2e.log $O/prefill_shapes_e2e.log
bash tools/pmc_prefill_attention.sh > $O/pmc_pa.txt 2>&1
cp $R/gpurun_out/pmc_pa/mfma.json $O/pmc_mfma_split.json; cp $R/gpurun_out/pmc_pa/lds.json $O/pmc_lds_split.json
cd $R; ls $O; tail -8 $O/prefill_shapes_e2e.txt
