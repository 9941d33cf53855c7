# Round-2 evidence refresh after the prompt-GEMM changes (k loop, epilogues, block shapes): the whole GPU suite, smoke, prompt-pass clocks,
# per-kernel stats and MFMA-busy counters of the 32 x 512 prompt pass, then the default bench line.  Small summaries only, into gpurun_out/fin3.
set -x
export TMPDIR=/tmp
O=gpurun_out/fin3
mkdir -p $O
R=$GRAFT_REPO_ROOT
timeout 400 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -3 $O/pytest_gpu.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
rm -f $O/prefill.log
for cfg in "1 48" "1 512" "4 512" "32 96" "32 512" "8 2000"; do timeout 60 python tools/prefill_probe.py $cfg 2>&1 | grep "prompt pass" | tail -1 >> $O/prefill.log; done
for cfg in "32 512" "8 2000"; do CTTS_PREFILL_GEMM=0 CTTS_PREFILL_ATTN=0 timeout 60 python tools/prefill_probe.py $cfg 2>&1 | grep "prompt pass" | tail -1 | sed 's/$/  (round-1 kernels: CTTS_PREFILL_GEMM=0 CTTS_PREFILL_ATTN=0)/' >> $O/prefill.log; done
cat $O/prefill.log
cd /tmp
timeout 100 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_pre -- python $R/tools/prefill_probe.py 32 512 > /tmp/prof_pre.log 2>&1
f=$(find /tmp/prof_pre -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $R/$O/pre_kernel_stats.csv
rm -rf /tmp/pmc_mfma_prefill
timeout 100 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES -d /tmp/pmc_mfma_prefill -- python $R/tools/prefill_probe.py 32 512 > /tmp/pmc_mfma_prefill.log 2>&1
db=$(find /tmp/pmc_mfma_prefill -name '*.db' | head -1)
[ -n "$db" ] && python $R/tools/rocpd_mfma.py $db $R/$O/pmc_mfma_prefill.json 2>> $R/$O/pmc_errors.log || echo "no db for mfma prefill" >> $R/$O/pmc_errors.log
cd $R
timeout 200 python bench.py > $O/bench_b1_fp16.json 2> $O/bench_b1_fp16.err; cut -c1-300 $O/bench_b1_fp16.json
ls -la $O; head -8 $O/pre_kernel_stats.csv | cut -c1-160
