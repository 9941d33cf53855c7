# Round 4, GPU call A: the whole -m gpu suite (incl. the 2-process real-engine sharding test), smoke, the driver's bench line (with the new
# sharded_request leg), the valu_rows A/B, PMC traffic at the TIMED WINDOW's context (a 293-token prompt puts steps 8..24 at mean context 309 --
# the full-length generation overruns rocprofv3's counter collection), and MFMA-busy for the parity path's split GEMMs.
set -x
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04a
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -15 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_b1_fp32_steps20.json 2> $O/bench.err; cut -c1-300 $O/bench_b1_fp32_steps20.json; tail -3 $O/bench.err
timeout 300 python tools/ab_options.py fp32 "valu_rows=0,4" --batches 1 2 3 4 --rounds 3 > $O/ab_valu_rows.jsonl 2> $O/ab_valu.err; cat $O/ab_valu_rows.jsonl; tail -2 $O/ab_valu.err
cd /tmp
for t in b1 b32; do
  [ $t = b1 ] && BA="--batch 1" || BA="--batch 32"
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_${t}_$c
    timeout 300 rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_${t}_$c -- python $R/bench.py $BA --prompt 293 --steps 16 --warmup 8 --gen-tokens 0 --cpu-steps 0 --no-extras > /tmp/pmc_${t}_$c.log 2>&1
    db=$(find /tmp/pmc_${t}_$c -name '*.db' | head -1)
    [ -n "$db" ] && python $R/tools/rocpd_pmc.py $db $c 14 $O/pmc_${t}_$c.json > /dev/null 2>> $O/pmc_errors.log || { echo "no db for $t $c" >> $O/pmc_errors.log; tail -3 /tmp/pmc_${t}_$c.log >> $O/pmc_errors.log; }
    grep '"metric"' /tmp/pmc_${t}_$c.log | cut -c1-600 > $O/pmc_${t}_${c}_bench.json
  done
done
# MFMA busy + LDS conflicts of the parity path's MFMA-bound kernels: prompt pass 32 x 512 (prefill_split_gemm / attn_prefill_split) and the vocoder (cnx_gemm)
for w in pre voc; do
  [ $w = pre ] && CMD="python $R/tools/prefill_probe.py 32 512 fp32" || CMD="python $R/tools/voc_batch_probe.py"
  rm -rf /tmp/pmc_mfma_$w /tmp/pmc_lds_$w
  timeout 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES -d /tmp/pmc_mfma_$w -- $CMD > /tmp/pmc_mfma_$w.log 2>&1
  db=$(find /tmp/pmc_mfma_$w -name '*.db' | head -1)
  [ -n "$db" ] && python $R/tools/rocpd_mfma.py $db $O/pmc_mfma_$w.json > /dev/null 2>> $O/pmc_errors.log || echo "no db mfma $w" >> $O/pmc_errors.log
  timeout 200 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES -d /tmp/pmc_lds_$w -- $CMD > /tmp/pmc_lds_$w.log 2>&1
  db=$(find /tmp/pmc_lds_$w -name '*.db' | head -1)
  [ -n "$db" ] && python $R/tools/rocpd_counters.py $db $O/pmc_lds_$w.json > /dev/null 2>> $O/pmc_errors.log || echo "no db lds $w" >> $O/pmc_errors.log
done
cd $R
ls -la $O; cat $O/pmc_errors.log 2>/dev/null | tail -5
