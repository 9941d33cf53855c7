mkdir -p gpurun_out; R=$PWD
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/pmc_sq1 /tmp/pmc_sq2
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES -d /tmp/pmc_sq1 -- python $R/tools/prefill_probe.py 32 512 > /tmp/sq1.log 2>&1
db=$(find /tmp/pmc_sq1 -name '*.db' | head -1); python $R/tools/rocpd_counters.py $db $R/gpurun_out/pmc_sq1.json prefill > $R/gpurun_out/pmc_sq1.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS -d /tmp/pmc_sq2 -- python $R/tools/prefill_probe.py 32 512 > /tmp/sq2.log 2>&1
db=$(find /tmp/pmc_sq2 -name '*.db' | head -1); python $R/tools/rocpd_counters.py $db $R/gpurun_out/pmc_sq2.json prefill > $R/gpurun_out/pmc_sq2.txt 2>&1
cd $R; cat gpurun_out/pmc_sq1.txt gpurun_out/pmc_sq2.txt; tail -3 /tmp/sq2.log
