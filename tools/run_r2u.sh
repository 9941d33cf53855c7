mkdir -p gpurun_out; R=$PWD
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/pmc_sq3
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES -d /tmp/pmc_sq3 -- python $R/bench.py --batch 32 --steps 64 --warmup 16 --gen-tokens 0 --cpu-steps 0 --no-extras > /tmp/sq3.log 2>&1
db=$(find /tmp/pmc_sq3 -name '*.db' | head -1); python $R/tools/rocpd_counters.py $db $R/gpurun_out/pmc_sq_b32.json > $R/gpurun_out/pmc_sq_b32.txt 2>&1
rm -rf /tmp/pmc_sq4
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES -d /tmp/pmc_sq4 -- python $R/bench.py --batch 1 --steps 64 --warmup 16 --gen-tokens 0 --cpu-steps 0 --no-extras > /tmp/sq4.log 2>&1
db=$(find /tmp/pmc_sq4 -name '*.db' | head -1); python $R/tools/rocpd_counters.py $db $R/gpurun_out/pmc_sq_b1.json > $R/gpurun_out/pmc_sq_b1.txt 2>&1
rm -rf /tmp/pmc_sq5
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES -d /tmp/pmc_sq5 -- python $R/tools/prefill_probe.py 32 512 > /tmp/sq5.log 2>&1
db=$(find /tmp/pmc_sq5 -name '*.db' | head -1); python $R/tools/rocpd_counters.py $db $R/gpurun_out/pmc_sq_prefill.json prefill > $R/gpurun_out/pmc_sq_prefill.txt 2>&1
cd $R; head -50 gpurun_out/pmc_sq_b32.txt
