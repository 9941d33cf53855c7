#!/bin/bash
# MFMA-busy and LDS bank-conflict counters of the parity engine's prompt pass (32 x 512), one counter group per rocprofv3 pass, kernel trace only
# -> gpurun_out/pmc_pa/{mfma,lds}.json   (bash tools/pmc_prefill_attention.sh; about one GPU-minute)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/pmc_pa
mkdir -p $O
cd /tmp
rm -rf /tmp/pmc_pa_mfma /tmp/pmc_pa_lds
timeout 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES -d /tmp/pmc_pa_mfma -- python $R/tools/prefill_probe.py 32 512 fp32 > /tmp/pmc_pa_mfma.log 2>&1
db=$(find /tmp/pmc_pa_mfma -name '*.db' | head -1); [ -n "$db" ] && python $R/tools/rocpd_mfma.py $db $O/mfma.json > $O/mfma.txt 2>&1 || tail -5 /tmp/pmc_pa_mfma.log > $O/mfma.txt
timeout 200 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS -d /tmp/pmc_pa_lds -- python $R/tools/prefill_probe.py 32 512 fp32 > /tmp/pmc_pa_lds.log 2>&1
db=$(find /tmp/pmc_pa_lds -name '*.db' | head -1); [ -n "$db" ] && python $R/tools/rocpd_counters.py $db $O/lds.json attn_prefill > $O/lds.txt 2>&1 || tail -5 /tmp/pmc_pa_lds.log > $O/lds.txt
cat $O/mfma.txt | head -30; cat $O/lds.txt | head -30; ls $O
