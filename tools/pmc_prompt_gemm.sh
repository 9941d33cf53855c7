#!/bin/bash
# Counter passes that name what the prompt GEMM's k loop waits for: one small counter group per rocprofv3 pass,
# kernel trace only (no other trace domains), on the 32 x 512 prompt pass.  Per-kernel averages -> gpurun_out/pmc_pf/<group>.json
#   bash tools/pmc_prompt_gemm.sh            (about one GPU-minute)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/pmc_pf
mkdir -p $O
cd /tmp
declare -A GROUPS=(
  [lds]="SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS"
  [vmem]="SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES"
  [l2]="TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCP_TCC_READ_REQ_sum"
  [ta]="TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum SQ_BUSY_CYCLES"
)
for g in lds vmem l2 ta; do
  rm -rf /tmp/pmc_pf_$g
  timeout 120 rocprofv3 --kernel-trace --pmc ${GROUPS[$g]} -d /tmp/pmc_pf_$g -- python $R/tools/prefill_probe.py 32 512 > /tmp/pmc_pf_$g.log 2>&1
  db=$(find /tmp/pmc_pf_$g -name '*.db' | head -1)
  if [ -n "$db" ]; then python $R/tools/rocpd_counters.py $db $O/$g.json prefill_gemm > $O/$g.txt 2>&1; else echo "no db for $g" > $O/$g.txt; tail -5 /tmp/pmc_pf_$g.log >> $O/$g.txt; fi
done
cat $O/*.txt | head -80
