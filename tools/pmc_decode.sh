#!/bin/bash
# Counter passes on the decode step (batch 1 and 32): what the skinny GEMM / attention kernels wait for.  One small counter group per
# rocprofv3 pass, kernel trace only, short generation (the full-length pass overruns rocprofv3's counter collection after ~30 k dispatches).
# Per-kernel averages -> gpurun_out/pmc_dec/<batch>_<group>.json      bash tools/pmc_decode.sh      (about two GPU-minutes)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/pmc_dec
mkdir -p $O
cd /tmp
declare -A GROUPS=(
  [lds]="SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS"
  [vmem]="SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES"
  [l2]="TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCP_TCC_READ_REQ_sum"
  [ta]="TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum SQ_BUSY_CYCLES SQ_WAVES"
)
for b in 1 32; do
  for g in lds vmem l2 ta; do
    rm -rf /tmp/pmc_dec_${b}_$g
    timeout 150 rocprofv3 --kernel-trace --pmc ${GROUPS[$g]} -d /tmp/pmc_dec_${b}_$g -- python $R/bench.py --batch $b --steps 48 --warmup 8 --gen-tokens 0 --cpu-steps 0 --no-extras > /tmp/pmc_dec_${b}_$g.log 2>&1
    db=$(find /tmp/pmc_dec_${b}_$g -name '*.db' | head -1)
    if [ -n "$db" ]; then python $R/tools/rocpd_counters.py $db $O/${b}_$g.json > $O/${b}_$g.txt 2>&1; else echo "no db for $b $g" > $O/${b}_$g.txt; tail -5 /tmp/pmc_dec_${b}_$g.log >> $O/${b}_$g.txt; fi
  done
done
ls -la $O
