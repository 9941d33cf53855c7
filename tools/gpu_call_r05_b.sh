# Round 5, call B: the whole GPU suite with the persistent MFMA stack as the default for 5..32 rows (every batch-5..32 golden runs through it) + marks at the bench window's context.
set -x
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r05_b
mkdir -p $O
cd $R
timeout 300 python tools/pm_probe.py --batches 32 --no-check --no-time --prompt 293 --marks 32 > $O/pm_marks_ctx300.jsonl 2> $O/pm_marks.err; cat $O/pm_marks_ctx300.jsonl
timeout 300 python tools/pm_probe.py --batches 32 --no-check --time-steps 32 --opt mfma_delay_0=0 --opt mfma_delay_1=0 --opt mfma_delay_2=0 --opt mfma_delay_3=0 --opt mfma_delay_4=0 > $O/pm_delay0.jsonl 2>/dev/null; cat $O/pm_delay0.jsonl
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -15 $O/pytest_gpu.log
