set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04n; mkdir -p $O
cd /tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_b1 -- python $R/bench.py --steps 64 --warmup 16 --gen-tokens 0 --prompt 293 --cpu-steps 0 --no-extras > /tmp/tr_b1.log 2>&1
python $R/tools/trace_gaps.py /tmp/tr_b1 180 > $O/gaps_b1_persistent.txt 2>&1; cat $O/gaps_b1_persistent.txt
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_b1c -- python $R/bench.py --steps 64 --warmup 16 --gen-tokens 0 --prompt 293 --cpu-steps 0 --no-extras --option persistent_rows=0 > /tmp/tr_b1c.log 2>&1
python $R/tools/trace_gaps.py /tmp/tr_b1c 600 > $O/gaps_b1_launch_chain.txt 2>&1; head -12 $O/gaps_b1_launch_chain.txt
