# Round 5: the driver's bench command (+ the --force-pg rehearsal of the N > 1 path on RCCL) on the final code.
set -x
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r05_bench
mkdir -p $O
cd $R
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_steps20.json 2> $O/bench_steps20.err ) 2> $O/bench_steps20.time; tail -3 $O/bench_steps20.time; cut -c1-300 $O/bench_steps20.json
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --force-pg > $O/bench_steps20_force_pg.json 2> $O/bench_force_pg.err ) 2> $O/bench_force_pg.time; tail -3 $O/bench_force_pg.time; cut -c1-200 $O/bench_steps20_force_pg.json; tail -3 $O/bench_force_pg.err
