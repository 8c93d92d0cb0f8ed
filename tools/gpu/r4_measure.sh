#!/bin/bash
# Round-4 measurement pass on the GPU box: kernel trace, PMC traffic (first: the bench line then quotes it), bench lines
# (default, bf16-decoder mode, configs[2] at N=1).    usage: bash tools/gpu/r4_measure.sh <tag>      (writes gpurun_out/<tag>_*)
set -u
TAG=${1:-r4}
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
cd /tmp
D=$GRAFT_REPO_ROOT/$O/prof_${TAG}
export PGT_RANGE_CHECK=0      # the profiled passes: steady-state forwards only (no range-telemetry pass over every tensor)
timeout 300 rocprofv3 --kernel-trace --stats -d $D/trace -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 1 --lanes 1 --resident --no-cpu-baseline --no-roofline > $D.trace.log 2>&1
PGT_SIDE_STREAM=0 timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $D/fetch -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-graph --lanes 1 --resident --no-cpu-baseline --no-roofline > $D.fetch.log 2>&1
PGT_SIDE_STREAM=0 timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $D/write -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-graph --lanes 1 --resident --no-cpu-baseline --no-roofline > $D.write.log 2>&1
cd $GRAFT_REPO_ROOT
unset PGT_RANGE_CHECK
T=$(find $D/trace -name '*_results.db' | head -1); F=$(find $D/fetch -name '*_results.db' | head -1); W=$(find $D/write -name '*_results.db' | head -1)
python tools/rocpd_stats.py $T $O/${TAG}_x3f16_b32_kernel_stats.csv @32
python tools/pmc_traffic.py $F $W $O/${TAG}_igemm_traffic_pmc.json x3f16 32
python tools/pmc_table.py $O/${TAG}_pmc_by_kernel.json $F $W | head -3
cp $O/${TAG}_igemm_traffic_pmc.json profiles/${TAG}_igemm_traffic_pmc.json      # (this box's copy of the tree: bench.py quotes it below)
rm -rf $D/trace $D/fetch $D/write      # the sqlite traces are large; the summaries are what travels back
PGT_DUMP_SHAPES=$O/${TAG}_conv_shapes_x3f16_b32.txt timeout 400 python bench.py --steps 20 --warmup 3 > $O/${TAG}_bench_x3f16_b32.json 2> $O/${TAG}_bench.err
timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --resident --precision bf16x3 > $O/${TAG}_bench_bf16x3_b32.json 2>> $O/${TAG}_bench.err
timeout 200 python bench.py --steps 2 --warmup 1 --clip-frames 256 > $O/${TAG}_bench_configs2_n1.json 2>> $O/${TAG}_bench.err
tail -c 600 $O/${TAG}_bench_x3f16_b32.json | head -c 600; echo; head -c 400 $O/${TAG}_bench_bf16x3_b32.json; echo; head -c 300 $O/${TAG}_bench_configs2_n1.json; echo; tail -3 $O/${TAG}_bench.err
