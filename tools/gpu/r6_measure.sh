#!/bin/bash
# Round-6 measurement pass (the round-5 pass with the binary stamp: lib_sha16 = the source sha compiled into the loaded library) on the GPU box: kernel trace (+ the traced conv / linear family json bench.py quotes), PMC traffic
# (first: the bench line then quotes it), MFMA utilisation per kernel, the rocprofv3 sweep of BASELINE config 5, bench lines
# (default, bf16-decoder mode, configs[2] at N=1).    usage: bash tools/gpu/r6_measure.sh <tag>      (writes gpurun_out/<tag>_*)
set -u
TAG=${1:-r6}
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp
D=$R/$O/prof_${TAG}
export PGT_RANGE_CHECK=0      # the profiled passes: steady-state forwards only (no range-telemetry pass over every tensor)
timeout 300 rocprofv3 --kernel-trace --stats -d $D/trace -o t -- python $R/bench.py --steps 8 --warmup 1 --lanes 1 --resident --no-cpu-baseline --no-roofline --no-extras > $D.trace.log 2>&1
PGT_SIDE_STREAM=0 timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $D/fetch -o r -- python $R/bench.py --steps 2 --warmup 1 --no-graph --lanes 1 --resident --no-cpu-baseline --no-roofline --no-extras > $D.fetch.log 2>&1
PGT_SIDE_STREAM=0 timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $D/write -o r -- python $R/bench.py --steps 2 --warmup 1 --no-graph --lanes 1 --resident --no-cpu-baseline --no-roofline --no-extras > $D.write.log 2>&1
for C in SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE; do
  PGT_SIDE_STREAM=0 timeout 300 rocprofv3 --kernel-trace --pmc $C -d $D/$C -o r -- python $R/bench.py --steps 1 --warmup 1 --no-graph --lanes 1 --resident --no-cpu-baseline --no-roofline --no-extras > $D.$C.log 2>&1
done
# BASELINE config 5: window attention 3x8x8, C = 512, fp16, nW in {64, 256, 512, 1024} - a kernel trace of the sweep
timeout 200 rocprofv3 --kernel-trace --stats -d $D/cfg5 -o c -- python $R/tools/config5_sweep.py > $D.cfg5.log 2>&1
cd $R
unset PGT_RANGE_CHECK
T=$(find $D/trace -name '*_results.db' | head -1); F=$(find $D/fetch -name '*_results.db' | head -1); W=$(find $D/write -name '*_results.db' | head -1)
A=$(find $D/SQ_VALU_MFMA_BUSY_CYCLES -name '*_results.db' | head -1); B=$(find $D/GRBM_GUI_ACTIVE -name '*_results.db' | head -1)
C5=$(find $D/cfg5 -name '*_results.db' | head -1)
python tools/rocpd_stats.py $T $O/${TAG}_x3f16_b32_kernel_stats.csv @32 $O/${TAG}_traced_family.json x3f16
python tools/pmc_traffic.py $F $W $O/${TAG}_igemm_traffic_pmc.json x3f16 32
python tools/pmc_table.py $O/${TAG}_pmc_by_kernel.json $F $W | head -3
python tools/pmc_mfma.py $A $B $O/${TAG}_mfma_utilisation.json | tail -3
python tools/config5_sweep.py --summarise $C5 $O/${TAG}_config5_sweep.csv
cp $O/${TAG}_igemm_traffic_pmc.json profiles/${TAG}_igemm_traffic_pmc.json      # (this box's copy of the tree: bench.py quotes both below)
cp $O/${TAG}_traced_family.json profiles/${TAG}_traced_family.json
rm -rf $D/trace $D/fetch $D/write $D/SQ_VALU_MFMA_BUSY_CYCLES $D/GRBM_GUI_ACTIVE $D/cfg5      # the sqlite traces are large; the summaries travel back
PGT_DUMP_SHAPES=$O/${TAG}_conv_shapes_x3f16_b32.txt PGT_DUMP_OPS=$O/${TAG}_ops_x3f16_b32.txt timeout 500 python bench.py --steps 20 --warmup 3 > $O/${TAG}_bench_x3f16_b32.json 2> $O/${TAG}_bench.err
timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-extras --resident --precision bf16x3 > $O/${TAG}_bench_bf16x3_b32.json 2>> $O/${TAG}_bench.err
timeout 200 python bench.py --steps 2 --warmup 1 --clip-frames 256 > $O/${TAG}_bench_configs2_n1.json 2>> $O/${TAG}_bench.err
tail -c 900 $O/${TAG}_bench_x3f16_b32.json | head -c 900; echo; head -c 400 $O/${TAG}_bench_bf16x3_b32.json; echo; head -c 300 $O/${TAG}_bench_configs2_n1.json; echo; tail -3 $O/${TAG}_bench.err; cat $O/${TAG}_config5_sweep.csv
