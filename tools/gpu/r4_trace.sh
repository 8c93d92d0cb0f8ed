#!/bin/bash
# kernel trace of the current build (bench.py --lanes 1, one forward at a time) -> gpurun_out/<tag>_kernel_stats.csv
set -u
TAG=${1:-r4}
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
cd /tmp
D=$GRAFT_REPO_ROOT/$O/prof_${TAG}
timeout 400 rocprofv3 --kernel-trace --stats -d $D/trace -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 1 --lanes 1 --resident --no-cpu-baseline --no-roofline > $D.trace.log 2>&1
cd $GRAFT_REPO_ROOT
T=$(find $D/trace -name '*_results.db' | head -1)
python tools/rocpd_stats.py $T $O/${TAG}_x3f16_b32_kernel_stats.csv @32
rm -rf $D/trace
head -45 $O/${TAG}_x3f16_b32_kernel_stats.csv | cut -c1-200
tail -3 $D.trace.log
