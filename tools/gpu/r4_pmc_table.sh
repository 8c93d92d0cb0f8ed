#!/bin/bash
# per-kernel HBM traffic (+ L2 hit rate) of one eager forward: three PMC passes -> gpurun_out/<tag>_pmc_by_kernel.json
set -u
TAG=${1:-r4}
export TMPDIR=/tmp PGT_RANGE_CHECK=0 PGT_SIDE_STREAM=0
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd /tmp
D=$O/prof_${TAG}_tbl
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  N=$(echo $C | cut -d' ' -f1)
  timeout 300 rocprofv3 --kernel-trace --pmc $C -d $D/$N -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-graph --lanes 1 --resident --no-cpu-baseline --no-roofline > $D.$N.log 2>&1
  tail -2 $D.$N.log
done
cd $GRAFT_REPO_ROOT
python tools/pmc_table.py $O/${TAG}_pmc_by_kernel.json $(find $D -name '*_results.db' | sort)
rm -rf $D
