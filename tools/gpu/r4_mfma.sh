#!/bin/bash
# per-kernel MFMA utilisation + effective clock of one eager forward (two PMC passes) -> gpurun_out/<tag>_mfma_utilisation.json
set -u
TAG=${1:-r4}
export TMPDIR=/tmp PGT_RANGE_CHECK=0 PGT_SIDE_STREAM=0
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd /tmp
D=$O/prof_${TAG}_mfma
for C in SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $C -d $D/$C -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-graph --lanes 1 --resident --no-cpu-baseline --no-roofline > $D.$C.log 2>&1
done
cd $GRAFT_REPO_ROOT
A=$(find $D/SQ_VALU_MFMA_BUSY_CYCLES -name '*_results.db' | head -1); B=$(find $D/GRBM_GUI_ACTIVE -name '*_results.db' | head -1)
python tools/pmc_mfma.py $A $B $O/${TAG}_mfma_utilisation.json
rm -rf $D
