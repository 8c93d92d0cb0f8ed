#!/bin/bash
# window attention: where a wave's cycles go (SQ counters over tools/bench_mha.py 32) -> gpurun_out/r6_i_mha_sq_counters.json
set -u
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd /tmp
D=$O/prof_mha
i=0
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD" "SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $C -d $D/p$i -o r -- python $GRAFT_REPO_ROOT/tools/bench_mha.py 32 > $D.p$i.log 2>&1
  tail -1 $D.p$i.log | cut -c1-200
done
cd $GRAFT_REPO_ROOT
python tools/pmc_table.py $O/r6_i_mha_sq_counters.json $(find $D -name '*_results.db' | sort) | head -8
rm -rf $D
