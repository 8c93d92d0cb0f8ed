#!/bin/bash
# round 4, first GPU pass: parity of the fused token-row chains, their micro-benchmark, then a same-box A/B of the bench
set -u
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_rowchain.py -x -q 2>&1 | tail -15 > $O/r4a_rowchain_tests.txt
cat $O/r4a_rowchain_tests.txt
for V in auto r1w8 r2w8 r1w16; do
  if [ $V = auto ]; then timeout 300 python tools/bench_rowchain.py >> $O/r4a_rowchain_micro.jsonl 2>>$O/r4a_err.txt
  else PGT_RC_LN=$V timeout 300 python tools/bench_rowchain.py >> $O/r4a_rowchain_micro.jsonl 2>>$O/r4a_err.txt; fi
done
cat $O/r4a_rowchain_micro.jsonl
bash tools/gpu/ab_env.sh "" "PGT_ROWCHAIN=0" "PGT_ROWCHAIN=1" 2>&1 | tee $O/r4a_ab.txt
tail -5 $O/r4a_err.txt
