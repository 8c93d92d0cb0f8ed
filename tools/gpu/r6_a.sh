#!/bin/bash
# round 6, first pass: the exact-weight kernels (parity, timing), the contract figure with / without them at points 1-3, bench A/B
mkdir -p gpurun_out
python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "exact_weights or banded_bias or c64_ring or epilogue_groupnorm" 2>&1 | tail -15
timeout 600 python tools/bench_w2.py gpurun_out/r6_a_w2_micro.jsonl 2>&1 | tail -12
for P in 2 1; do
  for V in "PGT_EXACT_W=512,32" "PGT_EXACT_W=" "PGT_EXACT_W=512" "PGT_EXACT_W=32"; do
    R5_POINT=$P env $V timeout 400 python tools/gpu/second_point_spread.py gpurun_out/r6_a_spread.jsonl 2>&1 | tail -1
  done
done
bash tools/gpu/ab_env.sh "--no-extras" "PGT_EXACT_W=512,32" "PGT_EXACT_W="
