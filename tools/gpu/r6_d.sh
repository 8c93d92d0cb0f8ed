#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "exact_weights" 2>&1 | grep -v "^$" | tail -40
for P in 2 1 3; do
  R5_POINT=$P timeout 500 python tools/gpu/second_point_spread.py gpurun_out/r6_d_spread.jsonl 2>&1 | tail -1
  R5_POINT=$P PGT_K_ORDER=tap timeout 500 python tools/gpu/second_point_spread.py gpurun_out/r6_d_spread.jsonl 2>&1 | tail -1
done
bash tools/gpu/ab_env.sh "--no-extras" "PGT_K_ORDER=x" "PGT_K_ORDER=tap"
