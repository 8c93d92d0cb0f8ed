#!/bin/bash
# compensation only where the weight rounding matters (decoder stages 512 and 32): contract figure at the four points, frames/s
mkdir -p gpurun_out
for V in "PGT_WCOMP_STAGES=512,32" "PGT_WCOMP_STAGES=512,32,64" ; do
  for P in 2 1 3; do R5_POINT=$P env $V timeout 500 python tools/gpu/second_point_spread.py gpurun_out/r6_k_wcomp_stages_spread.jsonl 2>&1 | tail -1; done
done
bash tools/gpu/ab_env.sh "--no-extras" "PGT_X=all" "PGT_WCOMP_STAGES=512,32"
