#!/bin/bash
# Round 5, same-box A/B of this round's switches (interleaved twice): ring kernel + fused GroupNorm apply, one-launch frame bias
mkdir -p gpurun_out
(timeout 300 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "frame_bias or bias_per_frame or c64_ring" 2>&1 | tail -5) > gpurun_out/r5_c_tests.txt
cat gpurun_out/r5_c_tests.txt
for i in 1 2; do
  for V in "PGT_X=1" "PGT_FRAME_BIAS=0" "PGT_C64_RING=0 PGT_FUSE_GN_APPLY=0" "PGT_C64_RING=0 PGT_FUSE_GN_APPLY=0 PGT_FRAME_BIAS=0"; do
    env $V timeout 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --resident --no-roofline 2>/dev/null | python -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('[$V]', b['value'], 'fps', b['ms_per_step'], 'ms/step')"
  done
done | tee gpurun_out/r5_c_ab.txt
