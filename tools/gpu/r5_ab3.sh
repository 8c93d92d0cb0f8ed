#!/bin/bash
# Round 5: banded mean field of the compensation (16 bias vectors per frame) - parity at both operating points, then a same-box A/B
mkdir -p gpurun_out
(timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -q -m gpu -k "frame_bias or psnr_contract or bias_per_frame or exported_program or whole_model_default or teacher" 2>&1 | tail -4) | tee gpurun_out/r5_j_tests.txt
for V in "PGT_X=1"; do env $V timeout 200 python tools/gpu/second_point_spread.py gpurun_out/r5_j_bands_spread.jsonl 2>&1 | tail -1; done
for i in 1 2; do
  for V in "PGT_X=1" "PGT_WCOMP_BANDS=1"; do
    env $V timeout 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --resident --no-roofline --no-extras 2>/dev/null | python -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('[$V]', b['value'], 'fps', b['ms_per_step'], 'ms/step')"
  done
done | tee gpurun_out/r5_j_ab_bands.txt
