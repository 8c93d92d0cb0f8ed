#!/bin/bash
# Round 5: frame_bias forms on one box: HEAD (1024 threads, bands of an image in one workgroup, parallel tail) against the per-band
# 256-thread form (alternate library, PGT_LIB_PATH), each with 16 bands and with 1
mkdir -p gpurun_out
(timeout 300 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "frame_bias" 2>&1 | tail -1)
env PGT_X=1 timeout 200 python tools/gpu/second_point_spread.py gpurun_out/r5_o_spread.jsonl 2>&1 | tail -1
ALT=$GRAFT_REPO_ROOT/pgtformer_amd/lib/alt/libpgt_perband.so
for i in 1 2; do
  for V in "PGT_X=1" "PGT_WCOMP_BANDS=1" "PGT_LIB_PATH=$ALT" "PGT_LIB_PATH=$ALT PGT_WCOMP_BANDS=1"; do
    env $V timeout 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --resident --no-roofline --no-extras 2>/dev/null | python -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('[${V##*/}]', b['value'], 'fps', b['ms_per_step'], 'ms/step')"
  done
done | tee gpurun_out/r5_o_ab_frame_bias_forms.txt
