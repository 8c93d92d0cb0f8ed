#!/bin/bash
# round 6, second pass: where the contract figure of the third point comes from once stages 512 / 32 are exact (all stages exact,
# remaining stages uncompensated), the block-major K order of igemm4 (A/B), exact-weight kernels' parity
mkdir -p gpurun_out
python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "exact_weights or banded_bias or c64_ring or epilogue_groupnorm or conv" 2>&1 | tail -8
for V in "PGT_EXACT_W=512,256,128,64,32" "PGT_EXACT_W=512,256,128,64,32 PGT_WCOMP=0" "PGT_EXACT_W=512,32 PGT_WCOMP=0" "PGT_EXACT_W= PGT_WCOMP=0" "PGT_EXACT_W=512,32 PGT_K_ORDER=tap"; do
  R5_POINT=2 env $V timeout 500 python tools/gpu/second_point_spread.py gpurun_out/r6_b_spread.jsonl 2>&1 | tail -1
done
R5_POINT=1 PGT_EXACT_W=512,256,128,64,32 timeout 500 python tools/gpu/second_point_spread.py gpurun_out/r6_b_spread.jsonl 2>&1 | tail -1
bash tools/gpu/ab_env.sh "--no-extras" "PGT_K_ORDER=cb" "PGT_K_ORDER=tap"
PGT_DUMP_SHAPES=gpurun_out/r6_b_conv_shapes_cb.txt timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --resident --no-extras > gpurun_out/r6_b_bench_cb.json 2>/dev/null
PGT_K_ORDER=tap PGT_DUMP_SHAPES=gpurun_out/r6_b_conv_shapes_tap.txt timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --resident --no-extras > gpurun_out/r6_b_bench_tap.json 2>/dev/null
