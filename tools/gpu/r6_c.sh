#!/bin/bash
# round 6, third pass: where the half decoder picks up a mean error (default mode against the fp32 mode of the build, block by block)
mkdir -p gpurun_out
python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "exact_weights" 2>&1 | tail -3
R5_POINT=2 PGT_EXACT_W=512,256,128,64,32 timeout 900 python tools/gpu/dc_bias_probe.py 11077 3 gpurun_out/r6_c_dc_probe_all_exact.json 2>&1 | tail -80
R5_POINT=2 PGT_EXACT_W= timeout 900 python tools/gpu/dc_bias_probe.py 11077 3 gpurun_out/r6_c_dc_probe_r5cfg.json 2>&1 | tail -3
