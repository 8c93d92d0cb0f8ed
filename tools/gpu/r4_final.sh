#!/bin/bash
# round-4 final evidence pass: full GPU test suite, 32-window PSNR sweep against the fp32 build, the measurement pass
set -u
O=gpurun_out; mkdir -p $O
TAG=${1:-r4_v5}
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6 | tee $O/${TAG}_gpu_tests.txt
for f in model ops r2 r2b x3 rowchain; do cp $O/parity_$f.json $O/${TAG}_parity_$f.json 2>/dev/null; done
PGT_SWEEP_MODES=x3f16 timeout 900 python tools/gpu/psnr_sweep.py $O/${TAG}_psnr_sweep_32_windows.json 4 10 2>/dev/null | tail -4
bash tools/gpu/r4_measure.sh $TAG 2>&1 | tail -6
