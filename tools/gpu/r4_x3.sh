#!/bin/bash
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_rowchain.py -x -q 2>&1 | tail -8
timeout 600 python -m pytest tests/test_gpu_model.py -x -q -k "range_telemetry or psnr_contract_at or default_mode" 2>&1 | tail -8
bash tools/gpu/ab_env.sh "" "PGT_ROWCHAIN_X3=0" "PGT_ROWCHAIN_X3=1" 2>&1 | tee $O/r4i_ab.txt
