#!/bin/bash
O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_rowchain.py -x -q 2>&1 | tail -3
PGT_RC_MLP=w4 PGT_RC_LN=r2w4 timeout 600 python -m pytest tests/test_gpu_rowchain.py -x -q 2>&1 | tail -3
PGT_RC_MLP=w4 PGT_RC_LN=r1w4 timeout 600 python -m pytest tests/test_gpu_rowchain.py -x -q -k "ln_linear" 2>&1 | tail -3
for V in "r2w8 w8" "r2w4 w4" "r1w4 w4" "r1w16 w8"; do set -- $V
  timeout 200 python tools/rowchain_probe.py --probes 256 --ln $1 --mlp $2 2>>$O/r4g_err.txt | tee -a $O/r4g_probe.jsonl
done
timeout 200 python tools/rowchain_probe.py --probes 256 --ln r2w4 --mlp w4 --rows 393216 2>>$O/r4g_err.txt | tee -a $O/r4g_probe.jsonl
timeout 200 python tools/rowchain_probe.py --probes 256 --ln r2w8 --mlp w8 --rows 393216 2>>$O/r4g_err.txt | tee -a $O/r4g_probe.jsonl
tail -3 $O/r4g_err.txt
