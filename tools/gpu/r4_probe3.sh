#!/bin/bash
O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_rowchain.py -x -q 2>&1 | tail -5
for D in "RC_PREFETCH=0 RC_PREFETCH0=0" "RC_PREFETCH=1 RC_PREFETCH0=1" "RC_PREFETCH=1 RC_PREFETCH0=1 RC_PIPE=0"; do
 for V in r1w8 r2w8 r1w16; do
  timeout 200 python tools/rowchain_probe.py --probes 0 --define $D --ln $V 2>>$O/r4d_err.txt | tee -a $O/r4d_probe.jsonl
 done
done
timeout 300 python tools/rowchain_probe.py --probes 1 2 8 32 64 128 --ln r2w8 2>>$O/r4d_err.txt | tee -a $O/r4d_probe.jsonl
tail -3 $O/r4d_err.txt
