#!/bin/bash
O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_rowchain.py -x -q 2>&1 | tail -5
for D in "RC_PIPE=0 RC_PREFETCH=0" "RC_PIPE=1 RC_PREFETCH=0" "RC_PIPE=0 RC_PREFETCH=1" "RC_PIPE=1 RC_PREFETCH=1"; do
  timeout 200 python tools/rowchain_probe.py --probes 0 --define $D 2>>$O/r4c_err.txt | tee -a $O/r4c_probe.jsonl
done
for K in 2 4 8; do for V in r1w8 r2w8 r1w16; do
  timeout 200 python tools/rowchain_probe.py --probes 0 --define RC_KG0=$K --ln $V 2>>$O/r4c_err.txt | tee -a $O/r4c_probe.jsonl
done; done
tail -3 $O/r4c_err.txt
