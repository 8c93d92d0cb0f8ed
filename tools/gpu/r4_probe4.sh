#!/bin/bash
O=gpurun_out; mkdir -p $O
for D in "RC_KG1=2 RC_PREFETCH=0" "RC_KG1=4 RC_PREFETCH=0" "RC_KG1=8 RC_PREFETCH=0" "RC_KG1=8 RC_PREFETCH=0 RC_PIPE=0" "RC_KG1=4 RC_PREFETCH=0 RC_PIPE=0"; do
  timeout 200 python tools/rowchain_probe.py --probes 256 --define $D --ln r2w8 2>>$O/r4f_err.txt | tee -a $O/r4f_probe.jsonl
done
tail -3 $O/r4f_err.txt
