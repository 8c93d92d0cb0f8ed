#!/bin/bash
# window attention: persistent kernel parity + same-box A/B against the per-window kernel
set -u
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x -k "window or attn or enclayer or swin or kernel_forms" 2>&1 | tail -4
for rep in 1 2; do
  for v in 0 1; do
    PGT_WATTN_PERSIST=$v timeout 120 python tools/bench_wattn.py 2>/dev/null | sed "s/}$/, \"persistent\": $v}/" | tee -a $O/r4i_wattn_persist.jsonl | python -c "
import sys,json
r=[json.loads(l) for l in sys.stdin]
print('persist $v', ' '.join(f\"{x['us']:.0f}\" for x in r))"
  done
done
for v in 0 1 0 1; do
  PGT_WATTN_PERSIST=$v timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline --resident 2>> $O/r4i_bench.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('persist=$v', d['value'], d['ms_per_step'])" | tee -a $O/r4i_bench_ab.txt
done
