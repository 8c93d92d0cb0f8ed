#!/bin/bash
# window attention: heads per workgroup (PGT_WATTN_HPW) parity + same-box A/B
set -u
O=gpurun_out; mkdir -p $O
for H in 4 8; do PGT_WATTN_HPW=$H timeout 600 python -m pytest tests -m gpu -q -k "window or attn or enclayer or swin" 2>&1 | tail -2; done
for rep in 1 2; do
  for v in 1 2 4 8; do
    PGT_WATTN_HPW=$v timeout 120 python tools/bench_wattn.py 2>/dev/null | sed "s/}$/, \"hpw\": $v}/" | tee -a $O/r4h_wattn_hpw.jsonl | python -c "
import sys,json
r=[json.loads(l) for l in sys.stdin]
print('hpw $v', ' '.join(f\"{x['us']:.0f}\" for x in r))"
  done
done
