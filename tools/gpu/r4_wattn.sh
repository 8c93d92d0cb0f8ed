#!/bin/bash
# window attention: same-box A/B of library variants (tools/_probe/wa/libpgt_*.so)
set -u
O=gpurun_out; mkdir -p $O
for rep in 1 2; do
  for v in bq0 bq1 bq2; do
    PGT_LIB_PATH=$PWD/tools/_probe/wa/libpgt_$v.so timeout 120 python tools/bench_wattn.py 2>/dev/null | sed "s/}$/, \"variant\": \"$v\"}/" | tee -a $O/r4j_wattn_bq.jsonl | python -c "
import sys,json
r=[json.loads(l) for l in sys.stdin]
print('$v', ' '.join(f\"{x['us']:.0f}\" for x in r))"
  done
done
