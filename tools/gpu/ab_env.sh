#!/bin/bash
# A/B/C... of environment settings inside ONE gpurun call (same box, interleaved twice):
#   bash tools/gpu/ab_env.sh "<bench.py args>" "VAR=a" "VAR=b OTHER=c" ...
ARGS=$1; shift
for i in 1 2; do
  for V in "$@"; do
    env $V timeout 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --resident $ARGS 2>/dev/null | python -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=b.get('roofline',{})
print('[$V]', b['value'], 'fps', ' '.join('%s=%.2f' % (k['name'][:9].replace(' ','_'), k['ms_per_step']) for k in r.get('kernels',[])[:2]))"
  done
done
