#!/bin/bash
# A/B of an environment switch inside ONE gpurun call (same box, interleaved): bash tools/gpu/ab.sh "VAR=a" "VAR=b" [bench args]
A=$1; B=$2; shift 2
for i in 1 2; do
  for V in "$A" "$B"; do
    env $V timeout 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --resident "$@" 2>/dev/null | python -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=b.get('roofline',{})
print('$V', b['value'], 'fps', b['ms_per_step'], 'ms/step', ' '.join('%s=%.2f' % (k['name'][:12].replace(' ','_'), k['ms_per_step']) for k in r.get('kernels',[])))"
  done
done
