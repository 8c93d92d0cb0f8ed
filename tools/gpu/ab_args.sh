#!/bin/bash
# A/B of bench.py argument sets inside ONE gpurun call (same box, interleaved twice): bash tools/gpu/ab_args.sh "args a" "args b" ...
for i in 1 2; do
  for V in "$@"; do
    timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --resident --no-roofline $V 2>/dev/null | python -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('[$V]', b['value'], 'fps', b['ms_per_step'], 'ms/step')"
  done
done
