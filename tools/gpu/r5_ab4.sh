#!/bin/bash
# Round 5: sample size of a band of the compensation's mean field (cells of 16 pixels per band): parity spread of point 2 and point 1, A/B
mkdir -p gpurun_out
for V in "PGT_WCOMP_CELLS=16" "PGT_WCOMP_CELLS=8" "PGT_WCOMP_CELLS=4"; do
  env $V timeout 200 python tools/gpu/second_point_spread.py gpurun_out/r5_l_cells_spread.jsonl 2>&1 | tail -1
  env $V timeout 400 python -m pytest tests/test_gpu_model.py -q -m gpu -k "psnr_contract and not second" 2>&1 | tail -1
  python - <<'P'
import json
d=json.load(open('gpurun_out/parity_model.json'))
vals=[]
for k,v in d.items():
    if k.startswith('operating_point') and isinstance(v,dict):
        if 'windows' in v:
            vals += [w['dpsnr_db'] for w in v['windows'] if w.get('differing_tokens',0)==0 and 'dpsnr_db' in w]
        elif 'x3f16' in k and 'dpsnr_db' in v: vals.append(v['dpsnr_db'])
print("point 1: max |dpsnr|", max(abs(x) for x in vals), len(vals))
P
done
for i in 1 2; do
  for V in "PGT_WCOMP_CELLS=16" "PGT_WCOMP_CELLS=8" "PGT_WCOMP_CELLS=4" "PGT_WCOMP_BANDS=1"; do
    env $V timeout 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --resident --no-roofline --no-extras 2>/dev/null | python -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('[$V]', b['value'], 'fps', b['ms_per_step'], 'ms/step')"
  done
done | tee gpurun_out/r5_l_ab_cells.txt
