mkdir -p gpurun_out
for B in 4 16 64; do
  PGT_WCOMP_BANDS=$B timeout 200 python tools/gpu/second_point_spread.py gpurun_out/r5_i_bands.jsonl 2>&1 | tail -1
  PGT_WCOMP_BANDS=$B timeout 400 python -m pytest tests/test_gpu_model.py -q -m gpu -k "psnr_contract and not second" 2>&1 | tail -2
  python - <<'P'
import json
d=json.load(open('gpurun_out/parity_model.json'))
vals=[]
for k,v in d.items():
    if k.startswith('operating_point') and isinstance(v,dict):
        if 'windows' in v:
            vals += [(f"{w.get('clip_seed')}w{w.get('window')}", w['dpsnr_db']) for w in v['windows'] if w.get('differing_tokens',0)==0 and 'dpsnr_db' in w]
        elif 'x3f16' in k and 'dpsnr_db' in v: vals.append((k, v['dpsnr_db']))
print("point 1: max |dpsnr|", max(abs(x) for _,x in vals), "mean", sum(x for _,x in vals)/len(vals), len(vals))
P
done
