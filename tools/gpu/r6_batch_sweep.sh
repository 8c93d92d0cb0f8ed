#!/bin/bash
# windows per forward x forwards in flight, same box, back to back (each line: bench.py's own JSON, cut down)
mkdir -p gpurun_out
for rep in 1 2; do
for cfg in "32 2" "48 2" "64 2" "32 3" "24 2" "32 2"; do
  set -- $cfg
  timeout 600 python bench.py --steps 8 --warmup 2 --no-roofline --no-cpu-baseline --no-extras --windows-per-forward $1 --lanes $2 2>gpurun_out/bs_err.log \
    | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(json.dumps({'windows_per_forward':$1,'lanes':$2,'fps':d['value'],'ms_per_step':d['ms_per_step']}))" \
    || tail -3 gpurun_out/bs_err.log
done; done | tee gpurun_out/r6_batch_sweep.jsonl
