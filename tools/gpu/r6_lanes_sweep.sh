for rep in 1 2; do
for cfg in "32 2" "32 3" "32 4" "48 3" "64 3" "64 2"; do
  set -- $cfg
  timeout 600 python bench.py --steps 8 --warmup 2 --no-roofline --no-cpu-baseline --no-extras --windows-per-forward $1 --lanes $2 2>/dev/null \
    | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(json.dumps({'windows_per_forward':$1,'lanes':$2,'fps':d['value'],'ms_per_step':d['ms_per_step']}))"
done; done
