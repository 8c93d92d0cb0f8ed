cd $GRAFT_REPO_ROOT
O=gpurun_out/c5; mkdir -p $O
for B in 24 32; do
  timeout 900 python bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-roofline --windows-per-forward $B > $O/bench_x3_b$B.json 2> $O/bench_x3_b$B.err
  head -c 300 $O/bench_x3_b$B.json; echo; tail -3 $O/bench_x3_b$B.err
done
( timeout 600 python -m pytest tests/test_gpu_ops.py -q -k "groupnorm_statistics or epilogue_statistics" 2>&1 | tail -5 ) > $O/tests.log; cat $O/tests.log
