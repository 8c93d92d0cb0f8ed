cd $GRAFT_REPO_ROOT
O=gpurun_out/c7; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_x3.py -q -k "mha" 2>&1 | tail -4 ) > $O/tests_mha.log; cat $O/tests_mha.log
timeout 900 python bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-roofline > $O/bench_x3.json 2> $O/bench_x3.err; head -c 230 $O/bench_x3.json; echo
PGT_SIDE_STREAM=1 timeout 900 python bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-roofline > $O/bench_x3_side.json 2> $O/bench_x3_side.err; head -c 230 $O/bench_x3_side.json; echo; tail -2 $O/bench_x3_side.err
timeout 600 python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -4 $O/smoke.log
