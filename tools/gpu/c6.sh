cd $GRAFT_REPO_ROOT
O=gpurun_out/c6; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_x3.py tests/test_gpu_model.py -q 2>&1 | tail -12 ) > $O/tests.log; tail -4 $O/tests.log
PGT_DUMP_SHAPES=$O/shapes_x3.txt timeout 900 python bench.py --steps 12 --warmup 2 --no-cpu-baseline > $O/bench_x3.json 2> $O/bench_x3.err
head -c 250 $O/bench_x3.json; echo
