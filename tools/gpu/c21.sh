cd $GRAFT_REPO_ROOT
O=gpurun_out/c21; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_x3.py -m gpu -q 2>&1 | tail -8 ) > $O/tests.log; tail -4 $O/tests.log
( timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q -k "default_mode or other_weights or overlap or parsing or middle" 2>&1 | tail -8 ) > $O/tests2.log; tail -4 $O/tests2.log
timeout 900 python bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-roofline > $O/bench_fold.json 2> $O/bench_fold.err; head -c 120 $O/bench_fold.json; echo
PGT_X3_FOLD=0 timeout 900 python bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-roofline > $O/bench_nofold.json 2> $O/bench_nofold.err; head -c 120 $O/bench_nofold.json; echo
timeout 900 python bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-roofline > $O/bench_fold2.json 2> $O/bench_fold2.err; head -c 120 $O/bench_fold2.json; echo
PGT_X3_FOLD=0 timeout 900 python bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-roofline > $O/bench_nofold2.json 2> $O/bench_nofold2.err; head -c 120 $O/bench_nofold2.json; echo
