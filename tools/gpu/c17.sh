cd $GRAFT_REPO_ROOT
O=gpurun_out/c17; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q -k "119 or batched or overlap" 2>&1 | tail -8 ) > $O/tests.log; tail -4 $O/tests.log
export PGT_AUTOTUNE_CACHE=$GRAFT_REPO_ROOT/$O/tune.json
timeout 900 python bench.py --no-cpu-baseline --no-roofline > $O/bench_l2.json 2> $O/bench_l2.err; head -c 200 $O/bench_l2.json; echo
timeout 900 python bench.py --no-cpu-baseline --no-roofline --lanes 1 > $O/bench_l1.json 2> $O/bench_l1.err; head -c 200 $O/bench_l1.json; echo
timeout 900 python bench.py --no-cpu-baseline --no-roofline --lanes 3 > $O/bench_l3.json 2> $O/bench_l3.err; head -c 200 $O/bench_l3.json; echo
timeout 900 python bench.py --no-cpu-baseline --no-roofline --lanes 2 --precision bf16 > $O/bench_bf16_l2.json 2> $O/bench_bf16_l2.err; head -c 200 $O/bench_bf16_l2.json; echo
