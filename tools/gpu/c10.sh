cd $GRAFT_REPO_ROOT
O=gpurun_out/c10; mkdir -p $O
( timeout 1200 python -m pytest tests/test_gpu_model.py -q 2>&1 | tail -30 ) > $O/tests.log; tail -6 $O/tests.log
cp gpurun_out/parity_model.json $O/ 2>/dev/null
timeout 900 python bench.py --steps 12 --warmup 2 --no-cpu-baseline > $O/bench_x3.json 2> $O/bench_x3.err; head -c 230 $O/bench_x3.json; echo; tail -2 $O/bench_x3.err
timeout 900 python bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-roofline --precision bf16 > $O/bench_bf16.json 2> $O/bench_bf16.err; head -c 230 $O/bench_bf16.json; echo
