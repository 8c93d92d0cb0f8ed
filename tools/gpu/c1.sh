cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c1
( timeout 900 python -m pytest tests -m gpu -q --deselect tests/test_gpu_model.py 2>&1 | tail -60 ) > gpurun_out/c1/tests_ops.log
( timeout 900 python -m pytest tests/test_gpu_model.py -q 2>&1 | tail -80 ) > gpurun_out/c1/tests_model.log
cp gpurun_out/parity_model.json gpurun_out/c1/ 2>/dev/null; cp gpurun_out/parity_x3.json gpurun_out/c1/ 2>/dev/null
PGT_DUMP_SHAPES=gpurun_out/c1/shapes_x3.txt timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/c1/bench_x3.json 2> gpurun_out/c1/bench_x3.err
timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --precision bf16 > gpurun_out/c1/bench_bf16.json 2> gpurun_out/c1/bench_bf16.err
tail -5 gpurun_out/c1/tests_ops.log gpurun_out/c1/tests_model.log; cat gpurun_out/c1/bench_x3.json | cut -c1-600
