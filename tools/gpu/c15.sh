cd $GRAFT_REPO_ROOT
O=gpurun_out/c15; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_r2.py tests/test_gpu_r2b.py -m gpu -q 2>&1 | tail -6 ) > $O/tests.log; tail -3 $O/tests.log
( timeout 600 python -m pytest tests/test_gpu_model.py -m gpu -q -k "stage1 or default_mode or public" 2>&1 | tail -6 ) > $O/tests2.log; tail -3 $O/tests2.log
timeout 300 python tools/bench_micro.py --iters 10 > $O/micro.jsonl 2> $O/micro.err; grep -E '"rq_lookup".*bfloat16.*true' $O/micro.jsonl | cut -c1-200
