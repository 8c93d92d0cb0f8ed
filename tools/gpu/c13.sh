cd $GRAFT_REPO_ROOT
O=gpurun_out/c13; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_x3.py tests/test_gpu_r2.py -m gpu -q -k "mha or fused or nearest or rq or transformer or attention" 2>&1 | tail -12 ) > $O/tests.log; tail -4 $O/tests.log
timeout 300 python tools/bench_micro.py --iters 10 > $O/micro.jsonl 2> $O/micro.err; grep -E "rq_lookup.*bfloat16.*true" $O/micro.jsonl | cut -c1-200
