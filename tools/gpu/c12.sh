cd $GRAFT_REPO_ROOT
O=gpurun_out/c12; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_x3.py tests/test_gpu_r2.py -m gpu -q -k "mha or fused or nearest or rq or transformer or attention" 2>&1 | tail -12 ) > $O/tests.log; tail -4 $O/tests.log
timeout 300 python tools/bench_micro.py --iters 10 > $O/micro.jsonl 2> $O/micro.err; grep -E "rq|mha" $O/micro.jsonl | head -c 1500
timeout 900 python bench.py --steps 12 --warmup 2 --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err; head -c 230 $O/bench_default.json; echo
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o x3 -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-roofline > $GRAFT_REPO_ROOT/$O/prof_bench.json 2> $GRAFT_REPO_ROOT/$O/prof.err
cd $GRAFT_REPO_ROOT
DB=$(find $O/prof -name "*results.db" | head -1)
python tools/rocpd_stats.py "$DB" $O/x3_kernel_stats.csv @16 >> $O/prof.err 2>&1
rm -rf $O/prof
head -25 $O/x3_kernel_stats.csv
