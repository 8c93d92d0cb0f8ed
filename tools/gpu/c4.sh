cd $GRAFT_REPO_ROOT
O=gpurun_out/c4; mkdir -p $O
( timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -70 ) > $O/tests.log
for f in parity_model parity_x3 parity_r2 parity_ops; do cp gpurun_out/$f.json $O/ 2>/dev/null; done
export PGT_AUTOTUNE_CACHE=$GRAFT_REPO_ROOT/$O/tune.json
PGT_DUMP_SHAPES=$O/shapes_x3.txt timeout 900 python bench.py --steps 12 --warmup 2 --no-cpu-baseline > $O/bench_x3.json 2> $O/bench_x3.err
unset PGT_AUTOTUNE_CACHE
PGT_EPILOGUE_GN=0 timeout 900 python bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-roofline > $O/bench_x3_nogn.json 2> $O/bench_x3_nogn.err
timeout 900 python bench.py --steps 12 --warmup 2 --no-cpu-baseline --precision bf16 > $O/bench_bf16.json 2> $O/bench_bf16.err
PGT_EPILOGUE_GN=0 timeout 900 python bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-roofline --precision bf16 > $O/bench_bf16_nogn.json 2> $O/bench_bf16_nogn.err
export PGT_AUTOTUNE_CACHE=$GRAFT_REPO_ROOT/$O/tune.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o x3 -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-roofline --resident > $GRAFT_REPO_ROOT/$O/prof_bench.json 2> $GRAFT_REPO_ROOT/$O/prof.err
cd $GRAFT_REPO_ROOT
DB=$(find $O/prof -name "*results.db" | head -1)
python tools/rocpd_stats.py "$DB" $O/x3_kernel_stats.csv @16 >> $O/prof.err 2>&1
rm -rf $O/prof
unset PGT_AUTOTUNE_CACHE
timeout 300 python tools/bench_micro.py --iters 10 > $O/micro.jsonl 2> $O/micro.err
tail -n 8 $O/tests.log; for f in bench_x3 bench_x3_nogn bench_bf16 bench_bf16_nogn; do head -c 200 $O/$f.json; echo; done; grep rq_lookup $O/micro.jsonl | grep 262144 | head -4
