cd $GRAFT_REPO_ROOT
O=gpurun_out/c8; mkdir -p $O
( timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -12 ) > $O/tests.log; tail -5 $O/tests.log
for f in parity_model parity_x3 parity_r2 parity_ops; do cp gpurun_out/$f.json $O/ 2>/dev/null; done
export PGT_AUTOTUNE_CACHE=$GRAFT_REPO_ROOT/$O/tune.json
PGT_DUMP_SHAPES=$O/shapes_x3.txt timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; head -c 230 $O/bench_default.json; echo
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o x3 -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-roofline --resident > $GRAFT_REPO_ROOT/$O/prof_bench.json 2> $GRAFT_REPO_ROOT/$O/prof.err
cd $GRAFT_REPO_ROOT
DB=$(find $O/prof -name "*results.db" | head -1)
python tools/rocpd_stats.py "$DB" $O/x3_kernel_stats.csv @16 >> $O/prof.err 2>&1
rm -rf $O/prof
head -12 $O/x3_kernel_stats.csv | cut -c1-150
