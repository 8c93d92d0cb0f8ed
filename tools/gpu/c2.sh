cd $GRAFT_REPO_ROOT
O=gpurun_out/c2; mkdir -p $O
rocminfo | grep -E "Marketing Name|Compute Unit|Max Clock" | head -8 > $O/box.txt; lscpu | grep -E "Model name|^CPU\(s\)|Core|Socket" >> $O/box.txt
( timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -70 ) > $O/tests.log
for f in parity_model parity_x3 parity_r2 parity_ops; do cp gpurun_out/$f.json $O/ 2>/dev/null; done
export PGT_AUTOTUNE_CACHE=$GRAFT_REPO_ROOT/$O/tune.json
PGT_DUMP_SHAPES=$O/shapes_x3.txt timeout 900 python bench.py --steps 20 --warmup 3 > $O/bench_default.json 2> $O/bench_default.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o x3 -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-roofline --resident > $GRAFT_REPO_ROOT/$O/prof_bench.json 2> $GRAFT_REPO_ROOT/$O/prof.err
cd $GRAFT_REPO_ROOT
DB=$(find $O/prof -name "*results.db" | head -1); echo "db=$DB" >> $O/prof.err
python tools/rocpd_stats.py "$DB" $O/x3_kernel_stats.csv @16 >> $O/prof.err 2>&1
rm -rf $O/prof
unset PGT_AUTOTUNE_CACHE
timeout 600 python tools/bench_micro.py --iters 10 > $O/micro.jsonl 2> $O/micro.err
timeout 400 python tools/cpu_threads_probe.py 32 64 > $O/cpu_threads.jsonl 2> $O/cpu_threads.err
tail -n 6 $O/tests.log; head -c 400 $O/bench_default.json; echo; head -n 12 $O/x3_kernel_stats.csv | cut -c1-160
