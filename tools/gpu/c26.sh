cd $GRAFT_REPO_ROOT
O=gpurun_out/c26; mkdir -p $O
( timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -12 ) > $O/tests.log; tail -4 $O/tests.log
for f in parity_model parity_x3 parity_r2 parity_r2b parity_ops; do cp gpurun_out/$f.json $O/ 2>/dev/null; done
export PGT_AUTOTUNE_CACHE=$GRAFT_REPO_ROOT/$O/tune.json
PGT_DUMP_SHAPES=$O/shapes_x3.txt timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; head -c 230 $O/bench_default.json; echo
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o x3 -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-roofline --lanes 1 > $GRAFT_REPO_ROOT/$O/prof_bench.json 2> $GRAFT_REPO_ROOT/$O/prof.err
cd $GRAFT_REPO_ROOT
DB=$(find $O/prof -name "*results.db" | head -1)
python tools/rocpd_stats.py "$DB" $O/x3_kernel_stats.csv @16 >> $O/prof.err 2>&1
rm -rf $O/prof
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  PGT_SIDE_STREAM=0 timeout 600 rocprofv3 --kernel-trace --pmc $C -d $GRAFT_REPO_ROOT/$O/pmc_$C -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-roofline --resident > /dev/null 2> $GRAFT_REPO_ROOT/$O/pmc_$C.err
done
cd $GRAFT_REPO_ROOT
python tools/pmc_traffic.py $(find $O/pmc_FETCH_SIZE -name "*results.db" | head -1) $(find $O/pmc_WRITE_SIZE -name "*results.db" | head -1) $O/igemm_traffic_pmc.json bf16x3 16 > $O/pmc.log 2>&1
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
unset PGT_AUTOTUNE_CACHE
timeout 900 python bench.py --steps 12 --warmup 2 --no-cpu-baseline --precision bf16 > $O/bench_bf16.json 2> $O/bench_bf16.err; head -c 230 $O/bench_bf16.json; echo
timeout 900 python bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-roofline --full-tail > $O/bench_x3_fulltail.json 2> $O/bench_x3_fulltail.err; head -c 230 $O/bench_x3_fulltail.json; echo
timeout 300 python tools/bench_micro.py --iters 10 > $O/micro.jsonl 2> $O/micro.err
head -c 400 $O/pmc.log
timeout 900 python bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-roofline --lanes 1 > $O/bench_x3_lanes1.json 2> $O/bench_x3_lanes1.err; head -c 230 $O/bench_x3_lanes1.json; echo
DBDIR=$O/prof_gaps
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$DBDIR -o x3 -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-roofline --lanes 1 --resident > /dev/null 2> $GRAFT_REPO_ROOT/$O/prof_gaps.err
cd $GRAFT_REPO_ROOT
python tools/rocpd_gaps.py "$(find $DBDIR -name '*results.db' | head -1)" $O/gaps.json > /dev/null
rm -rf $DBDIR
