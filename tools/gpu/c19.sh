cd $GRAFT_REPO_ROOT
O=gpurun_out/c19; mkdir -p $O
cp profiles/r2_v7_autotune_table_b16.json $O/tune.json
export PGT_AUTOTUNE_CACHE=$GRAFT_REPO_ROOT/$O/tune.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o x3 -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-roofline --lanes 1 > $GRAFT_REPO_ROOT/$O/prof_bench.json 2> $GRAFT_REPO_ROOT/$O/prof.err
cd $GRAFT_REPO_ROOT
DB=$(find $O/prof -name "*results.db" | head -1)
python tools/rocpd_stats.py "$DB" $O/x3_kernel_stats_lanes1.csv @16 >> $O/prof.err 2>&1
python tools/rocpd_gaps.py "$DB" $O/gaps_lanes1.json
rm -rf $O/prof
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof2 -o x3 -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-roofline --lanes 2 > $GRAFT_REPO_ROOT/$O/prof_bench2.json 2> $GRAFT_REPO_ROOT/$O/prof2.err
cd $GRAFT_REPO_ROOT
DB=$(find $O/prof2 -name "*results.db" | head -1)
python tools/rocpd_gaps.py "$DB" $O/gaps_lanes2.json
rm -rf $O/prof2
