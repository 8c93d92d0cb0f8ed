cd $GRAFT_REPO_ROOT
O=gpurun_out/c16; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o x3 -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-roofline --resident > $GRAFT_REPO_ROOT/$O/prof_bench.json 2> $GRAFT_REPO_ROOT/$O/prof.err
cd $GRAFT_REPO_ROOT
DB=$(find $O/prof -name "*results.db" | head -1)
python tools/rocpd_gaps.py "$DB" $O/gaps.json
rm -rf $O/prof
