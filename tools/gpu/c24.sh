cd $GRAFT_REPO_ROOT
O=gpurun_out/c25; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_x3.py -m gpu -q -k "mha or transformer or sa_layer" 2>&1 | tail -6 ) > $O/tests.log; tail -3 $O/tests.log
( timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q -k "default_mode or other_weights or f32_matches or golden" 2>&1 | tail -6 ) > $O/tests2.log; tail -3 $O/tests2.log
cp profiles/r2_v8_autotune_table_b16.json $O/tune.json
export PGT_AUTOTUNE_CACHE=$GRAFT_REPO_ROOT/$O/tune.json
timeout 900 python bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-roofline > $O/bench.json 2> $O/bench.err; head -c 120 $O/bench.json; echo
timeout 900 python bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-roofline --lanes 1 > $O/bench_l1.json 2> $O/bench_l1.err; head -c 120 $O/bench_l1.json; echo
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o x3 -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-roofline --lanes 1 > /dev/null 2> $GRAFT_REPO_ROOT/$O/prof.err
cd $GRAFT_REPO_ROOT
python tools/rocpd_stats.py "$(find $O/prof -name '*results.db' | head -1)" $O/stats.csv @16 > /dev/null 2>&1
rm -rf $O/prof
grep -E "mha" $O/stats.csv | cut -c1-60,190-
