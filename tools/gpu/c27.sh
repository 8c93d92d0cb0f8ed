cd $GRAFT_REPO_ROOT
O=gpurun_out/c27; mkdir -p $O
cp profiles/r2_v9_autotune_table_b16.json $O/tune.json
export PGT_AUTOTUNE_CACHE=$GRAFT_REPO_ROOT/$O/tune.json
for i in 1 2; do
timeout 900 python bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-roofline > $O/bench$i.json 2> $O/bench$i.err; head -c 120 $O/bench$i.json; echo
timeout 900 python bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-roofline --lanes 1 > $O/bench_l1_$i.json 2> $O/bench_l1_$i.err; head -c 120 $O/bench_l1_$i.json; echo
done
timeout 900 python bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-roofline --precision bf16 > $O/bench_bf16.json 2> $O/bench_bf16.err; head -c 120 $O/bench_bf16.json; echo
