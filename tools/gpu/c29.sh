cd $GRAFT_REPO_ROOT
O=gpurun_out/c29; mkdir -p $O
cp profiles/r2_v9_autotune_table_b16.json $O/tune.json
export PGT_AUTOTUNE_CACHE=$GRAFT_REPO_ROOT/$O/tune.json
for i in 1 2; do
PGT_SIDE_STREAM=0 timeout 900 python bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-roofline > $O/noside$i.json 2> $O/noside$i.err; echo "no side stream: $(head -c 75 $O/noside$i.json)"
timeout 900 python bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-roofline > $O/side$i.json 2> $O/side$i.err; echo "side stream:    $(head -c 75 $O/side$i.json)"
done
