cd $GRAFT_REPO_ROOT
O=gpurun_out/c23; mkdir -p $O
export PGT_AUTOTUNE_CACHE=$GRAFT_REPO_ROOT/$O/tune.json
for sw in "1 0" "2 1" "5 2" "7 1"; do
  set -- $sw
  timeout 900 python bench.py --gpus 1 --steps $1 --warmup $2 --no-cpu-baseline --no-roofline > $O/bench_s$1_w$2.json 2> $O/bench_s$1_w$2.err; echo "steps $1 warmup $2 rc=$?"; head -c 150 $O/bench_s$1_w$2.json; echo
done
MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --steps 4 --warmup 1 --no-cpu-baseline --no-roofline > $O/bench_torchrun.json 2> $O/bench_torchrun.err; echo "torchrun rc=$?"; tail -c 300 $O/bench_torchrun.json | head -c 200; echo
