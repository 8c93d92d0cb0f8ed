cd $GRAFT_REPO_ROOT
for s in "48 128 128 256 256 3 5 256" "48 256 256 128 128 3 5 128" "48 64 64 256 256 3 5 256" "1 1 49152 512 1024 1 5 256"; do
  tools/_probe/r2p96 $s | grep -v split
done
python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "phased" 2>&1 | tail -30
