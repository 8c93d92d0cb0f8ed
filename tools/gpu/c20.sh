cd $GRAFT_REPO_ROOT
O=gpurun_out/c20; mkdir -p $O
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
for B in 8 24 32; do
  timeout 900 python bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-roofline --windows-per-forward $B > $O/bench_b$B.json 2> $O/bench_b$B.err; head -c 120 $O/bench_b$B.json; echo
done
