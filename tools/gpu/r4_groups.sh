#!/bin/bash
# residual blocks over frame groups (PGT_BLOCK_GROUP_MIB): bit-equality test, micro-benchmark, same-box A/B of the bench line
set -u
O=gpurun_out; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_ops.py -q -k "frame_groups" 2>&1 | tail -4
timeout 400 python tools/bench_block_groups.py > $O/r4g_block_groups.jsonl 2> $O/r4g_block_groups.err; tail -3 $O/r4g_block_groups.err
cat $O/r4g_block_groups.jsonl
for M in 0 48 96 0 48; do
  PGT_BLOCK_GROUP_MIB=$M timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline --resident 2>> $O/r4g_bench.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('MIB=$M', d['value'], d['ms_per_step'])" | tee -a $O/r4g_bench_ab.txt
done
tail -3 $O/r4g_bench.err
