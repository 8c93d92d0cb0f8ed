#!/usr/bin/env python
"""BASELINE.json configs[4]: "Swin window-attention microbench: 8x8 windows, T=3, C=512, fp16 MFMA, rocprof roofline sweep".
    rocprofv3 --kernel-trace --stats -d <dir> -o c -- python tools/config5_sweep.py            (the sweep under the profiler)
    python tools/config5_sweep.py --summarise <dir>/c_results.db profiles/r5_config5_sweep.csv (trace -> one row per nW)
nW in {64, 256, 512, 1024} windows of 3 x 8 x 8 = 192 tokens, C = 512, 8 heads, fp16, un-shifted and shifted (0, 4, 4): 10 launches
each after 3 warm-ups.  The summary takes the kernel durations of the trace in launch order: algorithmic bytes = the q | k | v rows
read once + the output rows written once (pgt_window_attention3d reads 3C and writes C halves per token), against 8 TB/s."""
import csv
import sqlite3
import sys

SWEEP = [(3, 64, 64), (3, 128, 128), (6, 128, 128), (3, 256, 256)]      # (D, H, W) token grids: 64 / 256 / 512 / 1024 windows
SHIFTS = [(0, 0, 0), (0, 4, 4)]
WARM, REPS = 3, 10
C, HEADS, WIN = 512, 8, (3, 8, 8)


def run():
    import os

    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from pgtformer_amd import ops
    n = WIN[0] * WIN[1] * WIN[2]
    bias = (0.02 * torch.randn((HEADS, n, n), device="cuda")).float()
    for (d, h, w) in SWEEP:
        qkv = torch.randn((d * h * w, 3 * C), device="cuda").to(torch.float16)
        for shift in SHIFTS:
            for _ in range(WARM + REPS):
                ops.window_attention3d(qkv, bias, 1, d, h, w, C, HEADS, WIN, shift)
            torch.cuda.synchronize()


def summarise(db, out):
    c = sqlite3.connect(db)
    rows = c.execute("select name, duration from kernels where name like '%window_attn%' order by start").fetchall()
    per = WARM + REPS
    assert len(rows) == per * len(SWEEP) * len(SHIFTS), (len(rows), per)
    with open(out, "w", newline="") as f:
        wr = csv.writer(f)
        wr.writerow(["kernel", "nW", "tokens", "C", "dtype", "shift", "launches", "avg_us", "min_us", "algorithmic_MB", "GB_per_s", "frac_of_8TBps", "TFLOP_per_s_QK_PV"])
        i = 0
        for (d, h, w) in SWEEP:
            nw = (d // WIN[0]) * (h // WIN[1]) * (w // WIN[2])
            for shift in SHIFTS:
                chunk = rows[i + WARM:i + per]
                i += per
                us = [r[1] / 1e3 for r in chunk]
                avg = sum(us) / len(us)
                byts = d * h * w * C * 2 * 4
                flops = 4.0 * 192 * 192 * C * nw
                wr.writerow([chunk[0][0][:60], nw, d * h * w, C, "fp16", "x".join(map(str, shift)), len(us), round(avg, 2), round(min(us), 2),
                             round(byts / 1e6, 2), round(byts / avg / 1e3, 1), round(byts / avg / 1e3 / 8000, 3), round(flops / avg / 1e6, 1)])
    print(open(out).read())


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--summarise":
        summarise(sys.argv[2], sys.argv[3])
    else:
        run()
