#!/usr/bin/env python
"""Bandwidth of the GroupNorm passes (statistics, normalise+activation) per feature-map shape, next to a plain
device copy of the same tensor: python tools/bench_norms.py [--batch 4]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pgtformer_amd import ops  # noqa: E402


def timeit(fn, iters=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    a = ap.parse_args()
    n = 3 * a.batch
    print(f"{'shape':22s} {'MB':>7s} | stats us  GB/s | affine+silu us  GB/s | copy us  GB/s")
    for h, c in ((512, 64), (256, 128), (128, 256), (64, 256), (32, 512)):
        x = torch.randn((n, h, h, c), device="cuda").to(torch.bfloat16)
        y = torch.empty_like(x)
        g = torch.ones(c, device="cuda")
        b = torch.zeros(c, device="cuda")
        mb = x.numel() * 2 / 1e6
        sc, sh = ops.groupnorm_affine(x, g, b)
        t1 = timeit(lambda: ops.groupnorm_affine(x, g, b))
        t2 = timeit(lambda: ops.affine_act(x, sc, sh, act=ops.ACT_SILU, out=y))
        t3 = timeit(lambda: y.copy_(x))
        print(f"({n},{h},{h},{c})".ljust(22) + f" {mb:7.1f} | {t1:8.1f} {mb / t1 * 1e3 / 1e3:5.0f} | {t2:8.1f} {2 * mb / t2:11.0f} | {t3:7.1f} {2 * mb / t3:5.0f}")


if __name__ == "__main__":
    main()
