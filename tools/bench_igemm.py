#!/usr/bin/env python
"""Micro-benchmark of the implicit-GEMM conv kernel over the layer shapes of one PGTFormer window:
times every workgroup tile (BM x BN) and both epilogue paths per shape with events on the launch
stream, prints TFLOP/s and the best configuration (used to tune the dispatch heuristic in igemm.hip).

    python tools/bench_igemm.py [--dtype bf16|f32] [--iters 20] > gpurun_out/igemm_shapes.txt
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pgtformer_amd import ops  # noqa: E402

# (name, N, H, W, Cin, Cout, k, stride, ups, count per window)
SHAPES = [
    ("c3 64>64 @512", 3, 512, 512, 64, 64, 3, 1, 0, 7),
    ("c3 128>64 @512", 3, 512, 512, 128, 64, 3, 1, 0, 1),
    ("c3 128>128 ups@512", 3, 256, 256, 128, 128, 3, 1, 1, 1),
    ("c3 128>128 @256", 3, 256, 256, 128, 128, 3, 1, 0, 12),
    ("c3 288>128 @256", 3, 256, 256, 288, 128, 3, 1, 0, 1),
    ("c3 256>256 @128", 3, 128, 128, 256, 256, 3, 1, 0, 20),
    ("c3 544>256 @128", 3, 128, 128, 544, 256, 3, 1, 0, 1),
    ("c3 256>256 @64", 3, 64, 64, 256, 256, 3, 1, 0, 18),
    ("c3 512>512 @32", 3, 32, 32, 512, 512, 3, 1, 0, 26),
    ("c3 1056>512 @32", 3, 32, 32, 1056, 512, 3, 1, 0, 1),
    ("lin 256>768 @49152", 1, 1, 49152, 256, 768, 1, 1, 0, 6),
    ("lin 256>256 @49152", 1, 1, 49152, 256, 256, 1, 1, 0, 18),
    ("lin 256>256 @12288", 1, 1, 12288, 256, 256, 1, 1, 0, 24),
    ("lin 512>512 @3072", 1, 1, 3072, 512, 512, 1, 1, 0, 50),
    ("lin 512>1536 @3072", 1, 1, 3072, 512, 1536, 1, 1, 0, 10),
    ("lin 512>1024 @3072", 1, 1, 3072, 512, 1024, 1, 1, 0, 19),
    ("lin 1024>512 @3072", 1, 1, 3072, 1024, 512, 1, 1, 0, 9),
    ("c3 64>3 @512", 3, 512, 512, 64, 3, 3, 1, 0, 1),
    ("c3 8>64 @512", 3, 512, 512, 8, 64, 3, 1, 0, 1),
    ("c7 8>64 s2 @512", 3, 512, 512, 8, 64, 7, 2, 0, 1),
]
TILES = [(64, 64), (64, 128), (128, 64), (128, 128)]


def timeit(fn, iters):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3  # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--batch", type=int, default=1, help="multiply the frame count N (windows per forward)")
    args = ap.parse_args()
    dt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    dev = "cuda"
    total_best = total_auto = 0.0
    print(f"{'shape':24s} {'GFLOP':>7s} | " + " ".join(f"{bm}x{bn:<3d}(vec/scalar us)".rjust(24) for bm, bn in TILES) + " | auto us  TF/s")
    for name, n, h, w, cin, cout, k, stride, ups, cnt in SHAPES:
        if n > 1:
            n *= args.batch
        else:
            w *= args.batch
        x = torch.randn((n, h, w, cin), device=dev).to(dt)
        wt = (torch.randn((cout, k * k * cin), device=dev) / (k * k * cin) ** 0.5).to(dt)
        b = torch.randn((cout,), device=dev)
        pad = (k // 2,) * 4
        hv, wv = (h * 2, w * 2) if ups else (h, w)
        ho, wo = (hv + 2 * (k // 2) - k) // stride + 1, (wv + 2 * (k // 2) - k) // stride + 1
        res = torch.randn((n, ho, wo, cout), device=dev).to(dt)
        flops = 2.0 * n * ho * wo * cout * k * k * cin
        cells = []
        best = 1e30
        for bm, bn in TILES:
            if args.quick and (bm, bn) == (128, 128):
                continue
            tv = timeit(lambda: ops.conv2d(x, wt, b, kh=k, kw=k, stride=stride, pad=pad, ups=bool(ups), res=res, tile=(bm, bn)), args.iters)
            ts = timeit(lambda: ops.conv2d(x, wt, b, kh=k, kw=k, stride=stride, pad=pad, ups=bool(ups), res=res, tile=(bm, bn), scalar_epi=True), args.iters)
            best = min(best, tv, ts)
            cells.append(f"{tv:9.1f}/{ts:9.1f}".rjust(24))
        v2 = ""
        if dt == torch.bfloat16 and cin % 64 == 0 and cout % 8 == 0:
            for bn in (64, 128):
                for nst in (2, 3, 4):   # for kernel=2 the BM slot of `tile` carries the LDS stage count
                    t2 = timeit(lambda: ops.conv2d(x, wt, b, kh=k, kw=k, stride=stride, pad=pad, ups=bool(ups), res=res, kernel=2, tile=(nst, bn)), args.iters)
                    best = min(best, t2)
                    v2 += f" v2/{bn}s{nst}={t2:6.1f}"
        cells.append(v2)
        v3 = ""
        if dt == torch.bfloat16 and cin % 64 == 0 and cout % 8 == 0:
            for bn in (256, 128):
                t4 = timeit(lambda: ops.conv2d(x, wt, b, kh=k, kw=k, stride=stride, pad=pad, ups=bool(ups), res=res, kernel=4, tile=(0, bn)), args.iters)
                best = min(best, t4)
                v3 += f" v4/{bn}={t4:6.1f}"
        if dt == torch.bfloat16 and cin % 64 == 0 and cout % 8 == 0 and stride == 1 and not ups and k == 3 and w >= 32:
            t5 = timeit(lambda: ops.conv2d(x, wt, b, kh=k, kw=k, stride=stride, pad=pad, res=res, kernel=5), args.iters)
            best = min(best, t5)
            v3 += f" v5/256={t5:6.1f}"
        if dt == torch.bfloat16 and cin == 64 and cout <= 64 and cout % 8 == 0 and stride == 1 and not ups and k == 3:
            t6 = timeit(lambda: ops.conv2d(x, wt, b, kh=k, kw=k, stride=stride, pad=pad, res=res, kernel=6), args.iters)
            best = min(best, t6)
            v3 += f" v6={t6:6.1f}"
        cells.append(v3)
        ta = timeit(lambda: ops.conv2d(x, wt, b, kh=k, kw=k, stride=stride, pad=pad, ups=bool(ups), res=res), args.iters)
        total_best += best * cnt
        total_auto += ta * cnt
        print(f"{name:24s} {flops / 1e9:7.2f} | " + " ".join(cells) + f" | {ta:8.1f} {flops / ta / 1e6:6.1f}")
    print(f"weighted per-window: auto {total_auto / 1e3:.2f} ms, best-of-tiles {total_best / 1e3:.2f} ms")


if __name__ == "__main__":
    main()
