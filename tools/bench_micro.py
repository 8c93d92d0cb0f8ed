#!/usr/bin/env python
"""BASELINE.json configs 4 and 5 micro-benchmarks.

  config 5  window attention: (nW, N, C) sweep — shipping 3x4x4 windows (N=48; C=256/512) and the
            3x8x8 / C=512 point (N=192) — bf16, shifted and un-shifted; reports us, algorithmic GB/s
            (qkv read + out write) against the 8 TB/s HBM peak and TFLOP/s of QK^T+PV.
  config 4  nearest-code lookup (RQ-VAE): tokens X~N(0,1) (Ntok,512), codebook (1024,512), depth 1 and 4;
            distance GEMM (MFMA) + arg-min (wave reduction) + gather/residual update; reports us and
            tokens/s, checks lowest-index tie-breaking with duplicated codebook rows.
Prints one JSON line per case.   python tools/bench_micro.py [--iters 20]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pgtformer_amd import ops  # noqa: E402


def timeit(fn, iters):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def window_attention3d(iters):
    """BASELINE.json configs[4]: Video-Swin windows 3x8x8 (N = 192), C = 512, fp16 and bf16 MFMA, shifted / un-shifted,
    nW sweep (modules/swin.py parametrisation: pgt_window_attention3d)."""
    heads, c, win = 8, 512, (3, 8, 8)
    n = win[0] * win[1] * win[2]
    for dt in (torch.float16, torch.bfloat16):
        for (d, h, w) in [(3, 64, 64), (3, 128, 128), (3, 256, 256), (6, 128, 128)]:
            nw = (d // win[0]) * (h // win[1]) * (w // win[2])
            qkv = torch.randn((d * h * w, 3 * c), device="cuda").to(dt)
            bias = (0.02 * torch.randn((heads, n, n), device="cuda")).float()
            for shift in ((0, 0, 0), (0, 4, 4)) + (((1, 4, 4),) if d > 3 else ()):
                us = timeit(lambda: ops.window_attention3d(qkv, bias, 1, d, h, w, c, heads, win, shift), iters)
                byts = d * h * w * c * 2 * 4
                flops = 4.0 * n * n * c * nw
                print(json.dumps({"bench": "window_attention3d", "dtype": str(dt).replace("torch.", ""), "nW": nw, "N": n, "C": c,
                                  "shift": list(shift), "us": round(us, 1), "GBps": round(byts / us / 1e3, 1),
                                  "hbm_frac": round(byts / us / 1e3 / 8000, 3), "TFLOPs": round(flops / us / 1e6, 1)}))


def window_attention(iters):
    dt = torch.bfloat16
    heads = 8
    for (t, h, w, c, win) in [(3, 128, 128, 256, (4, 4)), (3, 64, 64, 256, (4, 4)), (3, 32, 32, 512, (4, 4)),
                              (3, 64, 64, 512, (8, 8)), (3, 128, 128, 512, (8, 8)), (3, 256, 256, 512, (8, 8))]:
        n = t * win[0] * win[1]
        nw = (h // win[0]) * (w // win[1])
        qkv = torch.randn((t * h * w, 3 * c), device="cuda").to(dt)
        bias = (0.02 * torch.randn((heads, n, n), device="cuda")).float()
        for shift in ((0, 0), (win[0] // 2, win[1] // 2)):
            us = timeit(lambda: ops.window_attention(qkv, bias, 1, t, h, w, c, heads, win, shift), iters)
            byts = t * h * w * c * 2 * 4
            flops = 4.0 * n * n * c * nw
            print(json.dumps({"bench": "window_attention", "dtype": "bf16", "nW": nw, "N": n, "C": c, "shift": list(shift),
                              "us": round(us, 1), "GBps": round(byts / us / 1e3, 1), "hbm_frac": round(byts / us / 1e3 / 8000, 3),
                              "TFLOPs": round(flops / us / 1e6, 1)}))


def rq_lookup(iters):
    torch.manual_seed(0)
    book = torch.randn((1025, 512), device="cuda")
    book[1024] = 0
    book[700] = book[13]          # duplicated rows: arg-min must return the lowest index
    enorm = book[:-1].pow(2).sum(1).contiguous()
    for dt in (torch.bfloat16, torch.float32):
        book_t = book[:-1].to(dt).contiguous()
        for ntok in (3072, 32768, 262144):
            x = torch.randn((ntok, 512), device="cuda").to(dt)
            x[5] = book[700].to(dt)
            for depth in (1, 4):
                for fused in ((True, False) if dt == torch.bfloat16 else (False,)):
                    def run():
                        resid = x.clone() if depth > 1 else x
                        agg = torch.empty_like(x)
                        codes = None
                        for i in range(depth):
                            if fused:      # arg-min inside the distance GEMM (csrc/rq_nearest.hip)
                                codes = ops.rq_nearest(resid, book_t, ops.row_sumsq(resid), enorm)
                            else:          # distance GEMM with fp32 output + row arg-min
                                dot = ops.linear(resid, book_t, None, out_f32=True)
                                codes = ops.rq_argmin(dot, ops.row_sumsq(resid), enorm)
                            ops.embed_rows(book, codes, dt, out=agg, accumulate=i > 0, resid=resid if depth > 1 else None)
                        return codes
                    us = timeit(run, max(2, iters // (1 + ntok // 65536)))
                    codes = run() if depth == 1 else None
                    # the search kernel(s) alone: |x|^2, the gather of the chosen rows and the residual update excluded
                    xn = ops.row_sumsq(x)
                    if fused:
                        k_us = timeit(lambda: ops.rq_nearest(x, book_t, xn, enorm), max(2, iters // (1 + ntok // 65536)))
                    else:
                        k_us = timeit(lambda: ops.rq_argmin(ops.linear(x, book_t, None, out_f32=True), xn, enorm),
                                      max(2, iters // (1 + ntok // 65536)))
                    tie_ok = bool(codes[5].item() == 13) if codes is not None else None
                    flops = 2.0 * ntok * 1024 * 512 * depth
                    # algorithmic HBM bytes: x in + quantised out (+ residual r/w at depth > 1) + codes + codebook
                    es = 2 if dt == torch.bfloat16 else 4
                    byts = depth * (ntok * 512 * es * (2 if depth == 1 else 4) + ntok * 4) + 1024 * 512 * es
                    print(json.dumps({"bench": "rq_lookup", "dtype": str(dt).replace("torch.", ""), "Ntok": ntok, "depth": depth,
                                      "fused_argmin": fused, "us": round(us, 1), "search_kernels_us_per_level": round(k_us, 1),
                                      "search_mfma_frac": round(2.0 * ntok * 1024 * 512 / k_us / 1e6 / (2500 if es == 2 else 157.3), 3),
                                      "Mtok_per_s": round(ntok / us, 2),
                                      "TFLOPs": round(flops / us / 1e6, 1), "mfma_frac": round(flops / us / 1e6 / (2500 if es == 2 else 157.3), 3),
                                      "algorithmic_GBps": round(byts / us / 1e3, 1), "lowest_index_tie": tie_ok}))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    window_attention(a.iters)
    window_attention3d(a.iters)
    rq_lookup(a.iters)
