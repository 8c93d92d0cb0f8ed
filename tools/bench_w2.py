#!/usr/bin/env python
"""Timing of the exact-weight forms (pgt_conv_desc::w2: two weight planes, two MFMAs per product) against the
single-plane launches they replace, at the shapes of the decoder's 512 x 512 and 32 x 32 stages (32 windows per forward):
    python tools/bench_w2.py [out.jsonl]
Rates are ALGORITHMIC (every reference product once); an exact-weight launch executes two MFMAs per product."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pgtformer_amd import ops  # noqa: E402


def timeit(fn, it=10):
    fn()
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3


out_path = sys.argv[1] if len(sys.argv) > 1 else None
dt = torch.float16
# (name, N, H, W, Cin, Cout, k, residual, fused apply)
SHAPES = [("512^2 64->64 3x3 (ring)", 32, 512, 512, 64, 64, 3, True, True),
          ("512^2 64->3 3x3 conv_out (ring)", 32, 512, 512, 64, 3, 3, False, True),
          ("512^2 128->64 3x3", 32, 512, 512, 128, 64, 3, False, False),
          ("512^2 128->64 1x1 nin_shortcut", 32, 512, 512, 128, 64, 1, False, False),
          ("32^2 512->512 3x3", 96, 32, 32, 512, 512, 3, True, False),
          ("32^2 1088->512 3x3", 96, 32, 32, 1088, 512, 3, False, False),
          ("32^2 512->1024 3x3", 96, 32, 32, 512, 1024, 3, False, False),
          ("32^2 512->512 2x2 sub-pixel", 96, 32, 32, 512, 512, 2, False, False),
          ("tokens 98304 x 512->512", 1, 1, 98304, 512, 512, 1, True, False),
          ("tokens 98304 x 512->1536", 1, 1, 98304, 512, 1536, 1, False, False)]
for name, n, h, w_, cin, cout, k, with_res, fuse in SHAPES:
    x = torch.randn((n, h, w_, cin), device="cuda").to(dt)
    w4 = torch.randn((cout, cin, k, k), device="cuda") / (cin * k * k) ** 0.5
    b = torch.zeros(cout, device="cuda")
    pw1, pw2 = ops.pack_conv_weight(w4, dt), ops.pack_conv_weight(w4, dt, w2=True)
    pad = (k // 2, (k - 1) // 2, k // 2, (k - 1) // 2)
    kw = dict(kh=k, kw=k, pad=pad)
    res = torch.randn((n, h, w_, cout), device="cuda").to(dt) if with_res else None
    sc, sh = torch.rand((n, cin), device="cuda") + 0.5, torch.randn((n, cin), device="cuda") * 0.1
    fl = 2.0 * n * h * w_ * cin * cout * k * k
    rec = {"shape": name, "gflop": round(fl / 1e9, 1)}
    t1 = timeit(lambda: ops.conv2d(x, pw1, b, res=res, **kw))
    t2 = timeit(lambda: ops.conv2d(x, pw2, b, res=res, w2=cout, **kw))
    rec.update(single_us=round(t1, 1), exact_us=round(t2, 1), single_tflops=round(fl / t1 / 1e6, 1), exact_tflops=round(fl / t2 / 1e6, 1),
               exact_executed_tflops=round(2 * fl / t2 / 1e6, 1), ratio=round(t2 / t1, 3))
    if fuse:
        a = (sc, sh, ops.ACT_SILU)
        f1 = timeit(lambda: ops.conv2d(x, pw1, b, res=res, affine_in=a, **kw))
        f2 = timeit(lambda: ops.conv2d(x, pw2, b, res=res, affine_in=a, w2=cout, **kw))
        rec.update(fused_single_us=round(f1, 1), fused_exact_us=round(f2, 1))
    print(json.dumps(rec), flush=True)
    if out_path:
        with open(out_path, "a") as f:
            f.write(json.dumps(rec) + "\n")
    del x, res
    torch.cuda.empty_cache()
