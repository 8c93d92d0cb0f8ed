#!/usr/bin/env python
"""Timing of the 64-channel 3x3 conv kernels at the model's full-resolution shape (N x 512 x 512 x 64, half):
    python tools/bench_c64.py [N] [out.jsonl]
igemm6 (three halo images per 128-pixel tile) against igemm8 (ring of row images, register epilogue; round 5), plain, with a
residual, with the GroupNorm apply + SiLU as a separate pass and fused into the operand load (pgt_conv2d_affine_in), and the
64 -> 3 output conv.  Rates: algorithmic FLOP / time; bytes = input once + output once (+ residual, + the apply pass's round trip)."""
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


n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
out_path = sys.argv[2] if len(sys.argv) > 2 else None
dt = torch.float16
x = torch.randn((n, 512, 512, 64), device="cuda").to(dt)
w = (torch.randn((64, 576), device="cuda") / 24).to(dt)
w3 = (torch.randn((3, 576), device="cuda") / 24).to(dt)
b = torch.zeros(64, device="cuda")
b3 = torch.zeros(3, device="cuda")
fb = torch.zeros((n, 64), device="cuda")
res = torch.randn_like(x)
sc, sh = torch.rand((n, 64), device="cuda") + 0.5, torch.randn((n, 64), device="cuda") * 0.1
kw = dict(kh=3, kw=3, pad=(1, 1, 1, 1))
gb = x.numel() * 2 / 1e9
tmp = torch.empty_like(x)


def two_pass(kernel, **k):
    ops.affine_act(x, sc, sh, ops.ACT_SILU, out=tmp)
    return ops.conv2d(tmp, w, b, kernel=kernel, **k, **kw)


cases = [
    ("v6", lambda: ops.conv2d(x, w, b, kernel=6, **kw), 2 * gb, 64),
    ("v8 ring", lambda: ops.conv2d(x, w, b, kernel=8, **kw), 2 * gb, 64),
    ("v6 + residual", lambda: ops.conv2d(x, w, b, res=res, kernel=6, **kw), 3 * gb, 64),
    ("v8 + residual", lambda: ops.conv2d(x, w, b, res=res, kernel=8, **kw), 3 * gb, 64),
    ("v8 + residual + frame bias", lambda: ops.conv2d(x, w, fb, res=res, kernel=8, **kw), 3 * gb, 64),
    ("apply pass alone", lambda: ops.affine_act(x, sc, sh, ops.ACT_SILU, out=tmp), 2 * gb, 0),
    ("apply + v6 + residual", lambda: two_pass(6, res=res), 5 * gb, 64),
    ("apply + v8 + residual", lambda: two_pass(8, res=res), 5 * gb, 64),
    ("v8 fused apply + residual", lambda: ops.conv2d(x, w, b, res=res, affine_in=(sc, sh, ops.ACT_SILU), **kw), 3 * gb, 64),
    ("v8 fused apply", lambda: ops.conv2d(x, w, b, affine_in=(sc, sh, ops.ACT_SILU), **kw), 2 * gb, 64),
    ("v6 64->3 fp32 out", lambda: ops.conv2d(x, w3, b3, out_f32=True, kernel=6, **kw), gb, 3),
    ("v8 64->3 fp32 out", lambda: ops.conv2d(x, w3, b3, out_f32=True, kernel=8, **kw), gb, 3),
    ("v8 fused apply 64->3 fp32 out", lambda: ops.conv2d(x, w3, b3, out_f32=True, affine_in=(sc, sh, ops.ACT_SILU), **kw), gb, 3),
]
rows = []
for name, fn, traffic, cout in cases:
    t = timeit(fn)
    fl = 2.0 * n * 512 * 512 * 64 * 9 * cout
    print(f"{name:32s} {t:8.1f} us  {fl / t / 1e6:7.1f} TFLOP/s  {traffic / t * 1e3:5.2f} TB/s", flush=True)
    rows.append({"case": name, "n": n, "us": round(t, 1), "tflops": round(fl / t / 1e6, 1), "tb_s": round(traffic / t * 1e3, 2)})
if out_path:
    with open(out_path, "a") as f:
        for r in rows:
            f.write(json.dumps(r) + "\n")
