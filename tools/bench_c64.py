#!/usr/bin/env python
"""Timing of the 64-channel 3x3 conv kernel (igemm6) at the model's full-resolution shape: python tools/bench_c64.py [N]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pgtformer_amd import ops  # noqa: E402


def timeit(fn, it=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3


n = int(sys.argv[1]) if len(sys.argv) > 1 else 48
x = torch.randn((n, 512, 512, 64), device="cuda").to(torch.bfloat16)
w = (torch.randn((64, 576), device="cuda") / 24).to(torch.bfloat16)
b = torch.zeros(64, device="cuda")
res = torch.randn_like(x)
kw = dict(kh=3, kw=3, pad=(1, 1, 1, 1))
gb = x.numel() * 2 / 1e9
for name, kws, traffic in (("v6 + residual", dict(res=res, kernel=6), 3 * gb), ("v6", dict(kernel=6), 2 * gb),
                           ("v1 + residual", dict(res=res, kernel=1), 3 * gb)):
    t = timeit(lambda: ops.conv2d(x, w, b, **kws, **kw))
    print(f"{name:14s} {t:8.1f} us  {2.0 * x.numel() * 576 / t / 1e6:6.1f} TFLOP/s  {traffic / t * 1e3:5.2f} TB/s")
