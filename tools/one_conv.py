#!/usr/bin/env python
"""Run ONE conv shape repeatedly with a chosen kernel/tile (target for rocprofv3 --pmc runs).
usage: one_conv.py N H W Cin Cout k kernel bm bn [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pgtformer_amd import ops  # noqa: E402

n, h, w, cin, cout, k, kernel, bm, bn = [int(a) for a in sys.argv[1:10]]
iters = int(sys.argv[10]) if len(sys.argv) > 10 else 5
dt = torch.bfloat16
x = torch.randn((n, h, w, cin), device="cuda").to(dt)
wt = (torch.randn((cout, k * k * cin), device="cuda") / (k * k * cin) ** 0.5).to(dt)
b = torch.randn((cout,), device="cuda")
for _ in range(iters):
    y = ops.conv2d(x, wt, b, kh=k, kw=k, pad=(k // 2,) * 4, kernel=kernel, tile=(bm, bn))
torch.cuda.synchronize()
print("ok", tuple(y.shape))
