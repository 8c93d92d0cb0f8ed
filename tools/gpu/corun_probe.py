#!/usr/bin/env python
"""Do an MFMA-bound conv (igemm4: one 8-wave workgroup per CU, 232-250 VGPRs) and an HBM-bound element-wise pass (affine_act: 31 VGPRs)
launched on two streams share the CUs, or does the second one only get the CUs the conv leaves free?  Times N launches of each alone
and both together (events around the whole batch on a third "join" stream).
    python tools/gpu/corun_probe.py [out.jsonl]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pgtformer_amd import ops  # noqa: E402

dt = torch.float16
CONVS = {"igemm4<2,4> (96,128^2,256->256,3x3)": ((96, 128, 128, 256), 256, 3),
         "igemm4<4,2> (96,256^2,128->128,3x3)": ((96, 256, 256, 128), 128, 3)}
y = torch.randn((32, 512, 512, 64), device="cuda").to(dt)
yo = torch.empty_like(y)
sc, sh = torch.rand((32, 64), device="cuda") + 0.5, torch.randn((32, 64), device="cuda") * 0.1
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()


def timed(fa, fb, n=10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        if fa:
            with torch.cuda.stream(sa):
                fa()
        if fb:
            with torch.cuda.stream(sb):
                fb()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for name, (shape, cout, k) in CONVS.items():
    x = torch.randn(shape, device="cuda").to(dt)
    w = ops.pack_conv_weight(torch.randn((cout, shape[3], k, k), device="cuda") / (shape[3] * k * k) ** 0.5, dt)
    b = torch.zeros(cout, device="cuda")
    conv = lambda: ops.conv2d(x, w, b, kh=k, kw=k, pad=(1, 1, 1, 1))          # noqa: E731
    apply_ = lambda: [ops.affine_act(y, sc, sh, ops.ACT_SILU, out=yo) for _ in range(3)]      # noqa: E731  (~1.1 ms of HBM-bound work)
    for f in (conv, apply_):
        with torch.cuda.stream(sa):
            f()
    ta, tb, tab = timed(conv, None), timed(None, apply_), timed(conv, apply_)
    rec = {"conv": name, "conv_alone_ms": round(ta, 3), "apply_alone_ms": round(tb, 3), "both_ms": round(tab, 3),
           "sum_ms": round(ta + tb, 3), "overlap_fraction_of_the_shorter": round((ta + tb - tab) / min(ta, tb), 3)}
    print(json.dumps(rec), flush=True)
    if len(sys.argv) > 1:
        with open(sys.argv[1], "a") as f:
            f.write(json.dumps(rec) + "\n")
