#!/usr/bin/env python
"""Probe: B windows per forward (B independent 3-frame windows stacked on the frame axis) — equality with
B separate forwards, and ms per window under HIP-graph replay."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pgtformer_amd.driver import load_architecture  # noqa: E402
from pgtformer_amd.synth import make_clip  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
model = load_architecture(prec)
lq, _ = make_clip(8, 512, seed=1234)
lq = torch.from_numpy(lq).cuda()
wins = [lq[i:i + 3] for i in range(4)]
outs1 = [model.forward_nhwc(w, w=1.0)[0].float().clone() for w in wins]
for B in (1, 2, 4):
    x = torch.cat(wins[:B], 0).contiguous()
    out = model.forward_nhwc(x, w=1.0)[0].float()
    err = max((out[3 * b:3 * b + 3] - outs1[b]).abs().max().item() for b in range(B))
    static = x.clone()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        model.forward_nhwc(static, w=1.0)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        o = model.forward_nhwc(static, w=1.0)[0]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        g.replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print(f"B={B}: max|batched - separate| = {err:.3e}; {dt * 1e3:.2f} ms/forward = {dt * 1e3 / B:.2f} ms/window "
          f"-> {B / dt:.1f} frames/s")
