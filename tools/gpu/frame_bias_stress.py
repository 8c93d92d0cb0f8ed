#!/usr/bin/env python
"""Determinism stress of pgt_frame_bias's cross-workgroup hand-over (agent-scope relaxed atomics, no fences): the same call repeated
with L2-thrashing launches in between must return the same bits every time.   python tools/gpu/frame_bias_stress.py [repeats]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pgtformer_amd import ops  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
torch.manual_seed(0)
bad = 0
big = torch.randn((64, 256, 256, 128), device="cuda").half()
sc, sh = torch.rand((64, 128), device="cuda") + 0.5, torch.randn((64, 128), device="cuda") * 0.1
tmp = torch.empty_like(big)
for (n, h, w, k, cout, bands) in [(96, 128, 128, 256, 256, 16), (96, 32, 32, 512, 512, 2), (32, 512, 512, 64, 64, 16), (96, 64, 64, 1056, 512, 8)]:
    x = (torch.randn((n, h, w, k), device="cuda") + 0.25).half()
    d = torch.randn((k, cout), device="cuda") * 1e-3
    b = torch.randn((cout,), device="cuda")
    xb, nb = ops.banded(x, bands)
    want = ops.frame_bias(xb, d, b, scale_div=nb, sample_cells=ops.band_sample_cells(nb)).clone()
    torch.cuda.synchronize()
    mism = 0
    for i in range(reps):
        if i % 3 == 0:
            ops.affine_act(big, sc, sh, ops.ACT_SILU, out=tmp)          # thrash the L2s / the Infinity Cache
        got = ops.frame_bias(xb, d, b, scale_div=nb, sample_cells=ops.band_sample_cells(nb))
        if not torch.equal(got, want):
            mism += 1
            if mism <= 3:
                print("  mismatch at repeat", i, "max abs diff", float((got - want).abs().max()), "elements", int((got != want).sum()))
    print(f"frame_bias {n}x{h}x{w}x{k}->{cout} bands {nb}: {mism} of {reps} repeats differ", flush=True)
    bad += mism
print("TOTAL mismatches", bad)
sys.exit(1 if bad else 0)
