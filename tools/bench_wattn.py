"""Window attention at the model's shapes: B = 32 windows of T = 3 frames, C = 256, 8 heads, 4x4 windows, on 128x128 and 64x64 maps,
half and split-half rows, un-shifted and shifted.  Prints one JSON line per case (us, algorithmic TB/s)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pgtformer_amd import ops  # noqa: E402


def timeit(fn, iters=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


g = torch.Generator(device="cuda").manual_seed(0)
bias = 0.02 * torch.randn((8, 48, 48), device="cuda", generator=g)
for hw in (128, 64):
    rows = 32 * 3 * hw * hw
    for x3 in (False, True):
        qkv = torch.randn((rows, 768 * (2 if x3 else 1)), device="cuda", dtype=torch.float16, generator=g)
        for shift in ((0, 0), (2, 2)):
            us = timeit(lambda: ops.window_attention(qkv, bias, 32, 3, hw, hw, 256, 8, (4, 4), shift, x3=x3))
            nbytes = rows * (768 + 256) * 2 * (2 if x3 else 1)
            print(json.dumps({"map": hw, "x3": x3, "shift": shift, "us": round(us, 1), "tbs": round(nbytes / us / 1e6, 2),
                              "variant": os.environ.get("WA_TAG", "")}), flush=True)
        del qkv
