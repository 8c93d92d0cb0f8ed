#!/usr/bin/env python
"""Timing of the code transformer's attention (mha_mfma_kernel, split-half rows): B windows of L = 3072 tokens, 8 heads x 64.
    python tools/bench_mha.py [B] [out.jsonl]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pgtformer_amd import ops  # noqa: E402
from tools.bench_micro import timeit  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
L, heads, hd = 3072, 8, 64
e = heads * hd
torch.manual_seed(1234)
qkv = (torch.randn((B * L, 6 * e), device="cuda") * 0.5).to(torch.float16)      # [hi q k v | lo q k v]
qkv[:, 3 * e:] *= 2.0 ** -11
q, k, v = qkv[:, :e], qkv[:, e:2 * e], qkv[:, 2 * e:3 * e]
us = timeit(lambda: ops.mha(q, k, v, B, L, heads, hd, hd ** -0.5, x3=(3 * e, 3 * e, 3 * e)), 10)
out = ops.mha(q, k, v, B, L, heads, hd, hd ** -0.5, x3=(3 * e, 3 * e, 3 * e))
fl = 4.0 * B * L * L * e
import hashlib
rec = {"B": B, "us": round(us, 1), "sha256_of_output": hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:16], "tflops_algorithmic": round(fl / us / 1e6, 1), "tflops_executed": round(3 * fl / us / 1e6, 1),
       "checksum": float(out.float().abs().sum()), "env": {k_: v_ for k_, v_ in os.environ.items() if k_.startswith("PGT_")}}
print(json.dumps(rec))
if len(sys.argv) > 2:
    with open(sys.argv[2], "a") as f:
        f.write(json.dumps(rec) + "\n")
