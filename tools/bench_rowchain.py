"""Micro-benchmark of the fused token-row chains (rowchain.hip) against the launches they replace, at the model's shapes:
1 572 864 rows (96 frames of 128 x 128 tokens) and 393 216 rows (64 x 64), C = 256, half.  Prints one JSON line per case.
Usage: python tools/bench_rowchain.py [--rows N] [--iters K]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pgtformer_amd import ops  # noqa: E402


def timeit(fn, iters):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3      # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, nargs="*", default=[1572864, 393216])
    ap.add_argument("--iters", type=int, default=10)
    a = ap.parse_args()
    dev, H = "cuda", torch.float16
    g = torch.Generator(device=dev).manual_seed(0)
    for rows in a.rows:
        frames = rows // 16384 if rows % 16384 == 0 else 0
        x = torch.randn((rows, 256), device=dev, dtype=H, generator=g)
        sc = torch.randn((rows, 256), device=dev, dtype=H, generator=g)
        wq = (0.06 * torch.randn((768, 256), device=dev, generator=g)).to(H)
        w3 = (0.06 * torch.randn((768, 256), device=dev, generator=g)).to(H)
        bq = torch.randn((frames, 768), device=dev, generator=g) if frames else torch.randn((768,), device=dev, generator=g)
        bp = torch.randn((frames, 256), device=dev, generator=g) if frames else torch.randn((256,), device=dev, generator=g)
        b1, b2 = torch.randn((256,), device=dev, generator=g), torch.randn((256,), device=dev, generator=g)
        ones, zeros = torch.ones(256, device=dev), torch.zeros(256, device=dev)
        qkv = torch.empty((rows, 768), device=dev, dtype=H)
        out = torch.empty((rows, 256), device=dev, dtype=H)
        res = {"rows": rows, "variant_ln": os.environ.get("PGT_RC_LN", "auto")}
        res["ln_linear_us"] = timeit(lambda: ops.ln_linear(x, wq, bq, out=qkv), a.iters)
        res["ln_linear_tbs"] = rows * 2048 / res["ln_linear_us"] / 1e6
        res["ln_linear_tflops"] = 2.0 * rows * 256 * 768 / res["ln_linear_us"] / 1e6

        def unfused_a():
            ln = ops.layernorm(x, ones, zeros)
            ops.linear(ln, wq, bq, out=qkv)
        res["layernorm+qkv_us"] = timeit(unfused_a, a.iters)
        res["proj_mlp_us"] = timeit(lambda: ops.attn_proj_mlp(x, sc, w3, bp, b1, b2, out=out), a.iters)
        res["proj_mlp_tbs"] = rows * 1536 / res["proj_mlp_us"] / 1e6
        res["proj_mlp_tflops"] = 6.0 * rows * 256 * 256 / res["proj_mlp_us"] / 1e6
        wp, wf1, wf2 = w3[:256].contiguous(), w3[256:512].contiguous(), w3[512:].contiguous()

        def unfused_c():
            x1 = ops.linear(x, wp, bp, res=sc)
            ln = ops.layernorm(x1, ones, zeros)
            h = ops.linear(ln, wf1, b1, act=ops.ACT_GELU)
            ops.linear(h, wf2, b2, res=x1, out=out)
        res["proj+ln+fc1+fc2_us"] = timeit(unfused_c, a.iters)
        print(json.dumps({k: (round(v, 2) if isinstance(v, float) else v) for k, v in res.items()}), flush=True)
        del x, sc, qkv, out
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
