"""Micro-benchmark of TDResnetBlock over frame groups (PGT_BLOCK_GROUP_MIB): the decoder's residual
blocks at the model's shapes, whole-tensor against groups of several sizes, each replayed from a HIP graph.  One JSON line per
(shape, group size).  Usage: python tools/bench_block_groups.py [--mib 0 24 48 96 160] [--iters 6]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pgtformer_amd import ops  # noqa: E402
from pgtformer_amd.modules import rstt_layers as RL  # noqa: E402

SHAPES = [(96, 128, 128, 256, 256), (96, 64, 64, 256, 256), (96, 256, 256, 128, 128), (48, 256, 256, 256, 128),
          (96, 32, 32, 512, 512), (32, 512, 512, 64, 64)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mib", type=float, nargs="*", default=[0, 24, 48, 96, 160])
    ap.add_argument("--iters", type=int, default=6)
    ap.add_argument("--shapes", type=int, nargs="*", default=list(range(len(SHAPES))))
    a = ap.parse_args()
    dev, H = "cuda", torch.float16
    gen = torch.Generator(device=dev).manual_seed(0)
    for si in a.shapes:
        n, h, w, cin, cout = SHAPES[si]
        torch.manual_seed(si)
        blk = RL.TDResnetBlock(in_channels=cin, out_channels=cout)
        for p_ in blk.parameters():
            torch.nn.init.normal_(p_, std=0.03)
        blk.prepare(dev, H)
        x = torch.randn((n, h, w, cin), device=dev, dtype=H, generator=gen)
        base = None
        ref = None
        for mib in a.mib:
            RL.BLOCK_GROUP_MIB = mib
            per = RL._chunk_frames(x, max(cin, cout))
            if mib > 0 and per is None:
                continue
            y = blk(x, gn_next=True)                       # eager warm-up (allocations, autotune-free)
            torch.cuda.synchronize()
            if ref is None:
                ref = y.clone()
            equal = bool(torch.equal(y, ref))
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=s):
                    y = blk(x, gn_next=True)
                for _ in range(2):
                    g.replay()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(s)
                for _ in range(a.iters):
                    g.replay()
                e1.record(s)
            e1.synchronize()
            us = e0.elapsed_time(e1) / a.iters * 1e3
            base = us if mib == 0 else base
            flops = 2.0 * n * h * w * 9 * (cin * cout + cout * cout) + (2.0 * n * h * w * cin * cout if cin != cout else 0)
            print(json.dumps({"shape_NHWCinCout": [n, h, w, cin, cout], "group_mib": mib, "frames_per_group": per or n,
                              "us": round(us, 1), "tflops": round(flops / us / 1e6, 1),
                              "vs_whole": None if not base else round(us / base, 4), "equal_to_whole": equal}), flush=True)
            del g
    return 0


if __name__ == "__main__":
    sys.exit(main())
