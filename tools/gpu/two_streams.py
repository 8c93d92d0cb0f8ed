"""Experiment: two forwards in flight (two HIP graphs of B windows each on two streams) against one graph at a time."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pgtformer_amd import PGTFormer, default_config
from pgtformer_amd.driver import WindowRunner
from pgtformer_amd.manifest import pgtformer_manifest
from pgtformer_amd.synth import make_clip
from pgtformer_amd.weightgen import generate_state_dict

dev = torch.device("cuda", 0)
cfg = default_config()
model = PGTFormer(**cfg)
model.load_state_dict(generate_state_dict(pgtformer_manifest(cfg), cfg, seed=0), strict=True)
model.prepare(dev, sys.argv[1] if len(sys.argv) > 1 else "bf16x3")
lq, _ = make_clip(8, 512, seed=1234)
NL = int(sys.argv[2]) if len(sys.argv) > 2 else 2
for B in (16,):
    clip = torch.from_numpy(np.concatenate([lq] * 4, 0)[:B + 2]).to(dev)
    rs = [WindowRunner(model, 1.0, True, 512, 512, batch=B) for _ in range(NL)]
    for r in rs:
        r.static_in.copy_(clip)
    ss = [torch.cuda.Stream(device=dev) for _ in range(NL)]
    torch.cuda.synchronize()

    def one(n):
        t0 = time.perf_counter()
        for _ in range(n):
            rs[0].graph.replay()
        torch.cuda.synchronize()
        return n * B / (time.perf_counter() - t0)

    def two(n):
        t0 = time.perf_counter()
        for _ in range(n):
            for r, s in zip(rs, ss):
                with torch.cuda.stream(s):
                    r.graph.replay()
        torch.cuda.synchronize()
        return NL * n * B / (time.perf_counter() - t0)

    one(2); two(2)
    a = one(10); b = two(6); a2 = one(10); b2 = two(6)
    ref = rs[0].static_res.clone(); 
    print(f"NL={NL} B={B}: one graph at a time {a:.1f} / {a2:.1f} fps; NL graphs on NL streams {b:.1f} / {b2:.1f} fps; "
          f"outputs equal {bool(torch.equal(rs[0].static_res, rs[1].static_res))}", flush=True)
    del rs
    torch.cuda.empty_cache()
