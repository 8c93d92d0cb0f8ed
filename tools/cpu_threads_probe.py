#!/usr/bin/env python
"""CPU-baseline thread-count probe (reported in BASELINE.md): the oracle (port of the reference's fp32 eager CPU path)
timed on this host at several torch thread counts, 1 warm-up + 1 timed window each.  python tools/cpu_threads_probe.py 32 64"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pgt_oracle as O  # noqa: E402
from pgtformer_amd import default_config  # noqa: E402
from pgtformer_amd.manifest import pgtformer_manifest  # noqa: E402
from pgtformer_amd.synth import make_clip  # noqa: E402
from pgtformer_amd.weightgen import generate_state_dict  # noqa: E402

cfg = default_config()
sd = generate_state_dict(pgtformer_manifest(cfg), cfg, seed=0)
lq, _ = make_clip(3, 512, seed=1234)
x = torch.from_numpy(lq.astype(np.float32) / np.float32(255.0)).permute(0, 3, 1, 2).contiguous()
for n in [int(a) for a in sys.argv[1:]] or [32, 64]:
    torch.set_num_threads(n)
    ts = []
    for _ in range(2):
        t0 = time.time()
        O.pgtformer_forward(sd, cfg, x, w=1.0)
        ts.append(time.time() - t0)
    print(json.dumps({"threads": n, "warm_s": round(ts[0], 2), "timed_s": round(ts[1], 2), "fps": round(1 / ts[1], 4)}), flush=True)
