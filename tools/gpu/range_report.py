"""Which layers reach the IEEE-half saturation limit at the fitted-tail operating point (PGTFormer.check_range)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pgtformer_amd import PGTFormer, default_config  # noqa: E402
from pgtformer_amd.manifest import pgtformer_manifest  # noqa: E402
from pgtformer_amd.synth import make_clip  # noqa: E402
from pgtformer_amd.weightgen import generate_state_dict  # noqa: E402
from tests.golden.r3_scheme import fitted_tail_state_dict  # noqa: E402

cfg = default_config()
sd = fitted_tail_state_dict(generate_state_dict(pgtformer_manifest(cfg), cfg, seed=0))
m = PGTFormer(**cfg)
m.load_state_dict(sd, strict=True)
for prec in ("x3f16",):
    m.prepare("cuda", prec)
    for seed in (1234, 4077):
        lq, _ = make_clip(4, 512, seed=seed)
        bad = m.check_range(torch.from_numpy(lq).cuda(), w=1.0, win=m.window_index(2, 3, "cuda"))
        print(prec, seed, "tensors checked:", m.last_range_launches, "saturating:", len(bad))
        for b in bad:
            print("   ", b)
