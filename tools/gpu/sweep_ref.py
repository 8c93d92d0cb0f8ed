"""The 12 sweep windows against the reference fixtures (tests/golden/r4_golden_sweep.npz): dPSNR on the fixture rows, codes.
Run under different environment switches (PGT_ROWCHAIN, PGT_WCOMP_LINEAR, ...) to attribute changes."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from pgtformer_amd import PGTFormer, default_config  # noqa: E402
from pgtformer_amd.manifest import pgtformer_manifest  # noqa: E402
from pgtformer_amd.synth import make_clip  # noqa: E402
from pgtformer_amd.weightgen import generate_state_dict  # noqa: E402
from tests.golden.r3_scheme import fitted_tail_state_dict  # noqa: E402


def psnr(a, b):
    return float(-10.0 * torch.log10(((a.double() - b.double()) ** 2).mean()))


cfg = default_config()
sd = fitted_tail_state_dict(generate_state_dict(pgtformer_manifest(cfg), cfg, seed=0))
m = PGTFormer(**cfg)
m.load_state_dict(sd, strict=True)
m.prepare("cuda", os.environ.get("PREC", "x3f16"))
g = np.load(os.path.join(ROOT, "tests", "golden", "r4_golden_sweep.npz"))
out = []
for seed in (4077, 5077):
    lq_u8, gt = make_clip(8, 512, seed=seed)
    fr = torch.from_numpy(lq_u8).cuda()
    for s0 in range(0, 6, 2):
        y, _, _ = m.forward_nhwc(fr[s0:s0 + 4], w=1.0, win=m.window_index(2, 3, "cuda"), middle_only=True)
        codes = m.last_codes.cpu().reshape(2, -1)
        for u in range(2):
            j = s0 + u
            tag = f"c{seed}w{j + 1}"
            ref_rows = torch.from_numpy(g[f"{tag}.out_mid_rows"]).double()
            rows = y[u].float().cpu().permute(2, 0, 1)[:, ::8, :].double()
            gt_rows = torch.from_numpy(gt[j + 1]).permute(2, 0, 1)[:, ::8, :].double()
            ndiff = int((codes[u].long() != torch.from_numpy(g[f"{tag}.codes"].astype(np.int64)).reshape(-1)).sum())
            out.append((tag, ndiff, round(psnr(rows, gt_rows) - psnr(ref_rows, gt_rows), 6), round(psnr(rows, ref_rows), 2)))
tagenv = {k: v for k, v in os.environ.items() if k.startswith("PGT_") or k == "PREC"}
print(json.dumps({"env": tagenv, "max_abs_dpsnr": max(abs(o[2]) for o in out if o[1] == 0), "windows": out}))
