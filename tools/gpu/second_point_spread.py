#!/usr/bin/env python
"""How much the PSNR-contract figure of the second operating point (tests/golden/r5_*; R5_POINT=2: the third) moves with summation-order-level changes of
the build: run once per setting of the A/B switches (they are read at import):
    for V in "PGT_X=1" "PGT_FRAME_BIAS=0" "PGT_C64_RING=0 PGT_FUSE_GN_APPLY=0" "PGT_WCOMP=0"; do env $V python tools/gpu/second_point_spread.py out.jsonl; done
Per window: PSNR(build, GT) - PSNR(reference, GT) on the fixture rows of the middle frame, default mode, benchmarked path."""
import json
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from pgtformer_amd import PGTFormer, default_config  # noqa: E402
from pgtformer_amd.manifest import pgtformer_manifest  # noqa: E402
from pgtformer_amd.synth import make_clip  # noqa: E402
from pgtformer_amd.weightgen import generate_state_dict  # noqa: E402
from tests.golden.r5_scheme import POINTS, point_state_dict  # noqa: E402

POINT = int(os.environ.get("R5_POINT", "1"))      # weight seed of the operating point: 1 (second point), 2 (third point)


def psnr(a, b):
    return float(-10.0 * np.log10(float(((a.double() - b.double()) ** 2).mean())))


cfg = default_config()
sd = point_state_dict(generate_state_dict(pgtformer_manifest(cfg), cfg, seed=POINT), POINT)
m = PGTFormer(**cfg)
m.load_state_dict(sd, strict=True)
m.prepare("cuda", "x3f16")
g = np.load(os.path.join(REPO, "tests", "golden", POINTS[POINT]["golden"]))
tags = sorted({k.split(".")[0] for k in g.files})
clips, rows_out = {}, {}
for tag in tags:
    seed, i = (int(v) for v in tag[1:].split("w"))
    if seed not in clips:
        clips[seed] = make_clip(POINTS[POINT]["clip_frames"][seed], 512, seed=seed)
    lq_u8, gt = clips[seed]
    out, _, _ = m.forward_nhwc(torch.from_numpy(lq_u8[i - 1:i + 2]).to("cuda"), w=1.0, win=m.window_index(1, 3, "cuda"), middle_only=True)
    rows = out[0].float().cpu().permute(2, 0, 1)[:, ::8, :]
    ref = torch.from_numpy(g[f"{tag}.out_mid_rows"])
    gt_rows = torch.from_numpy(gt[i]).permute(2, 0, 1)[:, ::8, :]
    codes = m.last_codes.cpu().numpy().astype(np.int64).reshape(-1)
    rows_out[tag] = {"dpsnr_db": psnr(rows, gt_rows) - psnr(ref, gt_rows), "psnr_build_vs_ref_db": psnr(rows, ref),
                     "differing_tokens": int((codes != g[f"{tag}.codes"].astype(np.int64).reshape(-1)).sum())}
env = {k: v for k, v in os.environ.items() if k.startswith("PGT_")}
rec = {"env": env, "weight_seed": POINT, "max_abs_dpsnr_db": max(abs(r["dpsnr_db"]) for r in rows_out.values()), "windows": rows_out}
print(json.dumps({"env": env, "max_abs_dpsnr_db": rec["max_abs_dpsnr_db"], "dpsnr": {t: round(r["dpsnr_db"], 6) for t, r in rows_out.items()}}))
if len(sys.argv) > 1:
    with open(sys.argv[1], "a") as f:
        f.write(json.dumps(rec) + "\n")
