#!/usr/bin/env python
"""Operand-level product study of the code transformer's attention: P on its hi plane only in P.V
(PGT_MHA_PLITE=1: 5 MFMA products per key tile instead of 6).  Run once per setting of the switch (it is read once per process):
    PGT_MHA_PLITE=0 python tools/gpu/mha_plite_study.py out.jsonl ; PGT_MHA_PLITE=1 python tools/gpu/mha_plite_study.py out.jsonl
Per window: max |logits(default mode) - logits(fp32 mode of the same build)| over all 3072 x 1024 logits (the fp32 build is
6e-6 from the reference), codes against the REFERENCE fixtures (r4_golden_sweep.npz: 12 windows of clips 4077 / 5077) with the
reference's margin at every differing token, and the mha launches' time."""
import json
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from pgtformer_amd import PGTFormer, default_config, ops  # noqa: E402
from pgtformer_amd.manifest import pgtformer_manifest  # noqa: E402
from pgtformer_amd.synth import make_clip  # noqa: E402
from pgtformer_amd.weightgen import generate_state_dict  # noqa: E402
from tests.golden.r3_scheme import fitted_tail_state_dict  # noqa: E402

dev = "cuda"
cfg = default_config()
sd = fitted_tail_state_dict(generate_state_dict(pgtformer_manifest(cfg), cfg, seed=0))
models = {}
for prec in ("x3f16", "fp32"):
    m = PGTFormer(**cfg)
    m.load_state_dict(sd, strict=True)
    models[prec] = m.prepare(dev, prec)
g = np.load(os.path.join(REPO, "tests", "golden", "r4_golden_sweep.npz"))
rows = []
for seed in (4077, 5077):
    lq_u8, _ = make_clip(8, 512, seed=seed)
    for i in range(1, 7):
        x = torch.from_numpy(lq_u8[i - 1:i + 2].astype(np.float32) / np.float32(255.0)).permute(0, 3, 1, 2).contiguous().to(dev)
        lg = {}
        for prec in ("x3f16", "fp32"):
            lg[prec] = models[prec](x, code_only=True)[0].float().reshape(-1, 1024)
        err = float((lg["x3f16"] - lg["fp32"]).abs().max())
        codes = lg["x3f16"].argmax(-1).cpu().numpy()
        tag = f"c{seed}w{i}"
        ref = g[f"{tag}.codes"].astype(np.int64).reshape(-1)
        diff = np.nonzero(codes != ref)[0]
        rows.append({"window": tag, "logits_err_vs_fp32_build": err, "differing_tokens": int(diff.size),
                     "reference_margin_at_differing_tokens": [float(g[f"{tag}.top2_margin"].reshape(-1)[j]) for j in diff]})
        print(rows[-1], flush=True)
# time of the 9 attention launches of one forward (event-bracketed, launches alone)
recs = []
ops.PROFILE = recs
models["x3f16"](x, code_only=True)
torch.cuda.synchronize()
ops.PROFILE = None
mha_ms = sum(r["events"][0].elapsed_time(r["events"][1]) for r in recs if r["kernel"] == "mha")
out = {"PGT_MHA_PLITE": os.environ.get("PGT_MHA_PLITE", "0"), "max_logits_err": max(r["logits_err_vs_fp32_build"] for r in rows),
       "windows_with_differing_codes": sum(1 for r in rows if r["differing_tokens"]), "mha_ms_one_window": round(mha_ms, 3), "windows": rows}
print(json.dumps({k: v for k, v in out.items() if k != "windows"}))
if len(sys.argv) > 1:
    with open(sys.argv[1], "a") as f:
        f.write(json.dumps(out) + "\n")
