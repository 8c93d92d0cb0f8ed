"""TEST INFRASTRUCTURE (a study, not a test): the contract figure of one window of an operating point through the CPU EMULATION of the
default mode's arithmetic (tests/emu_ops.py over pgtformer_amd.ops: the build's own host code, half decoder with / without exact-weight
stages and compensation), next to the DC decomposition of the figure.  Settings come from the same environment switches the GPU
build reads (PGT_EXACT_W, PGT_WCOMP, PGT_WCOMP_BANDS ...):
    R5_POINT=2 PGT_EXACT_W=512,32 python tests/precision_study4.py 11077 3 [out.jsonl]"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pgtformer_amd import PGTFormer, default_config  # noqa: E402
from pgtformer_amd.manifest import pgtformer_manifest  # noqa: E402
from pgtformer_amd.synth import make_clip  # noqa: E402
from pgtformer_amd.weightgen import generate_state_dict  # noqa: E402
from tests import emu_ops  # noqa: E402
from tests.golden.r5_scheme import POINTS, point_state_dict  # noqa: E402


class _Patch:
    def setattr(self, obj, name, val):
        setattr(obj, name, val)


POINT = int(os.environ.get("R5_POINT", "2"))
CLIP, WIN = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (11077, 3)
torch.set_num_threads(int(os.environ.get("STUDY_THREADS", "6")))
emu_ops.install(_Patch())
cfg = default_config()
sd = point_state_dict(generate_state_dict(pgtformer_manifest(cfg), cfg, seed=POINT), POINT)
m = PGTFormer(**cfg)
m.load_state_dict(sd, strict=True)
m.prepare("cpu", os.environ.get("STUDY_PRECISION", "x3f16"))
g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", POINTS[POINT]["golden"]))
lq_u8, gt = make_clip(POINTS[POINT]["clip_frames"][CLIP], 512, seed=CLIP)
tag = f"c{CLIP}w{WIN}"
t0 = time.time()
out, _, _ = m.forward_nhwc(torch.from_numpy(lq_u8[WIN - 1:WIN + 2]), w=1.0, win=m.window_index(1, 3, "cpu"), middle_only=True)
rows = out[0].float().permute(2, 0, 1)[:, ::8, :].double()
ref = torch.from_numpy(g[f"{tag}.out_mid_rows"]).double()
gtr = torch.from_numpy(gt[WIN]).permute(2, 0, 1)[:, ::8, :].double()
psnr = lambda a, b: float(-10 * torch.log10(((a - b) ** 2).mean()))  # noqa: E731
e, r = rows - ref, ref - gtr
m2 = float((r ** 2).mean())
dc = [float(e[c].mean()) for c in range(3)]
dc_part = sum(-8.686 * 2 * dc[c] * float(r[c].mean()) / (3 * m2) for c in range(3))
codes = m.last_codes.numpy().astype(np.int64).reshape(-1)
rec = {"env": {k: v for k, v in os.environ.items() if k.startswith("PGT_")}, "point": POINT, "window": tag,
       "dpsnr_db": psnr(rows, gtr) - psnr(ref, gtr), "psnr_build_vs_ref_db": psnr(rows, ref), "dc_error_per_channel": dc,
       "dpsnr_from_dc_db": dc_part, "ref_minus_gt_mean_per_channel": [float(r[c].mean()) for c in range(3)],
       "differing_tokens": int((codes != g[f"{tag}.codes"].astype(np.int64).reshape(-1)).sum()), "seconds": round(time.time() - t0)}
print(json.dumps(rec))
if len(sys.argv) > 3:
    with open(sys.argv[3], "a") as f:
        f.write(json.dumps(rec) + "\n")
