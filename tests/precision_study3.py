"""TEST INFRASTRUCTURE (a study, not a test): where the contract figure of the THIRD operating point comes from
(tests/golden/r5_scheme.py POINTS[2], window clip 11077 w3: 9.5e-4 dB on the GPU build).  Oracle forward with the
decoder's WEIGHTS and / or ACTIVATIONS (conv / linear operands, stored tensors) rounded to IEEE half, all stages or one stage at a
time (encoder side and codes exact).  No compensation is emulated: the weight rows show the raw weight-rounding effect the mean
field has to remove, the activation rows what no bias correction can reach.
    python tests/precision_study3.py [clip window]   ->  profiles/r5_u_third_point_oracle_ablation.md"""
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pgt_oracle as O  # noqa: E402
from pgtformer_amd.config import default_config  # noqa: E402
from pgtformer_amd.manifest import pgtformer_manifest  # noqa: E402
from pgtformer_amd.synth import make_clip, window_from_clip  # noqa: E402
from pgtformer_amd.weightgen import generate_state_dict  # noqa: E402
from tests.golden.r5_scheme import POINTS, point_state_dict  # noqa: E402

POINT = int(os.environ.get("R5_POINT", "2"))
CLIP, WIN = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (11077, 3)
cfg = default_config()
sd = point_state_dict(generate_state_dict(pgtformer_manifest(cfg), cfg, seed=POINT), POINT)
lq_u8, gt = make_clip(POINTS[POINT]["clip_frames"][CLIP], 512, seed=CLIP)
x = torch.from_numpy(window_from_clip(lq_u8, WIN).astype(np.float32) / np.float32(255.0)).permute(0, 3, 1, 2).contiguous()
g_mid = torch.from_numpy(gt[WIN]).permute(2, 0, 1)

STAGES = {"512": ("decoder.up.0.", "decoder.conv_out", "decoder.norm_out"),
          "256": ("decoder.up.1.", "fuse_convs_dict.256."),
          "128": ("decoder.up.2.", "fuse_convs_dict.128."),
          "64": ("decoder.up.3.", "fuse_convs_dict.64."),
          "32": ("decoder.up.4.", "decoder.mid.", "decoder.conv_in", "fuse_convs_dict.32.")}


def stage_of(p):
    for s, pre in STAGES.items():
        if p.startswith(pre):
            return s
    return None


def h(t):
    return t.to(torch.float16).float()


def h_diffused(w):
    """(O, C, KH, KW) -> halves whose rounding errors cancel over the taps of every (o, c) filter: the taps are rounded in order of
    decreasing magnitude, each one to the half nearest to (w - carried error), so that sum_taps (q - w) ends below half an ulp of the
    SMALLEST tap - the defect D = q - w then has no response to the spatially smooth part of the activations (what the per-band mean
    field only removes for the band mean), at the price of up to one ulp instead of half on single taps"""
    if w.dim() != 4 or w.shape[2] * w.shape[3] == 1:
        return h(w)
    o, c, kh, kw = w.shape
    flat = w.reshape(o * c, kh * kw).double()
    order = flat.abs().argsort(dim=1, descending=True)
    ws = flat.gather(1, order)
    q = torch.empty_like(ws)
    e = torch.zeros(o * c, dtype=torch.float64)
    for t in range(kh * kw):
        q[:, t] = (ws[:, t] - e).to(torch.float16).double()
        e = e + q[:, t] - ws[:, t]
    out = torch.empty_like(q)
    out.scatter_(1, order, q)
    return out.reshape(o, c, kh, kw).float()


H_W = {"plain": h, "diffused": h_diffused}
W_MODE = {"mode": "plain"}


def run(w_stages=(), a_stages=()):
    """w_stages / a_stages: decoder stages whose weights / activations (operands and stored outputs) are rounded to half"""
    oc, ol, odec = O._conv, O._lin, O.decoder_forward
    active = {"on": False}
    seen = set()

    def conv(sd_, p, xx, stride=1, padding=0):
        s = stage_of(p) if active["on"] else None
        if s is None:
            if active["on"]:
                seen.add(p)
            return oc(sd_, p, xx, stride, padding)
        w = sd_[p + ".weight"]
        y = F.conv2d(h(xx) if s in a_stages else xx, H_W[W_MODE["mode"]](w) if s in w_stages else w, sd_.get(p + ".bias"), stride=stride, padding=padding)
        return h(y) if s in a_stages else y

    def lin(sd_, p, xx):
        s = stage_of(p) if active["on"] else None
        if s is None:
            if active["on"]:
                seen.add(p)
            return ol(sd_, p, xx)
        w = sd_[p + ".weight"]
        y = F.linear(h(xx) if s in a_stages else xx, h(w) if s in w_stages else w, sd_.get(p + ".bias"))
        return h(y) if s in a_stages else y

    def dec(*a, **k):
        active["on"] = True
        try:
            return odec(*a, **k)
        finally:
            active["on"] = False
    O._conv, O._lin, O.decoder_forward = conv, lin, dec
    try:
        out = O.pgtformer_forward(sd, cfg, x, w=1.0)[0]
    finally:
        O._conv, O._lin, O.decoder_forward = oc, ol, odec
    assert not seen, sorted(seen)[:5]
    return out[1]


def psnr(a, b):
    return float(-10 * torch.log10(((a.double() - b.double()) ** 2).mean()))


ALL = tuple(STAGES)
lines = []


def report(name, out, ref, p_ref, t0):
    e, r = (out - ref).double(), (ref - g_mid).double()
    rho = float((e * r).sum() / (e.norm() * r.norm() + 1e-300))
    line = (f"| {name} | {psnr(out, ref):.1f} | {psnr(out, g_mid) - p_ref:+.2e} | {rho:+.3f} | {float(e.mean()):+.2e} |")
    print(line + f"   ({time.time() - t0:.0f} s)", flush=True)
    lines.append(line)


t0 = time.time()
ref = run()
p_ref = psnr(ref, g_mid)
print(f"point {POINT}, clip {CLIP} w{WIN}: PSNR(reference, GT) = {p_ref:.3f} dB  ({time.time() - t0:.0f} s)", flush=True)
for name, (ws, as_) in ([("weights half, all stages", (ALL, ())), ("activations half, all stages", ((), ALL)), ("both, all stages", (ALL, ALL))]
                        + [(f"weights half, stage {s} only", ((s,), ())) for s in ALL]
                        + [(f"activations half, stage {s} only", ((), (s,))) for s in ALL]):
    t0 = time.time()
    report(name, run(ws, as_), ref, p_ref, t0)
W_MODE["mode"] = "diffused"
for name, (ws, as_) in ([("weights half with tap-diffused rounding (3x3 filters), all stages", (ALL, ())), ("the same + activations half, all stages", (ALL, ALL))]
                        + [(f"tap-diffused weights, stage {s} only", ((s,), ())) for s in ("512", "32")]):
    t0 = time.time()
    report(name, run(ws, as_), ref, p_ref, t0)
W_MODE["mode"] = "plain"
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r5_u_third_point_oracle_ablation.md")
with open(out, "a") as f:
    f.write(f"\n## weight seed {POINT}, clip {CLIP} window {WIN}: PSNR(reference, GT) {p_ref:.3f} dB (middle frame, all rows)\n\n"
            "| decoder arithmetic rounded to IEEE half in the oracle (no compensation) | PSNR(out, reference) dB | PSNR(out, GT) - PSNR(reference, GT) dB | "
            "correlation of the error with (reference - GT) | mean error |\n|---|---|---|---|---|\n" + "\n".join(lines) + "\n")
