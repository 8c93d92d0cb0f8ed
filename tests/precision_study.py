"""TEST INFRASTRUCTURE (a study, not a test): which decoder operand / storage precision holds |dPSNR| <= 1e-3 dB at the
fitted-tail operating point?  Oracle forward with the decoder's conv / linear operands and stored activations rounded to a
given format (encoder side exact).  `python tests/precision_study.py`  ->  profiles/r3_decoder_precision_study.md."""
import os, sys, time
import numpy as np, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pgt_oracle as O
from pgtformer_amd.config import default_config
from pgtformer_amd.manifest import pgtformer_manifest
from pgtformer_amd.synth import make_clip, window_from_clip
from pgtformer_amd.weightgen import generate_state_dict
from tests.golden.r3_scheme import fitted_tail_state_dict

cfg = default_config()
sd = fitted_tail_state_dict(generate_state_dict(pgtformer_manifest(cfg), cfg, seed=0))
lq_u8, gt = make_clip(4, 512, seed=1234)
win = window_from_clip(lq_u8, 1)
x = torch.from_numpy(win.astype(np.float32) / np.float32(255.0)).permute(0, 3, 1, 2).contiguous()
g = torch.from_numpy(gt[[0, 1, 2]]).permute(0, 3, 1, 2).contiguous()

def q_bf16(t): return t.to(torch.bfloat16).float()
def q_f16(t): return t.to(torch.float16).float()
def q_x3(t):
    hi = t.to(torch.bfloat16).float()
    return hi + (t - hi).to(torch.bfloat16).float()
FMT = {"fp32": lambda t: t, "bf16": q_bf16, "f16": q_f16, "x3": q_x3}

def run(op_q, store_q):
    """op_q: rounding of conv/linear operands (activations and weights); store_q: rounding of every stored activation"""
    oc, ol, og, oln = O._conv, O._lin, O._gn, O._ln
    active = {"on": False}
    def conv(sd_, p, xx, stride=1, padding=0):
        if not active["on"]: return oc(sd_, p, xx, stride, padding)
        y = F.conv2d(op_q(xx), op_q(sd_[p + ".weight"]), sd_.get(p + ".bias"), stride=stride, padding=padding)
        return store_q(y)
    def lin(sd_, p, xx):
        if not active["on"]: return ol(sd_, p, xx)
        return store_q(F.linear(op_q(xx), op_q(sd_[p + ".weight"]), sd_.get(p + ".bias")))
    def gn(sd_, p, xx, eps=1e-6):
        y = og(sd_, p, xx, eps)
        return store_q(y) if active["on"] else y
    def ln(sd_, p, xx, eps=1e-5):
        y = oln(sd_, p, xx, eps)
        return store_q(y) if active["on"] else y
    odec = O.decoder_forward
    def dec(*a, **k):
        active["on"] = True
        try: return odec(*a, **k)
        finally: active["on"] = False
    O._conv, O._lin, O._gn, O._ln, O.decoder_forward = conv, lin, gn, ln, dec
    try:
        return O.pgtformer_forward(sd, cfg, x, w=1.0)[0]
    finally:
        O._conv, O._lin, O._gn, O._ln, O.decoder_forward = oc, ol, og, oln, odec

def psnr(a, b): return float(-10 * torch.log10(((a.double() - b.double()) ** 2).mean()))
ref = run(FMT["fp32"], FMT["fp32"])
p_ref = psnr(ref, g)
print(f"reference: PSNR(ref, GT) = {p_ref:.4f} dB, range [{ref.min():.3f}, {ref.max():.3f}]", flush=True)
for name, (oq, sq) in {"bf16": ("bf16", "bf16"), "f16": ("f16", "f16"), "f16 ops / x3 storage": ("f16", "x3"),
                       "bf16 ops / x3 storage": ("bf16", "x3"), "x3": ("x3", "x3")}.items():
    t0 = time.time()
    out = run(FMT[oq], FMT[sq])
    rel = float((out - ref).double().pow(2).mean().sqrt() / ref.double().pow(2).mean().sqrt())
    print(f"{name:24s}: PSNR(build, ref) {psnr(out, ref):7.2f} dB  rel rms {rel:.2e}  PSNR(build, GT) {psnr(out, g):.4f}  "
          f"dPSNR {psnr(out, g) - p_ref:+.2e} dB   mid frame dPSNR {psnr(out[1], g[1]) - psnr(ref[1], g[1]):+.2e}  ({time.time() - t0:.0f} s)", flush=True)
