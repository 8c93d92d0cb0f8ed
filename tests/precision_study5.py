"""TEST INFRASTRUCTURE (a study, not a test): would STOCHASTIC rounding of the half decoder's stores (gfx950 has v_cvt_sr_f16_f32) remove the
spatially coherent rounding errors behind the contract figure's DC sensitivity?  CPU emulation of the default
mode with every fp32 -> half store dithered by one ulp (SR=1) against round-to-nearest (SR=0), several dither seeds:
    SR=1 SR_SEED=2 R5_POINT=2 python tests/precision_study5.py c11077w3 c12077w2 c10077w2
Answer (profiles/r6_e_stochastic_rounding_study.jsonl): no - the operating points' code prediction collapses to ONE code per window
(random-init transformer), the decoder's input is a constant field per channel, and AdaIN divides (q - mean) by sqrt(var + 1e-5):
round-to-nearest keeps a constant field constant (the numerator is exactly 0), a dither turns it into noise amplified 300x
(PSNR(build, reference) 76.5 -> 66 dB on the one-code window, the figure scatters to 3e-3 dB).  Not built."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import emu_ops
SR = os.environ.get("SR", "1") == "1"
GEN = torch.Generator().manual_seed(int(os.environ.get("SR_SEED", "1")))
def sr_half(v):
    """fp32 -> half with a uniform dither of one half-ulp width added before round-to-nearest (= stochastic rounding)"""
    v = v.float().clamp(-65504.0, 65504.0)
    if not SR:
        return v.half()
    m, e = torch.frexp(v)                       # v = m * 2^e, 0.5 <= |m| < 1
    e = e.clamp(min=-13)                        # subnormal halves: fixed ulp 2^-24
    ulp = torch.ldexp(torch.ones_like(v), e - 11)
    u = torch.rand(v.shape, generator=GEN) - 0.5
    return (v + u * ulp).half()
_orig_to = None
def _store(val, out, dtype):
    if out is None:
        return (sr_half(val) if dtype == torch.float16 else val.to(dtype)).contiguous()
    out.copy_(sr_half(val) if out.dtype == torch.float16 else val.to(out.dtype))
    return out
emu_ops._store = _store
# the other direct half conversions of the emulation (layernorm, rownorm, embed_rows, parity / row placement stores)
class _P:
    def setattr(self, o, n, v): setattr(o, n, v)
# patch functions that cast with .to(x.dtype): layernorm, embed_rows
def layernorm(x, gamma, beta, eps=1e-5, pos=None, x3=False):
    if x3:
        return emu_ops._ln_orig(x, gamma, beta, eps, pos, x3)
    import torch.nn.functional as F
    y = F.layer_norm(x.float(), (x.shape[-1],), gamma.float(), beta.float(), eps)
    cv = (lambda t: sr_half(t)) if x.dtype == torch.float16 else (lambda t: t.to(x.dtype))
    return cv(y) if pos is None else (cv(y), cv(y + pos.float()))
emu_ops._ln_orig = emu_ops.layernorm
emu_ops.layernorm = layernorm
_er = emu_ops.embed_rows
def embed_rows(codebook, codes, dtype, out=None, accumulate=False, resid=None):
    if dtype != torch.float16 or accumulate or resid is not None:
        return _er(codebook, codes, dtype, out, accumulate, resid)
    e = codebook.float()[codes.long()]
    r = sr_half(e)
    if out is None: return r
    out.copy_(r); return out
emu_ops.embed_rows = embed_rows
emu_ops.install(_P())
torch.set_num_threads(6)
from pgtformer_amd import PGTFormer, default_config
from pgtformer_amd.manifest import pgtformer_manifest
from pgtformer_amd.synth import make_clip
from pgtformer_amd.weightgen import generate_state_dict
from tests.golden.r5_scheme import POINTS, point_state_dict
POINT = int(os.environ.get("R5_POINT", "2"))
cfg = default_config()
sd = point_state_dict(generate_state_dict(pgtformer_manifest(cfg), cfg, seed=POINT), POINT)
m = PGTFormer(**cfg); m.load_state_dict(sd, strict=True); m.prepare("cpu", "x3f16")
g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', POINTS[POINT]["golden"]))
psnr = lambda a, b: float(-10 * torch.log10(((a - b) ** 2).mean()))
res = {}
for tag in (sys.argv[1:] or ["c11077w3"]):
    seed, i = (int(v) for v in tag[1:].split("w"))
    lq, gt = make_clip(POINTS[POINT]["clip_frames"][seed], 512, seed=seed)
    out, _, _ = m.forward_nhwc(torch.from_numpy(lq[i - 1:i + 2]), w=1.0, win=m.window_index(1, 3, "cpu"), middle_only=True)
    rows = out[0].float().permute(2, 0, 1)[:, ::8, :].double()
    ref = torch.from_numpy(g[f"{tag}.out_mid_rows"]).double()
    gtr = torch.from_numpy(gt[i]).permute(2, 0, 1)[:, ::8, :].double()
    e = rows - ref
    res[tag] = {"dpsnr": round(psnr(rows, gtr) - psnr(ref, gtr), 6), "psnr_vs_ref": round(psnr(rows, ref), 2), "dc": [float(e[c].mean()) for c in range(3)]}
print(json.dumps({"SR": SR, "seed": os.environ.get("SR_SEED", "1"), "EXACT_W": os.environ.get("PGT_EXACT_W"), "res": res}))
