"""TEST INFRASTRUCTURE (a study, not a test): per-level decoder precision policies at the fitted-tail operating point, on
several windows.  The encoder side of the oracle runs once per window (exact, cached under /tmp); the decoder is re-run
with the conv / linear ACTIVATION operands, WEIGHT operands and STORED activations rounded by a policy chosen per layer
group (512x512 level, 256->512 up-sampling, 256x256 level + fusion, the rest).

    python tests/precision_study2.py [levels|groups|compensation|planes] [seed:window ...]   (default windows 1234:1 1077:4 2077:1)

  levels        per-level policies (which levels need more than half precision)
  groups        22-bit weights in one decoder level at a time / in all but one: whose weight rounding carries the systematic error
  compensation  half everywhere + the mean-field compensation of the weight rounding (what the product does: per-frame bias
                (W - W16) mean(x) from the library's pixel sample), for the residual-block / fusion convs only and for every layer
  products      TWO MFMA products instead of three in one sub-network of the code-prediction branch at a time (VERDICT round 3,
                item 4): single-plane WEIGHTS (x_hi w_hi + x_lo w_hi) or single-plane ACTIVATIONS (x_hi w_hi + x_hi w_lo), the rest
                of the branch on two half planes; logit error and flipped codes against the exact oracle
  planes        the code-prediction branch on two bf16 planes (the split type of rounds 2-3a) against two half planes (the product's): logit error
                and flipped codes against the exact oracle
Results: profiles/r3_psnr_sweep.md.
"""
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


def q_f16(t): return t.to(torch.float16).float()
def q_bf16(t): return t.to(torch.bfloat16).float()
def q_x3(t):
    hi = t.to(torch.bfloat16).float()
    return hi + (t - hi).to(torch.bfloat16).float()
def q_h2(t):       # two half planes: 22 significand bits
    hi = t.to(torch.float16).float()
    return hi + (t - hi).to(torch.float16).float()
def q_id(t): return t
Q = {"f16": q_f16, "bf16": q_bf16, "x3": q_x3, "h2": q_h2, "f32": q_id}


def group_of(p):
    if p.startswith("decoder.up.0.") or p.startswith("decoder.norm_out") or p.startswith("decoder.conv_out"):
        return "L512"
    if p.startswith("decoder.up.1.upsample"):
        return "U256"
    if p.startswith("decoder.up.1.") or p.startswith("fuse_convs_dict.256"):
        return "L256"
    return "rest"


def run_decoder(cache, policy):
    """policy: group -> (activation operand, weight operand, storage) format names"""
    oc, ol, og, oln = O._conv, O._lin, O._gn, O._ln
    def pol(p): return [Q[n] for n in policy[group_of(p)]]
    def conv(sd_, p, xx, stride=1, padding=0):
        a, w, s = pol(p)
        return s(F.conv2d(a(xx), w(sd_[p + ".weight"]), sd_.get(p + ".bias"), stride=stride, padding=padding))
    def lin(sd_, p, xx):
        a, w, s = pol(p)
        return s(F.linear(a(xx), w(sd_[p + ".weight"]), sd_.get(p + ".bias")))
    def gn(sd_, p, xx, eps=1e-6): return pol(p)[2](og(sd_, p, xx, eps))
    def ln(sd_, p, xx, eps=1e-5): return pol(p)[2](oln(sd_, p, xx, eps))
    O._conv, O._lin, O._gn, O._ln = conv, lin, gn, ln
    try:
        def fuse(fs, hcur):
            if fs in cache["connect"]:
                return O.fuse_sft(sd, f"fuse_convs_dict.{fs}", cache["enc_feats"][fs], hcur, 1.0)
            return hcur
        return O.decoder_forward(sd, cache["dd"], cache["zq"], cache["t"], fuse)
    finally:
        O._conv, O._lin, O._gn, O._ln = oc, ol, og, oln


def encoder_side(seed, i):
    path = f"/tmp/pgt_study_{seed}_{i}.pt"
    if os.path.exists(path):
        return torch.load(path)
    from pgtformer_amd.config import PGTFORMER_DEFAULTS
    lq_u8, gt = make_clip(i + 2, 512, seed=seed)
    win = window_from_clip(lq_u8, i)
    x = torch.from_numpy(win.astype(np.float32) / np.float32(255.0)).permute(0, 3, 1, 2).contiguous()
    full = dict(PGTFORMER_DEFAULTS); full.update(cfg)
    # run the oracle once with the decoder replaced by a recorder
    rec = {}
    odec = O.decoder_forward
    def grab(sd_, dd, zq, t, fuse):
        rec.update(dd=dd, zq=zq, t=t)
        return odec(sd_, dd, zq, t, fuse)
    ofuse = O.fuse_sft
    feats = {}
    def fuse_rec(sd_, p, enc, dec, w, tcc=32):
        feats[p.split(".")[-1]] = enc
        return ofuse(sd_, p, enc, dec, w, tcc)
    O.decoder_forward, O.fuse_sft = grab, fuse_rec
    try:
        ref = O.pgtformer_forward(sd, cfg, x, w=1.0)[0]
    finally:
        O.decoder_forward, O.fuse_sft = odec, ofuse
    cache = dict(rec, enc_feats=feats, connect=list(feats), ref=ref, gt=torch.from_numpy(gt[i]).permute(2, 0, 1).contiguous())
    torch.save(cache, path)
    return cache


def psnr(a, b): return float(-10 * torch.log10(((a.double() - b.double()) ** 2).mean()))

H = ("f16", "f16", "f16")
POLICIES = {
    "half everywhere": dict(L512=H, U256=H, L256=H, rest=H),
    "L512 weights h2": dict(L512=("f16", "h2", "f16"), U256=H, L256=H, rest=H),
    "L512 weights h2 + x3 storage": dict(L512=("f16", "h2", "x3"), U256=H, L256=H, rest=H),
    "L512 x3": dict(L512=("x3", "x3", "x3"), U256=H, L256=H, rest=H),
    "L512+U256 x3": dict(L512=("x3", "x3", "x3"), U256=("x3", "x3", "x3"), L256=H, rest=H),
    "L512+U256+L256 weights h2": dict(L512=("f16", "h2", "f16"), U256=("f16", "h2", "f16"), L256=("f16", "h2", "f16"), rest=H),
    "all weights h2": {g: ("f16", "h2", "f16") for g in ("L512", "U256", "L256", "rest")},
    "L512+U256+L256 x3": dict(L512=("x3",) * 3, U256=("x3",) * 3, L256=("x3",) * 3, rest=H),
    "all x3": {g: ("x3",) * 3 for g in ("L512", "U256", "L256", "rest")},
}

def run_groups(c):
    """22-bit ("h2") weights in one group of decoder layers at a time, and everywhere but one group"""
    global group_of
    G = ["up.0", "up.1", "up.2", "up.3", "up.4", "mid"]

    def by_level(p):
        if p.startswith("decoder.norm_out") or p.startswith("decoder.conv_out"):
            return "up.0"
        for k in range(6):
            if p.startswith(f"decoder.up.{k}."):
                return f"up.{k}"
        if p.startswith("fuse_convs_dict."):
            return {"256": "up.1", "128": "up.2", "64": "up.3", "32": "up.4"}[p.split(".")[1]]
        return "mid"
    old, group_of = group_of, by_level
    W2 = ("f16", "h2", "f16")
    pols = {"half everywhere": {g: H for g in G}, "all weights h2": {g: W2 for g in G}}
    pols.update({f"only {g} weights h2": {k: (W2 if k == g else H) for k in G} for g in G})
    pols.update({f"all but {g} weights h2": {k: (H if k == g else W2) for k in G} for g in G})
    try:
        for name, pol in pols.items():
            report(name, run_decoder(c, pol)[1], c)
    finally:
        group_of = old


def run_compensation(c):
    """half operands / storage everywhere; `which` layers get the per-frame bias (W - W16) mean(x), mean over the library's
    pixel sample of the frame (pgt_sampled_pixel)"""
    from tests.emu_ops import sampled_pixels
    oc, ol, og, oln = O._conv, O._lin, O._gn, O._ln
    for which in ("none", "convs of residual / fusion blocks", "every conv and linear"):
        def sel(p):
            blk = p.split(".")[-1] in ("conv1", "conv2", "conv_out") and p.startswith("decoder") or p.startswith("fuse")
            return which != "none" and (blk or which.startswith("every"))
        def conv(sd_, p, xx, stride=1, padding=0):
            W = sd_[p + ".weight"]; Wq = q_f16(W)
            y = F.conv2d(q_f16(xx), Wq, sd_.get(p + ".bias"), stride=stride, padding=padding)
            if sel(p):
                n, cch, hh, ww = xx.shape
                m = xx.reshape(n, cch, hh * ww)[:, :, sampled_pixels(hh * ww)].mean(2)               # (N, Cin)
                y = y + (m @ (W - Wq).sum(dim=(2, 3)).t()).view(n, -1, 1, 1)
            return q_f16(y)
        def lin(sd_, p, xx):
            W = sd_[p + ".weight"]; Wq = q_f16(W)
            y = F.linear(q_f16(xx), Wq, sd_.get(p + ".bias"))
            if sel(p):
                y = y + (W - Wq) @ xx.reshape(-1, xx.shape[-1]).mean(0)
            return q_f16(y)
        O._conv, O._lin = conv, lin
        O._gn = lambda sd_, p, xx, eps=1e-6: q_f16(og(sd_, p, xx, eps))
        O._ln = lambda sd_, p, xx, eps=1e-5: q_f16(oln(sd_, p, xx, eps))
        try:
            def fuse(fs, hcur):
                return O.fuse_sft(sd, f"fuse_convs_dict.{fs}", c["enc_feats"][fs], hcur, 1.0) if fs in c["connect"] else hcur
            out = O.decoder_forward(sd, c["dd"], c["zq"], c["t"], fuse)[1]
        finally:
            O._conv, O._lin, O._gn, O._ln = oc, ol, og, oln
        report(f"compensated: {which}", out, c)


def run_planes(seed, i):
    """code-prediction branch with operands and stored activations on two bf16 planes / two half planes"""
    lq_u8, _ = make_clip(i + 2, 512, seed=seed)
    x = torch.from_numpy(window_from_clip(lq_u8, i).astype(np.float32) / np.float32(255.0)).permute(0, 3, 1, 2).contiguous()
    oc, ol, og, oln = O._conv, O._lin, O._gn, O._ln

    def run(q0, part=lambda p: True):
        """q0 applied to the operands / stored activations of the layers whose parameter prefix satisfies `part`"""
        mx = [0.0]
        def q(p, t): return q0(t) if part(p) else t
        def conv(sd_, p, xx, stride=1, padding=0):
            y = F.conv2d(q(p, xx), q(p, sd_[p + ".weight"]), sd_.get(p + ".bias"), stride=stride, padding=padding)
            mx[0] = max(mx[0], float(xx.abs().max()), float(y.abs().max()))
            return q(p, y)
        def lin(sd_, p, xx):
            y = F.linear(q(p, xx), q(p, sd_[p + ".weight"]), sd_.get(p + ".bias"))
            mx[0] = max(mx[0], float(xx.abs().max()), float(y.abs().max()))
            return q(p, y)
        O._conv, O._lin = conv, lin
        O._gn = lambda sd_, p, xx, eps=1e-6: q(p, og(sd_, p, xx, eps))
        O._ln = lambda sd_, p, xx, eps=1e-5: q(p, oln(sd_, p, xx, eps))
        try:
            return O.pgtformer_forward(sd, cfg, x, w=1.0, code_only=True)[0], mx[0]
        finally:
            O._conv, O._lin, O._gn, O._ln = oc, ol, og, oln
    ref, mx = run(q_id)
    top2 = ref.topk(2, -1).values
    gap = top2[..., 0] - top2[..., 1]
    print(f"== clip {seed} window {i}: max |activation| {mx:.1f}, smallest top-2 logit gap {float(gap.min()):.2e}, gaps < 1e-4: {int((gap < 1e-4).sum())}", flush=True)
    parts = {"whole code branch": lambda p: True,
             "encoder + quant_conv only": lambda p: p.startswith("encoder") or p.startswith("quant_conv"),
             "BiSeNet + convpos only": lambda p: p.startswith("conditionnet") or p.startswith("convpos"),
             "feat_emb + transformer + head only": lambda p: p.startswith("feat_emb") or p.startswith("ft_layers") or p.startswith("idx_pred")}
    for pname, part in parts.items():
        for name, q in (("two bf16 planes (16 bits)", q_x3), ("two half planes (22 bits)", q_h2)):
            lg = run(q, part)[0]
            print(f"  {pname:36s} {name:28s} logits: max err {float((lg - ref).abs().max()):.2e}  rms {float((lg - ref).pow(2).mean().sqrt()):.2e}   "
                  f"flipped codes {int((lg.argmax(-1) != ref.argmax(-1)).sum())}/{ref.argmax(-1).numel()}", flush=True)


def run_products(seed, i):
    """one sub-network of the code branch with a single-plane operand (two products instead of three), everything else split"""
    lq_u8, _ = make_clip(i + 2, 512, seed=seed)
    x = torch.from_numpy(window_from_clip(lq_u8, i).astype(np.float32) / np.float32(255.0)).permute(0, 3, 1, 2).contiguous()
    oc, ol, og, oln = O._conv, O._lin, O._gn, O._ln

    def run(part, qa_part, qw_part):
        """layers of `part`: activation operand qa_part, weight operand qw_part; all other layers and all stored tensors: h2"""
        def conv(sd_, p, xx, stride=1, padding=0):
            qa, qw = (qa_part, qw_part) if part(p) else (q_h2, q_h2)
            return q_h2(F.conv2d(qa(xx), qw(sd_[p + ".weight"]), sd_.get(p + ".bias"), stride=stride, padding=padding))
        def lin(sd_, p, xx):
            qa, qw = (qa_part, qw_part) if part(p) else (q_h2, q_h2)
            return q_h2(F.linear(qa(xx), qw(sd_[p + ".weight"]), sd_.get(p + ".bias")))
        O._conv, O._lin = conv, lin
        O._gn = lambda sd_, p, xx, eps=1e-6: q_h2(og(sd_, p, xx, eps))
        O._ln = lambda sd_, p, xx, eps=1e-5: q_h2(oln(sd_, p, xx, eps))
        try:
            return O.pgtformer_forward(sd, cfg, x, w=1.0, code_only=True)[0]
        finally:
            O._conv, O._lin, O._gn, O._ln = oc, ol, og, oln
    ref = O.pgtformer_forward(sd, cfg, x, w=1.0, code_only=True)[0]
    top2 = ref.topk(2, -1).values
    gap = top2[..., 0] - top2[..., 1]
    print(f"== clip {seed} window {i}: smallest top-2 logit gap of the exact oracle {float(gap.min()):.2e}, gaps < 1e-4: {int((gap < 1e-4).sum())}", flush=True)
    parts = {"none (three products everywhere)": lambda p: False,
             "BiSeNet + convpos": lambda p: p.startswith("conditionnet") or p.startswith("convpos"),
             "encoder levels 0-1 (per-frame front)": lambda p: p.startswith("encoder.conv_in") or p.startswith("encoder.down.0") or p.startswith("encoder.down.1"),
             "encoder (all)": lambda p: p.startswith("encoder") or p.startswith("quant_conv"),
             "feat_emb + transformer + head": lambda p: p.startswith("feat_emb") or p.startswith("ft_layers") or p.startswith("idx_pred")}
    for pname, part in parts.items():
        for name, qa, qw in (("single-plane weights", q_h2, q_f16), ("single-plane activations", q_f16, q_h2)):
            lg = run(part, qa, qw)
            print(f"  {pname:40s} {name:26s} logits: max err {float((lg - ref).abs().max()):.2e}  rms {float((lg - ref).pow(2).mean().sqrt()):.2e}   "
                  f"flipped codes {int((lg.argmax(-1) != ref.argmax(-1)).sum())}/{ref.argmax(-1).numel()}", flush=True)
            if pname.startswith("none"):
                break


def report(name, out, c):
    ref, gt = c["ref"][1], c["gt"]
    e, r = (out - ref).double(), (ref - gt).double()
    rho = float((e * r).sum() / (e.norm() * r.norm()))
    print(f"  {name:40s} PSNR(build, ref) {psnr(out, ref):6.2f} dB   dPSNR {psnr(out, gt) - psnr(ref, gt):+.2e} dB   corr(e, r) {rho:+.4f}", flush=True)


if __name__ == "__main__":
    mode = sys.argv[1] if len(sys.argv) > 1 and ":" not in sys.argv[1] else "levels"
    sys.argv = [a for a in sys.argv if a != mode]
    wins = [tuple(int(v) for v in a.split(":")) for a in sys.argv[1:]] or [(1234, 1), (1077, 4), (2077, 1)]
    if mode in ("planes", "products"):
        torch.set_num_threads(8)
        for seed, i in wins:
            (run_planes if mode == "planes" else run_products)(seed, i)
        sys.exit(0)
    if mode in ("groups", "compensation"):
        torch.set_num_threads(8)
        for seed, i in wins:
            c = encoder_side(seed, i)
            print(f"== clip {seed} window {i}: PSNR(ref, GT) mid = {psnr(c['ref'][1], c['gt']):.3f} dB", flush=True)
            (run_groups if mode == "groups" else run_compensation)(c)
        sys.exit(0)
    only = os.environ.get("PGT_STUDY_ONLY")
    torch.set_num_threads(8)
    for seed, i in wins:
        t0 = time.time()
        c = encoder_side(seed, i)
        ref, gt = c["ref"][1], c["gt"]
        print(f"== clip {seed} window {i}: PSNR(ref, GT) mid = {psnr(ref, gt):.3f} dB ({time.time() - t0:.0f} s)", flush=True)
        for name, pol in POLICIES.items():
            if only and only not in name:
                continue
            t0 = time.time()
            out = run_decoder(c, pol)[1]
            e, r = (out - ref).double(), (ref - gt).double()
            rho = float((e * r).sum() / (e.norm() * r.norm()))
            print(f"  {name:36s} PSNR(build, ref) {psnr(out, ref):6.2f} dB   dPSNR {psnr(out, gt) - psnr(ref, gt):+.2e} dB   "
                  f"corr(e, r) {rho:+.4f}   ({time.time() - t0:.0f} s)", flush=True)
