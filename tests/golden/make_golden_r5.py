"""Round-5 fixtures from the REFERENCE (imported from /root/reference in the build container): a SECOND, independent
operating point for north_star's "within 1e-3 PSNR" contract.
    python tests/golden/make_golden_r5.py  ->  tests/golden/r5_tail_s1.npz, tests/golden/r5_golden_s1.npz   (about 35 min)
    R5_POINT=2 python tests/golden/make_golden_r5.py  ->  r5_tail_s2.npz, r5_golden_s2.npz: a THIRD point (weight seed 2, tail fitted on
    clip 10077 w2, 8 windows of clips 10077 / 11077 / 12077), generated after every constant of the build was fixed
    R5_POINT=3 / R5_POINT=4  ->  r6_tail_s3 / r6_golden_s3, r6_tail_s4 / r6_golden_s4 (round 6; about 8 min each): a FOURTH point (seed 3) and a
    FIFTH one (seed 4) whose code transformer keeps the tokens apart (r5_scheme.predamp): ~100 distinct codes per window instead of 1 - 13;
    R5_POINT=5 -> r6_tail_s5 / r6_golden_s5: the fifth point's scheme at weight seed 5

Why: every PSNR-contract window of rounds 3 / 4 runs ONE weight set (seed-0 weights + one tail fitted on clip 1234 w1).  The
half decoder's rounding errors, the defects D = W - half(W) behind the mean-field compensation and the
activation ranges IEEE half has to hold are properties of the weights, so the contract was shown at one draw of them.  This
script draws ALL 961 tensors again (weightgen seed 1), re-calibrates the SFT gains on the reference for that draw (same
targets as seed 0: rms(scale) = 0.25, rms(shift) = rms(dec) at each fusion), fits its own decoder tail on a window of a
DIFFERENT clip (7077 w2), and records the reference's outputs on 8 windows of 3 clips (one fitted, seven held out):
codes, the reference's own top-2 logit margins, every 8th fp32 row of the middle frame, PSNR(reference, GT).
The oracle is checked against the reference on the fitted window (bit-identical), as in make_golden_r3.py."""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(HERE, "_refshim"))
sys.path.insert(0, HERE)
REF = "/root/reference"

import make_golden_r3 as R3                                            # fit_tail, tail_forward, train_tail, psnr, rms  # noqa: E402
from tests.golden import r5_scheme as S5                               # noqa: E402

POINT = int(os.environ.get("R5_POINT", "1"))                         # weight seed of the operating point (r5_scheme.POINTS): 1 .. 5
PT = S5.POINTS[POINT]
WINDOWS = PT["windows"]                                              # (clip seed, window)
CLIP_FRAMES = PT["clip_frames"]
TRAIN_CLIP, TRAIN_WINDOW = PT["train"]
OUT = os.environ.get("R5_OUT", HERE)                                 # R5_OUT=/tmp/x R3_STEPS=2: a dry run that leaves the fixtures alone


def calibrate_and_capture(O, sd, cfg, x):
    """One oracle forward over `sd` (modified IN PLACE): at each fusion the raw scale / shift branches are measured first, the
    last convs of the branches re-scaled to the calibration targets, then the fusion runs with the corrected weights.  Returns
    (gains {"32.scale": c, ...}, enc256, dec256)."""
    held, gains = {}, {}
    orig_fuse, orig_conv = O.fuse_sft, O._conv

    def fuse(sd_, p, enc, dec, w, tcc=32):
        raw = {}

        def conv(sd2, q, xx, stride=1, padding=0):
            y = orig_conv(sd2, q, xx, stride, padding)
            if q in (p + ".scale.2", p + ".shift.2"):
                raw[q] = y
            return y
        O._conv = conv
        orig_fuse(sd_, p, enc, dec, w, tcc)                            # measuring pass
        a = R3.SCALE_RMS / R3.rms(raw[p + ".scale.2"])
        b = R3.SHIFT_REL * R3.rms(dec) / R3.rms(raw[p + ".shift.2"])
        size = p.rsplit(".", 1)[1]
        gains[f"{size}.scale"], gains[f"{size}.shift"] = float(np.float32(a)), float(np.float32(b))
        sd_.update(S5.apply_gains(sd_, {f"{size}.scale": gains[f"{size}.scale"], f"{size}.shift": gains[f"{size}.shift"]}))
        raw.clear()
        out = orig_fuse(sd_, p, enc, dec, w, tcc)                      # the fusion as the calibrated weights compute it
        O._conv = orig_conv
        print(f"  {p}: rms dec {R3.rms(dec):.2f}; corrections scale x{a:.3f} shift x{b:.3f} -> rms scale "
              f"{R3.rms(raw[p + '.scale.2']):.3f} shift {R3.rms(raw[p + '.shift.2']):.2f}", flush=True)
        if p.endswith(".256"):
            held["enc"], held["dec"] = enc.clone(), dec.clone()
        return out

    O.fuse_sft = fuse
    try:
        O.pgtformer_forward(sd, cfg, x, w=1.0)
    finally:
        O.fuse_sft, O._conv = orig_fuse, orig_conv
    return gains, held["enc"], held["dec"]


def main():
    os.chdir(REF)
    sys.path.insert(0, REF)
    from archs.pgtformer_arch import PGTFormer                       # reference

    from oracle import pgt_oracle as O
    from pgtformer_amd.config import default_config
    from pgtformer_amd.manifest import pgtformer_manifest
    from pgtformer_amd.synth import make_clip, window_from_clip
    from pgtformer_amd.weightgen import generate_state_dict

    torch.manual_seed(0)
    torch.use_deterministic_algorithms(True)
    cfg = default_config()
    model = PGTFormer(**cfg)
    model.eval()
    sd1 = generate_state_dict(pgtformer_manifest(cfg), cfg, seed=POINT)
    clips = {seed: make_clip(n, 512, seed=seed) for seed, n in CLIP_FRAMES.items()}

    def window(seed, i):
        lq_u8, gt = clips[seed]
        n = lq_u8.shape[0]
        idx = [max(i - 1, 0), i, min(i + 1, n - 1)]
        x = torch.from_numpy(window_from_clip(lq_u8, i).astype(np.float32) / np.float32(255.0)).permute(0, 3, 1, 2).contiguous()
        return x, torch.from_numpy(gt[idx]).permute(0, 3, 1, 2).contiguous()

    xt, gtt = window(TRAIN_CLIP, TRAIN_WINDOW)
    print("PSNR(LQ input, GT) on the fitted window = %.2f dB" % R3.psnr(xt, gtt))
    sd = dict(S5.predamp(sd1, POINT))                                 # (points with damped code-transformer branches: r5_scheme.POINTS)
    print(f"SFT gains re-calibrated for the seed-{POINT} draw (one oracle forward, each fusion measured then corrected):")
    gains, enc, dec = calibrate_and_capture(O, sd, cfg, xt)
    with torch.no_grad():
        feat = R3.tail_forward(O, sd, enc, dec, upto_features=True)
    sd["decoder.conv_out.weight"], sd["decoder.conv_out.bias"] = R3.fit_tail(feat, gtt)
    with torch.no_grad():
        print("least-squares read-out of the random tail: PSNR(ref, GT) = %.2f dB" % R3.psnr(R3.tail_forward(O, sd, enc, dec), gtt))
    print(f"training the tail ({R3.STEPS} Adam steps):")
    trained = R3.train_tail(O, sd, enc, dec, gtt)
    payload = {k: v.numpy() for k, v in trained.items()}
    payload.update({S5.GAIN_KEY + k: np.float32(v) for k, v in gains.items()})
    np.savez_compressed(os.path.join(OUT, PT["fixture"]), **payload)
    print("tail tensors:", len(trained), "with", sum(v.numel() for v in trained.values()), "values; gains:", gains)

    sd = S5.point_state_dict(sd1, POINT, here=OUT)                   # the scheme exactly as the tests apply it
    model.load_state_dict(sd, strict=True)
    full = {}
    for seed, i in (WINDOWS[1:2] if OUT != HERE else WINDOWS):
        x, g = window(seed, i)
        t0 = time.time()
        with torch.no_grad():
            out, logits, _ = model(x.clone(), w=1.0)
        lg = logits.reshape(-1, logits.shape[-1])
        top2 = lg.topk(2, dim=-1).values
        tag = f"c{seed}w{i}"
        sat = float(((out[1] < 0) | (out[1] > 1)).float().mean())
        msg = (f"{tag}: middle frame range [{out[1].min().item():.3f}, {out[1].max().item():.3f}] ({sat:.2e} outside [0, 1]), "
               f"PSNR(ref, GT) {R3.psnr(out[1], g[1]):.3f} dB, smallest top-2 margin {float((top2[:, 0] - top2[:, 1]).min()):.2e}")
        if (seed, i) == (TRAIN_CLIP, TRAIN_WINDOW):
            o_out, o_logits, _ = O.pgtformer_forward(sd, cfg, x, w=1.0)
            msg += f"; reference vs oracle max|d| {(out - o_out).abs().max().item():.3e}"
        print(msg + f"  ({time.time() - t0:.0f} s)", flush=True)
        full[f"{tag}.out_mid_rows"] = out[1, :, ::8, :].numpy()
        full[f"{tag}.codes"] = lg.argmax(-1).numpy().astype(np.int16)
        full[f"{tag}.top2_margin"] = (top2[:, 0] - top2[:, 1]).numpy().astype(np.float32)
        full[f"{tag}.psnr_ref_vs_gt_db"] = np.array([R3.psnr(out[1], g[1])])
        full[f"{tag}.out_stats"] = np.array([[o.mean().item(), o.std().item(), o.min().item(), o.max().item()] for o in out])
    np.savez_compressed(os.path.join(OUT, PT["golden"]), **full)
    for fn in (PT["fixture"], PT["golden"]):
        print(fn, os.path.getsize(os.path.join(OUT, fn)) // 1024, "KiB")


if __name__ == "__main__":
    main()
