"""The SECOND operating point of the PSNR contract (TEST DATA, see make_golden_r5.py): seed-1 synthetic weights
(pgtformer_amd.weightgen.generate_state_dict(seed=1): every one of the 961 tensors differs from the seed-0 set the round-3 / 4
fixtures use) with
  * the last convs of the SFT `scale` / `shift` branches of the 32 / 64 / 128 fusions re-scaled by the factors the generator
    measured on the reference (`__gain__.*` entries of r5_tail_s1.npz: weightgen.SFT_GAINS were calibrated on seed 0, the same
    targets - rms(scale) = 0.25, rms(shift) = rms(dec) - need slightly different factors for another draw), and
  * the decoder's last stage (256x256 fusion block .. conv_out) TRAINED on a window of another clip (seed 7077, window 2) than
    the first operating point's (seed 1234, window 1), stored as halves and de-quantised to generic fp32 values.
Weight rounding errors, the mean-field compensation's defects and the activation ranges of the half decoder are properties of
the weights: this is an independent draw of all of them."""
import os

import numpy as np
import torch

from .r3_scheme import _dither as _dither_r3

HERE = os.path.dirname(os.path.abspath(__file__))
GAIN_KEY = "__gain__."
# the operating points made by make_golden_r5.py (R5_POINT=<weight seed>): weight seed -> fitted window, fixture files, the 8
# windows the reference ran (clip seed, window) and the clip lengths.  Seed 1 = "the second operating point"; seed 2 = a THIRD
# independent draw, generated after the compensation's mean field was resolved in bands - a point that
# had no part in choosing any constant of the build.
POINTS = {
    1: dict(train=(7077, 2), fixture="r5_tail_s1.npz", golden="r5_golden_s1.npz", dither="r5:",
            windows=((7077, 1), (7077, 2), (7077, 3), (8077, 1), (8077, 2), (8077, 3), (9077, 2), (9077, 5)),
            clip_frames={7077: 5, 8077: 5, 9077: 7}, min_psnr_ref_gt_db=25.0),
    2: dict(train=(10077, 2), fixture="r5_tail_s2.npz", golden="r5_golden_s2.npz", dither="r5s2:",
            windows=((10077, 1), (10077, 2), (10077, 3), (11077, 1), (11077, 2), (11077, 3), (12077, 2), (12077, 5)),
            clip_frames={10077: 5, 11077: 5, 12077: 7}, min_psnr_ref_gt_db=23.0),      # PSNR(reference, GT) 23.7 - 28.9 dB
    # round 6: a FOURTH draw (weight seed 3, tail fitted on clip 13077 w2).  The fixtures come from the reference alone; the build was
    # first run on them after the exact-weight layers and every constant of the compensation were fixed.
    3: dict(train=(13077, 2), fixture="r6_tail_s3.npz", golden="r6_golden_s3.npz", dither="r6s3:",
            windows=((13077, 1), (13077, 2), (13077, 3), (14077, 1), (14077, 2), (14077, 3), (15077, 2), (15077, 5)),
            clip_frames={13077: 5, 14077: 5, 15077: 7}, min_psnr_ref_gt_db=22.0,
            gen_threads=5),      # (torch threads of the generating process: CPU fp32 sums depend on it at the 4e-7 level; default 8)
    # round 6: a FIFTH draw in the regime of a trained checkpoint - DIVERSE codes.  The random-init code transformer of the other points
    # makes all tokens alike (1 - 13 distinct codes per window); here its residual branches (self_attn.out_proj, linear2 of the nine
    # ft_layers) are scaled by 0.1 before everything else, so that the tokens keep their identity: ~100 distinct codes per window, the
    # most frequent one on a quarter of the tokens (probed on the oracle).  Weight seed 4, tail fitted on clip 16077 w2.
    4: dict(train=(16077, 2), fixture="r6_tail_s4.npz", golden="r6_golden_s4.npz", dither="r6s4:",
            windows=((16077, 1), (16077, 2), (16077, 3), (17077, 1), (17077, 2), (17077, 3), (18077, 2), (18077, 5)),
            clip_frames={16077: 5, 17077: 5, 18077: 7}, min_psnr_ref_gt_db=22.0, damp=0.1),
    # round 6: a SIXTH draw, the fifth one's scheme repeated at another weight seed (5; tail fitted on clip 19077 w2) - is the diverse-code
    # figure a property of the regime or of one draw?
    5: dict(train=(19077, 2), fixture="r6_tail_s5.npz", golden="r6_golden_s5.npz", dither="r6s5:",
            windows=((19077, 1), (19077, 2), (19077, 3), (20077, 1), (20077, 2), (20077, 3), (21077, 2), (21077, 5)),
            clip_frames={19077: 5, 20077: 5, 21077: 7}, min_psnr_ref_gt_db=22.0, damp=0.1),
}
DAMPED = ("self_attn.out_proj.weight", "self_attn.out_proj.bias", "linear2.weight", "linear2.bias")


def predamp(sd, seed):
    """the weightgen state dict of a point with the `damp` factor of its scheme applied (a copy; points without one: unchanged)"""
    f = POINTS[seed].get("damp")
    if not f:
        return sd
    out = dict(sd)
    for k in sd:
        if k.startswith("ft_layers.") and k.endswith(DAMPED):
            out[k] = (sd[k] * np.float32(f)).contiguous()
    return out
SEED = 1
TRAIN_CLIP, TRAIN_WINDOW = POINTS[1]["train"]
FIXTURE = POINTS[1]["fixture"]


def _dither(name, half, tag="r5:"):
    return _dither_r3(tag + name, half)      # another stream than the first operating point's


def apply_gains(sd, gains):
    """gains: {"32.scale": c, "32.shift": c, ...} -> scales weight AND bias of fuse_convs_dict.S.{scale,shift}.2 in a copy"""
    out = dict(sd)
    for key, c in gains.items():
        size, branch = key.split(".")
        for leaf in ("weight", "bias"):
            name = f"fuse_convs_dict.{size}.{branch}.2.{leaf}"
            out[name] = (sd[name] * np.float32(c)).contiguous()
    return out


def point_state_dict(sd, seed, here=None):
    """sd: the state dict pgtformer_amd.weightgen draws for weight seed `seed` -> the state dict the reference ran for that point's
    fixture (re-calibrated SFT gains applied, fitted decoder tail put in place)"""
    pt = POINTS[seed]
    sd = predamp(sd, seed)
    fix = np.load(os.path.join(here or HERE, pt["fixture"]))
    gains = {k[len(GAIN_KEY):]: float(fix[k]) for k in fix.files if k.startswith(GAIN_KEY)}
    out = apply_gains(sd, gains)
    for key in fix.files:
        if key.startswith(GAIN_KEY):
            continue
        t = torch.from_numpy(_dither(key, fix[key], pt["dither"]))
        assert t.shape == sd[key].shape, key
        out[key] = t
    return out


def second_point_state_dict(sd1):
    """sd1: the seed-1 state dict of pgtformer_amd.weightgen -> the state dict the reference ran for r5_golden_s1.npz"""
    return point_state_dict(sd1, 1)
