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
SEED = 1
TRAIN_CLIP, TRAIN_WINDOW = 7077, 2
FIXTURE = "r5_tail_s1.npz"
GAIN_KEY = "__gain__."


def _dither(name, half):
    return _dither_r3("r5:" + name, half)      # another stream than the first operating point's


def apply_gains(sd, gains):
    """gains: {"32.scale": c, "32.shift": c, ...} -> scales weight AND bias of fuse_convs_dict.S.{scale,shift}.2 in a copy"""
    out = dict(sd)
    for key, c in gains.items():
        size, branch = key.split(".")
        for leaf in ("weight", "bias"):
            name = f"fuse_convs_dict.{size}.{branch}.2.{leaf}"
            out[name] = (sd[name] * np.float32(c)).contiguous()
    return out


def second_point_state_dict(sd1):
    """sd1: the seed-1 state dict of pgtformer_amd.weightgen -> the state dict the reference ran for r5_golden_s1.npz"""
    fix = np.load(os.path.join(HERE, FIXTURE))
    gains = {k[len(GAIN_KEY):]: float(fix[k]) for k in fix.files if k.startswith(GAIN_KEY)}
    out = apply_gains(sd1, gains)
    for key in fix.files:
        if key.startswith(GAIN_KEY):
            continue
        t = torch.from_numpy(_dither(key, fix[key]))
        assert t.shape == sd1[key].shape, key
        out[key] = t
    return out
