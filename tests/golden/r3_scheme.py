"""The "fitted tail" weight scheme (TEST DATA, see make_golden_r3.py): seed-0 synthetic weights (pgtformer_amd.weightgen, SFT gains
included) with the trained tensors of the decoder's last stage (256x256 fusion block .. conv_out) from
tests/golden/r3_tail.npz.  With these weights the reference's restored frames sit inside [0, 1] at
PSNR(reference, GT) >= 25 dB, so |PSNR(build, GT) - PSNR(reference, GT)| actually measures decoder arithmetic."""
import os

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def _dither(name, half):
    """fp16-stored value -> a generic fp32 value that rounds back to it: v * (1 + 2^-11 * u), u ~ U[-0.45, 0.45) from a PCG64
    stream keyed by the tensor name.  The fixture stays small (2 bytes per value) without handing an IEEE-half decoder
    weights that are exactly representable in its own storage type (no weight rounding error in the trained layers)."""
    import hashlib

    h = hashlib.sha256(("r3-tail:" + name).encode()).digest()
    g = np.random.Generator(np.random.PCG64(int.from_bytes(h[:16], "little")))
    u = g.uniform(-0.45, 0.45, size=half.shape).astype(np.float32)
    return (half.astype(np.float32) * (np.float32(1.0) + np.float32(2.0 ** -11) * u)).astype(np.float32)


def fitted_tail_state_dict(sd):
    """sd: the seed-0 state dict of pgtformer_amd.weightgen -> a new dict with the trained tail tensors replaced"""
    fix = np.load(os.path.join(HERE, "r3_tail.npz"))
    out = dict(sd)
    for key in fix.files:
        t = torch.from_numpy(_dither(key, fix[key]))
        assert t.shape == sd[key].shape, key
        out[key] = t
    return out
