"""Deterministic synthetic weights, keyed by state-dict key name.

The trained checkpoint (`kepeng/pgtformer-base`) is not in the reference repo and there is no network
(SURVEY.md §8c), so parity and benchmarks run on random-init weights of the exact architecture. Every
tensor is drawn from a numpy PCG64 stream seeded by SHA-256(name) so that the same 130 M parameters
are regenerated bit-identically in the build container (oracle / golden generation) and on the GPU
box, without committing ~520 MB of weights.

Distributions are chosen so activations stay O(1) through the ~90 conv / 40 attention layers:
conv/linear weights ~ N(0, 1/fan_in), biases ~ N(0, 0.05^2), norm gains 1+0.1N, norm shifts 0.1N,
BatchNorm running stats (mean 0.1N, var U(0.5,1.5)), relative-position-bias table 0.2N,
codebook rows 0.3N with the padding row zero (nn.Embedding padding_idx; reference:
archs/tdcrqvae3_arch.py:83-97).

The last convs of the SFT `scale` / `shift` branches (`fuse_convs_dict.S.{scale,shift}.2`) carry the gains of SFT_GAINS:
fused = dec + w (dec * scale + shift) is MULTIPLICATIVE in the decoder trunk (archs/pgtformer_arch.py:478-479), and with
unit-gain random branches the trunk squares its magnitude at each of the four fusions (rms 3 -> 4.5e5 at 256x256, measured
on the reference): no trained checkpoint behaves like that (the reference's constructor even carries a commented-out
`last_zero_init` for exactly these convs, :450-451), it turns every decoder error figure into chaos-amplified noise and
overflows IEEE half.  The gains were calibrated once on the reference (tests/golden/make_golden_r3.py re-measures them):
rms(scale) = 0.25 and rms(shift) = rms(dec) at every fusion, so the trunk stays O(10).
"""
import hashlib

import numpy as np

from .manifest import I64


# (scale.2 gain, shift.2 gain) per fusion size; see the module docstring
SFT_GAINS = {"32": (0.113, 1.48), "64": (0.086, 1.56), "128": (0.0541, 2.07), "256": (0.0408, 1.74)}


def _sft_gain(name):
    parts = name.split(".")
    if len(parts) == 5 and parts[0] == "fuse_convs_dict" and parts[2] in ("scale", "shift") and parts[3] == "2":
        g = SFT_GAINS.get(parts[1])
        if g is not None:
            return np.float32(g[0] if parts[2] == "scale" else g[1])
    return None


def _rng(name, seed):
    h = hashlib.sha256(f"{seed}:{name}".encode()).digest()
    return np.random.Generator(np.random.PCG64(int.from_bytes(h[:16], "little")))


def relative_position_index(frames, win):
    """Restates the buffer built in WindowAttention3D.__init__ (reference:
    modules/rstt_layers.py:167-184): index into the (2D-1)(2Wh-1)(2Ww-1) bias table for every
    (query token, key token) pair of a (D, Wh, Ww) window, tokens ordered d-major then h then w."""
    d, wh, ww = frames, win[0], win[1]
    dd, hh, ww_ = np.meshgrid(np.arange(d), np.arange(wh), np.arange(ww), indexing="ij")
    coords = np.stack([dd.ravel(), hh.ravel(), ww_.ravel()])  # 3, N
    rel = coords[:, :, None] - coords[:, None, :]  # 3, N, N  (query - key)
    rel = rel.transpose(1, 2, 0).copy()
    rel[:, :, 0] += d - 1
    rel[:, :, 1] += wh - 1
    rel[:, :, 2] += ww - 1
    rel[:, :, 0] *= (2 * wh - 1) * (2 * ww - 1)
    rel[:, :, 1] *= 2 * ww - 1
    return rel.sum(-1).astype(np.int64)


def _is_norm_key(name):
    parts = name.split(".")
    leaf_parent = parts[-2] if len(parts) > 1 else ""
    return (leaf_parent.startswith("norm") or leaf_parent.startswith("bn") or leaf_parent == "bn"
            or name.startswith("idx_pred_layer.0.") or ".downsample.1." in name)


def generate_tensor(name, shape, dtype, cfg, seed=0):
    """Return the numpy array for one state-dict entry."""
    if dtype == I64:
        if name.endswith("relative_position_index"):
            dd = cfg["ddconfig"]
            # all levels share num_frames and window size in every shipped config; use the shape
            n = shape[0]
            win = dd["window_sizes"][0]
            frames = n // (win[0] * win[1])
            return relative_position_index(frames, win)
        return np.zeros(shape, np.int64)  # num_batches_tracked
    gain = _sft_gain(name)
    if gain is not None:
        return (_generate_plain(name, shape, cfg, seed) * gain).astype(np.float32)
    return _generate_plain(name, shape, cfg, seed)


def _generate_plain(name, shape, cfg, seed):
    g = _rng(name, seed)
    leaf = name.split(".")[-1]
    if leaf == "running_mean":
        return (0.1 * g.standard_normal(shape, dtype=np.float32)).astype(np.float32)
    if leaf == "running_var":
        return g.uniform(0.5, 1.5, size=shape).astype(np.float32)
    if leaf == "relative_position_bias_table":
        return (0.2 * g.standard_normal(shape, dtype=np.float32)).astype(np.float32)
    if name.startswith("quantizer.codebooks."):
        base = name.rsplit(".", 1)[0] + ".weight"
        if cfg.get("shared_codebook", False):
            base = "quantizer.codebooks.0.weight"
        if leaf == "cluster_size_ema":
            return np.zeros(shape, np.float32)
        gw = _rng(base, seed)
        rows = shape[0] if leaf == "embed_ema" else shape[0] - 1
        w = (0.3 * gw.standard_normal((rows, shape[1]), dtype=np.float32)).astype(np.float32)
        if leaf == "embed_ema":
            return w
        return np.concatenate([w, np.zeros((1, shape[1]), np.float32)], 0)
    if len(shape) == 1:
        if _is_norm_key(name):
            if leaf == "weight":
                return (1.0 + 0.1 * g.standard_normal(shape, dtype=np.float32)).astype(np.float32)
            return (0.1 * g.standard_normal(shape, dtype=np.float32)).astype(np.float32)
        return (0.05 * g.standard_normal(shape, dtype=np.float32)).astype(np.float32)
    fan_in = int(np.prod(shape[1:]))
    w = g.standard_normal(shape, dtype=np.float32)
    return (w * np.float32(1.0 / np.sqrt(fan_in))).astype(np.float32)


def generate_state_dict(manifest, cfg, seed=0, as_torch=True, only=None):
    """Generate all (or the `only` subset of) tensors of `manifest` (name -> (shape, dtype))."""
    out = {}
    for name, (shape, dtype) in manifest.items():
        if only is not None and not only(name):
            continue
        out[name] = generate_tensor(name, shape, dtype, cfg, seed)
    if as_torch:
        import torch

        out = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in out.items()}
    return out
