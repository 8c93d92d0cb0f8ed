"""Round-2 golden cases: the Video-Swin window attention of modules/swin.py (BASELINE.json configs[4]), the
VectorQuantizer look-up of archs/vqgan_arch.py and the remaining stage-I API of archs/tdcrqvae3_arch.py.

`make_golden_r2.py` runs the REFERENCE on these inputs (build container only) and stores the outputs in
`r2_golden.npz`; tests re-create the same inputs / weights (numpy seeds, platform independent) and run the oracle (CPU)
or the HIP path (GPU) against the stored outputs."""
import numpy as np
import torch


def rnd(shape, seed, scale=1.0):
    g = np.random.default_rng(seed)
    return torch.from_numpy((scale * g.standard_normal(shape)).astype(np.float32))


# name -> (dim, heads, window (wd,wh,ww), shift (sd,sh,sw), feature map (B,D,H,W), qkv_bias, seed)
SWIN = {
    # BASELINE point: 8x8 windows over T = 3 frames (the D axis is one window), C = 512: N = 192 tokens per window
    "swin_3x8x8_c512": (512, 8, (3, 8, 8), (0, 4, 4), (1, 3, 16, 16), False, 301),
    "swin_3x8x8_c512_noshift": (512, 8, (3, 8, 8), (0, 0, 0), (1, 3, 16, 16), False, 302),
    # windows AND shift along the depth axis: the 27-region mask
    "swin_2x4x6_c256_dshift": (256, 8, (2, 4, 6), (1, 2, 3), (1, 4, 8, 12), True, 303),
    "swin_2x4x6_c256_b2": (256, 8, (2, 4, 6), (1, 0, 3), (2, 4, 8, 12), True, 304),
}


def swin_params(name):
    """Deterministic parameters of one SwinTransformerBlock3D (norm1 + WindowAttention3D), keys as in its state dict."""
    dim, heads, ws, ss, fmap, qkv_bias, seed = SWIN[name]
    n_tab = (2 * ws[0] - 1) * (2 * ws[1] - 1) * (2 * ws[2] - 1)
    p = {"norm1.weight": 1 + 0.1 * rnd((dim,), seed + 1), "norm1.bias": 0.1 * rnd((dim,), seed + 2),
         "attn.qkv.weight": rnd((3 * dim, dim), seed + 3, dim ** -0.5),
         "attn.proj.weight": rnd((dim, dim), seed + 4, dim ** -0.5), "attn.proj.bias": 0.1 * rnd((dim,), seed + 5),
         "attn.relative_position_bias_table": rnd((n_tab, heads), seed + 6, 0.3)}
    if qkv_bias:
        p["attn.qkv.bias"] = 0.1 * rnd((3 * dim,), seed + 7)
    return p


def swin_input(name):
    dim, heads, ws, ss, fmap, qkv_bias, seed = SWIN[name]
    return rnd(tuple(fmap) + (dim,), seed)


# VectorQuantizer.forward: (codebook_size, emb_dim, z shape (B,C,H,W), seed); duplicated code rows exercise the tie rule
VQ = {"vq_1024x256": (1024, 256, (2, 256, 8, 8), 401), "vq_512x64_ties": (512, 64, (1, 64, 6, 5), 402)}


def vq_case(name):
    k, c, zshape, seed = VQ[name]
    w = rnd((k, c), seed, 1.0 / k ** 0.5)
    z = rnd(zshape, seed + 1, 0.05)
    if name.endswith("ties"):
        w[77] = w[5]                                  # identical code vectors: the lower index must win
        z[0, :, 1, 2] = w[77]
    return w, z
