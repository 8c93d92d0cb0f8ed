"""Round-2 golden cases of SURVEY §8 row (f)4: the training-side quantiser (VQEmbedding EMA codebook update,
archs/tdcrqvae3_arch.py:128-186) and the Video-Swin `BasicLayer` stage of TDRQVAE (modules/swin.py:326-409).

`make_golden_r2b.py` runs the REFERENCE on these inputs (build container only) and stores the outputs in
`r2b_golden.npz`; tests re-create the same inputs / weights (numpy seeds, platform independent)."""
import numpy as np
import torch



def rnd(shape, seed, scale=1.0):
    g = np.random.default_rng(seed)
    return torch.from_numpy((scale * g.standard_normal(shape)).astype(np.float32))

# name -> (n_embed, embed_dim, batch vectors, decay, restart_unused_codes, steps, seed)
EMA = {
    # more vectors than codes; 3 consecutive training steps so that dead codes (EMA count < 1) restart
    "ema_256x64": (256, 64, 1200, 0.99, True, 3, 501),
    # fewer vectors than codes: _tile_with_noise feeds the restart
    "ema_512x32_tiled": (512, 32, 300, 0.9, True, 2, 502),
    "ema_64x512_norestart": (64, 512, 700, 0.99, False, 2, 503),
}


def ema_case(name):
    """(initial weight (K+1, D) with a zero padding row, list of per-step batches (n, D))"""
    k, d, n, decay, restart, steps, seed = EMA[name]
    w = torch.cat([rnd((k, d), seed, 1.0 / k), torch.zeros(1, d)], 0)
    return w, [rnd((n, d), seed + 10 + s, 0.02 + 0.01 * s) for s in range(steps)]


# name -> (dim, depth, heads, window, feature map (B, D, H, W), mlp_ratio, qkv_bias, seed)
LAYER = {
    "layer_c128_2x4x4": (128, 2, 4, (2, 4, 4), (1, 4, 8, 8), 4.0, False, 601),        # BasicLayer defaults
    "layer_c256_3x8x8_clamped": (256, 2, 8, (3, 8, 8), (2, 3, 16, 8), 2.0, True, 602),  # D and W not larger than the window: clamped, no shift there
    # the reference's TDRQVAE stage (options/release_test_stage_IIII...yml:77-79: window [5,5,5], 8 heads) on a 3 x 32 x 32
    # latent: D clamped to 3, H and W padded 32 -> 35, 75 tokens per window, shift (0, 2, 2); depth 2 of its 4
    "layer_c512_5x5x5_tdrqvae": (512, 2, 8, (5, 5, 5), (1, 3, 32, 32), 4.0, False, 603),
    "layer_c64_2x3x3_padded_bias": (64, 2, 4, (2, 3, 3), (2, 3, 7, 8), 1.0, True, 604),   # padding on all three axes + qkv bias
}


def layer_params(name):
    dim, depth, heads, ws, fmap, mlp_ratio, qkv_bias, seed = LAYER[name]
    hid = int(dim * mlp_ratio)
    n_tab = (2 * ws[0] - 1) * (2 * ws[1] - 1) * (2 * ws[2] - 1)
    p = {}
    for i in range(depth):
        s = seed + 20 * i
        b = f"blocks.{i}."
        p[b + "norm1.weight"] = 1 + 0.1 * rnd((dim,), s + 1)
        p[b + "norm1.bias"] = 0.1 * rnd((dim,), s + 2)
        p[b + "attn.relative_position_bias_table"] = rnd((n_tab, heads), s + 3, 0.3)
        p[b + "attn.qkv.weight"] = rnd((3 * dim, dim), s + 4, dim ** -0.5)
        if qkv_bias:
            p[b + "attn.qkv.bias"] = 0.1 * rnd((3 * dim,), s + 5)
        p[b + "attn.proj.weight"] = rnd((dim, dim), s + 6, dim ** -0.5)
        p[b + "attn.proj.bias"] = 0.1 * rnd((dim,), s + 7)
        p[b + "norm2.weight"] = 1 + 0.1 * rnd((dim,), s + 8)
        p[b + "norm2.bias"] = 0.1 * rnd((dim,), s + 9)
        p[b + "mlp.fc1.weight"] = rnd((hid, dim), s + 10, dim ** -0.5)
        p[b + "mlp.fc1.bias"] = 0.1 * rnd((hid,), s + 11)
        p[b + "mlp.fc2.weight"] = rnd((dim, hid), s + 12, hid ** -0.5)
        p[b + "mlp.fc2.bias"] = 0.1 * rnd((dim,), s + 13)
    return p


def layer_input(name):
    dim, depth, heads, ws, fmap, mlp_ratio, qkv_bias, seed = LAYER[name]
    b, d, h, w = fmap
    return rnd((b, dim, d, h, w), seed)


# leading channels of the layer output kept in the fixture
KEEP = {"layer_c128_2x4x4": 64, "layer_c256_3x8x8_clamped": 64, "layer_c512_5x5x5_tdrqvae": 24, "layer_c64_2x3x3_padded_bias": 64}
