"""Shared definition of the per-function golden cases (SURVEY.md §8a rows at reduced shapes).

`make_golden.py` runs the REFERENCE module for each case (build container only) and stores the
outputs in `ops_golden.npz`; tests re-create the same inputs/weights and run the oracle (CPU) or
the HIP path (GPU) against the stored outputs.
"""
import numpy as np
import torch

from pgtformer_amd.config import default_config
from pgtformer_amd.manifest import pgtformer_manifest
from pgtformer_amd.weightgen import generate_state_dict

CFG = default_config()
MANIFEST = pgtformer_manifest(CFG)


def weights_for(prefix, seed=0):
    """Generated weights of one sub-module, keys relative to nothing (full names kept)."""
    return generate_state_dict(MANIFEST, CFG, seed=seed, only=lambda k: k.startswith(prefix))


def rand(shape, seed, scale=1.0):
    g = np.random.default_rng(seed)
    return torch.from_numpy((scale * g.standard_normal(shape)).astype(np.float32))


# name -> (kind, prefix, input shape(s), seed)
CASES = {
    # a6 TDResnetBlock: identity shortcut (5-D), nin_shortcut (5-D, B=1 broadcast), 4-D input
    "resblock_64": ("resblock", "encoder.down.0.block.0", (1, 3, 64, 12, 12), 11),
    "resblock_64_128_nin": ("resblock", "encoder.down.1.block.0", (1, 3, 64, 8, 8), 12),
    "resblock_512_4d": ("resblock", "decoder.mid.block_1", (3, 512, 8, 8), 13),
    # a7 / a8
    "downsample_128": ("downsample", "encoder.down.1.downsample", (1, 3, 128, 12, 12), 14),
    "upsample_256": ("upsample", "decoder.up.2.upsample", (1, 3, 256, 4, 4), 15),
    # a9-a13 EncoderLayer (two blocks: unshifted + shifted, mask, rel-pos bias), C=256 and C=512
    "enclayer_256": ("enclayer", "encoder.down.2.attn.0", (1, 3, 256, 8, 12), 16),
    "enclayer_512": ("enclayer", "encoder.mid.attn_1", (1, 3, 512, 8, 8), 17),
    # a16 TransformerSALayer at L=192 (T=3 x 8 x 8 tokens)
    "sa_layer": ("salayer", "ft_layers.0", (192, 1, 512), 18),
    # a22/a23 Fuse_sft_block, C=128 (ResBlock in-channels 288 -> 9 ch / group)
    "fuse_256": ("fuse", "fuse_convs_dict.256", (1, 3, 128, 8, 8), 19),
    "fuse_32": ("fuse", "fuse_convs_dict.32", (1, 3, 512, 4, 4), 20),
    # a19 AdaIN, a18 embed_code, a25 nearest-code lookup
    "adain": ("adain", "", (3, 512, 8, 8), 21),
    "embed_code": ("embed", "quantizer", (1, 32, 32, 1), 22),
    "rq_nearest": ("rq", "quantizer", (3, 8, 8, 512), 23),
}


def case_inputs(name):
    kind, prefix, shape, seed = CASES[name]
    if kind == "fuse":
        return [rand(shape, seed), rand(shape, seed + 1000)]
    if kind == "salayer":
        return [rand(shape, seed), rand(shape, seed + 1000, 0.5)]
    if kind == "adain":
        return [rand(shape, seed), rand(shape, seed + 1000, 2.0) + 0.3]
    if kind == "embed":
        g = np.random.default_rng(seed)
        return [torch.from_numpy(g.integers(0, 1024, shape).astype(np.int64))]
    if kind == "rq":
        return [rand(shape, seed, 0.3)]
    return [rand(shape, seed)]


def run_oracle(name, sd=None):
    """Run the oracle function of one case; returns a list of output tensors."""
    from oracle import pgt_oracle as O

    kind, prefix, shape, seed = CASES[name]
    if sd is None:
        sd = weights_for(prefix) if prefix else {}
    x = case_inputs(name)
    if kind == "resblock":
        return [O.td_resblock(sd, prefix, x[0])]
    if kind == "downsample":
        return [O.downsample(sd, prefix, x[0])]
    if kind == "upsample":
        return [O.upsample(sd, prefix, x[0])]
    if kind == "enclayer":
        return [O.encoder_layer(sd, prefix, x[0], 8, (4, 4), 2)]
    if kind == "salayer":
        return [O.transformer_sa_layer(sd, prefix, x[0], x[1], 8)]
    if kind == "fuse":
        return [O.fuse_sft(sd, prefix, x[0], x[1], 1.0)]
    if kind == "adain":
        return [O.adain(x[0], x[1])]
    if kind == "embed":
        return [O.embed_code(sd, x[0], True)[:, ::4, ::4]]  # stored sub-sampled (fixture size)
    if kind == "rq":
        agg, codes = O.rq_quantize(sd, x[0], 1, True)
        return [agg, codes]
    raise KeyError(kind)
