"""State-dict manifest of the reference `PGTFormer`: every key, shape and dtype, derived from the config.

The reference model has 961 state-dict tensors at the shipping config. This module
enumerates them from the config alone so that (a) the synthetic weight generator can run anywhere,
(b) checkpoints can be validated key-for-key before the HIP weight repack. The enumeration is pinned
against the imported reference by tests/golden/state_dict_manifest.json.

Reference constructors followed: archs/pgtformer_arch.py:490-556 (PGTFormer), :34-397 (BiSeNet),
:409-458 (ResBlock, Fuse_sft_block); archs/tdcrqvae3_arch.py:460-538 (Encoder), :577-670 (Decoder),
:80-97 (VQEmbedding), :711-755 (TDCRQVAE3); modules/rstt_layers.py:139-193 (WindowAttention3D),
:236-282 (VSTSREncoderTransformerBlock), :835-873 (TDResnetBlock); archs/codeformer_arch.py:102-116.
"""
from collections import OrderedDict

F32 = "float32"
I64 = "int64"


class _M(OrderedDict):
    def add(self, key, shape, dtype=F32):
        assert key not in self, key
        self[key] = (tuple(int(s) for s in shape), dtype)


def _conv(m, p, cin, cout, k, bias=True):
    m.add(p + ".weight", (cout, cin, k, k))
    if bias:
        m.add(p + ".bias", (cout,))


def _linear(m, p, cin, cout, bias=True):
    m.add(p + ".weight", (cout, cin))
    if bias:
        m.add(p + ".bias", (cout,))


def _norm(m, p, c):
    m.add(p + ".weight", (c,))
    m.add(p + ".bias", (c,))


def _bn(m, p, c):
    m.add(p + ".weight", (c,))
    m.add(p + ".bias", (c,))
    m.add(p + ".running_mean", (c,))
    m.add(p + ".running_var", (c,))
    m.add(p + ".num_batches_tracked", (), I64)


def _td_resblock(m, p, cin, cout):
    _norm(m, p + ".norm1", cin)
    _conv(m, p + ".conv1", cin, cout, 3)
    _norm(m, p + ".norm2", cout)
    _conv(m, p + ".conv2", cout, cout, 3)
    if cin != cout:
        _conv(m, p + ".nin_shortcut", cin, cout, 1)


def _encoder_layer(m, p, dim, depth, heads, frames, win):
    n = frames * win[0] * win[1]
    for i in range(depth):
        b = f"{p}.blocks.{i}"
        _norm(m, b + ".norm1", dim)
        m.add(b + ".attn.relative_position_bias_table",
              ((2 * frames - 1) * (2 * win[0] - 1) * (2 * win[1] - 1), heads))
        m.add(b + ".attn.relative_position_index", (n, n), I64)
        _linear(m, b + ".attn.q", dim, dim)
        _linear(m, b + ".attn.kv", dim, 2 * dim)
        _linear(m, b + ".attn.proj", dim, dim)
        _norm(m, b + ".norm2", dim)
        _linear(m, b + ".mlp.fc1", dim, dim)  # mlp_ratio=1 (tdcrqvae3_arch.py:499)
        _linear(m, b + ".mlp.fc2", dim, dim)


def _encoder(m, p, dd):
    ch, mult = dd["ch"], list(dd["ch_mult"])
    nres, res = dd["num_res_blocks"], dd["resolution"]
    nlev = len(mult)
    _conv(m, p + ".conv_in", dd["in_channels"], ch, 3)
    in_mult = [1] + mult
    cur = res
    block_in = ch
    for lvl in range(nlev):
        block_in = ch * in_mult[lvl]
        block_out = ch * mult[lvl]
        k = 0
        for b in range(nres):
            _td_resblock(m, f"{p}.down.{lvl}.block.{b}", block_in, block_out)
            block_in = block_out
            if cur in dd["attn_resolutions"]:
                _encoder_layer(m, f"{p}.down.{lvl}.attn.{k}", block_in, dd["depths"][lvl],
                               dd["num_heads"][lvl], dd["num_frames"], dd["window_sizes"][lvl])
                k += 1
        if lvl != nlev - 1:
            _conv(m, f"{p}.down.{lvl}.downsample.conv", block_in, block_in, 3)
            cur //= 2
    _td_resblock(m, p + ".mid.block_1", block_in, block_in)
    _encoder_layer(m, p + ".mid.attn_1", block_in, dd["depths"][nlev - 1], dd["num_heads"][nlev - 1],
                   dd["num_frames"], dd["window_sizes"][nlev - 1])
    _td_resblock(m, p + ".mid.block_2", block_in, block_in)
    _norm(m, p + ".norm_out", block_in)
    zc = 2 * dd["z_channels"] if dd.get("double_z", True) else dd["z_channels"]
    _conv(m, p + ".conv_out", block_in, zc, 3)


def _decoder(m, p, dd):
    ch, mult = dd["ch"], list(dd["ch_mult"])
    nres, res = dd["num_res_blocks"], dd["resolution"]
    nlev = len(mult)
    block_in = ch * mult[nlev - 1]
    cur = res // 2 ** (nlev - 1)
    _conv(m, p + ".conv_in", dd["z_channels"], block_in, 3)
    _td_resblock(m, p + ".mid.block_1", block_in, block_in)
    _encoder_layer(m, p + ".mid.attn_1", block_in, dd["depths"][-1], dd["num_heads"][-1],
                   dd["num_frames"], dd["window_sizes"][-1])
    _td_resblock(m, p + ".mid.block_2", block_in, block_in)
    # state_dict order follows module registration: mid, then up.0 .. up.N (the list is built
    # top level first and prepended; reference: tdcrqvae3_arch.py:627-662)
    per_level = {}
    for lvl in reversed(range(nlev)):
        sub = _M()
        block_out = ch * mult[lvl]
        k = 0
        for b in range(nres + 1):
            _td_resblock(sub, f"{p}.up.{lvl}.block.{b}", block_in, block_out)
            block_in = block_out
        # attn modules registered after the blocks of the level (up.block then up.attn)
        if cur in dd["attn_resolutions"]:
            for b in range(nres + 1):
                _encoder_layer(sub, f"{p}.up.{lvl}.attn.{k}", block_in, dd["depths"][lvl],
                               dd["num_heads"][lvl], dd["num_frames"], dd["window_sizes"][lvl])
                k += 1
        if lvl != 0:
            _conv(sub, f"{p}.up.{lvl}.upsample.conv", block_in, block_in, 3)
            cur *= 2
        per_level[lvl] = sub
    for lvl in range(nlev):
        for k, v in per_level[lvl].items():
            m[k] = v
    _norm(m, p + ".norm_out", block_in)
    _conv(m, p + ".conv_out", block_in, dd["out_ch"], 3)


def _conv_bn_relu(m, p, cin, cout, k):
    _conv(m, p + ".conv", cin, cout, k, bias=False)
    _bn(m, p + ".bn", cout)


def _basic_block(m, p, cin, cout, stride):
    _conv(m, p + ".conv1", cin, cout, 3, bias=False)
    _bn(m, p + ".bn1", cout)
    _conv(m, p + ".conv2", cout, cout, 3, bias=False)
    _bn(m, p + ".bn2", cout)
    if cin != cout or stride != 1:
        _conv(m, p + ".downsample.0", cin, cout, 1, bias=False)
        _bn(m, p + ".downsample.1", cout)


def _bisenet(m, p, n_classes=19):
    r = p + ".cp.resnet"
    _conv(m, r + ".conv1", 3, 64, 7, bias=False)
    _bn(m, r + ".bn1", 64)
    for name, cin, cout, stride in (("layer1", 64, 64, 1), ("layer2", 64, 128, 2),
                                    ("layer3", 128, 256, 2), ("layer4", 256, 512, 2)):
        _basic_block(m, f"{r}.{name}.0", cin, cout, stride)
        _basic_block(m, f"{r}.{name}.1", cout, cout, 1)
    for name, cin in (("arm16", 256), ("arm32", 512)):
        a = f"{p}.cp.{name}"
        _conv_bn_relu(m, a + ".conv", cin, 128, 3)
        _conv(m, a + ".conv_atten", 128, 128, 1, bias=False)
        _bn(m, a + ".bn_atten", 128)
    _conv_bn_relu(m, p + ".cp.conv_head32", 128, 128, 3)
    _conv_bn_relu(m, p + ".cp.conv_head16", 128, 128, 3)
    _conv_bn_relu(m, p + ".cp.conv_avg", 512, 128, 1)
    _conv_bn_relu(m, p + ".ffm.convblk", 256, 256, 1)
    _conv(m, p + ".ffm.conv1", 256, 64, 1, bias=False)
    _conv(m, p + ".ffm.conv2", 64, 256, 1, bias=False)
    for name, cin, mid in (("conv_out", 256, 256), ("conv_out16", 128, 64), ("conv_out32", 128, 64)):
        _conv_bn_relu(m, f"{p}.{name}.conv", cin, mid, 3)
        _conv(m, f"{p}.{name}.conv_out", mid, n_classes, 1, bias=False)


FUSE_CHANNELS = {"16": 512, "32": 512, "64": 256, "128": 256, "256": 128, "512": 64}


def _fuse_block(m, p, c, t, tcc=32):
    e = p + ".encode_enc"
    cin = 2 * c + tcc
    _norm(m, e + ".norm1", cin)
    _conv(m, e + ".conv1", cin, c, 3)
    _norm(m, e + ".norm2", c)
    _conv(m, e + ".conv2", c, c, 3)
    _conv(m, e + ".conv_out", cin, c, 1)
    for name in ("scale", "shift"):
        _conv(m, f"{p}.{name}.0", c, c, 3)
        _conv(m, f"{p}.{name}.2", c, c, 3)
    _conv(m, p + ".tconvenc", c, tcc, 1)
    _conv(m, p + ".tconvdec", c, tcc, 1)
    _conv(m, p + ".tfusion0", 2 * t * tcc, tcc * t, 1)
    _conv(m, p + ".tfusion1", tcc, tcc, 1)


def tdcrqvae3_manifest(cfg):
    """Keys of `TDCRQVAE3` (stage-I RQ-VAE; reference: archs/tdcrqvae3_arch.py:711-755)."""
    m = _M()
    dd = cfg["ddconfig"]
    _encoder(m, "encoder", dd)
    _decoder(m, "decoder", dd)
    n_embed, embed_dim = cfg["n_embed"], cfg["embed_dim"]
    depth = cfg["code_shape"][-1]
    ls, cs = cfg["latent_shape"], cfg["code_shape"]
    vq_dim = (ls[0] * ls[1]) // (cs[0] * cs[1]) * ls[2]
    n_books = depth  # a shared codebook is the same module listed `depth` times (duplicate keys)
    for i in range(n_books):
        q = f"quantizer.codebooks.{i}"
        m.add(q + ".weight", (n_embed + 1, vq_dim))
        m.add(q + ".cluster_size_ema", (n_embed,))
        m.add(q + ".embed_ema", (n_embed, vq_dim))
    _conv(m, "quant_conv", dd["z_channels"], embed_dim, 1)
    _conv(m, "post_quant_conv", embed_dim, dd["z_channels"], 1)
    return m


def pgtformer_manifest(cfg):
    """Keys of `PGTFormer` in the reference's `state_dict()` order."""
    from .config import PGTFORMER_DEFAULTS

    full = dict(PGTFORMER_DEFAULTS)
    full.update(cfg)
    m = tdcrqvae3_manifest(full)
    t = full["tf"]
    dim, nl = full["dim_embd"], full["n_layers"]
    _bisenet(m, "conditionnet")
    _conv(m, "convpos", 57, 512, 1)
    _linear(m, "feat_emb", 512, dim)
    for i in range(nl):
        f = f"ft_layers.{i}"
        m.add(f + ".self_attn.in_proj_weight", (3 * dim, dim))
        m.add(f + ".self_attn.in_proj_bias", (3 * dim,))
        _linear(m, f + ".self_attn.out_proj", dim, dim)
        _linear(m, f + ".linear1", dim, 2 * dim)
        _linear(m, f + ".linear2", 2 * dim, dim)
        _norm(m, f + ".norm1", dim)
        _norm(m, f + ".norm2", dim)
    depth = full["code_shape"][-1]
    _norm(m, "idx_pred_layer.0", dim)
    _linear(m, "idx_pred_layer.1", dim, depth * full["n_embed"], bias=False)
    for fs in full["connect_list"]:
        _fuse_block(m, f"fuse_convs_dict.{fs}", FUSE_CHANNELS[fs], t)
    return m


def manifest_to_json(m):
    return {k: [list(s), "torch." + d] for k, (s, d) in m.items()}
