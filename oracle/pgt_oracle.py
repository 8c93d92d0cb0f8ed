"""ORACLE — test infrastructure only. Not part of the product path.

CPU (torch fp32, eager) restatement of the reference PGTFormer forward pass, written functionally over
a flat state dict. Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import this module, and only as the checker / reported CPU baseline. The product
(`pgtformer_amd`) never imports it and has no CPU fallback.

Pinning: the reference ships no tests or golden vectors. This restatement is
pinned against the reference ITSELF, imported in the build container by
`tests/golden/make_golden.py` (which also writes the fixtures under tests/golden/): whole-model
outputs agree bit-for-bit / to fp32 round-off on the shipping config (see tests/test_oracle_golden.py).

Each function cites the reference lines it follows. Layout is the reference's (NCHW, (B,D,C,H,W) for
frame stacks); leaf arithmetic is torch ATen CPU, the same library the reference's CPU path uses.
"""
import math

import numpy as np

import torch
import torch.nn.functional as F

T_FRAMES = 3


def _conv(sd, p, x, stride=1, padding=0):
    return F.conv2d(x, sd[p + ".weight"], sd.get(p + ".bias"), stride=stride, padding=padding)


def _gn(sd, p, x, eps=1e-6):
    # Normalize(): GroupNorm(32, C, eps=1e-6, affine)  (reference: modules/rstt_layers.py:754-755,
    # archs/pgtformer_arch.py:406-407)
    return F.group_norm(x, 32, sd[p + ".weight"], sd[p + ".bias"], eps)


def _ln(sd, p, x, eps=1e-5):
    c = x.shape[-1]
    return F.layer_norm(x, (c,), sd[p + ".weight"], sd[p + ".bias"], eps)


def _bn(sd, p, x, eps=1e-5):
    # eval-mode BatchNorm2d
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"],
                        sd[p + ".bias"], False, 0.0, eps)


def _lin(sd, p, x):
    return F.linear(x, sd[p + ".weight"], sd.get(p + ".bias"))


# ----------------------------------------------------------------------------------------------
# TDResnetBlock (reference: modules/rstt_layers.py:875-904)
# ----------------------------------------------------------------------------------------------
def td_resblock(sd, p, x):
    five_d = x.dim() == 5
    if five_d:
        b, d, c, h, w = x.shape
        inp = x.reshape(b * d, c, h, w)
    else:
        inp = x
    hdn = F.silu(_gn(sd, p + ".norm1", inp))
    hdn = _conv(sd, p + ".conv1", hdn, padding=1)
    hdn = F.silu(_gn(sd, p + ".norm2", hdn))
    hdn = _conv(sd, p + ".conv2", hdn, padding=1)
    if (p + ".nin_shortcut.weight") in sd:
        sc = _conv(sd, p + ".nin_shortcut", inp)
    else:
        sc = inp
    out = sc + hdn
    if five_d:
        out = out.reshape(b, d, -1, h, w)
    return out


# ----------------------------------------------------------------------------------------------
# Window attention stack (reference: modules/rstt_layers.py:55-88, 195-234, 284-338, 535-575)
# ----------------------------------------------------------------------------------------------
def _partition(x, ws):
    b, d, h, w, c = x.shape
    x = x.reshape(b, d, h // ws[0], ws[0], w // ws[1], ws[1], c)
    return x.permute(0, 2, 4, 1, 3, 5, 6).reshape(-1, d * ws[0] * ws[1], c)


def _unpartition(win, ws, b, d, h, w):
    x = win.reshape(b, h // ws[0], w // ws[1], d, ws[0], ws[1], -1)
    return x.permute(0, 3, 1, 4, 2, 5, 6).reshape(b, d, h, w, -1)


def shift_mask(d, h, w, ws, ss):
    """9-region shifted-window mask, values in {0,-100} (reference: rstt_layers.py:552-568)."""
    img = torch.zeros(1, d, h, w, 1)
    cnt = 0
    for hs in (slice(0, -ws[0]), slice(-ws[0], -ss[0]), slice(-ss[0], None)):
        for wsl in (slice(0, -ws[1]), slice(-ws[1], -ss[1]), slice(-ss[1], None)):
            img[:, :, hs, wsl, :] = cnt
            cnt += 1
    mw = _partition(img, ws).squeeze(-1)  # nW, N
    am = mw.unsqueeze(1) - mw.unsqueeze(2)
    return torch.where(am != 0, torch.full_like(am, -100.0), torch.zeros_like(am))


def window_attention(sd, p, xw, heads, mask):
    """WindowAttention3D.forward, self-attention case (reference: rstt_layers.py:195-234)."""
    bw, n, c = xw.shape
    hd = c // heads
    q = _lin(sd, p + ".q", xw).reshape(bw, n, heads, hd).permute(0, 2, 1, 3)
    kv = _lin(sd, p + ".kv", xw).reshape(bw, n, 2, heads, hd).permute(2, 0, 3, 1, 4)
    k, v = kv[0], kv[1]
    q = q * (hd ** -0.5)
    attn = q @ k.transpose(-2, -1)
    idx = sd[p + ".relative_position_index"].reshape(-1)
    bias = sd[p + ".relative_position_bias_table"][idx].reshape(n, n, heads).permute(2, 0, 1)
    attn = attn + bias.unsqueeze(0)
    if mask is not None:
        nw = mask.shape[0]
        attn = attn.reshape(bw // nw, nw, heads, n, n) + mask.unsqueeze(1).unsqueeze(0)
        attn = attn.reshape(-1, heads, n, n)
    attn = attn.softmax(-1)
    out = (attn @ v).transpose(1, 2).reshape(bw, n, c)
    return _lin(sd, p + ".proj", out)


def vstsr_block(sd, p, x, heads, ws, ss, mask):
    """VSTSREncoderTransformerBlock.forward (reference: rstt_layers.py:284-338). x: (B,D,H,W,C);
    H, W are multiples of the window in every shipped shape so the pad branch is a no-op."""
    b, d, h, w, c = x.shape
    assert h % ws[0] == 0 and w % ws[1] == 0
    shortcut = x
    y = _ln(sd, p + ".norm1", x)
    shifted = ss[0] > 0 or ss[1] > 0
    if shifted:
        y = torch.roll(y, shifts=(-ss[0], -ss[1]), dims=(2, 3))
    yw = _partition(y, ws)
    aw = window_attention(sd, p + ".attn", yw, heads, mask if shifted else None)
    y = _unpartition(aw, ws, b, d, h, w)
    if shifted:
        y = torch.roll(y, shifts=(ss[0], ss[1]), dims=(2, 3))
    x = shortcut + y
    m = _lin(sd, p + ".mlp.fc2", F.gelu(_lin(sd, p + ".mlp.fc1", _ln(sd, p + ".norm2", x))))
    return x + m


def encoder_layer(sd, p, x, heads=8, ws=(4, 4), depth=2):
    """EncoderLayer.forward (reference: rstt_layers.py:535-575). x: (B,D,C,H,W)."""
    b, d, c, h, w = x.shape
    ws = (min(ws[0], h), min(ws[1], w))
    ss = (ws[0] // 2 if h > ws[0] else 0, ws[1] // 2 if w > ws[1] else 0)
    y = x.permute(0, 1, 3, 4, 2)
    mask = shift_mask(d, h, w, ws, ss) if (ss[0] > 0 or ss[1] > 0) else None
    for i in range(depth):
        blk_ss = (0, 0) if i % 2 == 0 else ss
        y = vstsr_block(sd, f"{p}.blocks.{i}", y, heads, ws, blk_ss, mask)
    return y.permute(0, 1, 4, 2, 3)


# ----------------------------------------------------------------------------------------------
# Encoder / Decoder resampling (reference: archs/tdcrqvae3_arch.py:45-52, 67-76)
# ----------------------------------------------------------------------------------------------
def downsample(sd, p, x):
    b, d, c, h, w = x.shape
    y = F.pad(x.reshape(b * d, c, h, w), (0, 1, 0, 1))
    y = _conv(sd, p + ".conv", y, stride=2)
    return y.reshape(b, d, -1, h // 2, w // 2)


def upsample(sd, p, x):
    b, d, c, h, w = x.shape
    y = F.interpolate(x.reshape(b * d, c, h, w), scale_factor=2.0, mode="nearest")
    y = _conv(sd, p + ".conv", y, padding=1)
    return y.reshape(b, d, -1, 2 * h, 2 * w)


def encoder_forward(sd, dd, x, p="encoder"):
    """Encoder.forward with return_multi_res_feats=True (reference: tdcrqvae3_arch.py:540-573)."""
    b, d, c, h, w = x.shape
    nlev = len(dd["ch_mult"])
    hcur = _conv(sd, p + ".conv_in", x.reshape(b * d, c, h, w), padding=1).reshape(b, d, -1, h, w)
    feats = []
    res = dd["resolution"]
    for lvl in range(nlev):
        k = 0
        for blk in range(dd["num_res_blocks"]):
            hcur = td_resblock(sd, f"{p}.down.{lvl}.block.{blk}", hcur)
            if res in dd["attn_resolutions"]:
                hcur = encoder_layer(sd, f"{p}.down.{lvl}.attn.{k}", hcur, dd["num_heads"][lvl],
                                     tuple(dd["window_sizes"][lvl]), dd["depths"][lvl])
                k += 1
        feats.append(hcur)
        if lvl != nlev - 1:
            hcur = downsample(sd, f"{p}.down.{lvl}.downsample", hcur)
            res //= 2
    hcur = td_resblock(sd, p + ".mid.block_1", hcur)
    hcur = encoder_layer(sd, p + ".mid.attn_1", hcur, dd["num_heads"][-1],
                         tuple(dd["window_sizes"][-1]), dd["depths"][-1])
    hcur = td_resblock(sd, p + ".mid.block_2", hcur)
    b0, d0, c0, h0, w0 = hcur.shape
    y = F.silu(_gn(sd, p + ".norm_out", hcur.reshape(b0 * d0, c0, h0, w0)))
    y = _conv(sd, p + ".conv_out", y, padding=1)
    return y, feats


def decoder_forward(sd, dd, z, t=T_FRAMES, fuse=None, p="decoder"):
    """Decoder.forward (reference: tdcrqvae3_arch.py:672-707) and, with `fuse`, the inlined loop of
    PGTFormer.forward (reference: archs/pgtformer_arch.py:684-710). fuse(f_size:str, h)->h."""
    nlev = len(dd["ch_mult"])
    h = _conv(sd, p + ".conv_in", z, padding=1)
    h = td_resblock(sd, p + ".mid.block_1", h)  # 4-D input on purpose (reference :686)
    bd, c, hh, ww = h.shape
    h = h.reshape(bd // t, t, c, hh, ww)
    h = encoder_layer(sd, p + ".mid.attn_1", h, dd["num_heads"][-1], tuple(dd["window_sizes"][-1]),
                      dd["depths"][-1])
    h = td_resblock(sd, p + ".mid.block_2", h)
    res = dd["resolution"] // 2 ** (nlev - 1)
    for lvl in reversed(range(nlev)):
        for blk in range(dd["num_res_blocks"] + 1):
            h = td_resblock(sd, f"{p}.up.{lvl}.block.{blk}", h)
            if res in dd["attn_resolutions"]:
                h = encoder_layer(sd, f"{p}.up.{lvl}.attn.{blk}", h, dd["num_heads"][lvl],
                                  tuple(dd["window_sizes"][lvl]), dd["depths"][lvl])
        if fuse is not None:
            h = fuse(str(h.shape[-1]), h)
        if lvl != 0:
            h = upsample(sd, f"{p}.up.{lvl}.upsample", h)
            res *= 2
    b, d, c, hh, ww = h.shape
    y = F.silu(_gn(sd, p + ".norm_out", h.reshape(b * d, c, hh, ww)))
    return _conv(sd, p + ".conv_out", y, padding=1)


# ----------------------------------------------------------------------------------------------
# BiSeNet condition net (reference: archs/pgtformer_arch.py:40-68, 78-100, 138-153, 161-171,
# 191-207, 216-249, 304-334, 354-379)
# ----------------------------------------------------------------------------------------------
def _cbr(sd, p, x, k, stride=1):
    y = _conv(sd, p + ".conv", x, stride=stride, padding=k // 2)
    return F.relu(_bn(sd, p + ".bn", y))


def _basic_block(sd, p, x, stride):
    r = F.relu(_bn(sd, p + ".bn1", _conv(sd, p + ".conv1", x, stride=stride, padding=1)))
    r = _bn(sd, p + ".bn2", _conv(sd, p + ".conv2", r, padding=1))
    sc = x
    if (p + ".downsample.0.weight") in sd:
        sc = _bn(sd, p + ".downsample.1", _conv(sd, p + ".downsample.0", x, stride=stride))
    return F.relu(sc + r)


def _arm(sd, p, x):
    feat = _cbr(sd, p + ".conv", x, 3)
    att = F.avg_pool2d(feat, feat.shape[2:])
    att = torch.sigmoid(_bn(sd, p + ".bn_atten", _conv(sd, p + ".conv_atten", att)))
    return feat * att


def _bise_out(sd, p, x):
    return _conv(sd, p + ".conv_out", _cbr(sd, p + ".conv", x, 3))


def bisenet_forward(sd, x, p="conditionnet"):
    r = p + ".cp.resnet"
    y = F.relu(_bn(sd, r + ".bn1", _conv(sd, r + ".conv1", x, stride=2, padding=3)))
    y = F.max_pool2d(y, 3, 2, 1)
    for name, stride in (("layer1", 1), ("layer2", 2), ("layer3", 2), ("layer4", 2)):
        y = _basic_block(sd, f"{r}.{name}.0", y, stride)
        y = _basic_block(sd, f"{r}.{name}.1", y, 1)
        if name == "layer2":
            feat8 = y
        elif name == "layer3":
            feat16 = y
    feat32 = y
    cp = p + ".cp"
    avg = _cbr(sd, cp + ".conv_avg", F.avg_pool2d(feat32, feat32.shape[2:]), 1)
    avg_up = F.interpolate(avg, feat32.shape[2:], mode="nearest")
    f32 = _arm(sd, cp + ".arm32", feat32) + avg_up
    f32 = _cbr(sd, cp + ".conv_head32", F.interpolate(f32, feat16.shape[2:], mode="nearest"), 3)
    f16 = _arm(sd, cp + ".arm16", feat16) + f32
    f16 = _cbr(sd, cp + ".conv_head16", F.interpolate(f16, feat8.shape[2:], mode="nearest"), 3)
    feat_cp8, feat_cp16 = f16, f32
    # FeatureFusionModule (:324-334)
    fm = p + ".ffm"
    feat = _cbr(sd, fm + ".convblk", torch.cat([feat8, feat_cp8], 1), 1)
    att = F.avg_pool2d(feat, feat.shape[2:])
    att = torch.sigmoid(_conv(sd, fm + ".conv2", F.relu(_conv(sd, fm + ".conv1", att))))
    fuse = feat * att + feat
    out = _bise_out(sd, p + ".conv_out", fuse)
    out16 = _bise_out(sd, p + ".conv_out16", feat_cp8)
    out32 = _bise_out(sd, p + ".conv_out32", feat_cp16)
    out = F.interpolate(out, (32, 32), mode="bilinear", align_corners=True)
    out16 = F.interpolate(out16, (32, 32), mode="bilinear", align_corners=True)
    return torch.cat([out, out16, out32], 1)


# ----------------------------------------------------------------------------------------------
# Code-prediction transformer (reference: archs/codeformer_arch.py:121-137, nn.MultiheadAttention)
# ----------------------------------------------------------------------------------------------
def mha(sd, p, q_in, k_in, v_in, heads):
    """nn.MultiheadAttention forward, seq-first (L,B,E), no masks, need_weights path: q scaled by
    hd^-0.5 before QK^T, softmax, PV, out_proj."""
    l, b, e = q_in.shape
    hd = e // heads
    w, bias = sd[p + ".in_proj_weight"], sd[p + ".in_proj_bias"]
    q = F.linear(q_in, w[:e], bias[:e])
    k = F.linear(k_in, w[e:2 * e], bias[e:2 * e])
    v = F.linear(v_in, w[2 * e:], bias[2 * e:])
    q = q.reshape(l, b * heads, hd).transpose(0, 1) * math.sqrt(1.0 / hd)
    k = k.reshape(l, b * heads, hd).transpose(0, 1)
    v = v.reshape(l, b * heads, hd).transpose(0, 1)
    attn = torch.bmm(q, k.transpose(1, 2)).softmax(-1)
    o = torch.bmm(attn, v).transpose(0, 1).reshape(l, b, e)
    return _lin(sd, p + ".out_proj", o)


def transformer_sa_layer(sd, p, tgt, pos, heads):
    t2 = _ln(sd, p + ".norm1", tgt)
    qk = t2 + pos
    tgt = tgt + mha(sd, p + ".self_attn", qk, qk, t2, heads)
    t2 = _ln(sd, p + ".norm2", tgt)
    t2 = _lin(sd, p + ".linear2", F.gelu(_lin(sd, p + ".linear1", t2)))
    return tgt + t2


# ----------------------------------------------------------------------------------------------
# Quantiser pieces
# ----------------------------------------------------------------------------------------------
def embed_code(sd, codes, shared=True):
    """RQBottleneck.embed_code for shape_divisor 1 (reference: tdcrqvae3_arch.py:355-368)."""
    depth = codes.shape[-1]
    outs = []
    for i in range(depth):
        book = sd[f"quantizer.codebooks.{0 if shared else i}.weight"]
        outs.append(F.embedding(codes[..., i], book))
    return torch.stack(outs, -2).sum(-2)


def rq_quantize(sd, x, depth, shared=True):
    """RQBottleneck.quantize + VQEmbedding nearest lookup in eval mode (reference:
    tdcrqvae3_arch.py:100-126, 294-328). x: (B,h,w,D). Returns (aggregated quant, codes)."""
    resid = x.clone()
    agg = torch.zeros_like(x)
    codes = []
    for i in range(depth):
        book = sd[f"quantizer.codebooks.{0 if shared else i}.weight"][:-1]
        flat = resid.reshape(-1, resid.shape[-1])
        cb_t = book.t()
        dist = torch.addmm(flat.pow(2.0).sum(1, keepdim=True) + cb_t.pow(2.0).sum(0, keepdim=True),
                           flat, cb_t, alpha=-2.0)
        idx = dist.argmin(-1).reshape(resid.shape[:-1])
        q = F.embedding(idx, sd[f"quantizer.codebooks.{0 if shared else i}.weight"])
        resid = resid - q
        agg = agg + q
        codes.append(idx.unsqueeze(-1))
    return agg, torch.cat(codes, -1)


def adain(content, style, eps=1e-5):
    """adaptive_instance_normalization (reference: archs/codeformer_arch.py:15-46); unbiased var."""
    b, c = content.shape[:2]

    def ms(f):
        v = f.reshape(b, c, -1).var(dim=2) + eps
        return f.reshape(b, c, -1).mean(dim=2).reshape(b, c, 1, 1), v.sqrt().reshape(b, c, 1, 1)

    sm, ss = ms(style)
    cm, cs = ms(content)
    return (content - cm) / cs * ss + sm


# ----------------------------------------------------------------------------------------------
# SFT fusion (reference: archs/pgtformer_arch.py:421-432, 460-484)
# ----------------------------------------------------------------------------------------------
def res_block(sd, p, x):
    h = _gn(sd, p + ".norm1", x)
    h = _conv(sd, p + ".conv1", h * torch.sigmoid(h), padding=1)
    h = _gn(sd, p + ".norm2", h)
    h = _conv(sd, p + ".conv2", h * torch.sigmoid(h), padding=1)
    if (p + ".conv_out.weight") in sd:
        x = _conv(sd, p + ".conv_out", x)
    return h + x


def fuse_sft(sd, p, enc, dec, w, tcc=32):
    b, d, c, h, wf = enc.shape
    enc = enc.reshape(b * d, c, h, wf)
    dec = dec.reshape(b * d, c, h, wf)
    et = _conv(sd, p + ".tconvenc", enc).reshape(b, d * tcc, h, wf)
    dt = _conv(sd, p + ".tconvdec", dec).reshape(b, d * tcc, h, wf)
    fut = _conv(sd, p + ".tfusion0", torch.cat([et, dt], 1)).reshape(b * d, tcc, h, wf)
    fut = _conv(sd, p + ".tfusion1", fut)
    e = res_block(sd, p + ".encode_enc", torch.cat([enc, dec, fut], 1))

    def branch(name):
        y = F.leaky_relu(_conv(sd, f"{p}.{name}.0", e, padding=1), 0.2)
        return _conv(sd, f"{p}.{name}.2", y, padding=1)

    scale, shift = branch("scale"), branch("shift")
    out = dec + w * (dec * scale + shift)
    return out.reshape(b, d, -1, h, wf)


# ----------------------------------------------------------------------------------------------
# Whole model (reference: archs/pgtformer_arch.py:598-714)
# ----------------------------------------------------------------------------------------------
IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)
FUSE_IDX = {"512": 0, "256": 1, "128": 2, "64": 3, "32": 4, "16": 5}


@torch.no_grad()
def pgtformer_forward(sd, cfg, x, w=1.0, adain_on=None, code_only=False, taps=None):
    """x: (B*T,3,512,512) fp32 in [0,1]. Returns (out, logits, lq_feat_nhwc) like the reference."""
    from pgtformer_amd.config import PGTFORMER_DEFAULTS

    full = dict(PGTFORMER_DEFAULTS)
    full.update(cfg)
    dd = full["ddconfig"]
    t = full["tf"]
    heads = full["n_head"]
    if adain_on is None:
        adain_on = full["adain"]
    mean = torch.tensor(IMAGENET_MEAN).view(1, 3, 1, 1)
    std = torch.tensor(IMAGENET_STD).view(1, 3, 1, 1)
    cond = bisenet_forward(sd, (x - mean) / std)
    if taps is not None:
        taps["cond"] = cond
    cond = _conv(sd, "convpos", cond)
    tb, tc, th, tw = cond.shape
    b = tb // t
    pos = cond.reshape(b, t, tc, th, tw).permute(0, 2, 1, 3, 4).reshape(b, tc, t * th * tw)
    pos = pos.permute(2, 0, 1)  # (T*H*W, B, C), t-major

    bt, c, h, wf = x.shape
    z, feats = encoder_forward(sd, dd, x.reshape(b, t, c, h, wf))
    enc_feats = {}
    for fs in full["connect_list"]:
        f = feats[FUSE_IDX[fs]]
        enc_feats[str(f.shape[-1])] = f
    lq = _conv(sd, "quant_conv", z)
    if taps is not None:
        taps["lq_feat"] = lq
        taps["enc_feats"] = feats

    fe = _lin(sd, "feat_emb", lq.flatten(2).permute(2, 0, 1))  # (HW, BT, C)
    cc = fe.shape[-1]
    q = fe.reshape(th * tw, b, t, cc).permute(2, 0, 1, 3).reshape(t * th * tw, b, cc)
    for i in range(full["n_layers"]):
        q = transformer_sa_layer(sd, f"ft_layers.{i}", q, pos, heads)
    if taps is not None:
        taps["query_emb"] = q
    q = q.reshape(t, th * tw, b, cc).permute(1, 2, 0, 3).reshape(th * tw, b * t, cc)
    logits = F.linear(_ln(sd, "idx_pred_layer.0", q), sd["idx_pred_layer.1.weight"])
    code_shape = tuple(full["code_shape"])
    logits = logits.transpose(0, 1).reshape(b * t, *code_shape, full["n_embed"])
    lq_nhwc = lq.permute(0, 2, 3, 1)
    if code_only:
        return logits, lq_nhwc

    codes = logits.argmax(-1)
    quant = embed_code(sd, codes, full["shared_codebook"]).permute(0, 3, 1, 2).contiguous()
    if taps is not None:
        taps["codes"] = codes
    if adain_on:
        quant = adain(quant, lq)
    zq = _conv(sd, "post_quant_conv", quant)

    def fuse(fs, hcur):
        if fs in full["connect_list"] and w > 0:
            return fuse_sft(sd, f"fuse_convs_dict.{fs}", enc_feats[fs], hcur, w)
        return hcur

    out = decoder_forward(sd, dd, zq, t, fuse)
    return out, logits, lq_nhwc


@torch.no_grad()
def tdcrqvae3_forward(sd, cfg, x):
    """TDCRQVAE3.forward in eval mode (reference: tdcrqvae3_arch.py:760-783, 330-352).
    Returns (out, commitment_loss, codes)."""
    dd = cfg["ddconfig"]
    t = cfg.get("tf", T_FRAMES)
    bt, c, h, w = x.shape
    z, _ = encoder_forward(sd, dd, x.reshape(bt // t, t, c, h, w))
    z_e = _conv(sd, "quant_conv", z).permute(0, 2, 3, 1).contiguous()
    depth = cfg["code_shape"][-1]
    z_q, loss, codes = rq_forward(sd, z_e, depth, cfg.get("shared_codebook", False))
    z_q = _conv(sd, "post_quant_conv", z_q.permute(0, 3, 1, 2).contiguous())
    return decoder_forward(sd, dd, z_q, t), loss, codes


def rq_forward(sd, x, depth, shared=True):
    """RQBottleneck.forward in eval mode (reference: tdcrqvae3_arch.py:330-352; code/latent shapes coincide for
    PGTFormer): (x + (quant - x), mean_i mean((x - agg_i)^2), codes)."""
    resid = x.clone()
    agg = torch.zeros_like(x)
    codes, losses = [], []
    for i in range(depth):
        q, idx = _nearest(sd, resid, 0 if shared else i)
        resid = resid - q
        agg = agg + q
        losses.append((x - agg).pow(2.0).mean())
        codes.append(idx.unsqueeze(-1))
    return x + (agg - x), torch.mean(torch.stack(losses)), torch.cat(codes, -1)


def _distances(sd, flat, book_idx):
    """VQEmbedding.compute_distances (reference :100-117): addmm(|x|^2 + |e|^2, x, e^T, alpha=-2), padding row excluded."""
    cb_t = sd[f"quantizer.codebooks.{book_idx}.weight"][:-1].t()
    return torch.addmm(flat.pow(2.0).sum(1, keepdim=True) + cb_t.pow(2.0).sum(0, keepdim=True), flat, cb_t, alpha=-2.0)


def _nearest(sd, resid, book_idx):
    dist = _distances(sd, resid.reshape(-1, resid.shape[-1]), book_idx)
    idx = dist.argmin(-1).reshape(resid.shape[:-1])
    return F.embedding(idx, sd[f"quantizer.codebooks.{book_idx}.weight"]), idx


def rq_soft_codes(sd, x, depth, shared=True, temp=1.0):
    """RQBottleneck.get_soft_codes, stochastic=False (reference :429-457): (soft (B,h,w,d,K), codes (B,h,w,d))."""
    resid = x.clone()
    softs, codes = [], []
    for i in range(depth):
        bi = 0 if shared else i
        dist = _distances(sd, resid.reshape(-1, resid.shape[-1]), bi).reshape(*resid.shape[:-1], -1)
        soft = F.softmax(-dist / temp, dim=-1)
        code = dist.argmin(-1)
        resid = resid - F.embedding(code, sd[f"quantizer.codebooks.{bi}.weight"])
        codes.append(code.unsqueeze(-1))
        softs.append(soft.unsqueeze(-2))
    return torch.cat(softs, -2), torch.cat(codes, -1)


def vector_quantizer(weight, z, beta=0.25):
    """VectorQuantizer.forward in eval mode (reference: archs/vqgan_arch.py:42-84): z (B,C,H,W), codebook `weight`
    (K,C).  Distances are formed as |z|^2 + |e|^2 - 2 z.e^T (this association, not addmm), the nearest code by
    topk(k=1, largest=False).  Returns (z_q (B,C,H,W), loss, indices (B*H*W,1), mean distance)."""
    zp = z.permute(0, 2, 3, 1).contiguous()
    flat = zp.view(-1, weight.shape[1])
    d = (flat ** 2).sum(dim=1, keepdim=True) + (weight ** 2).sum(1) - 2 * torch.matmul(flat, weight.t())
    scores, idx = torch.topk(d, 1, dim=1, largest=False)
    onehot = torch.zeros(idx.shape[0], weight.shape[0]).to(zp)
    onehot.scatter_(1, idx, 1)
    z_q = torch.matmul(onehot, weight).view(zp.shape)
    loss = torch.mean((z_q - zp) ** 2) + beta * torch.mean((z_q - zp) ** 2)
    z_q = zp + (z_q - zp)
    return z_q.permute(0, 3, 1, 2).contiguous(), loss, idx, torch.mean(d)


# ----------------------------------------------------------------------------------------------
# Video-Swin window attention (reference: modules/swin.py) - BASELINE.json configs[4]
# ----------------------------------------------------------------------------------------------
def swin_window_partition(x, ws):
    """(B,D,H,W,C) -> (B*nW, wd*wh*ww, C) (reference: swin.py:38-49)."""
    b, d, h, w, c = x.shape
    x = x.view(b, d // ws[0], ws[0], h // ws[1], ws[1], w // ws[2], ws[2], c)
    return x.permute(0, 1, 3, 5, 2, 4, 6, 7).contiguous().view(-1, ws[0] * ws[1] * ws[2], c)


def swin_window_reverse(win, ws, b, d, h, w):
    """inverse of swin_window_partition (reference: swin.py:52-64)."""
    x = win.view(b, d // ws[0], h // ws[1], w // ws[2], ws[0], ws[1], ws[2], -1)
    return x.permute(0, 1, 4, 2, 5, 3, 6, 7).contiguous().view(b, d, h, w, -1)


def swin_compute_mask(d, h, w, ws, ss):
    """27-region shift mask (nW, N, N) in {0, -100} (reference: swin.py:311-323)."""
    img = torch.zeros((1, d, h, w, 1))
    cnt = 0
    for ds in (slice(-ws[0]), slice(-ws[0], -ss[0]), slice(-ss[0], None)):
        for hs in (slice(-ws[1]), slice(-ws[1], -ss[1]), slice(-ss[1], None)):
            for wsl in (slice(-ws[2]), slice(-ws[2], -ss[2]), slice(-ss[2], None)):
                img[:, ds, hs, wsl, :] = cnt
                cnt += 1
    mw = swin_window_partition(img, ws).squeeze(-1)
    am = mw.unsqueeze(1) - mw.unsqueeze(2)
    return am.masked_fill(am != 0, float(-100.0)).masked_fill(am == 0, float(0.0))


def swin_relative_position_index(ws):
    """(N, N) int64 index into the (2wd-1)(2wh-1)(2ww-1)-row bias table (reference: swin.py:103-116)."""
    coords = torch.stack(torch.meshgrid(torch.arange(ws[0]), torch.arange(ws[1]), torch.arange(ws[2]), indexing="ij"))
    cf = torch.flatten(coords, 1)
    rel = (cf[:, :, None] - cf[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += ws[0] - 1
    rel[:, :, 1] += ws[1] - 1
    rel[:, :, 2] += ws[2] - 1
    rel[:, :, 0] *= (2 * ws[1] - 1) * (2 * ws[2] - 1)
    rel[:, :, 1] *= (2 * ws[2] - 1)
    return rel.sum(-1)


def swin_window_attention(p, xw, heads, ws, mask=None):
    """WindowAttention3D.forward (reference: swin.py:128-167).  p: dict with qkv.weight [, qkv.bias], proj.weight,
    proj.bias, relative_position_bias_table; xw (B_, N, C).  ws: the window the module was CONSTRUCTED with - when the
    feature map clamps the window, the reference still slices that window's index table, index[:N, :N] (:150)."""
    b_, n, c = xw.shape
    qkv = F.linear(xw, p["qkv.weight"], p.get("qkv.bias")).reshape(b_, n, 3, heads, c // heads).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    q = q * (c // heads) ** -0.5
    attn = q @ k.transpose(-2, -1)
    idx = swin_relative_position_index(ws)
    bias = p["relative_position_bias_table"][idx[:n, :n].reshape(-1)].reshape(n, n, -1).permute(2, 0, 1).contiguous()
    attn = attn + bias.unsqueeze(0)
    if mask is not None:
        nw = mask.shape[0]
        attn = attn.view(b_ // nw, nw, heads, n, n) + mask.unsqueeze(1).unsqueeze(0)
        attn = attn.view(-1, heads, n, n)
    attn = attn.softmax(-1)
    x = (attn @ v).transpose(1, 2).reshape(b_, n, c)
    return F.linear(x, p["proj.weight"], p["proj.bias"])


def swin_block_part1(p, x, heads, ws, ss, ws_ctor=None):
    """SwinTransformerBlock3D.forward_part1 (reference: swin.py:212-246): norm1 -> zero-pad (D,H,W) up to multiples of the
    window -> cyclic shift -> partition -> attention (+ mask of the PADDED grid when shifted) -> reverse -> shift back -> crop.
    ws / ss: the window and shift in effect (already clamped to the map); ws_ctor: the constructor's window (bias index)."""
    b, d, h, w, c = x.shape
    x = F.layer_norm(x, (c,), p["norm1.weight"], p["norm1.bias"], 1e-5)
    pd, pb, pr = (ws[0] - d % ws[0]) % ws[0], (ws[1] - h % ws[1]) % ws[1], (ws[2] - w % ws[2]) % ws[2]
    x = F.pad(x, (0, 0, 0, pr, 0, pb, 0, pd))
    dp, hp, wp = d + pd, h + pb, w + pr
    shifted = any(i > 0 for i in ss)
    if shifted:
        x = torch.roll(x, shifts=(-ss[0], -ss[1], -ss[2]), dims=(1, 2, 3))
    mask = swin_compute_mask(dp, hp, wp, ws, ss) if shifted else None
    attn_p = {k[len("attn."):]: v for k, v in p.items() if k.startswith("attn.")}
    aw = swin_window_attention(attn_p, swin_window_partition(x, ws), heads, ws_ctor or ws, mask)
    x = swin_window_reverse(aw.view(-1, ws[0], ws[1], ws[2], c), ws, b, dp, hp, wp)
    if shifted:
        x = torch.roll(x, shifts=(ss[0], ss[1], ss[2]), dims=(1, 2, 3))
    return x[:, :d, :h, :w, :].contiguous()


def swin_block(p, x, heads, ws, ss, ws_ctor=None):
    """SwinTransformerBlock3D.forward (reference: swin.py:251-270; drop_path = 0): x + part1(x), then + mlp(norm2(.)),
    Mlp = fc1 -> exact GELU -> fc2 (:14-35)."""
    c = x.shape[-1]
    x = x + swin_block_part1(p, x, heads, ws, ss, ws_ctor)
    y = F.layer_norm(x, (c,), p["norm2.weight"], p["norm2.bias"], 1e-5)
    y = F.linear(F.gelu(F.linear(y, p["mlp.fc1.weight"], p["mlp.fc1.bias"])), p["mlp.fc2.weight"], p["mlp.fc2.bias"])
    return x + y


def swin_basic_layer(p, x, depth, heads, ws):
    """BasicLayer.forward without down-sampling (reference: swin.py:389-409): x (B,C,D,H,W); block i uses shift (0,0,0)
    for even i and window // 2 for odd i (:361-375), windows clamped to the feature map (get_window_size :67-82).
    p: state-dict-style keys `blocks.{i}.<block key>`."""
    b, c, d, h, w = x.shape
    x = x.permute(0, 2, 3, 4, 1).contiguous()
    half = tuple(i // 2 for i in ws)
    for i in range(depth):
        ss = (0, 0, 0) if i % 2 == 0 else half
        use_w = tuple(min(s, wv) for s, wv in zip((d, h, w), ws))
        use_s = tuple(0 if s <= wv else sv for s, wv, sv in zip((d, h, w), ws, ss))
        bp = {k[len(f"blocks.{i}."):]: v for k, v in p.items() if k.startswith(f"blocks.{i}.")}
        x = swin_block(bp, x, heads, use_w, use_s, ws)
    return x.permute(0, 4, 1, 2, 3).contiguous()


# ----------------------------------------------------------------------------------------------
# Training-side quantiser: EMA codebook update (reference: tdcrqvae3_arch.py:128-186)
# ----------------------------------------------------------------------------------------------
def vq_tile_with_noise(x, target_n, noise):
    """VQEmbedding._tile_with_noise (:129-136) with the uniform noise given (`torch.rand_like` in the reference)."""
    b, d = x.shape
    n_rep = (target_n + b - 1) // b
    std = x.new_ones(d) * 0.01 / np.sqrt(d)
    return x.repeat(n_rep, 1) + noise * std


def vq_ema_step(weight, cluster_size_ema, embed_ema, vectors, idxs, decay=0.99, eps=1e-5, restart=True, perm=None,
                noise=None, reduce=None):
    """One training step of VQEmbedding on given assignments: _update_buffers (:138-177) then _update_embedding (:179-186).
    weight (K+1, D), cluster_size_ema (K,), embed_ema (K, D) are NOT modified; returns the new (weight, cluster_size_ema,
    embed_ema).  perm / noise: the draws of torch.randperm / torch.rand_like the reference makes (fixtures carry them);
    reduce: optional callable standing for the two all-reduces (:157-158)."""
    k, d = weight.shape[0] - 1, weight.shape[1]
    vectors = vectors.reshape(-1, d)
    idxs = idxs.reshape(-1).long()
    n_vec = vectors.shape[0]
    one_hot = vectors.new_zeros(k, n_vec)
    one_hot.scatter_(0, idxs.unsqueeze(0), vectors.new_ones(1, n_vec))
    cluster_size = one_hot.sum(1)
    vec_sum = one_hot @ vectors
    if reduce is not None:
        vec_sum, cluster_size = reduce(vec_sum), reduce(cluster_size)
    cs = cluster_size_ema.clone().mul_(decay).add_(cluster_size, alpha=1 - decay)
    em = embed_ema.clone().mul_(decay).add_(vec_sum, alpha=1 - decay)
    if restart:
        if n_vec < k:
            vectors = vq_tile_with_noise(vectors, k, noise)
        rnd = vectors[perm][:k]
        usage = (cs.view(-1, 1) >= 1).float()
        em.mul_(usage).add_(rnd * (1 - usage))
        cs.mul_(usage.view(-1))
        cs.add_(torch.ones_like(cs) * (1 - usage).view(-1))
    n = cs.sum()
    norm = n * (cs + eps) / (n + k * eps)
    new_w = weight.clone()
    new_w[:-1, :] = em / norm.reshape(-1, 1)
    return new_w, cs, em


# ----------------------------------------------------------------------------------------------
# Driver semantics (reference: inference.py:6-19, 38-74)
# ----------------------------------------------------------------------------------------------
def window_triples(n_frames):
    """Input-frame index triple for every output frame: replicate-pad at both clip ends."""
    if n_frames <= 0:
        return []
    return [(max(i - 1, 0), i, min(i + 1, n_frames - 1)) for i in range(n_frames)]


def frame_to_u8(frame_chw):
    """clamp(0,1) * 255 then TRUNCATE to uint8 (np.array(float, np.uint8); reference:
    inference.py:16-18)."""
    return (frame_chw.clamp(0, 1).permute(1, 2, 0) * 255).to(torch.uint8)
