"""GPU parity tests of the fused token-row chains (pgtformer_amd/csrc/rowchain.hip: pgt_ln_linear, pgt_attn_proj_mlp,
pgt_fold_layernorm, pgt_sampled_rownorm_mean) against the torch-CPU emulation (tests/emu_ops.py) and against the layer-by-layer
launches they replace (reference: modules/rstt_layers.py:284-338, 126-132, 195-234).

Tolerances: half results 5e-3 * max(1, max|want|) against the emulation (as tests/test_gpu_ops.py); fused against
layer-by-layer on the GPU: the same bound (they differ by fp32 summation order and by where gamma is rounded); fp32 results 2e-4.
"""
import json
import os

import numpy as np
import pytest
import torch

from tests import emu_ops as E

pytestmark = pytest.mark.gpu
DEV = "cuda"
_LOG = []


@pytest.fixture(scope="module", autouse=True)
def _dump_log():
    yield
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/parity_rowchain.json", "w") as f:
        json.dump(_LOG, f, indent=1)


def O():
    import pgtformer_amd.ops as ops
    return ops


def rnd(shape, seed, dtype=torch.float32, scale=1.0):
    g = np.random.default_rng(seed)
    return torch.from_numpy((scale * g.standard_normal(shape)).astype(np.float32)).to(dtype)


def check(name, got, want, tol):
    got, want = got.detach().float().cpu(), want.detach().float().cpu()
    assert got.shape == want.shape, (name, got.shape, want.shape)
    err = (got - want).abs().max().item()
    ref = max(1.0, want.abs().max().item())
    _LOG.append({"name": name, "max_abs_err": err, "ref_absmax": ref, "tol": tol * ref, "ok": bool(err <= tol * ref)})
    assert np.isfinite(err) and err <= tol * ref, f"{name}: max err {err:.3e} > tol {tol * ref:.3e}"


H = torch.float16


def test_fold_layernorm_and_sampled_rownorm_mean():
    ops = O()
    w, gm, bt, b = rnd((768, 256), 1, scale=0.06), 1 + 0.1 * rnd((256,), 2), 0.1 * rnd((256,), 3), rnd((768,), 4)
    ww, wb = E.fold_layernorm(w, gm, bt, b)
    gw, gb = ops.fold_layernorm(w.to(DEV), gm.to(DEV), bt.to(DEV), b.to(DEV))
    check("fold_w", gw, ww, 1e-6)
    check("fold_b", gb, wb, 2e-5)
    gw, gb = ops.fold_layernorm(w.to(DEV), gm.to(DEV), bt.to(DEV), None)
    check("fold_b_nobias", gb, E.fold_layernorm(w, gm, bt)[1], 2e-5)
    for (n, hw, c, cw) in [(3, 4096, 256, 256), (2, 16384, 256, 320), (2, 200, 256, 256), (2, 1024, 512, 512)]:
        buf = rnd((n, hw, cw), 20 + n, H) * 1.5 + 0.3
        want = E.sampled_rownorm_mean(buf[..., :c])
        got = ops.sampled_rownorm_mean(buf.to(DEV)[..., :c])
        check(f"rownorm_mean_{n}x{hw}x{c}", got, want, 2e-4)


@pytest.mark.parametrize("case", ["rows128", "ragged", "cout256", "cout128_views", "frames", "big_r2", "big_r1w16"])
def test_ln_linear(case, monkeypatch):
    """LayerNorm statistics + normalisation in registers, then the GEMM against weights streaming through the LDS ring: every
    column block of a 768-wide projection, ragged row counts, strided input / output views, per-frame bias, the two-row-tile
    (256-row) form chosen from 65 536 rows and the 16-wave form."""
    ops = O()
    rows, cout, frames = 128, 768, 0
    if case == "ragged":
        rows = 1000 + 13
    elif case == "cout256":
        rows, cout = 384, 256
    elif case == "cout128_views":
        rows, cout = 640, 128
    elif case == "frames":
        rows, frames = 4096, 4
    elif case in ("big_r2", "big_r1w16"):
        rows, frames = 131072, 8
        if case == "big_r1w16":
            monkeypatch.setenv("PGT_RC_LN", "r1w16")
    wide = rnd((rows, 640), 31, H) * 1.3 + 0.2
    x = wide[:, 128:384] if case == "cout128_views" else wide[:, :256].contiguous()
    w = rnd((cout, 256), 32, H, 0.06)
    w[:, 0] += (torch.arange(cout, dtype=torch.float32) * 0.002).to(H)          # asymmetric in the output column
    w[0, :] += (torch.arange(256, dtype=torch.float32) * 0.001).to(H)           # ... and in k
    bias = rnd((frames, cout), 33) if frames else rnd((cout,), 33)
    want = E.ln_linear(x, w, bias)
    xg = wide.to(DEV)[:, 128:384] if case == "cout128_views" else x.to(DEV)
    out = None
    if case == "cout128_views":
        outw = torch.full((rows, 384), 7.0, dtype=H, device=DEV)
        out = outw[:, 128:256]
    got = ops.ln_linear(xg, w.to(DEV), bias.to(DEV), out=out)
    check(f"ln_linear_{case}", got, want, 5e-3)
    if out is not None:
        assert bool((outw[:, :128] == 7.0).all()) and bool((outw[:, 256:] == 7.0).all()), "wrote outside the output view"
    # against the two launches it replaces (layernorm without affine = gamma 1, beta 0)
    ln = ops.layernorm(xg.contiguous(), torch.ones(256, device=DEV), torch.zeros(256, device=DEV))
    ref = ops.linear(ln, w.to(DEV), bias.to(DEV))
    check(f"ln_linear_{case}_vs_unfused", got, ref, 2e-3)


@pytest.mark.parametrize("case", ["rows128", "ragged", "views_frames", "big"])
def test_attn_proj_mlp(case):
    """proj + shortcut -> LN -> fc1 -> GELU -> fc2 + residual with the rows held in registers across the three GEMMs."""
    ops = O()
    rows, frames = 128, 0
    if case == "ragged":
        rows = 2048 + 77
    elif case == "views_frames":
        rows, frames = 2048, 4
    elif case == "big":
        rows, frames = 98304, 6
    c = 256
    ao = rnd((rows, c), 41, H)
    scw = rnd((rows, 2 * c), 42, H)
    sc = scw[:, c:] if case == "views_frames" else scw[:, :c].contiguous()
    w3 = rnd((3 * c, c), 43, H, 0.06)
    w3[:, 0] += (torch.arange(3 * c, dtype=torch.float32) * 0.001).to(H)
    w3[0, :] += (torch.arange(c, dtype=torch.float32) * 0.001).to(H)
    bp = rnd((frames, c), 44) if frames else rnd((c,), 44)          # the three biases: vectors, or all three per frame
    b1, b2 = (rnd((frames, c), 45), rnd((frames, c), 46)) if frames else (rnd((c,), 45), rnd((c,), 46))
    want = E.attn_proj_mlp(ao, sc, w3, bp, b1, b2)
    scg = scw.to(DEV)[:, c:] if case == "views_frames" else sc.to(DEV)
    out = None
    if case == "views_frames":
        outw = torch.full((rows, 3 * c), 5.0, dtype=H, device=DEV)
        out = outw[:, c:2 * c]
    got = ops.attn_proj_mlp(ao.to(DEV), scg, w3.to(DEV), bp.to(DEV), b1.to(DEV), b2.to(DEV), out=out)
    check(f"attn_proj_mlp_{case}", got, want, 5e-3)
    if out is not None:
        assert bool((outw[:, :c] == 5.0).all()) and bool((outw[:, 2 * c:] == 5.0).all()), "wrote outside the output view"
    # against the five launches it replaces
    w3g = w3.to(DEV)
    x1 = ops.linear(ao.to(DEV), w3g[:c].contiguous(), bp.to(DEV), res=scg)
    ln = ops.layernorm(x1, torch.ones(c, device=DEV), torch.zeros(c, device=DEV))
    h = ops.linear(ln, w3g[c:2 * c].contiguous(), b1.to(DEV), act=E.ACT_GELU)
    ref = ops.linear(h, w3g[2 * c:].contiguous(), b2.to(DEV), res=x1)
    check(f"attn_proj_mlp_{case}_vs_unfused", got, ref, 2e-3)
    # bit-reproducible
    again = ops.attn_proj_mlp(ao.to(DEV), scg, w3g, bp.to(DEV), b1.to(DEV), b2.to(DEV))
    assert torch.equal(again, got if out is None else out)


@pytest.mark.parametrize("case", ["hw4096_frames", "hw16384", "hw200"])
def test_attn_proj_mlp_sample(case):
    """The sampled pass of the fused block tail: per-frame channel means of fc1's operand (normalised x1) and fc2's operand
    (GELU'd hidden row) over the library's pixel sample, against the emulation of the same chain on the same sample."""
    ops = O()
    frames, hw, per_frame = {"hw4096_frames": (5, 4096, True), "hw16384": (2, 16384, False), "hw200": (3, 200, False)}[case]
    c, rows = 256, frames * hw
    ao, sc = rnd((rows, c), 81, H), rnd((rows, c), 82, H)
    w3 = rnd((3 * c, c), 83, H, 0.06)
    bp = rnd((frames, c), 84) if per_frame else rnd((c,), 84)
    b1 = rnd((c,), 85)
    want_ln, want_hid = E.attn_proj_mlp_sample(ao, sc, w3, bp, b1, frames)
    got_ln, got_hid = ops.attn_proj_mlp_sample(ao.to(DEV), sc.to(DEV), w3.to(DEV), bp.to(DEV), b1.to(DEV), frames)
    check(f"sample_mean_ln_{case}", got_ln, want_ln, 1e-3)          # means of half-rounded values: a few half ulps / sqrt(rows)
    check(f"sample_mean_hid_{case}", got_hid, want_hid, 1e-3)
    again = ops.attn_proj_mlp_sample(ao.to(DEV), sc.to(DEV), w3.to(DEV), bp.to(DEV), b1.to(DEV), frames)
    assert torch.equal(again[0], got_ln) and torch.equal(again[1], got_hid)


def test_block_fused_equals_layer_by_layer():
    """A VSTSREncoderTransformerBlock pair (un-shifted + shifted windows) of the model: the fused path (two chain launches around
    the attention, compensated per-frame biases) against the layer-by-layer path on the same weights."""
    from pgtformer_amd.modules.rstt_layers import EncoderLayer
    torch.manual_seed(0)
    layer = EncoderLayer(256, 2, 8, 3, window_size=(4, 4), mlp_ratio=1.0)
    with torch.no_grad():
        for n_, p_ in layer.named_parameters():
            if p_.dim() == 2 and "relative_position" not in n_:
                p_.normal_(0, 0.06)
            elif "norm" in n_ and n_.endswith("weight"):
                p_.copy_(1 + 0.1 * torch.randn_like(p_))
            elif p_.dim() == 1:
                p_.normal_(0, 0.1)
    layer.prepare(DEV, torch.float16)
    assert all(b.fused for b in layer.blocks)
    x = (rnd((6, 32, 32, 256), 7, H) * 1.2).to(DEV)          # 2 windows of 3 frames, 32 x 32 (frames of 1024 rows: compensated)
    y_f = layer(x).float().cpu()
    for b in layer.blocks:
        b.fused = False
    y_u = layer(x).float().cpu()
    check("block_fused_vs_unfused", y_f, y_u, 4e-3)


# ---- split-half forms (encoder-side blocks): against the fp32 emulation on the same split inputs, merged outputs, 2e-5 relative
#      (tests/test_gpu_x3.py: 22 significand bits on two half planes, three MFMA products, fp32 accumulation)
def _x3w(w):
    return E.pack_conv_weight(w, "f16x3")


@pytest.mark.parametrize("case", ["rows128", "ragged", "cout256", "big_r2", "big_r1"])
def test_ln_linear_x3(case, monkeypatch):
    ops = O()
    rows, cout = 128, 768
    if case == "ragged":
        rows = 1000 + 13
    elif case == "cout256":
        rows, cout = 384, 256
    elif case in ("big_r2", "big_r1"):
        rows = 131072
        monkeypatch.setenv("PGT_RC_LNX3", "r2" if case == "big_r2" else "r1")
    x = E.to_x3(rnd((rows, 256), 61) * 1.3 + 0.2)
    w = rnd((cout, 256), 62, scale=0.06)
    w[:, 0] += torch.arange(cout, dtype=torch.float32) * 0.002
    w[0, :] += torch.arange(256, dtype=torch.float32) * 0.001
    wp = _x3w(w)
    bias = rnd((cout,), 63)
    want = E.from_x3(E.ln_linear(x, wp, bias, x3=True))
    got = ops.ln_linear(x.to(DEV), wp.to(DEV), bias.to(DEV), x3=True)
    check(f"ln_linear_x3_{case}", ops.from_x3(got), want, 2e-5)
    # against the two launches it replaces
    ln = ops.layernorm(x.to(DEV), torch.ones(256, device=DEV), torch.zeros(256, device=DEV), x3=True)
    ref = ops.linear(ln, wp.to(DEV), bias.to(DEV), x3=True)
    check(f"ln_linear_x3_{case}_vs_unfused", ops.from_x3(got), ops.from_x3(ref), 2e-5)


@pytest.mark.parametrize("case", ["rows128", "ragged", "big"])
def test_ln_mlp_x3(case):
    ops = O()
    rows = {"rows128": 128, "ragged": 2048 + 77, "big": 98304}[case]
    c = 256
    x = E.to_x3(rnd((rows, c), 71) * 1.2)
    w1, w2 = rnd((c, c), 72, scale=0.06), rnd((c, c), 73, scale=0.06)
    w1[:, 0] += torch.arange(c, dtype=torch.float32) * 0.002
    w2[0, :] += torch.arange(c, dtype=torch.float32) * 0.001
    wp = torch.cat([_x3w(w1), _x3w(w2)], 0).contiguous()
    b1, b2 = rnd((c,), 74), rnd((c,), 75)
    want = E.from_x3(E.ln_mlp(x, wp, b1, b2))
    got = ops.ln_mlp(x.to(DEV), wp.to(DEV), b1.to(DEV), b2.to(DEV))
    check(f"ln_mlp_x3_{case}", ops.from_x3(got), want, 2e-5)
    xg = x.to(DEV)
    ln = ops.layernorm(xg, torch.ones(c, device=DEV), torch.zeros(c, device=DEV), x3=True)
    h = ops.linear(ln, wp[:c].contiguous().to(DEV), b1.to(DEV), act=E.ACT_GELU, x3=True)
    ref = ops.linear(h, wp[c:].contiguous().to(DEV), b2.to(DEV), res=xg, x3=True)
    check(f"ln_mlp_x3_{case}_vs_unfused", ops.from_x3(got), ops.from_x3(ref), 2e-5)
    assert torch.equal(ops.ln_mlp(xg, wp.to(DEV), b1.to(DEV), b2.to(DEV)), got)


def test_block_fused_equals_layer_by_layer_x3():
    from pgtformer_amd.modules.rstt_layers import EncoderLayer
    from pgtformer_amd.ops import X3
    torch.manual_seed(1)
    layer = EncoderLayer(256, 2, 8, 3, window_size=(4, 4), mlp_ratio=1.0)
    with torch.no_grad():
        for n_, p_ in layer.named_parameters():
            if p_.dim() == 2 and "relative_position" not in n_:
                p_.normal_(0, 0.06)
            elif "norm" in n_ and n_.endswith("weight"):
                p_.copy_(1 + 0.1 * torch.randn_like(p_))
            elif p_.dim() == 1:
                p_.normal_(0, 0.1)
    layer.prepare(DEV, X3)
    assert all(b.fused for b in layer.blocks)
    x = O().to_x3((rnd((6, 32, 32, 256), 9) * 1.2).to(DEV))
    y_f = O().from_x3(layer(x)).cpu()
    for b in layer.blocks:
        b.fused = False
    y_u = O().from_x3(layer(x)).cpu()
    check("block_fused_vs_unfused_x3", y_f, y_u, 2e-5)
