"""GPU parity tests, operator level: every HIP kernel (called through the C-ABI via pgtformer_amd.ops)
against the torch-CPU operator emulation (tests/emu_ops.py) on the same seeded inputs, in f32 (exact
MFMA, tight tolerance), bf16 and IEEE half (16-bit MFMA, fp32 accumulate).  Each comparison is also logged to
gpurun_out/parity_ops.json.

Tolerances (written here, per the parity contract):
  f32 : max|got-want| <= 2e-4 * max(1, max|want|)     (fp32 accumulation-order differences)
  bf16: max|got-want| <= 4e-2 * max(1, max|want|)     (inputs/outputs rounded to bf16, 8-bit mantissa)
  f16 : max|got-want| <= 5e-3 * max(1, max|want|)     (inputs/outputs rounded to IEEE half, 11-bit mantissa)
  integer results (codes, u8 frames): bit-exact (u8: +-1 allowed only where noted)
"""
import json
import os

import numpy as np
import pytest
import torch

from tests import emu_ops as E

pytestmark = pytest.mark.gpu

DEV = "cuda"
TOL = {torch.float32: 2e-4, torch.bfloat16: 4e-2, torch.float16: 5e-3}
DTYPES = [torch.float32, torch.bfloat16, torch.float16]
H16 = [torch.bfloat16, torch.float16]       # the two 16-bit MFMA operand types
_LOG = []


@pytest.fixture(scope="module", autouse=True)
def _dump_log():
    yield
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/parity_ops.json", "w") as f:
        json.dump(_LOG, f, indent=1)


def ops():
    import pgtformer_amd.ops as O
    return O


def rnd(shape, seed, dtype=torch.float32, scale=1.0):
    g = np.random.default_rng(seed)
    t = torch.from_numpy((scale * g.standard_normal(shape)).astype(np.float32))
    return t.to(dtype)


def check(name, got, want, dtype, tol_scale=1.0):
    got = got.detach().float().cpu()
    want = want.detach().float().cpu()
    assert got.shape == want.shape, (name, got.shape, want.shape)
    err = (got - want).abs().max().item()
    ref = max(1.0, want.abs().max().item())
    tol = TOL[dtype] * tol_scale * ref
    _LOG.append({"name": name, "dtype": str(dtype), "max_abs_err": err, "ref_absmax": ref, "tol": tol,
                 "ok": bool(err <= tol)})
    assert np.isfinite(err) and err <= tol, f"{name}: max err {err:.3e} > tol {tol:.3e}"


def g(t):
    return None if t is None else t.to(DEV)


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("tile", [(0, 0), (64, 64), (64, 128), (128, 64), (128, 128)])
def test_linear_asymmetric_all_tiles(dtype, tile):
    """GEMM with ragged M, N, K and an asymmetric weight: catches operand/accumulator layout swaps."""
    m, k, n = 200, 72, 150
    x = rnd((m, k), 1, dtype)
    w = rnd((n, k), 2, dtype, 0.2)
    w[:, 0] += torch.arange(n, dtype=torch.float32).to(dtype) * 0.01   # asymmetric in n
    b = rnd((n,), 3)
    res = rnd((m, n), 4, dtype)
    want = E.linear(x, w, b, act=E.ACT_GELU, res=res)
    x4 = g(x).reshape(1, 1, m, k)
    got = ops().conv2d(x4, g(w), g(b), act=E.ACT_GELU, res=g(res).reshape(1, 1, m, n), tile=tile)
    check(f"linear_tile{tile}", got.reshape(m, n), want, dtype)


CONV_CASES = [
    # name, N,H,W,Cin,Cout,k,stride,pad4,ups
    ("c3x3_s1", 2, 13, 11, 32, 48, 3, 1, (1, 1, 1, 1), False),
    ("c3x3_s2_asym", 3, 16, 16, 64, 64, 3, 2, (0, 1, 0, 1), False),
    ("c3x3_ups", 2, 6, 5, 128, 128, 3, 1, (1, 1, 1, 1), True),
    ("c7x7_s2_cin8", 1, 32, 32, 8, 64, 7, 2, (3, 3, 3, 3), False),
    ("c1x1_s2", 2, 12, 12, 64, 128, 1, 2, (0, 0, 0, 0), False),
    ("c3x3_288_128", 1, 8, 8, 288, 128, 3, 1, (1, 1, 1, 1), False),
    ("c3x3_cout3", 3, 20, 20, 64, 3, 3, 1, (1, 1, 1, 1), False),
    ("c3x3_s2_p1", 2, 16, 16, 64, 128, 3, 2, (1, 1, 1, 1), False),
    ("c3x3_big_m", 3, 64, 64, 64, 64, 3, 1, (1, 1, 1, 1), False),
    ("c3x3_cin8", 3, 24, 24, 8, 64, 3, 1, (1, 1, 1, 1), False),
]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv2d(dtype, case):
    name, n, h, w_, cin, cout, k, stride, pad4, ups = case
    x = rnd((n, h, w_, cin), 10, dtype)
    wt = rnd((cout, k * k * cin), 11, dtype, 1.0 / np.sqrt(k * k * cin))
    b = rnd((cout,), 12, torch.float32, 0.1)
    kw = dict(kh=k, kw=k, stride=stride, pad=pad4, ups=ups)
    want = E.conv2d(x, wt, b, act=E.ACT_SILU, **kw)
    got = ops().conv2d(g(x), g(wt), g(b), act=E.ACT_SILU, **kw)
    check(name, got, want, dtype)


@pytest.mark.parametrize("case", [("in3x3", 3, 24, 40, 3, 64, 3, 1, 1), ("in7x7_s2", 2, 32, 32, 3, 64, 7, 2, 3), ("k108", 2, 9, 10, 12, 72, 3, 1, 1),
                                  ("k20", 1, 1, 300, 20, 40, 1, 1, 0), ("k52", 2, 6, 6, 52, 24, 1, 1, 0)], ids=lambda c: c[0])
def test_conv2d_fp32_k_tail(case):
    """fp32 layers whose K is not a multiple of the 32-wide K tile: the steps of the last tile that lie past K are skipped (1, 2 or
    3 of 4 run).  The two input layers in their shipping form - 3 channels padded to ONE 16-byte chunk (4 channels in fp32: K = 36
    and 196) - against the emulation AND against the same conv on 8-channel padding (what bf16 / half inputs use): same products,
    paired differently inside the fp32 MFMA, so equal to rounding."""
    name, n, h, w_, cin, cout, k, stride, pad = case
    x3 = rnd((n, h, w_, cin), 20, torch.float32)
    w3 = rnd((cout, k * k, cin), 21, torch.float32, 1.0 / np.sqrt(k * k * cin))
    b = rnd((cout,), 22, torch.float32, 0.1)
    kw = dict(kh=k, kw=k, stride=stride, pad=(pad,) * 4)

    def padded(cp):
        xx = torch.zeros((n, h, w_, cp))
        xx[..., :cin] = x3
        ww = torch.zeros((cout, k * k, cp))
        ww[..., :cin] = w3
        return xx, ww.reshape(cout, -1).contiguous()
    cp = (cin + 3) // 4 * 4
    x4, w4 = padded(cp)
    want = E.conv2d(x4, w4, b, **kw)
    got = ops().conv2d(g(x4), g(w4), g(b), **kw)
    check(name, got, want, torch.float32)
    x8, w8 = padded((cin + 7) // 8 * 8)
    got8 = ops().conv2d(g(x8), g(w8), g(b), **kw)
    assert (got - got8).abs().max().item() <= 2e-6 * max(1.0, want.abs().max().item()), name


V2_CASES = [
    ("v2_c3x3", 2, 20, 24, 64, 128, 3, 1, (1, 1, 1, 1), False),
    ("v2_c3x3_s2_asym", 3, 16, 16, 128, 64, 3, 2, (0, 1, 0, 1), False),
    ("v2_c3x3_ups", 2, 9, 7, 192, 200, 3, 1, (1, 1, 1, 1), True),
    ("v2_lin_ragged_m", 1, 1, 333, 256, 72, 1, 1, (0, 0, 0, 0), False),
    ("v2_c1x1_s2", 2, 12, 12, 64, 136, 1, 2, (0, 0, 0, 0), False),
]


@pytest.mark.parametrize("bn", [64, 128])
@pytest.mark.parametrize("case", V2_CASES, ids=[c[0] for c in V2_CASES])
def test_conv2d_lds_dma_kernel(case, bn):
    """igemm2 (global_load_lds + XOR-swizzled LDS + double buffering) against the emulation, incl. all epilogues."""
    name, n, h, w_, cin, cout, k, stride, pad4, ups = case
    dtype = torch.bfloat16
    x = rnd((n, h, w_, cin), 110, dtype)
    wt = rnd((cout, k * k * cin), 111, dtype, 1.0 / np.sqrt(k * k * cin))
    wt[:, 0] += (torch.arange(cout, dtype=torch.float32) * 0.01).to(dtype)
    b = rnd((cout,), 112, torch.float32, 0.1)
    kw = dict(kh=k, kw=k, stride=stride, pad=pad4, ups=ups)
    want = E.conv2d(x, wt, b, act=E.ACT_SILU, **kw)
    res = rnd(tuple(want.shape), 113, dtype)
    got = ops().conv2d(g(x), g(wt), g(b), act=E.ACT_SILU, kernel=2, tile=(2, bn), **kw)
    check(f"{name}_bn{bn}", got, want, dtype)
    got = ops().conv2d(g(x), g(wt), g(b), res=g(res), post_relu=True, kernel=2, tile=(3, bn), **kw)
    check(f"{name}_bn{bn}_res", got, E.conv2d(x, wt, b, res=res, post_relu=True, **kw), dtype)
    v1 = ops().conv2d(g(x), g(wt), g(b), res=g(res), post_relu=True, kernel=1, **kw)
    check(f"{name}_bn{bn}_v1_vs_v2", got, v1, dtype, 0.2)
    dec, shf = rnd(tuple(want.shape), 114, dtype), rnd(tuple(want.shape), 115, dtype)
    got = ops().conv2d(g(x), g(wt), g(b), sft=(g(dec), g(shf), 0.6), out_f32=False, kernel=2, tile=(4, bn), **kw)
    check(f"{name}_bn{bn}_sft", got, E.conv2d(x, wt, b, sft=(dec, shf, 0.6), **kw), dtype)


@pytest.mark.parametrize("dtype", DTYPES)
def test_conv2d_split_k(dtype):
    """Deep-K / small-M layers run as K slices + a fixed-order reduction: same result as the single pass,
    deterministic, with every epilogue."""
    n, h, w_, cin, cout = 3, 16, 16, 256, 72
    x = rnd((n, h, w_, cin), 120, dtype)
    wt = rnd((cout, 9 * cin), 121, dtype, 1.0 / np.sqrt(9 * cin))
    b = rnd((cout,), 122, torch.float32, 0.1)
    res = rnd((n, h, w_, cout), 123, dtype)
    kw = dict(kh=3, kw=3, pad=(1, 1, 1, 1))
    want = E.conv2d(x, wt, b, act=E.ACT_SILU, res=res, **kw)
    one = ops().conv2d(g(x), g(wt), g(b), act=E.ACT_SILU, res=g(res), splitk=1, **kw)
    check("splitk_ref", one, want, dtype)
    for s in (2, 3, 5, 0):   # 0 = library heuristic (this shape splits)
        got = ops().conv2d(g(x), g(wt), g(b), act=E.ACT_SILU, res=g(res), splitk=s, **kw)
        check(f"splitk{s}", got, want, dtype)
        again = ops().conv2d(g(x), g(wt), g(b), act=E.ACT_SILU, res=g(res), splitk=s, **kw)
        assert torch.equal(got, again)
    dec, shf = rnd((n, h, w_, cout), 124, dtype), rnd((n, h, w_, cout), 125, dtype)
    check("splitk_sft", ops().conv2d(g(x), g(wt), g(b), sft=(g(dec), g(shf), 0.5), splitk=4, **kw),
          E.conv2d(x, wt, b, sft=(dec, shf, 0.5), **kw), dtype)
    check("splitk_f32out", ops().conv2d(g(x), g(wt), None, out_f32=True, splitk=4, **kw),
          E.conv2d(x, wt, None, out_f32=True, **kw), dtype)


@pytest.mark.parametrize("dtype", DTYPES)
def test_conv_epilogues_and_views(dtype):
    n, h, w_, c = 2, 10, 9, 64
    x = rnd((n, h, w_, c), 20, dtype)
    wt = rnd((c, 9 * c), 21, dtype, 0.04)
    b = rnd((c,), 22)
    res = rnd((n, h, w_, c), 23, dtype)
    kw = dict(kh=3, kw=3, pad=(1, 1, 1, 1))
    check("res_postrelu", ops().conv2d(g(x), g(wt), g(b), res=g(res), post_relu=True, **kw),
          E.conv2d(x, wt, b, res=res, post_relu=True, **kw), dtype)
    dec, shf = rnd((n, h, w_, c), 24, dtype), rnd((n, h, w_, c), 25, dtype)
    check("sft", ops().conv2d(g(x), g(wt), g(b), sft=(g(dec), g(shf), 0.7), **kw),
          E.conv2d(x, wt, b, sft=(dec, shf, 0.7), **kw), dtype)
    check("out_f32", ops().conv2d(g(x), g(wt), g(b), out_f32=True, **kw), E.conv2d(x, wt, b, out_f32=True, **kw), dtype)
    # the 16-byte (LDS-staged) epilogue and the element-wise one are the same arithmetic: bit-identical
    for extra in (dict(res=g(res), post_relu=True), dict(sft=(g(dec), g(shf), 0.7)), dict(act=E.ACT_GELU)):
        for tile in ((64, 64), (128, 128), (128, 64), (64, 128)):
            a = ops().conv2d(g(x), g(wt), g(b), tile=tile, **extra, **kw)
            bb = ops().conv2d(g(x), g(wt), g(b), tile=tile, scalar_epi=True, **extra, **kw)
            assert torch.equal(a, bb), (tile, list(extra))
    # channel-sliced input view and channel-sliced output view (concat buffers)
    wide = rnd((n, h, w_, 2 * c), 26, dtype)
    wide_d = g(wide)
    outbuf = torch.zeros((n, h, w_, 3 * c), device=DEV, dtype=dtype)
    ops().conv2d(wide_d[..., c:], g(wt), g(b), act=E.ACT_LEAKY02, out=outbuf[..., c:2 * c], **kw)
    want = E.conv2d(wide[..., c:], wt, b, act=E.ACT_LEAKY02, **kw)
    check("sliced_views", outbuf[..., c:2 * c], want, dtype)
    assert outbuf[..., :c].abs().max().item() == 0 and outbuf[..., 2 * c:].abs().max().item() == 0
    # 19-channel output slice at an odd offset (BiSeNet heads)
    w19 = rnd((19, c), 27, dtype, 0.1)
    cond = torch.zeros((n, h, w_, 64), device=DEV, dtype=dtype)
    ops().conv2d(g(x), g(w19), None, out=cond[..., 19:38])
    check("cout19_slice", cond[..., 19:38], E.conv2d(x, w19, None), dtype)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(3, 24, 24, 64), (2, 9, 7, 288), (1, 6, 6, 1056), (3, 16, 16, 512), (3, 5, 5, 544)])
def test_groupnorm_silu(dtype, shape):
    x = rnd(shape, 30, dtype) * 1.5 + 0.4
    x = x.to(dtype)
    gam, bet = 1 + 0.1 * rnd((shape[3],), 31), 0.1 * rnd((shape[3],), 32)
    want = E.groupnorm_act(x, gam, bet)
    got = ops().groupnorm_act(g(x), g(gam), g(bet))
    check(f"gn_silu{shape}", got, want, dtype)
    s_w, b_w = E.groupnorm_affine(x, gam, bet)
    s_g, b_g = ops().groupnorm_affine(g(x), g(gam), g(bet))
    check(f"gn_scale{shape}", s_g, s_w, torch.float32, 5.0)
    check(f"gn_shift{shape}", b_g, b_w, torch.float32, 5.0)


def test_groupnorm_is_deterministic():
    x = g(rnd((3, 64, 64, 128), 33))
    gam, bet = g(torch.ones(128)), g(torch.zeros(128))
    a = ops().groupnorm_affine(x, gam, bet)
    b = ops().groupnorm_affine(x, gam, bet)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("c", [256, 512])
def test_layernorm(dtype, c):
    x = rnd((77, c), 40, dtype) * 2 + 0.3
    x = x.to(dtype)
    pos = rnd((77, c), 41, dtype)
    gam, bet = 1 + 0.1 * rnd((c,), 42), 0.1 * rnd((c,), 43)
    y_w, y2_w = E.layernorm(x, gam, bet, 1e-5, pos)
    y_g, y2_g = ops().layernorm(g(x), g(gam), g(bet), 1e-5, g(pos))
    check(f"ln{c}", y_g, y_w, dtype)
    check(f"ln_pos{c}", y2_g, y2_w, dtype)
    check(f"ln_nopos{c}", ops().layernorm(g(x), g(gam), g(bet)), E.layernorm(x, gam, bet), dtype)


@pytest.mark.parametrize("dtype", DTYPES)
def test_adain_and_stats(dtype):
    c_ = (rnd((3, 8, 8, 512), 50, dtype) * 0.3).to(dtype)
    s_ = (rnd((3, 8, 8, 512), 51, torch.float32) * 2.0 + 0.3)
    mc, vc = ops().channel_stats(g(c_))
    mw, vw = E.channel_stats(c_)
    check("stats_mean", mc, mw, torch.float32, 5.0)
    check("stats_var", vc, vw, torch.float32, 5.0)
    from pgtformer_amd.archs.codeformer_arch import adaptive_instance_normalization as adain
    got = adain(g(c_), g(s_))
    mm, vv = E.channel_stats(s_)
    sc, sh = E.adain_affine(mw, vw, mm, vv)
    check("adain", got, E.affine_act(c_, sc, sh), dtype)


WA_CASES = [(1, 3, 8, 12, 256, (4, 4), (0, 0)), (1, 3, 8, 12, 256, (4, 4), (2, 2)), (2, 3, 8, 8, 512, (4, 4), (2, 2)),
            (1, 3, 16, 16, 256, (4, 4), (2, 2)), (1, 3, 8, 8, 512, (4, 4), (0, 0))]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", WA_CASES)
def test_window_attention(dtype, case):
    b, t, h, w_, c, win, shift = case
    heads = 8
    n = t * win[0] * win[1]
    qkv = rnd((b * t * h * w_, 3 * c), 60, dtype)
    bias = rnd((heads, n, n), 61, torch.float32, 0.5)
    want = E.window_attention(qkv, bias, b, t, h, w_, c, heads, win, shift)
    got = ops().window_attention(g(qkv), g(bias), b, t, h, w_, c, heads, win, shift)
    check(f"winattn{case}", got, want, dtype)


@pytest.mark.parametrize("dtype", DTYPES)
def test_copy_into_channel_slices_vector_and_element_paths(dtype):
    """pgt_copy2d: the 16-byte-chunk kernel (same type, widths / pitches / addresses multiples of 16 bytes) and the element
    kernel (everything else, incl. conversions) write exactly the source values, into channel slices of wider buffers, and
    nothing outside them."""
    O = ops()
    for rows, c, cw, off in ((3 * 16 * 16, 64, 160, 0), (2 * 8 * 8, 64, 160, 64), (5 * 4 * 4, 24, 56, 8), (77, 20, 36, 3), (4, 8, 8, 0)):
        src = rnd((rows, c + 8), 70 + c, dtype)[:, :c] if off != 3 else rnd((rows, c), 70 + c, dtype)
        dst = torch.full((rows, cw), -7.0, dtype=dtype)
        want = dst.clone()
        want[:, off:off + c] = src
        d = g(dst)
        O.copy_into(g(src) if off == 3 else g(rnd((rows, c + 8), 70 + c, dtype))[:, :c], d[:, off:off + c])
        assert torch.equal(d.cpu(), want), (rows, c, cw, off, dtype)
    x = rnd((2, 8, 8, 32), 79, dtype)
    for to in DTYPES:      # conversions stay on the element kernel
        got = O.cast(g(x), to)
        assert got.dtype == to and torch.equal(got.cpu(), x.to(to))


def test_window_attention_heads_per_workgroup_forms_store_the_same_bits():
    """The model's window attention (48 tokens, 8 heads x 32) runs the 8 heads of a window in one workgroup (window_attn_mfma.hip:
    HPW); the 1- / 2- / 4-head forms stay selectable (PGT_WATTN_HPW, read once per process: separate interpreters).  Every head's
    arithmetic is the same instruction sequence in all of them - the outputs are bit-equal, on power-of-two and other window
    grids (shift / mask index arithmetic by shifts or by division), shifted and not, half and split rows."""
    import subprocess
    import sys
    import tempfile
    prog = (
        "import sys, torch\n"
        "sys.path.insert(0, %r)\n"
        "import pgtformer_amd.ops as O\n"
        "g = torch.Generator(device='cuda').manual_seed(3)\n"
        "outs = []\n"
        "for (b, h, w, shift, x3) in ((2, 16, 16, (2, 2), False), (1, 8, 12, (0, 0), False), (1, 8, 12, (2, 2), True), (2, 16, 8, (0, 0), True),\n"
        "                             (5, 32, 32, (2, 2), False), (3, 32, 64, (2, 2), True), (1, 4, 4, (0, 0), False)):\n"
        "    qkv = torch.randn((b * 3 * h * w, 768 * (2 if x3 else 1)), device='cuda', dtype=torch.float16, generator=g)\n"
        "    if x3: qkv[:, 768:] *= 2.0 ** -11\n"
        "    bias = 0.5 * torch.randn((8, 48, 48), device='cuda', generator=g)\n"
        "    outs.append(O.window_attention(qkv, bias, b, 3, h, w, 256, 8, (4, 4), shift, x3=x3).cpu())\n"
        "torch.save(outs, sys.argv[1])\n") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    with tempfile.TemporaryDirectory() as d:
        for hpw in ("8", "1", "2", "4"):
            f = os.path.join(d, f"o{hpw}.pt")
            subprocess.run([sys.executable, "-c", prog, f], check=True, env=dict(os.environ, PGT_WATTN_HPW=hpw), timeout=300)
            res[hpw] = torch.load(f)
    for hpw in ("8", "2", "4"):
        for i, (a, b_) in enumerate(zip(res["1"], res[hpw])):
            assert torch.isfinite(a.float()).all()
            assert torch.equal(a, b_), (hpw, i)
    _LOG.append({"name": "window_attention_heads_per_workgroup_1_2_4_8", "bit_equal": True, "cases": len(res["1"])})


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("L", [192, 200, 640, 777])
def test_mha(dtype, L):
    b, heads, hd = 2, 8, 64
    e = heads * hd
    qk = rnd((b * L, 2 * e), 70, dtype)
    v = rnd((b * L, e), 71, dtype)
    qk[5, :64] *= 6.0   # a spiky query row: exercises the online-softmax rescale
    want = E.mha(qk[:, :e], qk[:, e:], v, b, L, heads, hd, 0.125)
    qk_d = g(qk)
    got = ops().mha(qk_d[:, :e], qk_d[:, e:], g(v), b, L, heads, hd, 0.125)
    check(f"mha{L}", got, want, dtype)


def test_argmax_argmin_ties_and_embed():
    logits = rnd((70, 1024), 80)
    logits[3, 100] = logits[3, 900] = 50.0       # tie -> lowest index (torch rule)
    logits[4, 1023] = 60.0
    logits[5, 0] = 60.0
    got = ops().argmax_rows(g(logits)).cpu()
    assert torch.equal(got, logits.argmax(-1).to(torch.int32)) and got[3] == 100
    book = rnd((1025, 512), 81)
    book[77] = book[5]                              # duplicated code vector -> argmin must return 5
    book[1024] = 0
    x = rnd((64, 512), 82, torch.float32, 0.3)
    x[9] = book[77] + 1e-3
    for dtype in DTYPES:
        xd = g(x.to(dtype))
        bt = g(book[:-1].to(dtype))
        dot = ops().linear(xd, bt, None, out_f32=True)
        codes = ops().rq_argmin(dot, ops().row_sumsq(xd), g(book[:-1].pow(2).sum(1))).cpu()
        want = E.rq_argmin(E.linear(x.to(dtype), book[:-1].to(dtype), None, out_f32=True),
                           E.row_sumsq(x.to(dtype)), book[:-1].pow(2).sum(1))
        agree = (codes == want).float().mean().item()
        _LOG.append({"name": f"rq_argmin_{dtype}", "agree": agree})
        assert codes[9] == 5
        assert agree >= (1.0 if dtype == torch.float32 else 0.95)
        out = ops().embed_rows(g(book), g(want), dtype)
        check(f"embed_rows_{dtype}", out, book[want.long()], dtype)
        resid = xd.clone()
        ops().embed_rows(g(book), g(want), dtype, out=out, accumulate=True, resid=resid)
        check(f"embed_acc_{dtype}", out, 2 * book[want.long()].to(dtype).float(), dtype)
        check(f"embed_resid_{dtype}", resid, x.to(dtype).float() - book[want.long()], dtype)


@pytest.mark.parametrize("dtype", DTYPES)
def test_glue_ops(dtype):
    x = rnd((2, 17, 16, 64), 90, dtype)
    check("maxpool", ops().maxpool3x3s2(g(x)), E.maxpool3x3s2(x), dtype)
    gate, av = rnd((2, 64), 91, dtype), rnd((2, 64), 92, dtype)
    at = rnd((2, 17, 16, 64), 93, dtype)
    check("gate_add", ops().gate_add(g(x), gate=g(gate), addvec=g(av), addt=g(at)),
          E.gate_add(x, gate=gate, addvec=av, addt=at), dtype)
    y = rnd((2, 64, 64, 19), 94, dtype)
    buf = torch.zeros((2, 32, 32, 64), device=DEV, dtype=dtype)
    ops().resize_bilinear_ac(g(y), 32, 32, out=buf[..., 19:38])
    check("bilinear", buf[..., 19:38], E.resize_bilinear_ac(y, 32, 32), dtype)
    check("cast", ops().cast(g(x), torch.float32), x.float(), dtype)
    check("nchw", ops().nhwc_to_nchw_f32(g(x)), E.nhwc_to_nchw_f32(x), dtype)


def test_driver_edges():
    gen = np.random.default_rng(5)
    u8 = torch.from_numpy(gen.integers(0, 256, (3, 20, 24, 3), dtype=np.uint8))
    for dtype in DTYPES:
        raw, norm = ops().prep_input(g(u8), dtype)
        rw, nw = E.prep_input(u8, dtype)
        check("prep_raw_u8", raw, rw, dtype, 0.1 if dtype == torch.float32 else 1.0)
        check("prep_norm_u8", norm, nw, dtype, 0.1 if dtype == torch.float32 else 1.0)
        xf = (u8.float() / 255).permute(0, 3, 1, 2).contiguous()
        raw2, _ = ops().prep_input(g(xf), dtype)
        assert torch.equal(raw2, raw)
    fr = torch.tensor([-0.2, 0.0, 0.5, 0.999, 1.0, 1.7, 0.25, 0.75]).repeat(6).reshape(4, 4, 3)
    got = ops().frame_to_u8(g(fr)).cpu()
    assert torch.equal(got, E.frame_to_u8(fr))


def test_errors_are_reported_not_fatal():
    from pgtformer_amd import hip
    x = g(rnd((1, 4, 4, 12), 99))      # Cin=12 f32 is fine (multiple of 4); bf16 Cin=12 is not (multiple of 8)
    w = g(rnd((8, 12), 98))
    ops().conv2d(x, w)
    with pytest.raises(hip.PgtError, match="Cin"):
        ops().conv2d(x.to(torch.bfloat16), w.to(torch.bfloat16))
    with pytest.raises(hip.PgtError):
        ops().conv2d(torch.zeros(1, 4, 4, 12), torch.zeros(8, 12))   # CPU tensors: no fallback


SWIN_CASES = [(1, 3, 16, 16, 512, (8, 8), (0, 0)), (1, 3, 16, 24, 512, (8, 8), (4, 4)), (2, 3, 8, 16, 256, (8, 8), (4, 4)),
              (1, 6, 8, 8, 256, (4, 4), (2, 2))]


@pytest.mark.parametrize("case", SWIN_CASES)
def test_window_attention_large_windows_bf16(case):
    """BASELINE config 5 point: 3x8x8 windows (N = 192 tokens, C = 512) and other N % 48 == 0 shapes run on the
    MFMA kernel (modules/swin.py parametrisation of the same attention)."""
    b, t, h, w_, c, win, shift = case
    heads, dtype = 8, torch.bfloat16
    n = t * win[0] * win[1]
    qkv = rnd((b * t * h * w_, 3 * c), 160, dtype)
    bias = rnd((heads, n, n), 161, torch.float32, 0.5)
    want = E.window_attention(qkv, bias, b, t, h, w_, c, heads, win, shift)
    got = ops().window_attention(g(qkv), g(bias), b, t, h, w_, c, heads, win, shift)
    check(f"winattn_large{case}", got, want, dtype)


def test_window_attention_f32_rejects_large_windows():
    from pgtformer_amd import hip
    qkv = g(rnd((3 * 8 * 8, 3 * 256), 170))
    bias = g(rnd((8, 192, 192), 171))
    with pytest.raises(hip.PgtError, match="tokens per window"):
        ops().window_attention(qkv, bias, 1, 3, 8, 8, 256, 8, (8, 8), (0, 0))


V3_SHAPES = [("v3_c3x3", 2, 20, 24, 64, 128, 3), ("v3_lin_ragged", 1, 1, 333, 256, 72, 1), ("v3_c3x3_wide", 3, 16, 16, 128, 264, 3)]

V4_SHAPES = [(n, a, b, c, d, e, f, 1, False) for (n, a, b, c, d, e, f) in V3_SHAPES] + [
    ("v4_1tile_k64", 1, 8, 8, 64, 64, 1, 1, False), ("v4_odd_ktiles", 1, 24, 40, 192, 320, 3, 1, False),
    ("v4_big", 4, 64, 64, 256, 256, 3, 1, False), ("v4_ups", 2, 20, 28, 128, 128, 3, 1, True),
    ("v4_ups_odd", 1, 9, 13, 64, 72, 3, 1, True), ("v4_stride2", 2, 32, 40, 128, 128, 3, 2, False)]


@pytest.mark.parametrize("dtype", H16, ids=["bf16", "f16"])
@pytest.mark.parametrize("bn", [256, 128])
@pytest.mark.parametrize("shape", V4_SHAPES, ids=[s[0] for s in V4_SHAPES])
def test_conv2d_phased_kernel(shape, bn, dtype):
    """igemm4 (256x256 / 512x128 tiles, 8 waves, phase-interleaved LDS-DMA schedule, strided and up-sampled inputs)
    against the emulation and v1, plus a repeat-run screen: the hand-placed vmcnt/barrier schedule must give
    bit-identical results on every launch."""
    name, n, h, w_, cin, cout, k, stride, ups = shape
    x = rnd((n, h, w_, cin), 310, dtype)
    wt = rnd((cout, k * k * cin), 311, dtype, 1.0 / np.sqrt(k * k * cin))
    wt[:, 0] += (torch.arange(cout, dtype=torch.float32) * 0.01).to(dtype)
    b = rnd((cout,), 312, torch.float32, 0.1)
    pad = (0, 1, 0, 1) if stride == 2 else (k // 2,) * 4   # Downsample pads bottom/right only (rstt_layers.py:891)
    kw = dict(kh=k, kw=k, pad=pad, stride=stride, ups=ups)
    v4 = dict(kernel=4, tile=(0, bn))
    gx, gw, gb = g(x), g(wt), g(b)
    want = E.conv2d(x, wt, b, act=E.ACT_SILU, **kw)
    check(f"{name}_v4", ops().conv2d(gx, gw, gb, act=E.ACT_SILU, **v4, **kw), want, dtype)
    res = rnd(tuple(want.shape), 313, dtype)
    gres = g(res)
    got = ops().conv2d(gx, gw, gb, res=gres, post_relu=True, **v4, **kw)
    check(f"{name}_v4_res", got, E.conv2d(x, wt, b, res=res, post_relu=True, **kw), dtype)
    v1 = ops().conv2d(gx, gw, gb, res=gres, post_relu=True, kernel=1, **kw)
    check(f"{name}_v4_v1", got, v1, dtype, 0.2)
    for _ in range(20):
        again = ops().conv2d(gx, gw, gb, res=gres, post_relu=True, **v4, **kw)
        assert torch.equal(again, got), f"{name}: igemm4 is not run-to-run deterministic"
    dec, shf = rnd(tuple(want.shape), 314, dtype), rnd(tuple(want.shape), 315, dtype)
    check(f"{name}_v4_sft", ops().conv2d(gx, gw, gb, sft=(g(dec), g(shf), 0.6), **v4, **kw),
          E.conv2d(x, wt, b, sft=(dec, shf, 0.6), **kw), dtype)


V5_SHAPES = [("v5_w32", 2, 32, 32, 64, 256, 3), ("v5_w64_c128", 1, 64, 64, 128, 200, 3), ("v5_w128", 1, 32, 128, 192, 256, 3),
             ("v5_w256", 1, 8, 256, 64, 72, 3), ("v5_w512_ragged_m", 1, 2, 512, 64, 64, 3), ("v5_big", 4, 64, 64, 256, 256, 3),
             ("v5_1x3", 1, 32, 64, 128, 128, (1, 3))]


@pytest.mark.parametrize("dtype", H16, ids=["bf16", "f16"])
@pytest.mark.parametrize("shape", V5_SHAPES, ids=[s[0] for s in V5_SHAPES])
def test_conv2d_tap_reuse_kernel(shape, dtype):
    """igemm5 (igemm4 schedule, one LDS input image shared by the three horizontal taps, reads shifted by kx) against
    the emulation and v1, all epilogues, plus the repeat-run determinism screen of the hand-placed schedule."""
    name, n, h, w_, cin, cout, k = shape
    kh, kw_ = k if isinstance(k, tuple) else (k, k)
    x = rnd((n, h, w_, cin), 410, dtype)
    wt = rnd((cout, kh * kw_ * cin), 411, dtype, 1.0 / np.sqrt(kh * kw_ * cin))
    wt[:, 0] += (torch.arange(cout, dtype=torch.float32) * 0.01).to(dtype)
    b = rnd((cout,), 412, torch.float32, 0.1)
    kw = dict(kh=kh, kw=kw_, pad=(kh // 2, kh // 2, 1, 1))
    gx, gw, gb = g(x), g(wt), g(b)
    want = E.conv2d(x, wt, b, act=E.ACT_SILU, **kw)
    check(f"{name}_v5", ops().conv2d(gx, gw, gb, act=E.ACT_SILU, kernel=5, **kw), want, dtype)
    res = rnd(tuple(want.shape), 413, dtype)
    gres = g(res)
    got = ops().conv2d(gx, gw, gb, res=gres, post_relu=True, kernel=5, **kw)
    check(f"{name}_v5_res", got, E.conv2d(x, wt, b, res=res, post_relu=True, **kw), dtype)
    v1 = ops().conv2d(gx, gw, gb, res=gres, post_relu=True, kernel=1, **kw)
    check(f"{name}_v5_v1", got, v1, dtype, 0.2)
    for _ in range(20):
        again = ops().conv2d(gx, gw, gb, res=gres, post_relu=True, kernel=5, **kw)
        assert torch.equal(again, got), f"{name}: igemm5 is not run-to-run deterministic"
    dec, shf = rnd(tuple(want.shape), 414, dtype), rnd(tuple(want.shape), 415, dtype)
    check(f"{name}_v5_sft", ops().conv2d(gx, gw, gb, sft=(g(dec), g(shf), 0.6), kernel=5, **kw),
          E.conv2d(x, wt, b, sft=(dec, shf, 0.6), **kw), dtype)


V6_SHAPES = [("v6_w32", 2, 32, 32, 64, 64), ("v6_w64_c40", 1, 16, 64, 64, 40), ("v6_w128", 3, 8, 128, 64, 64),
             ("v6_w512_ragged_m", 1, 2, 512, 64, 64), ("v6_many_tiles", 6, 128, 128, 64, 64), ("v6_c8", 1, 32, 64, 64, 8), ("v6_c3_scalar_epilogue", 2, 32, 64, 64, 3)]


@pytest.mark.parametrize("dtype", H16, ids=["bf16", "f16"])
@pytest.mark.parametrize("shape", V6_SHAPES, ids=[s[0] for s in V6_SHAPES])
def test_conv2d_c64_kernel(shape, dtype):
    """igemm6 (3x3, Cin = 64, Cout <= 64: register-resident weights, persistent workgroups, halo images) against the
    emulation and v1, all epilogues, repeat-run determinism."""
    name, n, h, w_, cin, cout = shape
    x = rnd((n, h, w_, cin), 510, dtype)
    wt = rnd((cout, 9 * cin), 511, dtype, 1.0 / np.sqrt(9 * cin))
    wt[:, 0] += (torch.arange(cout, dtype=torch.float32) * 0.01).to(dtype)
    b = rnd((cout,), 512, torch.float32, 0.1)
    kw = dict(kh=3, kw=3, pad=(1, 1, 1, 1))
    gx, gw, gb = g(x), g(wt), g(b)
    want = E.conv2d(x, wt, b, act=E.ACT_SILU, **kw)
    check(f"{name}_v6", ops().conv2d(gx, gw, gb, act=E.ACT_SILU, kernel=6, **kw), want, dtype)
    res = rnd(tuple(want.shape), 513, dtype)
    gres = g(res)
    got = ops().conv2d(gx, gw, gb, res=gres, post_relu=True, kernel=6, **kw)
    check(f"{name}_v6_res", got, E.conv2d(x, wt, b, res=res, post_relu=True, **kw), dtype)
    v1 = ops().conv2d(gx, gw, gb, res=gres, post_relu=True, kernel=1, **kw)
    check(f"{name}_v6_v1", got, v1, dtype, 0.2)
    for _ in range(10):
        again = ops().conv2d(gx, gw, gb, res=gres, post_relu=True, kernel=6, **kw)
        assert torch.equal(again, got), f"{name}: igemm6 is not run-to-run deterministic"
    dec, shf = rnd(tuple(want.shape), 514, dtype), rnd(tuple(want.shape), 515, dtype)
    check(f"{name}_v6_sft", ops().conv2d(gx, gw, gb, sft=(g(dec), g(shf), 0.6), kernel=6, **kw),
          E.conv2d(x, wt, b, sft=(dec, shf, 0.6), **kw), dtype)


V8_SHAPES = [("v8_w128", 3, 8, 128, 64), ("v8_w512_two_strips", 2, 128, 512, 64), ("v8_w256_c40", 1, 16, 256, 40), ("v8_h4", 2, 4, 128, 64),
             ("v8_c8", 1, 32, 128, 8), ("v8_c3_scalar_epilogue", 2, 32, 128, 3), ("v8_c24", 1, 64, 256, 24)]


@pytest.mark.parametrize("dtype", H16, ids=["bf16", "f16"])
@pytest.mark.parametrize("shape", V8_SHAPES, ids=[s[0] for s in V8_SHAPES])
def test_conv2d_c64_ring_kernel(shape, dtype):
    """igemm8 (round 5; 3x3, Cin = 64, Cout <= 64, W >= 128: a ring of row images in LDS, weights as the MFMA A operand, epilogue
    straight from the accumulators) against the emulation, against igemm6 on the same launch, with a per-frame bias and a residual,
    fp32 output, channel-slice views, repeat-run determinism - and the FUSED operand: conv(act(x * scale[n, c] + shift[n, c]))
    (pgt_conv2d_affine_in: the GroupNorm apply + SiLU of the Normalize in front of the conv) equals affine_act followed by the
    same kernel BIT FOR BIT (same operand bits, same accumulation order)."""
    name, n, h, w_, cout = shape
    cin = 64
    O = ops()
    x = rnd((n, h, w_, cin), 810, dtype)
    wt = rnd((cout, 9 * cin), 811, dtype, 1.0 / np.sqrt(9 * cin))
    wt[:, 0] += (torch.arange(cout, dtype=torch.float32) * 0.01).to(dtype)
    b = rnd((cout,), 812, torch.float32, 0.1)
    kw = dict(kh=3, kw=3, pad=(1, 1, 1, 1))
    gx, gw, gb = g(x), g(wt), g(b)
    want = E.conv2d(x, wt, b, **kw)
    got = O.conv2d(gx, gw, gb, kernel=8, **kw)
    check(f"{name}_v8", got, want, dtype)
    check(f"{name}_v8_auto", O.conv2d(gx, gw, gb, **kw), want, dtype)
    check(f"{name}_v8_v6", got, O.conv2d(gx, gw, gb, kernel=6, **kw), dtype, 0.2)
    res = rnd(tuple(want.shape), 813, dtype)
    gres = g(res)
    got = O.conv2d(gx, gw, gb, res=gres, kernel=8, **kw)
    check(f"{name}_v8_res", got, E.conv2d(x, wt, b, res=res, **kw), dtype)
    for _ in range(10):
        assert torch.equal(O.conv2d(gx, gw, gb, res=gres, kernel=8, **kw), got), f"{name}: igemm8 is not run-to-run deterministic"
    check(f"{name}_v8_f32out", O.conv2d(gx, gw, gb, out_f32=True, kernel=8, **kw), want, dtype)
    if (h * w_) % 512 == 0:
        fb = rnd((n, cout), 814, torch.float32, 0.3)
        check(f"{name}_v8_frame_bias", O.conv2d(gx, gw, g(fb), res=gres, kernel=8, **kw), E.conv2d(x, wt, fb, res=res, **kw), dtype)
    if cout % 8 == 0:           # input and output as channel slices of wider buffers
        wide_in, wide_out = g(rnd((n, h, w_, cin + 16), 815, dtype)), torch.zeros((n, h, w_, cout + 8), dtype=dtype, device=DEV)
        wide_in[..., 8:8 + cin] = gx
        O.conv2d(wide_in[..., 8:8 + cin], gw, gb, out=wide_out[..., 8:], kernel=8, **kw)
        check(f"{name}_v8_views", wide_out[..., 8:], want, dtype)
        assert float(wide_out[..., :8].abs().max()) == 0.0
    # ---- the fused operand
    sc = (1.0 + 0.3 * rnd((n, cin), 816)).contiguous()
    sh = (0.2 * rnd((n, cin), 817)).contiguous()
    for act in (E.ACT_SILU, E.ACT_NONE):
        xa = O.affine_act(gx, g(sc), g(sh), act)
        two = O.conv2d(xa, gw, gb, res=gres, kernel=8, **kw)
        one = O.conv2d(gx, gw, gb, res=gres, affine_in=(g(sc), g(sh), act), **kw)
        assert torch.equal(one, two), f"{name}: fused operand (act {act}) differs from affine_act + conv: {(one.float() - two.float()).abs().max().item():.3e}"
        check(f"{name}_v8_fused_act{act}", one, E.conv2d(E.affine_act(x, sc, sh, act), wt, b, res=res, **kw), dtype)
    # no fused form for a launch the ring kernel does not cover (W = 64): the library says so and refuses the call
    import ctypes as _C

    from pgtformer_amd import hip as _hip
    dd = _hip.ConvDesc()
    dd.dtype, dd.N, dd.H, dd.W, dd.Cin, dd.ldx = (1 if dtype == torch.bfloat16 else 3), 1, 32, 64, 64, 64
    dd.KH = dd.KW = 3
    dd.stride = dd.pad_t = dd.pad_l = 1
    dd.Ho, dd.Wo, dd.Cout, dd.ldy = 32, 64, cout, cout
    assert _hip.lib().pgt_conv2d_affine_in_ok(_C.byref(dd)) == 0
    d_x = g(rnd((1, 32, 64, 64), 1, dtype))
    yy = torch.empty((1, 32, 64, cout), dtype=dtype, device=DEV)
    rc = _hip.lib().pgt_conv2d_affine_in(_C.byref(dd), d_x.data_ptr(), g(sc).data_ptr(), g(sh).data_ptr(), 3, gw.data_ptr(), None, None,
                                         yy.data_ptr(), None)
    assert rc == -22 and b"fused-operand" in _hip.lib().pgt_last_error()
    dd.W = dd.Wo = dd.ldx * 2
    dd.H = dd.Ho = 16
    assert _hip.lib().pgt_conv2d_affine_in_ok(_C.byref(dd)) == (1 if (cout % 8 == 0 or cout <= 32) else 0)
    # ... and ops.conv2d then runs the apply pass itself
    small = O.conv2d(d_x, gw, gb, affine_in=(g(sc[:1].contiguous()), g(sh[:1].contiguous()), E.ACT_SILU), **kw)
    assert torch.equal(small, O.conv2d(O.affine_act(d_x, g(sc[:1].contiguous()), g(sh[:1].contiguous()), E.ACT_SILU), gw, gb, **kw))


def _exact_conv_f64(x, w32, b, kh, kw, pad, res=None, act=E.ACT_NONE, stride=1):
    """conv of the (half) operand x with the EXACT fp32 weights, in float64: what an exact-weight layer approximates"""
    import torch.nn.functional as F
    cout = w32.shape[0]
    xi = F.pad(x.double().permute(0, 3, 1, 2), (pad[2], pad[3], pad[0], pad[1]))
    wt = w32.double().reshape(cout, kh, kw, -1).permute(0, 3, 1, 2)
    y = F.conv2d(xi, wt, b.double(), stride=stride).permute(0, 2, 3, 1)
    y = E._act(y.float(), act).double() if act != E.ACT_NONE else y
    return y if res is None else y + res.double()


W2_CASES = [
    # name, (n, h, w, cin), cout, k, what
    ("ring_64_64", (3, 16, 128, 64), 64, 3, "ring"),
    ("ring_64_3", (2, 8, 256, 64), 3, 3, "ring"),
    ("ring_64_16", (2, 8, 128, 64), 16, 3, "ring"),
    ("ring_64_40", (2, 8, 128, 64), 40, 3, "ring"),
    ("v4_128_64_3x3", (2, 32, 32, 128), 64, 3, "v4"),
    ("v4_512_512_3x3", (3, 32, 32, 512), 512, 3, "v4"),
    ("v4_192_96_1x1", (2, 32, 32, 192), 96, 1, "v4"),
    ("v4_64_64_small_map", (2, 32, 32, 64), 64, 3, "v4"),
    ("v4_256_24_1x1", (1, 16, 64, 256), 24, 1, "v4"),
]


@pytest.mark.parametrize("case", W2_CASES, ids=[c[0] for c in W2_CASES])
def test_conv2d_exact_weights(case):
    """The exact-weight form of the IEEE-half layers (pgt_conv_desc::w2, round 6): two weight planes
    (w_hi | (w - w_hi) * 2048), two MFMAs per product, y = acc_hi + acc_lo / 2048 - in the 64-channel ring kernel (16 output channels
    per wave, both planes in registers) and in the phased LDS-DMA kernel (a wave's 64 tile columns = hi and lo rows of 32 output
    channels).  With fp32 output the result must sit at fp32-ACCUMULATION distance from the conv with the exact fp32 weights (2e-5 of
    the output scale), where the single-plane layer of the same launch is 2^-12-per-weight away (asserted: > 5x further) - the test
    would not pass with the lo plane dropped, mis-scaled or paired with the wrong channel.  Also: half output = one rounding of that
    result, residual / activation / per-frame bias / SFT epilogues, channel-slice views, epilogue GroupNorm statistics, run-to-run bits."""
    name, (n, h, w_, cin), cout, k, kind = case
    O = ops()
    x = rnd((n, h, w_, cin), 910, torch.float16)
    w32 = rnd((cout, k * k * cin), 911, torch.float32, 1.0 / np.sqrt(k * k * cin))
    w32[:, 0] += torch.arange(cout, dtype=torch.float32) * 0.01          # asymmetric in the output channel
    w4 = w32.reshape(cout, k, k, cin).permute(0, 3, 1, 2).contiguous()    # the reference's (Cout, Cin, KH, KW)
    b = rnd((cout,), 912, torch.float32, 0.1)
    pad = (k // 2,) * 4
    kw = dict(kh=k, kw=k, pad=pad)
    gx, gb = g(x), g(b)
    pw2 = O.pack_conv_weight(g(w4), torch.float16, w2=True)
    pw1 = O.pack_conv_weight(g(w4), torch.float16)
    assert tuple(pw2.shape) == (O.w2_rows(cout), k * k * cin)
    assert torch.equal(pw2.cpu(), E.pack_conv_weight(w4, torch.float16, w2=True)), "library and emulation pack different exact-weight operands"
    assert O.w2_ok(gx, cout, cin, k, k, 1, pad, bias=gb)
    assert (kind == "ring") == O.ring_covers(gx, cout, k, k, 1, pad, bias=gb)
    want64 = _exact_conv_f64(x, w32, b, k, k, pad)
    scale = max(1.0, float(want64.abs().max()))
    got = O.conv2d(gx, pw2, gb, w2=cout, out_f32=True, **kw)
    err2 = float((got.double().cpu() - want64).abs().max())
    one = O.conv2d(gx, pw1, gb, out_f32=True, **kw)
    err1 = float((one.double().cpu() - want64).abs().max())
    _LOG.append({"name": f"w2_{name}_f32out", "dtype": "float16 (two weight planes)", "max_abs_err": err2, "single_plane_err": err1,
                 "ref_absmax": scale, "tol": 2e-5 * scale, "ok": bool(err2 <= 2e-5 * scale)})
    assert err2 <= 2e-5 * scale, (name, err2, scale)
    assert err1 > 5 * err2, f"{name}: the single-plane launch is as close to the exact conv as the exact-weight one ({err1:.2e} vs {err2:.2e}): the test cannot see the lo plane"
    check(f"w2_{name}_emu_f32out", got, E.conv2d(x, E.pack_conv_weight(w4, torch.float16, w2=True), b, w2=cout, out_f32=True, **kw), torch.float32)
    # half output: ONE rounding of the fp32 result
    goth = O.conv2d(gx, pw2, gb, w2=cout, **kw)
    assert float((goth.float() - got.clamp(-65504, 65504).half().float()).abs().max()) == 0.0 or \
        float((goth.double().cpu() - want64).abs().max()) <= 6e-4 * scale
    for _ in range(5):
        assert torch.equal(O.conv2d(gx, pw2, gb, w2=cout, **kw), goth), f"{name}: not run-to-run deterministic"
    # residual (both kernels), fp32 out
    res = rnd(tuple(want64.shape), 913, torch.float16)
    gres = g(res)
    gotr = O.conv2d(gx, pw2, gb, w2=cout, res=gres, out_f32=True, **kw)
    assert float((gotr.double().cpu() - (want64 + res.double())).abs().max()) <= 2e-5 * scale
    if (h * w_) % 512 == 0 and (kind == "v4" or (h * w_) % (4 * w_) == 0):      # one bias vector per frame
        fb = rnd((n, cout), 914, torch.float32, 0.3)
        gotb = O.conv2d(gx, pw2, g(fb), w2=cout, out_f32=True, **kw)
        wantb = _exact_conv_f64(x, w32, torch.zeros(cout), k, k, pad) + fb.double()[:, None, None, :]
        assert float((gotb.double().cpu() - wantb).abs().max()) <= 2e-5 * scale
    if kind == "ring":
        # the fused operand (GroupNorm apply + SiLU in the operand load) with exact weights = apply pass + the same kernel, bit for bit
        sc = (1.0 + 0.3 * rnd((n, cin), 916)).contiguous()
        sh = (0.2 * rnd((n, cin), 917)).contiguous()
        xa = O.affine_act(gx, g(sc), g(sh), E.ACT_SILU)
        two = O.conv2d(xa, pw2, gb, w2=cout, res=gres if cout % 8 == 0 else None, **kw)
        fused = O.conv2d(gx, pw2, gb, w2=cout, res=gres if cout % 8 == 0 else None, affine_in=(g(sc), g(sh), E.ACT_SILU), **kw)
        assert torch.equal(fused, two), f"{name}: fused operand differs from affine_act + conv"
    else:
        if cout % 8 == 0:
            # activation, SFT epilogue, channel-slice views
            gota = O.conv2d(gx, pw2, gb, w2=cout, act=E.ACT_LEAKY02, out_f32=True, **kw)
            assert float((gota.double().cpu() - _exact_conv_f64(x, w32, b, k, k, pad, act=E.ACT_LEAKY02)).abs().max()) <= 2e-5 * scale
            dec, shf = rnd(tuple(want64.shape), 918, torch.float16), rnd(tuple(want64.shape), 919, torch.float16)
            gots = O.conv2d(gx, pw2, gb, w2=cout, sft=(g(dec), g(shf), 0.7), **kw)
            wants = dec.double() + 0.7 * (dec.double() * want64 + shf.double())
            assert float((gots.double().cpu() - wants).abs().max()) <= 2e-3 * max(1.0, float(wants.abs().max()))
            wide_in = g(rnd((n, h, w_, cin + 16), 920, torch.float16))
            wide_out = torch.zeros((n, h, w_, cout + 8), dtype=torch.float16, device=DEV)
            wide_in[..., 8:8 + cin] = gx
            O.conv2d(wide_in[..., 8:8 + cin], pw2, gb, w2=cout, out=wide_out[..., 8:], **kw)
            assert torch.equal(wide_out[..., 8:], goth) and float(wide_out[..., :8].abs().max()) == 0.0
        if cout % 32 == 0 and O._w2_gn_ok(h * w_, cout, 32) and O.gn_ok(n, h * w_, cout, 32, cin, k):
            # epilogue GroupNorm statistics of the exact-weight tile (half as many columns per wave)
            y = O.conv2d(gx, pw2, gb, w2=cout, gn=32, **kw)
            assert getattr(y, "_pgt_gn", None) is not None
            gam, bet = g(1.0 + 0.1 * rnd((cout,), 921)), g(0.1 * rnd((cout,), 922))
            sc1, sh1 = O.groupnorm_affine(y, gam, bet, 32, 1e-6)
            y2 = y.clone()            # (no statistics attached: the separate pass)
            sc2, sh2 = O.groupnorm_affine(y2, gam, bet, 32, 1e-6)
            # epilogue statistics are taken of the fp32 values before the store rounding, the pass reads the rounded tensor
            assert float((sc1 - sc2).abs().max()) <= 2e-3 * float(sc2.abs().max()) and float((sh1 - sh2).abs().max()) <= 2e-3 * max(1.0, float(sh2.abs().max()))
    # a launch the library has no exact-weight form for is refused, not silently run on one plane
    if kind == "v4" and cin % 64 == 0:
        odd = g(rnd((1, 8, 8, cin + 8), 923, torch.float16))[..., 4:4 + cin]      # operand rows not 16-byte aligned
        assert not O.w2_ok(odd, cout, cin, k, k, 1, pad)
        from pgtformer_amd.hip import PgtError
        with pytest.raises((PgtError, AssertionError)):
            O.conv2d(odd, pw2, gb, w2=cout, **kw)


def test_ring_kernel_banded_bias_with_long_strips():
    """the ring kernel takes ONE bias vector per strip of R rows, and R grows to 32 / 64 rows once a launch
    has enough strips (>= 128 frames at 256 x 256 on this chip); with a bias per BAND of the frame (16 bands: 16 rows at 256 x 256)
    a long strip straddled bands and the later bands got the first band's bias.  The launcher now clamps R to whole bands
    (ring_legal refuses bias bands that are not whole 4-row strips).  136 frames at 256 x 128: R would be 64, bands are 8 rows."""
    O = ops()
    n, h, w_, cin, cout = 136, 256, 128, 64, 64
    x = rnd((n, h, w_, cin), 930, torch.float16)
    wt = rnd((cout, 9 * cin), 931, torch.float16, 1.0 / np.sqrt(9 * cin))
    bands = 32
    fb = rnd((n * bands, cout), 932, torch.float32, 0.5)
    kw = dict(kh=3, kw=3, pad=(1, 1, 1, 1))
    got = O.conv2d(g(x), g(wt), g(fb), kernel=8, **kw)
    base = O.conv2d(g(x), g(wt), None, kernel=8, out_f32=True, **kw)
    want = (base.reshape(n * bands, -1, cout) + g(fb)[:, None, :]).reshape(base.shape)
    err = float((got.float() - want).abs().max())
    assert err <= 5e-3 * max(1.0, float(want.abs().max())), err
    # bands that are not whole 4-row strips: the ring kernel is not offered, the launch runs on another kernel with the same result
    fb2 = rnd((n * 128, cout), 933, torch.float32, 0.5)       # 2-row bands
    assert not O.ring_covers(g(x), cout, 3, 3, 1, (1, 1, 1, 1), bias=g(fb2))


def test_frame_bias_concurrent_streams_need_their_own_counters():
    """The arrival counters of pgt_frame_bias are shared by consecutive launches of ONE stream; launches that run concurrently must
    not share them (found in round 5: in the pure-bf16 mode BiSeNet's compensated convs on the side stream raced with the encoder's,
    a flaky bit-difference).  ops.BRANCH / ops.LANE select the counter set: two streams with their own sets always reproduce the
    single-stream result."""
    O = ops()
    x = (rnd((96, 64, 64, 256), 71, torch.float16) + 0.25)
    dt_, b = rnd((256, 256), 72) * 1e-3, rnd((256,), 73)
    gx, gd, gb = g(x), g(dt_), g(b)
    want = O.frame_bias(gx, gd, gb).clone()
    assert O._fb_counters(gx.device, 96) is not None
    keep = O.BRANCH
    try:
        O.BRANCH = 1
        c1 = O._fb_counters(gx.device, 96)
        O.BRANCH = 0
        assert c1.data_ptr() != O._fb_counters(gx.device, 96).data_ptr()
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        torch.cuda.synchronize()
        outs = []
        for i in range(200):
            with torch.cuda.stream(s1):
                O.BRANCH = 0
                outs.append(O.frame_bias(gx, gd, gb))
            with torch.cuda.stream(s2):
                O.BRANCH = 1
                outs.append(O.frame_bias(gx, gd, gb))
        torch.cuda.synchronize()
    finally:
        O.BRANCH = keep
    assert all(torch.equal(o, want) for o in outs)
    assert int(O._fb_counters(gx.device, 96).abs().sum()) == 0 and int(c1.abs().sum()) == 0      # every call left its counters at zero


@pytest.mark.parametrize("dtype", H16, ids=["bf16", "f16"])
def test_frame_bias_one_launch(dtype):
    """pgt_frame_bias (round 5): sampled channel mean + mean-field bias in ONE launch, against the emulation and against the two
    launches it replaces; with the fused-operand form (the sample is taken of act(x * scale + shift) rounded to the tensor's type),
    on channel slices, deterministic."""
    O = ops()
    for (n, h, w, c, cw, cout) in [(3, 32, 32, 64, 64, 64), (2, 128, 128, 256, 320, 256), (2, 512, 512, 64, 64, 3), (5, 4, 8, 72, 72, 40),
                                   (1, 64, 64, 576, 576, 128), (2, 32, 32, 1056, 1056, 512), (96, 32, 32, 512, 512, 512)]:
        buf = rnd((n, h, w, cw), 31 + c, dtype) + 0.25
        x = buf[..., :c]
        dt_ = rnd((c, cout), 32) * 1e-3
        b = rnd((cout,), 33)
        want = E.sampled_channel_mean(x) @ dt_ + b
        got = O.frame_bias(g(buf)[..., :c], g(dt_), g(b))
        check(f"frame_bias_{n}x{h}x{w}x{c}->{cout}", got, want, torch.float32, tol_scale=0.05)
        two = O.mean_field_bias(O.sampled_channel_mean(g(buf)[..., :c]), g(dt_), g(b))
        check(f"frame_bias_vs_two_launches_{c}", got, two, torch.float32, tol_scale=0.01)
        assert torch.equal(got, O.frame_bias(g(buf)[..., :c], g(dt_), g(b)))
        check(f"frame_bias_nobias_{c}", O.frame_bias(g(buf)[..., :c], g(dt_)), want - b, torch.float32, tol_scale=0.05)
        sc, sh = (1.0 + 0.3 * rnd((n, c), 34)).contiguous(), (0.2 * rnd((n, c), 35)).contiguous()
        xa = O.affine_act(g(buf)[..., :c], g(sc), g(sh), E.ACT_SILU)
        fused = O.frame_bias(g(buf)[..., :c], g(dt_), g(b), affine_in=(g(sc), g(sh), E.ACT_SILU))
        assert torch.equal(fused, O.frame_bias(xa, g(dt_), g(b))), f"frame_bias: fused-operand sample differs ({c})"
        if (h * w) % 1024 == 0:     # bands: the frames are horizontal bands of the images, the coefficient rows stay per image (scale_div)
            xb, nb = O.banded(g(buf)[..., :c], 2)
            assert nb == 2 and tuple(xb.shape) == (2 * n, h * w // 2, c)
            fb_bands = O.frame_bias(xb, g(dt_), g(b), affine_in=(g(sc), g(sh), E.ACT_SILU), scale_div=2)
            assert torch.equal(fb_bands, O.frame_bias(O.banded(xa, 2)[0], g(dt_), g(b), scale_div=2))      # (same lanes per band: same sums)
            check(f"frame_bias_bands_{c}", O.frame_bias(xb, g(dt_), g(b)), E.sampled_channel_mean(x.reshape(2 * n, h * w // 2, c)) @ dt_ + b,
                  torch.float32, tol_scale=0.05)
            # a sparser sample per band (sample_cells): 16 cells x 16 pixels, the pixels pgt_sampled_pixel_cells names
            idx = E.sampled_pixels(h * w // 2, 16)
            assert len(idx) == min(h * w // 2, 256) and len(set(idx.tolist())) == len(idx)
            check(f"frame_bias_bands_sparse_{c}", O.frame_bias(xb, g(dt_), g(b), sample_cells=16),
                  E.sampled_channel_mean(x.reshape(2 * n, h * w // 2, c), 16) @ dt_ + b, torch.float32, tol_scale=0.05)
        if cout % 4 == 0:       # G layers side by side (the four sub-pixel convolutions of an Upsample): (G, N, Csub), same numbers
            grouped = O.frame_bias(g(buf)[..., :c], g(dt_), g(b), groups=4)
            assert tuple(grouped.shape) == (4, n, cout // 4) and grouped.is_contiguous()
            assert torch.equal(grouped.permute(1, 0, 2).reshape(n, cout), got)


def test_conv2d_output_parity_placement_splitk():
    """Split-K with output placement: the fp32 slabs stay dense, the reduce kernel scatters the rows (small, deep-K layer:
    the first decoder up-sampling at 16x16)."""
    dtype = torch.bfloat16
    n, h, w_, cin, cout = 3, 16, 16, 512, 512
    x = rnd((n, h, w_, cin), 620, dtype)
    b = rnd((cout,), 622, torch.float32, 0.1)
    out = torch.zeros((n, 2 * h, 2 * w_, cout), dtype=dtype, device="cuda")
    guard = torch.zeros((1 << 20,), dtype=torch.float32, device="cuda")   # lands right after `out` in a fresh pool
    ref = torch.zeros((n, 2 * h, 2 * w_, cout), dtype=dtype)
    for py in (0, 1):
        for px in (0, 1):
            w2 = rnd((cout, 4 * cin), 623 + 2 * py + px, dtype, 1.0 / np.sqrt(4 * cin))
            kw = dict(kh=2, kw=2, pad=(1 - py, py, 1 - px, px), out_parity=(py, px))
            ops().conv2d(g(x), g(w2), g(b), out=out, splitk=4, **kw)
            E.conv2d(x, w2, b, out=ref, **kw)
    torch.cuda.synchronize()
    check("parity_splitk", out, ref, dtype)
    assert float(guard.abs().max()) == 0.0


@pytest.mark.parametrize("dtype", H16, ids=["bf16", "f16"])
@pytest.mark.parametrize("kernel", [0, 1, 4])
def test_conv2d_output_parity_placement(kernel, dtype):
    """2x2 sub-pixel convolution written to out[:, py::2, px::2, :] (pgt_conv_desc::orow_*) for all four parities; the
    merged-tap decomposition reproduces nearest-x2 + conv3x3."""
    n, h, w_, cin, cout = 2, 12, 32, 64, 128
    x = rnd((n, h, w_, cin), 610, dtype)
    w3 = rnd((cout, 3, 3, cin), 611, torch.float32, 1.0 / np.sqrt(9 * cin))
    b = rnd((cout,), 612, torch.float32, 0.1)
    rows = (((0,), (1, 2)), ((0, 1), (2,)))
    out = torch.zeros((n, 2 * h, 2 * w_, cout), dtype=dtype, device="cuda")
    ref = torch.zeros((n, 2 * h, 2 * w_, cout), dtype=dtype)
    for py in (0, 1):
        for px in (0, 1):
            w2 = torch.stack([torch.stack([sum(w3[:, ky, kx, :] for ky in rows[py][a] for kx in rows[px][bb])
                                           for bb in (0, 1)], 1) for a in (0, 1)], 1).reshape(cout, -1).to(dtype)
            kw = dict(kh=2, kw=2, pad=(1 - py, py, 1 - px, px), out_parity=(py, px))
            extra = dict(kernel=kernel, tile=(0, 128)) if kernel == 4 else dict(kernel=kernel)
            ops().conv2d(g(x), g(w2), g(b), out=out, **extra, **kw)
            E.conv2d(x, w2, b, out=ref, **kw)
    check(f"parity_k{kernel}", out, ref, dtype)
    full = E.conv2d(x, w3.reshape(cout, -1).to(dtype), b, kh=3, kw=3, pad=(1, 1, 1, 1), ups=True)
    check(f"parity_vs_ups_k{kernel}", out, full, dtype, 2.0)   # merged taps are rounded to bf16 once, not three times


# ------------------------------------------------------------------------------------------------
# GroupNorm statistics from the producing conv's epilogue (pgt_conv2d_gn + pgt_groupnorm_from_partials)
GN_CASES = [
    # name, N,H,W,Cin,Cout,k, kernel, tile
    ("gn_v1_64", 3, 32, 32, 32, 64, 3, 1, (64, 64)),
    ("gn_v1_128x128", 2, 32, 32, 64, 128, 3, 1, (128, 128)),
    ("gn_v4_256", 2, 32, 32, 128, 256, 3, 4, (0, 256)),
    ("gn_v4_512x128", 2, 32, 32, 128, 128, 3, 4, (0, 128)),
    ("gn_v4_two_n_tiles", 2, 32, 32, 128, 512, 3, 4, (0, 256)),
    ("gn_auto_512", 2, 32, 32, 512, 512, 1, 0, (0, 0)),
]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", GN_CASES, ids=[c[0] for c in GN_CASES])
def test_conv_epilogue_groupnorm_statistics(dtype, case):
    """The conv epilogue's per-group sum / sum of squares, finalised by pgt_groupnorm_from_partials, give the same GroupNorm
    coefficients as the separate statistics pass over the stored output (reference: Normalize, rstt_layers.py:754-755)."""
    name, n, h, w_, cin, cout, k, kernel, tile = case
    if dtype == torch.float32 and kernel != 1:
        pytest.skip("fp32 runs on the register-staged kernel only")
    O = ops()
    x = rnd((n, h, w_, cin), 200, dtype)
    wt = rnd((cout, k * k * cin), 201, dtype, 1.0 / np.sqrt(k * k * cin))
    b = rnd((cout,), 202, torch.float32, 0.3)
    res = rnd((n, h, w_, cout), 203, dtype)
    gam, bet = 1 + 0.1 * rnd((cout,), 204), 0.1 * rnd((cout,), 205)
    kw = dict(kh=k, kw=k, pad=(k // 2,) * 4, act=E.ACT_SILU, res=g(res), kernel=kernel, tile=tile)
    y = O.conv2d(g(x), g(wt), g(b), gn=32, **kw)
    st = getattr(y, "_pgt_gn", None)
    assert st is not None and st.groups == 32 and st.hw == h * w_
    plain = O.conv2d(g(x), g(wt), g(b), **kw)
    assert torch.equal(y, plain)                                        # the statistics do not touch the outputs
    s_e, b_e = O.groupnorm_affine(y, g(gam), g(bet))                    # from the epilogue statistics
    s_p, b_p = O.groupnorm_affine(plain, g(gam), g(bet))                # separate pass over the stored tensor
    s_w, b_w = E.groupnorm_affine(y.float().cpu(), gam, bet)
    tol = 1.0 if dtype == torch.float32 else 20.0                       # bf16: statistics of the un-rounded fp32 outputs
    check(f"{name}_scale", s_e, s_w, torch.float32, tol)
    check(f"{name}_shift", b_e, b_w, torch.float32, tol)
    check(f"{name}_scale_vs_pass", s_e, s_p, torch.float32, tol)
    again = O.groupnorm_affine(O.conv2d(g(x), g(wt), g(b), gn=32, **kw), g(gam), g(bet))
    assert torch.equal(again[0], s_e) and torch.equal(again[1], b_e)    # deterministic


@pytest.mark.parametrize("dtype", [torch.bfloat16])
def test_epilogue_statistics_linear_and_subpixel_and_x3(dtype):
    O = ops()
    # linear: rows = 3 images x 1024 tokens
    x, wt, b = rnd((3072, 256), 210, dtype), rnd((256, 256), 211, dtype, 1 / 16), rnd((256,), 212)
    res = rnd((3072, 256), 213, dtype)
    gam, bet = 1 + 0.1 * rnd((256,), 214), 0.1 * rnd((256,), 215)
    y = O.linear(g(x), g(wt), g(b), res=g(res), gn=(32, 3))
    assert getattr(y, "_pgt_gn", None) is not None
    y4 = y.reshape(3, 32, 32, 256)
    y._pgt_gn.bind(y4, 256)
    s_e, b_e = O.groupnorm_affine(y4, g(gam), g(bet))
    s_w, b_w = E.groupnorm_affine(y4.float().cpu(), gam, bet)
    check("gn_linear_scale", s_e, s_w, torch.float32, 20.0)
    check("gn_linear_shift", b_e, b_w, torch.float32, 20.0)
    # the four sub-pixel convolutions of Upsample write one tensor: 4 statistics sub-ranges
    from pgtformer_amd.archs.tdcrqvae3_arch import Upsample
    torch.manual_seed(5)
    up = Upsample(128, True)
    up.prepare(DEV, dtype)
    xin = rnd((2, 32, 32, 128), 216, dtype)
    out = up(g(xin))
    st = getattr(out, "_pgt_gn", None)
    assert st is not None and st.nsub == 4 and out.shape == (2, 64, 64, 128)
    gam, bet = 1 + 0.1 * rnd((128,), 217), 0.1 * rnd((128,), 218)
    s_e, b_e = O.groupnorm_affine(out, g(gam), g(bet))
    s_w, b_w = E.groupnorm_affine(out.float().cpu(), gam, bet)
    check("gn_subpixel_scale", s_e, s_w, torch.float32, 20.0)
    check("gn_subpixel_shift", b_e, b_w, torch.float32, 20.0)
    # split-half conv
    xs = E.to_x3(rnd((2, 32, 32, 128), 219))
    w3 = O.pack_x3_weight(rnd((256, 9, 128), 220, 1 / 34))
    ys = O.conv2d(g(xs), g(w3), None, kh=3, kw=3, pad=(1, 1, 1, 1), x3=True, gn=32)
    assert getattr(ys, "_pgt_gn", None) is not None
    gam, bet = 1 + 0.1 * rnd((256,), 221), 0.1 * rnd((256,), 222)
    s_e, b_e = O.groupnorm_affine(ys, g(gam), g(bet), x3=True)
    s_w, b_w = E.groupnorm_affine(ys.cpu(), gam, bet, x3=True)
    check("gn_x3_scale", s_e, s_w, torch.float32, 5.0)
    check("gn_x3_shift", b_e, b_w, torch.float32, 5.0)


def test_half_stores_saturate_instead_of_overflowing():
    """PGT_F16: every fp32 -> half store clamps to +-65504 (a value out of half range must not become inf and then NaN
    downstream); in-range values are rounded to nearest even as torch rounds them."""
    O = ops()
    x = torch.full((1, 8, 8, 64), 200.0, dtype=torch.float16)
    w = torch.full((64, 64), 100.0, dtype=torch.float16)                      # 64 * 200 * 100 = 1.28e6 > 65504
    y = O.conv2d(g(x), g(w), None).float().cpu()
    assert torch.isfinite(y).all() and float(y.min()) == 65504.0
    y = O.conv2d(g(x), g(-w), None).float().cpu()
    assert float(y.max()) == -65504.0
    big = torch.tensor([[1e6, -1e6, 70000.0, 65504.0, 65519.9, 1.00048828125, 3.1415927]], dtype=torch.float32).repeat(4, 1)
    got = O.cast(g(F_pad8(big)), torch.float16).float().cpu()[:, :7]
    want = big.clamp(-65504.0, 65504.0).to(torch.float16).float()
    assert torch.equal(got, want), (got[0], want[0])
    sc, sh = g(torch.full((1, 64), 1000.0)), g(torch.zeros(1, 64))
    z = O.affine_act(g(x), sc, sh).float().cpu()                               # 200 * 1000 = 2e5
    assert torch.isfinite(z).all() and float(z.max()) == 65504.0


def F_pad8(t):
    return torch.nn.functional.pad(t, (0, 8 - t.shape[1] % 8 if t.shape[1] % 8 else 0)).contiguous()


def test_weight_repack_behind_the_abi_matches_the_host_restatement():
    """pgt_pack_conv_weight / pgt_fold_batchnorm (the repack a non-Python host needs after loading a reference checkpoint:
    K-major rows, channel padding, BatchNorm fold, rounding, the split-half forms) against the torch restatement: packed
    operands bit for bit, the BatchNorm factors to one unit in the last place."""
    O = ops()
    w4 = rnd((72, 57, 3, 3), 950, torch.float32, 0.1)
    sc = 1.0 + 0.2 * rnd((72,), 951)
    for dt in (torch.float32, torch.bfloat16, torch.float16):
        got = O.pack_conv_weight(g(w4), dt, cin_pad=64, scale=g(sc)).cpu()
        want = E.pack_conv_weight(w4, dt, cin_pad=64, scale=sc)
        assert got.dtype == want.dtype and torch.equal(got, want), dt
    w2 = rnd((200, 136), 952, torch.float32, 0.1)                       # a Linear weight, no padding
    assert torch.equal(O.pack_conv_weight(g(w2), torch.float16).cpu(), E.pack_conv_weight(w2, torch.float16))
    wx = rnd((64, 128, 3, 3), 953, torch.float32, 0.05)
    assert torch.equal(O.pack_conv_weight(g(wx), O.X3).cpu(), E.pack_conv_weight(wx, O.X3))
    assert torch.equal(O.pack_conv_weight(g(wx), O.X3).cpu(), O.pack_x3_weight(wx.permute(0, 2, 3, 1).reshape(64, 9, 128)))
    assert torch.equal(O.pack_conv_weight(g(wx), O.X3, fold=True).cpu(), O.pack_x3_fold_weight(wx.permute(0, 2, 3, 1).reshape(64, 9, 128)))
    w57 = rnd((512, 57, 1, 1), 954, torch.float32, 0.1)                 # convpos: 57 -> 64 input channels, split-half
    assert torch.equal(O.pack_conv_weight(g(w57), O.X3, cin_pad=64).cpu(), E.pack_conv_weight(w57, O.X3, cin_pad=64))
    gam, bet, mu, var, b0 = 1 + 0.1 * rnd((72,), 955), 0.1 * rnd((72,), 956), 0.1 * rnd((72,), 957), rnd((72,), 958).abs() + 0.5, rnd((72,), 959)
    s_g, b_g = O.fold_batchnorm(g(gam), g(bet), g(mu), g(var), 1e-5, g(b0))
    s_w, b_w = E.fold_batchnorm(gam, bet, mu, var, 1e-5, b0)
    ulp = lambda a, b: float(((a.cpu() - b).abs() / b.abs().clamp_min(1e-30)).max())   # noqa: E731  (device sqrt / divide: <= 1 ulp)
    assert ulp(s_g, s_w) <= 2.4e-7 and float((b_g.cpu() - b_w).abs().max()) <= 5e-7, (ulp(s_g, s_w), float((b_g.cpu() - b_w).abs().max()))
    s_g, b_g = O.fold_batchnorm(g(gam), g(bet), g(mu), g(var), 1e-5, None)
    assert float((b_g.cpu() - E.fold_batchnorm(gam, bet, mu, var, 1e-5, None)[1]).abs().max()) <= 5e-7


def test_statistics_epilogue_only_on_kernels_1_and_4():
    """Round 2 left a wrong-result variant (igemm5 with a statistics epilogue) reverted; it is gone: a conv pinned to kernel 5
    or 6 leaves NO epilogue statistics (the following GroupNorm takes its separate pass), and the C-ABI rejects the request."""
    from pgtformer_amd import hip
    O = ops()
    x = rnd((2, 32, 32, 64), 900, torch.bfloat16)
    wt = rnd((64, 9 * 64), 901, torch.bfloat16, 0.04)
    for k in (5, 6):
        y = O.conv2d(g(x), g(wt), None, kh=3, kw=3, pad=(1, 1, 1, 1), kernel=k, gn=32)
        assert getattr(y, "_pgt_gn", None) is None
    st = O.GnStats(2, 1, 32 * 32, 64, 32, DEV)
    with pytest.raises(hip.PgtError, match="statistics epilogue"):
        O.conv2d(g(x), g(wt), None, kh=3, kw=3, pad=(1, 1, 1, 1), kernel=5, gn=(st, 0))


def test_model_with_and_without_epilogue_statistics(monkeypatch):
    """Whole TDResnetBlock / decoder-style chain: epilogue statistics on vs off (PGT_EPILOGUE_GN=0) agree to the rounding of
    the statistics (bf16: the stored tensor vs its un-rounded fp32 values)."""
    import pgtformer_amd.ops as O
    from pgtformer_amd.modules.rstt_layers import TDResnetBlock
    torch.manual_seed(7)
    blk = TDResnetBlock(in_channels=128, out_channels=256)
    for p_ in blk.parameters():
        torch.nn.init.normal_(p_, std=0.05)
    x = rnd((3, 32, 32, 128), 230)
    for dtype in DTYPES:
        blk.prepare(DEV, dtype)
        on = blk(g(x.to(dtype)), gn_next=True)
        assert getattr(on, "_pgt_gn", None) is not None
        monkeypatch.setattr(O, "USE_EPILOGUE_GN", False)
        off = blk(g(x.to(dtype)), gn_next=True)
        monkeypatch.setattr(O, "USE_EPILOGUE_GN", True)
        assert getattr(off, "_pgt_gn", None) is None
        check(f"block_gn_on_vs_off_{dtype}", on, off, dtype, 0.5)


@pytest.mark.parametrize("dtype", DTYPES)
def test_sampled_channel_mean_and_mean_field_bias(dtype):
    O = ops()
    for (n, h, w, c, cw) in [(3, 32, 32, 64, 64), (2, 128, 128, 256, 320), (2, 512, 512, 64, 64), (5, 4, 8, 72, 72), (1, 64, 64, 576, 576)]:
        buf = rnd((n, h, w, cw), 11 + c, dtype) + 0.25
        x = buf[..., :c]                                     # a channel slice of a wider buffer when cw > c
        want = E.sampled_channel_mean(x)
        got = O.sampled_channel_mean(g(buf)[..., :c])
        idx = E.sampled_pixels(h * w)
        assert len(idx) == min(h * w, 1024) and len(set(idx.tolist())) == len(idx) and int(idx.max()) < h * w
        check(f"sampled_mean_{n}x{h}x{w}x{c}", got, want, torch.float32, tol_scale=0.05)
        if h * w >= 4096:       # the sample stands for the frame: its mean is close to the full mean
            full = x.float().mean(dim=(1, 2))
            assert (want - full).abs().max() < 0.2
    mean = rnd((7, 200), 5) * 0.3
    dt = rnd((200, 96), 6) * 1e-3
    b = rnd((96,), 7)
    check("mean_field_bias", O.mean_field_bias(g(mean), g(dt), g(b)), mean @ dt + b, torch.float32, tol_scale=0.05)
    check("mean_field_bias_nobias", O.mean_field_bias(g(mean), g(dt)), mean @ dt, torch.float32, tol_scale=0.05)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", ["c64_3x3", "c256_3x3", "cout32", "splitk_32x32", "tokens", "kernel5", "sft"])
def test_conv2d_bias_per_frame(dtype, case):
    """pgt_conv_desc::bias_rows: one bias vector per frame, through every kernel the static selection uses (register-weight
    64-channel kernel, phased LDS-DMA kernel, register-staged tiles, split-K, token rows of a linear, SFT epilogue)."""
    O = ops()
    kw = {}
    if case == "c64_3x3":
        n, h, w, cin, cout, k = 3, 32, 64, 64, 64, 3
    elif case == "c256_3x3":
        n, h, w, cin, cout, k = 4, 32, 32, 256, 256, 3
    elif case == "cout32":
        n, h, w, cin, cout, k = 2, 32, 32, 96, 32, 3
    elif case == "splitk_32x32":
        n, h, w, cin, cout, k = 1, 32, 32, 1024, 128, 3
    elif case == "kernel5":
        n, h, w, cin, cout, k = 2, 32, 32, 128, 256, 3
        kw = {"kernel": 5} if dtype == torch.bfloat16 else {}
    elif case == "sft":
        n, h, w, cin, cout, k = 2, 32, 32, 128, 128, 3
    else:
        n, h, w, cin, cout, k = 1, 1, 6 * 1024, 256, 768, 1
    x = rnd((n, h, w, cin), 21, dtype)
    wt = rnd((cout, k * k * cin), 22, dtype, 0.05)
    frames = 6 if case == "tokens" else n
    bias = rnd((frames, cout), 23)
    pad = (1, 1, 1, 1) if k == 3 else (0, 0, 0, 0)
    if case == "sft":
        dec, sh = rnd((n, h, w, cout), 24, dtype), rnd((n, h, w, cout), 25, dtype)
        want = E.conv2d(x, wt, bias, kh=k, kw=k, pad=pad, sft=(dec, sh, 0.7))
        got = O.conv2d(g(x), g(wt), g(bias), kh=k, kw=k, pad=pad, sft=(g(dec), g(sh), 0.7))
    else:
        res = rnd((n, h, w, cout), 24, dtype)
        want = E.conv2d(x, wt, bias, kh=k, kw=k, pad=pad, act=E.ACT_SILU, res=res)
        got = O.conv2d(g(x), g(wt), g(bias), kh=k, kw=k, pad=pad, act=E.ACT_SILU, res=g(res), **kw)
    check(f"conv_bias_per_frame_{case}", got, want, dtype, tol_scale=2.0)
    # and it is not the shared-bias result (the frames' vectors differ)
    other = O.conv2d(g(x), g(wt), g(bias[0].contiguous()), kh=k, kw=k, pad=pad)
    assert frames == 1 or not torch.equal(other[-1], got[-1])
    if case == "c64_3x3":       # frames that are not whole 512-row tiles are refused, not mis-indexed
        with pytest.raises(Exception):
            O.conv2d(g(x[:, :8, :16]), g(wt), g(bias), kh=k, kw=k, pad=pad)


def test_compensated_half_conv_is_closer_to_fp32():
    """What the compensation is for: a half conv on inputs with a channel mean (post-SiLU activations) - the frame-mean
    of its error against the fp32 conv drops by an order of magnitude with the per-frame bias of _frame_bias."""
    from pgtformer_amd.modules.rstt_layers import Conv2d, prepare_tree
    torch.manual_seed(3)
    conv = Conv2d(128, 128, 3, padding=1)
    x32 = torch.nn.functional.silu(rnd((2, 64, 64, 128), 31))
    want = E.conv2d(x32, E.pack_conv_weight(conv.weight.detach(), torch.float32), conv.bias.detach(), kh=3, kw=3, pad=(1, 1, 1, 1))
    xh = g(x32.to(torch.float16))
    want_h = E.conv2d(xh.cpu().float(), E.pack_conv_weight(conv.weight.detach(), torch.float32), conv.bias.detach(), kh=3, kw=3, pad=(1, 1, 1, 1))
    errs = {}
    import pgtformer_amd.modules.rstt_layers as R
    for on in (False, True):
        old = R.USE_WCOMP
        R.USE_WCOMP = on
        try:
            prepare_tree(conv, torch.device(DEV), torch.float16)
            y = conv.run(xh).float().cpu()
        finally:
            R.USE_WCOMP = old
        e = (y - want_h)[:, 4:-4, 4:-4]                       # away from the zero-padded border
        errs[on] = float(e.mean(dim=(1, 2)).abs().mean())       # per-(frame, channel) mean error = the bias part
    _LOG.append({"name": "compensated_half_conv_bias_error", "off": errs[False], "on": errs[True]})
    assert errs[True] < 0.25 * errs[False], errs


def test_linear_rows_beyond_2gib():
    """A token matrix of 2 GiB or more (a 48-window batch at 128x128: 2.4 M rows of split channels) runs as equal row chunks -
    the kernels take 32-bit byte offsets - with a per-frame bias following its frames; rows sampled across every chunk are
    compared with the emulation (a linear is row-independent)."""
    O = ops()
    rows, cin, cout, frames = 3 * (1 << 20), 512, 64, 192           # 3 GiB of half; 16384 rows per frame
    x = torch.empty((rows, cin), dtype=torch.float16, device=DEV).normal_(generator=torch.Generator(device=DEV).manual_seed(5))
    w = rnd((cout, cin), 41, torch.float16, 0.05)
    bias = rnd((frames, cout), 42)
    res = torch.empty((rows, cout), dtype=torch.float16, device=DEV).normal_(generator=torch.Generator(device=DEV).manual_seed(6))
    got = O.linear(x, g(w), g(bias), act=E.ACT_GELU, res=res)
    pick = torch.cat([torch.arange(0, 64), torch.arange(rows // 2 - 32, rows // 2 + 32), torch.arange(rows - 64, rows),
                      torch.arange(0, rows, 99991)])
    fr = pick // (rows // frames)
    want = E.linear(x[pick.to(DEV)].cpu(), w, None, act=E.ACT_NONE).float() + bias[fr]
    want = E._act(want, E.ACT_GELU) + res[pick.to(DEV)].cpu().float()
    check("linear_3gib_rows", got[pick.to(DEV)], want.to(torch.float16), torch.float16)


@pytest.mark.parametrize("dtype", H16)
@pytest.mark.parametrize("case", ["ragged_256", "qkv_768", "views_resid", "tokens_frames"])
def test_linear_k256_streaming_kernel(dtype, case):
    """igemm7 (K = 256 linears on many rows: weights in registers, rows through a double-buffered LDS image, fp32 stage):
    ragged row counts, 768 output columns, strided operands with residual and GELU, a per-frame bias - forced (kernel=7)
    and, from 65 536 rows up, as the library's own choice; against the emulation and the phased kernel (kernel=4)."""
    O = ops()
    cout, rows, frames = 256, 4096 + 17, 0
    if case == "qkv_768":
        cout, rows = 768, 2048 + 31
    elif case == "tokens_frames":
        rows, frames = 131072, 8
    x = rnd((rows, 256), 51, dtype)
    w = rnd((cout, 256), 52, dtype, 0.06)
    bias = rnd((frames, cout), 53) if frames else rnd((cout,), 53)
    kw = {}
    want_kw = {}
    if case == "views_resid":
        wide = rnd((rows, 640), 54, dtype)
        x = wide[:, 128:384]                               # a channel slice: ldx = 640
        resw = rnd((rows, 512), 55, dtype)
        kw = dict(act=E.ACT_GELU, res=resw[:, 256:])
    elif case == "tokens_frames":
        kw = dict(act=E.ACT_GELU, res=rnd((rows, cout), 55, dtype))
    xg = g(wide)[:, 128:384] if case == "views_resid" else g(x)
    kwg = {k: (g(resw)[:, 256:] if (k == "res" and case == "views_resid") else (g(v) if torch.is_tensor(v) else v)) for k, v in kw.items()}
    want = E.linear(x, w, bias, **kw)
    x4 = xg.as_strided((1, 1, rows, 256), (0, 0, xg.stride(0), 1))
    r4 = None if "res" not in kwg else kwg["res"].as_strided((1, 1, rows, cout), (0, 0, kwg["res"].stride(0), 1))
    for kern in ((7, 4, 0) if rows >= 65536 else (7, 4)):
        got = O.conv2d(x4, g(w), g(bias), act=kw.get("act", E.ACT_NONE), res=r4, kernel=kern)
        check(f"linear_k256_{case}_k{kern}", got.reshape(rows, cout), want, dtype)
        if kern == 7:
            ref7 = got.clone()
    assert torch.equal(got.reshape(rows, cout), ref7.reshape(rows, cout)) or rows < 65536     # kernel 0 == kernel 7 from 65 536 rows


@pytest.mark.parametrize("dtype", H16, ids=["bf16", "f16"])
def test_weight_defect_behind_the_abi(dtype):
    """pgt_weight_defect (the D operand of the mean-field compensation) against the host restatement: a 3x3 conv with a
    BatchNorm-style output scale and padded input channels, a Linear, and the one-row-per-tap form."""
    O = ops()
    for (cout, cin, k, cp, scaled, sum_taps) in [(64, 57, 3, 64, True, True), (256, 256, 1, 256, False, True), (32, 192, 3, 192, False, False)]:
        w = rnd((cout, cin, k, k), 90 + cout, torch.float32, 0.05) if k > 1 else rnd((cout, cin), 90 + cout, torch.float32, 0.05)
        scale = (1 + 0.2 * rnd((cout,), 91)) if scaled else None
        pw = E.pack_conv_weight(w, dtype, cin_pad=cp, scale=scale)
        want = E.weight_defect(w, pw, scale=scale, sum_taps=sum_taps)
        got = O.weight_defect(g(w), g(pw), scale=g(scale), sum_taps=sum_taps)
        check(f"weight_defect_{cout}x{cin}x{k}", got, want, torch.float32, tol_scale=1e-3)
        assert want.abs().max() > 0
