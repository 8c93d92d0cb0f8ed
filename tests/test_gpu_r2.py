"""GPU parity tests of the round-2 rows: the Video-Swin window attention of modules/swin.py (BASELINE.json configs[4],
incl. windows / shift along the depth axis and fp16 storage), the VectorQuantizer look-up of archs/vqgan_arch.py, the
fused nearest-code kernel (arg-min inside the distance GEMM) and the remaining stage-I API of archs/tdcrqvae3_arch.py
(commitment loss, straight-through, soft codes, decode / decode_code / get_codes) - against the reference goldens of
tests/golden/r2_golden.npz and the fp32 emulation.

Tolerances: integer results bit-exact (codes; except tokens whose two nearest codes are closer than 1e-5 in distance when
the operand precision differs from the reference's fp32); fp32 paths 1e-3 * max|ref|; bf16 4e-2, fp16 6e-3 * max|ref|."""
import json
import os

import numpy as np
import pytest
import torch

from tests import emu_ops as E
from tests.golden import cases_r2 as C2

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
DEV = "cuda"
_LOG = {}


@pytest.fixture(scope="module", autouse=True)
def _dump_log():
    yield
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/parity_r2.json", "w") as f:
        json.dump(_LOG, f, indent=1)


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLD, "r2_golden.npz"))


def ops():
    import pgtformer_amd.ops as O
    return O


def rnd(shape, seed, scale=1.0):
    g = np.random.default_rng(seed)
    return torch.from_numpy((scale * g.standard_normal(shape)).astype(np.float32))


WA3 = [(1, 3, 16, 16, 512, (3, 8, 8), (0, 4, 4)), (1, 4, 8, 12, 256, (2, 4, 6), (1, 2, 3)), (2, 4, 8, 12, 256, (2, 4, 6), (1, 0, 3)),
       (1, 6, 8, 8, 512, (3, 4, 4), (1, 2, 2)), (1, 3, 8, 8, 256, (3, 4, 4), (0, 0, 0))]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("case", WA3)
def test_window_attention3d_vs_emulation(dtype, case):
    b, d, h, w, c, win, shift = case
    heads = 8
    n = win[0] * win[1] * win[2]
    qkv = rnd((b * d * h * w, 3 * c), 60).to(dtype)
    bias = rnd((heads, n, n), 61, 0.5)
    want = E.window_attention3d(qkv, bias, b, d, h, w, c, heads, win, shift).float()
    got = ops().window_attention3d(qkv.to(DEV), bias.to(DEV), b, d, h, w, c, heads, win, shift).float().cpu()
    err = (got - want).abs().max().item()
    tol = (4e-2 if dtype == torch.bfloat16 else 6e-3) * max(1.0, want.abs().max().item())
    _LOG[f"wa3d/{dtype}/{case}"] = {"max_abs_err": err, "tol": tol}
    assert err <= tol, (case, dtype, err)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("name", list(C2.SWIN))
def test_swin_block_part1_matches_reference_golden(gold, name, dtype):
    """SwinTransformerBlock3D.forward_part1 (modules/swin.py:212-246) = LayerNorm -> qkv Linear -> window attention (roll,
    partition, bias, 27-region mask: ONE kernel) -> proj Linear.  The projections run in exact fp32 so that the reduced
    precision is that of the attention kernel alone (its qkv input / output are rounded to bf16 / fp16)."""
    dim, heads, ws, ss, fmap, qkv_bias, seed = C2.SWIN[name]
    p = {k: v.to(DEV) for k, v in C2.swin_params(name).items()}
    x = C2.swin_input(name).to(DEV)
    b, d, h, w = fmap
    rows = b * d * h * w
    O = ops()
    ln = O.layernorm(x.reshape(rows, dim), p["norm1.weight"], p["norm1.bias"], 1e-5)
    qkv = O.linear(ln, p["attn.qkv.weight"].contiguous(), p.get("attn.qkv.bias"))
    from oracle import pgt_oracle as ORA
    n = ws[0] * ws[1] * ws[2]
    idx = ORA.swin_relative_position_index(ws).reshape(-1).to(DEV)
    bias = p["attn.relative_position_bias_table"][idx].reshape(n, n, heads).permute(2, 0, 1).contiguous()
    ao = O.window_attention3d(qkv.to(dtype), bias, b, d, h, w, dim, heads, ws, ss)
    y = O.linear(ao.float(), p["attn.proj.weight"].contiguous(), p["attn.proj.bias"]).reshape(b, d, h, w, dim)
    ref = torch.from_numpy(gold[f"{name}.out"])
    err = (y[..., :128].cpu() - ref).abs().max().item()
    tol = (4e-2 if dtype == torch.bfloat16 else 6e-3) * max(1.0, ref.abs().max().item())
    _LOG[f"swin/{name}/{dtype}"] = {"max_abs_err": err, "tol": tol, "ref_absmax": ref.abs().max().item()}
    assert err <= tol, (name, dtype, err)


@pytest.mark.parametrize("name", list(C2.VQ))
def test_vector_quantizer_matches_reference_golden(gold, name):
    from pgtformer_amd.archs.vqgan_arch import VectorQuantizer

    k, c, zshape, seed = C2.VQ[name]
    w, z = C2.vq_case(name)
    vq = VectorQuantizer(k, c, 0.25)
    vq.embedding.weight.data.copy_(w)
    vq.prepare(DEV, torch.float32)
    zq, loss, idx = vq.forward_nhwc(z.permute(0, 2, 3, 1).contiguous().to(DEV))
    ref_idx = gold[f"{name}.indices"].reshape(-1)
    agree = float((idx.cpu().numpy() == ref_idx).mean())
    _LOG[f"vq/{name}"] = {"index_agreement": agree, "loss": float(loss.item()), "ref_loss": float(gold[f"{name}.loss"][0])}
    assert agree == 1.0
    assert (zq.cpu().permute(0, 3, 1, 2) - torch.from_numpy(gold[f"{name}.z_q"])).abs().max().item() <= 1e-6
    assert abs(loss.item() - gold[f"{name}.loss"][0]) <= 1e-5 * max(1e-6, gold[f"{name}.loss"][0]) + 1e-9
    if name.endswith("ties"):
        assert int(idx.cpu().reshape(1, 6, 5)[0, 1, 2]) == 5


@pytest.mark.parametrize("shape", [(3072, 1024, 512), (1000, 1024, 512), (5000, 1024, 512), (700, 1000, 512), (300, 96, 512), (70, 1000, 256),
                                   (4096, 512, 64), (333, 1025, 128)])
def test_fused_nearest_code_equals_two_kernel_form(shape):
    """arg-min inside the distance GEMM == distance GEMM (fp32 out) + row arg-min on the same bf16 operands: same dot
    products (same k order), same association, same first-index tie rule -> identical codes."""
    rows, k, d = shape
    x = rnd((rows, d), 80, 0.3)
    book = rnd((k, d), 81)
    book[77] = book[5]                               # duplicated code vector: argmin must return 5
    x[9] = book[77] + 1e-3
    x[11] = book[k - 1]                              # the last code (ragged K: k % 32 != 0 for some shapes)
    xd, bd = x.to(DEV).to(torch.bfloat16), book.to(DEV).to(torch.bfloat16)
    en = bd.float().pow(2.0).sum(1).contiguous()
    O = ops()
    xn = O.row_sumsq(xd)
    two = O.rq_argmin(O.linear(xd, bd, None, out_f32=True), xn, en).cpu()
    fused = O.rq_nearest(xd, bd, xn, en).cpu()
    _LOG[f"rq_fused/{shape}"] = {"agree": float((two == fused).float().mean())}
    assert torch.equal(two, fused)
    assert int(fused[9]) == 5 and int(fused[11]) == k - 1
    want = E.rq_argmin(E.linear(xd.cpu(), bd.cpu(), None, out_f32=True), xd.cpu().float().pow(2).sum(1), en.cpu())
    assert (fused == want).float().mean().item() >= 0.995


def test_stage1_api_matches_reference_golden(gold, cfg, full_sd, golden_window):
    """TDCRQVAE3.forward(code_only) (z_q = x + (q - x), commitment loss, codes), get_codes, get_codesbt, get_soft_codes,
    decode_code (reference: archs/tdcrqvae3_arch.py:760-813, :330-352, :429-457) in fp32."""
    from pgtformer_amd import PGTFormer
    from pgtformer_amd.archs.tdcrqvae3_arch import TDCRQVAE3

    m = PGTFormer(**cfg)
    m.load_state_dict(full_sd, strict=True)
    m.prepare(DEV, "fp32")
    x, _, _ = golden_window
    xd = x.to(DEV)
    z_q, loss, codes = TDCRQVAE3.forward(m, xd, code_only=True)
    ref_codes = gold["stage1.codes"]
    assert np.array_equal(codes.cpu().numpy().astype(np.int16), ref_codes)
    assert np.array_equal(TDCRQVAE3.get_codes(m, xd).cpu().numpy().astype(np.int16), ref_codes)
    assert np.array_equal(TDCRQVAE3.get_codesbt(m, xd.reshape(1, 3, 3, 512, 512)).cpu().numpy().astype(np.int16), ref_codes)
    rec = {"loss": float(loss.item()), "ref_loss": float(gold["stage1.loss"][0]),
           "z_q_err": float(np.abs(z_q.float().cpu()[:, 12:20, 12:20, :64].numpy() - gold["stage1.z_q_crop"]).max())}
    soft, scode = TDCRQVAE3.get_soft_codes(m, xd, temp=0.5)
    assert soft.shape == (3, 32, 32, 1, 1024) and np.array_equal(scode.cpu().numpy().astype(np.int16), ref_codes)
    rec["soft_err"] = float(np.abs(soft.cpu()[:, :2, :2].numpy() - gold["stage1.soft_tok"]).max())
    rec["soft_max_err"] = float(np.abs(soft.max(-1).values.cpu().numpy() - gold["stage1.soft_max"]).max())
    dec = TDCRQVAE3.decode_code(m, torch.from_numpy(ref_codes.astype(np.int64)))
    rec["decode_code_err"] = float(np.abs(dec[1, :, 192:320, 192:320].cpu().numpy() - gold["stage1.decode_code_mid_crop"]).max())
    out, loss2, _ = TDCRQVAE3.forward(m, xd)
    full = np.load(os.path.join(GOLD, "full_golden.npz"))
    rec["forward_out_err"] = float(np.abs(out[1, :, 192:320, 192:320].cpu().numpy() - full["stage1_out_mid_crop"]).max())
    rec["forward_loss_vs_full_golden"] = abs(float(loss2.item()) - float(full["stage1_loss"][0]))
    _LOG["stage1_api/fp32"] = rec
    assert abs(rec["loss"] - rec["ref_loss"]) <= 1e-4 * max(1e-6, rec["ref_loss"])
    assert rec["forward_loss_vs_full_golden"] <= 1e-4 * max(1e-6, float(full["stage1_loss"][0]))
    assert rec["z_q_err"] <= 1e-4 and rec["soft_err"] <= 1e-4 and rec["soft_max_err"] <= 1e-4
    assert rec["decode_code_err"] <= 2e-3 and rec["forward_out_err"] <= 2e-3
    # stochastic soft codes (reference: torch.multinomial(soft_code, 1), :443-446): same soft codes, codes drawn from them -
    # reproducible for a seeded generator, and at temp -> 0 the draw collapses onto the nearest code
    gen = torch.Generator(device=DEV)
    gen.manual_seed(7)
    soft_s, code_a = TDCRQVAE3.get_soft_codes(m, xd, temp=0.5, stochastic=True, generator=gen)
    gen.manual_seed(7)
    _, code_b = TDCRQVAE3.get_soft_codes(m, xd, temp=0.5, stochastic=True, generator=gen)
    assert torch.equal(soft_s, soft) and torch.equal(code_a, code_b) and code_a.shape == scode.shape
    p_drawn = soft_s.reshape(-1, 1024).gather(1, code_a.reshape(-1, 1)).reshape(-1)
    assert float(p_drawn.min()) > 0.0
    _, code_cold = TDCRQVAE3.get_soft_codes(m, xd, temp=1e-4, stochastic=True, generator=gen)
    assert np.array_equal(code_cold.cpu().numpy().astype(np.int16), ref_codes)


def test_sample_rows_is_an_inverse_cdf_draw():
    """pgt_sample_rows: the drawn index brackets u * total in the cumulative sums, ties and zero-probability entries are
    never drawn, and the empirical frequencies follow the probabilities."""
    from tests import emu_ops as E
    from pgtformer_amd import ops as O

    g_ = torch.Generator().manual_seed(3)
    prob = torch.rand((4096, 1024), generator=g_) ** 8
    prob[:, 100:200] = 0.0
    prob = prob / prob.sum(-1, keepdim=True)
    u = torch.rand((4096,), generator=g_)
    got = O.sample_rows(prob.to(DEV), u.to(DEV)).cpu().long()
    c = prob.double().cumsum(-1)
    t = u.double() * c[:, -1]
    lo = torch.where(got > 0, c.gather(1, (got - 1).clamp_min(0).unsqueeze(1)).squeeze(1), torch.zeros_like(t))
    hi = c.gather(1, got.unsqueeze(1)).squeeze(1)
    assert bool(((lo <= t + 1e-6) & (hi >= t - 1e-6)).all())
    assert not bool(((got >= 100) & (got < 200)).any())
    assert float((got == E.sample_rows(prob, u).long()).float().mean()) > 0.995       # differs only by fp32 summation order
    one = torch.tensor([0.5, 0.25, 0.125, 0.125] + [0.0] * 60).repeat(20000, 1)
    d = O.sample_rows(one.to(DEV), torch.rand((20000,), generator=g_).to(DEV)).cpu()
    f = torch.bincount(d.long(), minlength=64).float() / 20000
    assert float((f[:4] - one[0, :4]).abs().max()) < 0.012 and float(f[4:].sum()) == 0.0
    edge = O.sample_rows(one[:2].to(DEV), torch.tensor([0.0, 0.99999994]).to(DEV)).cpu().tolist()
    assert edge == [0, 3]


def test_stage1_bf16x3_codes_match_reference(gold, cfg, full_sd, golden_window):
    """The default mode's stage-I path: split-half encoder -> fp32 quantiser: codes equal the reference's."""
    from pgtformer_amd import PGTFormer
    from pgtformer_amd.archs.tdcrqvae3_arch import TDCRQVAE3

    m = PGTFormer(**cfg)
    m.load_state_dict(full_sd, strict=True)
    m.prepare(DEV, "bf16x3")
    x, _, _ = golden_window
    codes = TDCRQVAE3.get_codes(m, x.to(DEV)).cpu().numpy().astype(np.int16)
    agree = float((codes == gold["stage1.codes"]).mean())
    _LOG["stage1_api/bf16x3"] = {"code_agreement": agree}
    assert agree >= 0.999
