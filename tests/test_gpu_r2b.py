"""GPU parity tests of SURVEY §8 row (f)4: the training-side quantiser (VQEmbedding EMA codebook update, csrc/vq_ema.hip)
and the Video-Swin BasicLayer stage (pgtformer_amd/modules/swin.py) - against the reference goldens of
tests/golden/r2b_golden.npz and the oracle.

Tolerances: codes and counts bit-exact; fp32 state 2e-6 * max|ref| (the reference forms the per-code sums with a GEMM whose
summation order is unspecified; the kernel adds in row order); BasicLayer bf16 2e-2, fp32 with fp16 / bf16 attention 1e-3 / 6e-3 * max|ref|."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import pgt_oracle as ORA
from tests import emu_ops as E
from tests.golden import cases_r2b as C

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
DEV = "cuda"
_LOG = {}


@pytest.fixture(scope="module", autouse=True)
def _dump_log():
    yield
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/parity_r2b.json", "w") as f:
        json.dump(_LOG, f, indent=1)


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLD, "r2b_golden.npz"))


def ops():
    import pgtformer_amd.ops as O
    return O


def _rel(a, ref):
    ref = torch.as_tensor(ref).float()
    return (torch.as_tensor(a).float().cpu() - ref).abs().max().item() / max(1e-6, ref.abs().max().item())


@pytest.mark.parametrize("shape", [(1200, 256, 64), (300, 512, 32), (5000, 1024, 512), (777, 100, 1024), (64, 8, 2048)])
def test_cluster_statistics_kernel(shape):
    """per-code vector sums (row order) and counts == the emulation; the same bits on every run (no float atomics)"""
    rows, k, d = shape
    g = torch.Generator().manual_seed(rows + k)
    x = torch.randn((rows, d), generator=g)
    codes = torch.randint(0, k, (rows,), generator=g, dtype=torch.int32)
    codes[:7] = k - 1
    want = E.vq_cluster_stats(x, codes, k)
    O = ops()
    got = O.vq_cluster_stats(x.to(DEV), codes.to(DEV), k)
    again = O.vq_cluster_stats(x.to(DEV), codes.to(DEV), k)
    assert torch.equal(got, again)
    got = got.cpu()
    assert torch.equal(got[k * d:], want[k * d:])                                    # counts: exact
    err = _rel(got[:k * d], want[:k * d])
    _LOG[f"cluster_stats/{shape}"] = {"rel_err": err}
    assert err <= 2e-6
    # sequential fp32 addition in row order: bit-equal to a python loop for one code
    j = int(codes[10])
    acc = torch.zeros(d)
    for r in (codes == j).nonzero().flatten().tolist():
        acc = acc + x[r]
    assert torch.equal(got[j * d:(j + 1) * d], acc)


@pytest.mark.parametrize("name", list(C.EMA))
def test_vqembedding_training_steps_match_reference(gold, name):
    """VQEmbedding.forward in training mode on the GPU, consecutive steps: codes, embeds (from the codebook BEFORE the update),
    then cluster_size_ema / embed_ema / codebook after EMA, restart of dead codes and renormalisation
    (reference: tdcrqvae3_arch.py:128-199), with the reference's permutation / noise draws from the fixture."""
    from pgtformer_amd.archs.tdcrqvae3_arch import VQEmbedding

    k, d, n, decay, restart, steps, seed = C.EMA[name]
    w, batches = C.ema_case(name)
    vq = VQEmbedding(k, d, decay=decay, restart_unused_codes=restart)
    with torch.no_grad():
        vq.weight.copy_(w)
        vq.embed_ema.copy_(w[:-1])
    vq.prepare(DEV, torch.float32)
    vq.train()
    rec = {}
    for s, x in enumerate(batches):
        old = vq.book.clone()
        perm = torch.from_numpy(gold[f"{name}.{s}.perm"]).long() if restart else None
        noise = torch.from_numpy(gold[f"{name}.{s}.noise"]) if f"{name}.{s}.noise" in gold else None
        emb, idx = vq(x.to(DEV), perm=perm, noise=noise)
        assert np.array_equal(idx.cpu().numpy(), gold[f"{name}.{s}.idxs"])
        assert torch.equal(emb, old[idx.long()])
        rec[s] = {"weight": _rel(vq.book, gold[f"{name}.{s}.weight"]),
                  "cluster_size_ema": _rel(vq.cs_ema_d, gold[f"{name}.{s}.cluster_size_ema"]),
                  "embed_ema": _rel(vq.embed_ema_d, gold[f"{name}.{s}.embed_ema"])}
        assert max(rec[s].values()) <= 2e-6, rec
        assert torch.equal(vq.book[-1].cpu(), torch.zeros(d))
        assert _rel(vq.enorm, vq.book[:-1].pow(2).sum(1).cpu()) <= 1e-6
    _LOG[f"ema/{name}"] = rec
    sd = vq.state_dict()
    assert sd["weight"].device.type == "cpu" and _rel(sd["weight"], gold[f"{name}.{steps - 1}.weight"]) <= 2e-6
    assert _rel(sd["embed_ema"], gold[f"{name}.{steps - 1}.embed_ema"]) <= 2e-6


def test_rq_bottleneck_trains_its_codebook_only_in_training_mode():
    """RQBottleneck.forward: eval leaves the codebook alone; train() folds the batch in (shared codebook, depth 2: two
    updates per call, statistics taken on the residual the search saw)."""
    from pgtformer_amd.archs.tdcrqvae3_arch import RQBottleneck

    torch.manual_seed(3)
    rq = RQBottleneck([8, 8, 64], [8, 8, 2], 128, shared_codebook=True)
    with torch.no_grad():
        rq.codebooks[0].weight[:-1].normal_(0, 0.05)
        rq.codebooks[0].embed_ema.copy_(rq.codebooks[0].weight[:-1])
    rq.prepare(DEV, torch.float32)
    x = (0.05 * torch.randn(2, 8, 8, 64)).to(DEV)
    book = rq.codebooks[0]
    w0 = book.book.clone()
    q0, l0, c0 = rq(x)
    assert torch.equal(w0, book.book)
    rq.train()
    q1, l1, c1 = rq(x)
    assert torch.equal(c0[..., 0], c1[..., 0]) and not torch.equal(w0, book.book)
    # oracle: two EMA steps on (x, residual) with the codes the build found
    sd_w = w0.cpu()
    cs, em = torch.zeros(128), sd_w[:-1].clone()
    book2 = RQBottleneck([8, 8, 64], [8, 8, 2], 128, shared_codebook=True).codebooks[0]   # for decay / eps defaults
    x2 = x.cpu().reshape(-1, 64)
    r = x2.clone()
    wcur = sd_w
    for i in range(2):
        idx = c1[..., i].reshape(-1).cpu().long()
        # restart draws are random: compare the codes that stayed alive only
        neww, cs, em = ORA.vq_ema_step(wcur, cs, em, r, idx, book2.decay, book2.eps, False)
        r = r - wcur[idx]
        wcur = neww
    alive = (book.cs_ema_d.cpu() != 1) & (cs >= 1)
    if int(alive.sum()):
        assert _rel(book.embed_ema_d.cpu()[alive], em[alive]) <= 1e-5


@pytest.mark.parametrize("mode", ["bf16", "fp32_fp16attn", "fp32_bf16attn"])
@pytest.mark.parametrize("name", list(C.LAYER))
def test_swin_basic_layer_matches_reference_golden(gold, name, mode):
    """BasicLayer.forward (modules/swin.py:389-409): depth blocks, alternating shift, windows clamped to the feature map,
    shortcut / MLP residual adds as GEMM epilogues - reference signature (B, C, D, H, W) in and out."""
    from pgtformer_amd.modules.swin import BasicLayer

    dim, depth, heads, ws, fmap, mlp_ratio, qkv_bias, seed = C.LAYER[name]
    layer = BasicLayer(dim, depth, heads, window_size=ws, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias)
    layer.load_state_dict({**layer.state_dict(), **C.layer_params(name)}, strict=True)
    layer.prepare(DEV, torch.bfloat16 if mode == "bf16" else torch.float32)
    if mode == "fp32_bf16attn":
        for blk in layer.blocks:
            blk.attn_dtype = torch.bfloat16
    y = layer(C.layer_input(name).to(DEV)).float().cpu()
    ref = torch.from_numpy(gold[f"{name}.out"])
    err = (y[:, :C.KEEP[name]] - ref).abs().max().item()
    tol = {"bf16": 2e-2, "fp32_fp16attn": 1e-3, "fp32_bf16attn": 6e-3}[mode] * max(1.0, ref.abs().max().item())
    _LOG[f"swin_layer/{name}/{mode}"] = {"max_abs_err": err, "tol": tol, "ref_absmax": ref.abs().max().item()}
    assert y.shape == tuple(C.layer_input(name).shape) and err <= tol, (name, mode, err)
    want = ORA.swin_basic_layer(C.layer_params(name), C.layer_input(name), depth, heads, ws)   # all channels, via the oracle
    assert (y - want).abs().max().item() <= tol


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("case", [(1, 3, 32, 32, 512, 8, (3, 5, 5), (0, 2, 2), False), (2, 3, 7, 8, 64, 4, (2, 3, 3), (1, 1, 1), True),
                                  (1, 4, 8, 8, 128, 4, (2, 4, 4), (1, 2, 2), False), (1, 5, 5, 6, 256, 8, (5, 5, 6), (0, 0, 0), True),
                                  (1, 2, 9, 9, 64, 4, (2, 4, 4), (0, 2, 2), True)])
def test_window_attention3d_general_kernel(case, dtype):
    """pgt_window_attention3d outside the MFMA kernel's shapes: windows whose token count is not a multiple of 48, feature
    maps that are not multiples of the window (padded at the far end, padding tokens carry `pad_row` = the qkv bias), fp32 /
    bf16 / fp16 storage - against the independent emulation."""
    b, d, h, w, c, heads, win, shift, with_pad_row = case
    g = torch.Generator().manual_seed(c + h)
    n = win[0] * win[1] * win[2]
    qkv = (0.5 * torch.randn((b * d * h * w, 3 * c), generator=g)).to(dtype)
    bias = 0.3 * torch.randn((heads, n, n), generator=g)
    pad = (0.2 * torch.randn((3 * c,), generator=g)).to(dtype) if with_pad_row else None
    want = E.window_attention3d(qkv, bias, b, d, h, w, c, heads, win, shift, pad).float()
    got = ops().window_attention3d(qkv.to(DEV), bias.to(DEV), b, d, h, w, c, heads, win, shift,
                                   None if pad is None else pad.to(DEV)).float().cpu()
    err = (got - want).abs().max().item()
    tol = {torch.float32: 2e-5, torch.bfloat16: 2e-2, torch.float16: 3e-3}[dtype] * max(1.0, want.abs().max().item())
    _LOG[f"wa3d_general/{dtype}/{case}"] = {"max_abs_err": err, "tol": tol}
    assert err <= tol, (case, dtype, err)
