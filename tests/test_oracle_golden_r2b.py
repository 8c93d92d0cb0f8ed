"""SURVEY §8 row (f)4 on CPU: the oracle restatements of the training-side quantiser (VQEmbedding EMA codebook update,
archs/tdcrqvae3_arch.py:128-199) and of the Video-Swin BasicLayer (modules/swin.py:326-409) against fixtures produced by the
imported reference (tests/golden/make_golden_r2b.py); the product's host logic for both through the CPU emulation of the
operators; the data-parallel EMA step (one all-reduce + one broadcast) with world_size 2 over gloo."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import pgt_oracle as O
from tests import emu_ops
from tests.golden import cases_r2b as C

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLD, "r2b_golden.npz"))


def _close(a, ref, rtol):
    ref = torch.as_tensor(ref)
    return (torch.as_tensor(a).float() - ref).abs().max().item() <= rtol * max(1e-6, ref.abs().max().item())


@pytest.mark.parametrize("name", list(C.EMA))
def test_ema_codebook_update_oracle_matches_reference(gold, name):
    k, d, n, decay, restart, steps, seed = C.EMA[name]
    w, batches = C.ema_case(name)
    cs, em = torch.zeros(k), w[:-1].clone()
    for s, x in enumerate(batches):
        idx = O._distances({"quantizer.codebooks.0.weight": w}, x, 0).argmin(-1)
        assert np.array_equal(idx.numpy().astype(np.int32), gold[f"{name}.{s}.idxs"])
        perm = torch.from_numpy(gold[f"{name}.{s}.perm"]).long() if restart else None
        noise = torch.from_numpy(gold[f"{name}.{s}.noise"]) if f"{name}.{s}.noise" in gold else None
        w, cs, em = O.vq_ema_step(w, cs, em, x, idx, decay, 1e-5, restart, perm, noise)
        assert _close(w, gold[f"{name}.{s}.weight"], 1e-6)
        assert _close(cs, gold[f"{name}.{s}.cluster_size_ema"], 1e-6)
        assert _close(em, gold[f"{name}.{s}.embed_ema"], 1e-6)
        assert torch.equal(w[-1], torch.zeros(d))                       # the padding row is never updated
    if restart:
        assert int((cs == 1).sum()) > 0                                  # the cases do restart codes


@pytest.mark.parametrize("name", list(C.LAYER))
def test_swin_basic_layer_oracle_matches_reference(gold, name):
    dim, depth, heads, ws, fmap, mlp_ratio, qkv_bias, seed = C.LAYER[name]
    y = O.swin_basic_layer(C.layer_params(name), C.layer_input(name), depth, heads, ws)
    assert _close(y[:, :C.KEEP[name]], gold[f"{name}.out"], 2e-6)


@pytest.mark.parametrize("name", list(C.EMA))
def test_vqembedding_training_step_host_logic(gold, name, monkeypatch):
    """VQEmbedding.forward in training mode (search, statistics, gather with the OLD codebook, EMA + restart + renormalise,
    refreshed search operands) through the CPU emulation of the operators, step by step against the reference's states."""
    from pgtformer_amd.archs.tdcrqvae3_arch import VQEmbedding

    emu_ops.install(monkeypatch)
    k, d, n, decay, restart, steps, seed = C.EMA[name]
    w, batches = C.ema_case(name)
    vq = VQEmbedding(k, d, decay=decay, restart_unused_codes=restart)
    assert not vq.training                                           # inference-first: the EMA path is opt-in
    with torch.no_grad():
        vq.weight.copy_(w)
        vq.embed_ema.copy_(w[:-1])
    vq.prepare("cpu", torch.float32)
    vq.train()
    for s, x in enumerate(batches):
        old = vq.book.clone()
        perm = torch.from_numpy(gold[f"{name}.{s}.perm"]).long() if restart else None
        noise = torch.from_numpy(gold[f"{name}.{s}.noise"]) if f"{name}.{s}.noise" in gold else None
        emb, idx = vq(x, perm=perm, noise=noise)
        assert np.array_equal(idx.numpy(), gold[f"{name}.{s}.idxs"])
        assert torch.equal(emb, old[idx.long()])                     # embeds come from the codebook BEFORE the update
        assert _close(vq.book, gold[f"{name}.{s}.weight"], 2e-6)
        assert _close(vq.cs_ema_d, gold[f"{name}.{s}.cluster_size_ema"], 2e-6)
        assert _close(vq.embed_ema_d, gold[f"{name}.{s}.embed_ema"], 2e-6)
        assert _close(vq.enorm, vq.book[:-1].pow(2).sum(1), 1e-6) and torch.equal(vq.book_t, vq.book[:-1])
    sd = vq.state_dict()                                             # checkpoints see the trained state
    assert _close(sd["weight"], gold[f"{name}.{steps - 1}.weight"], 2e-6)
    assert _close(sd["cluster_size_ema"], gold[f"{name}.{steps - 1}.cluster_size_ema"], 2e-6)
    vq.eval()
    before = vq.book.clone()
    vq(batches[0])
    assert torch.equal(before, vq.book)                              # eval mode: no update


@pytest.mark.parametrize("name", list(C.LAYER))
def test_swin_basic_layer_host_logic(gold, name, monkeypatch):
    """pgtformer_amd.modules.swin.BasicLayer (state-dict keys of the reference, window clamp, alternating shift, epilogue
    residual adds) through the CPU emulation of the operators."""
    from pgtformer_amd.modules.swin import BasicLayer

    emu_ops.install(monkeypatch)
    dim, depth, heads, ws, fmap, mlp_ratio, qkv_bias, seed = C.LAYER[name]
    layer = BasicLayer(dim, depth, heads, window_size=ws, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias)
    p = C.layer_params(name)
    sd = layer.state_dict()
    assert set(p) | {f"blocks.{i}.attn.relative_position_index" for i in range(depth)} == set(sd)
    for kk in sd:
        if kk.endswith("relative_position_index"):
            assert torch.equal(sd[kk], O.swin_relative_position_index(ws))
    layer.load_state_dict({**sd, **p}, strict=True)
    layer.prepare("cpu", torch.float32)
    for blk in layer.blocks:
        assert blk.attn_dtype == torch.float16                       # attention storage type of the fp32 mode
    y = layer(C.layer_input(name))
    ref = torch.from_numpy(gold[f"{name}.out"])
    err = (y[:, :C.KEEP[name]] - ref).abs().max().item()
    assert err <= 6e-3 * max(1.0, ref.abs().max().item()), err       # fp16 rounding of qkv / attention output only


_WORKER = r"""
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, os.environ["PGT_REPO"])
from tests import emu_ops
import pgtformer_amd.ops as real
for name in emu_ops.ALL:
    setattr(real, name, getattr(emu_ops, name))
from pgtformer_amd.archs.tdcrqvae3_arch import VQEmbedding
from tests.golden import cases_r2b as C
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
name = "ema_256x64"
k, d, n, decay, restart, steps, seed = C.EMA[name]
w, batches = C.ema_case(name)

def fresh():
    vq = VQEmbedding(k, d, decay=decay, restart_unused_codes=restart)
    with torch.no_grad():
        vq.weight.copy_(w); vq.embed_ema.copy_(w[:-1])
    return vq.prepare("cpu", torch.float32).train()

vq = fresh()
calls = {"all_reduce": 0, "broadcast": 0}
ar, bc = dist.all_reduce, dist.broadcast
def _ar(*a, **kw): calls["all_reduce"] += 1; return ar(*a, **kw)
def _bc(*a, **kw): calls["broadcast"] += 1; return bc(*a, **kw)
dist.all_reduce, dist.broadcast = _ar, _bc
x = batches[0]
half = x[rank * (n // 2):(rank + 1) * (n // 2)]
torch.manual_seed(100 + rank)                       # ranks draw DIFFERENT restart candidates: rank 0's are broadcast
vq(half)
dist.all_reduce, dist.broadcast = ar, bc
assert calls == {"all_reduce": 1, "broadcast": 1}, calls      # ONE all-reduce for sums + counts (the reference issues two)
# every rank ends with the same state ...
for t in (vq.book, vq.cs_ema_d, vq.embed_ema_d):
    g = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(g, t.contiguous())
    assert torch.equal(g[0], g[1])
# ... whose EMA part equals a single-process step on the whole batch (counts exactly, sums to round-off)
idx = real.rq_argmin(real.linear(x, w[:-1].contiguous(), None, out_f32=True), real.row_sumsq(x), w[:-1].pow(2).sum(1))
stats = emu_ops.vq_cluster_stats(x, idx, k)
cs_want = 0.0 * torch.zeros(k) + (1 - decay) * stats[k * d:]
live = cs_want >= 1
assert torch.equal(vq.cs_ema_d[~live], torch.ones(int((~live).sum())))
em_want = decay * w[:-1] + (1 - decay) * stats[:k * d].reshape(k, d)
assert torch.allclose(vq.embed_ema_d[live], em_want[live], rtol=1e-5, atol=1e-7)
if rank == 0:
    # restarted rows are rows of rank 0's half batch
    dead = (~live).nonzero().flatten()
    rows = {tuple(np.round(r.numpy(), 6)) for r in half}
    assert all(tuple(np.round(vq.embed_ema_d[j].numpy(), 6)) in rows for j in dead[:8])
dist.barrier()
dist.destroy_process_group()
print("OK", rank)
"""


def test_ema_update_world2_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29547", PGT_REPO=REPO)
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for p in procs:
        out, _ = p.communicate(timeout=180)
        assert p.returncode == 0, out.decode()
        assert b"OK" in out
