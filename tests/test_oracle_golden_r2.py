"""Round-2 oracle pins (CPU): the restatements of modules/swin.py (WindowAttention3D, forward_part1, compute_mask),
archs/vqgan_arch.py VectorQuantizer.forward and the remaining stage-I API against fixtures produced by the imported
reference (tests/golden/make_golden_r2.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import pgt_oracle as O
from tests.golden import cases_r2 as C2

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLD, "r2_golden.npz"))


@pytest.mark.parametrize("name", list(C2.SWIN))
def test_swin_block_part1_matches_reference(gold, name):
    dim, heads, ws, ss, fmap, qkv_bias, seed = C2.SWIN[name]
    y = O.swin_block_part1(C2.swin_params(name), C2.swin_input(name), heads, ws, ss)
    ref = torch.from_numpy(gold[f"{name}.out"])
    assert (y[..., :128] - ref).abs().max().item() <= 2e-6 * max(1.0, ref.abs().max().item())
    if any(ss):
        b, d, h, w = fmap
        m = O.swin_compute_mask(d, h, w, ws, ss)
        assert np.array_equal(m.numpy().astype(np.int8), gold[f"{name}.mask"])
        assert m.shape == ((d // ws[0]) * (h // ws[1]) * (w // ws[2]), ws[0] * ws[1] * ws[2], ws[0] * ws[1] * ws[2])


@pytest.mark.parametrize("name", list(C2.VQ))
def test_vector_quantizer_matches_reference(gold, name):
    w, z = C2.vq_case(name)
    zq, loss, idx, mean_d = O.vector_quantizer(w, z, 0.25)
    assert np.array_equal(idx.numpy().astype(np.int32), gold[f"{name}.indices"])
    assert (zq - torch.from_numpy(gold[f"{name}.z_q"])).abs().max().item() <= 1e-7
    assert abs(loss.item() - gold[f"{name}.loss"][0]) <= 1e-9 and abs(mean_d.item() - gold[f"{name}.loss"][1]) <= 1e-7
    if name.endswith("ties"):
        assert int(idx.reshape(1, 6, 5)[0, 1, 2]) == 5           # duplicated rows 5 / 77: the lower index wins


@pytest.mark.slow
def test_stage1_api_matches_reference(gold, cfg, full_sd, golden_window):
    x, _, _ = golden_window
    z, _ = O.encoder_forward(full_sd, cfg["ddconfig"], x.reshape(1, 3, 3, 512, 512))
    z_e = O._conv(full_sd, "quant_conv", z).permute(0, 2, 3, 1).contiguous()
    zq, loss, codes = O.rq_forward(full_sd, z_e, 1, True)
    soft, scode = O.rq_soft_codes(full_sd, z_e, 1, True, temp=0.5)
    assert np.array_equal(codes.numpy().astype(np.int16), gold["stage1.codes"]) and torch.equal(codes, scode)
    assert abs(loss.item() - gold["stage1.loss"][0]) <= 1e-7 * max(1.0, gold["stage1.loss"][0])
    assert np.abs(zq[:, 12:20, 12:20, :64].numpy() - gold["stage1.z_q_crop"]).max() <= 1e-6
    assert np.abs(soft[:, :2, :2].numpy() - gold["stage1.soft_tok"]).max() <= 1e-6
    assert np.abs(soft.max(-1).values.numpy() - gold["stage1.soft_max"]).max() <= 1e-6
