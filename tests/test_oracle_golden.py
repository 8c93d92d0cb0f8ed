"""The oracle (CPU restatement) against fixtures produced by the imported reference
(tests/golden/make_golden.py). These are the pins that make the oracle trustworthy."""
import json
import os

import numpy as np
import pytest
import torch

from tests.golden import cases

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_manifest_matches_reference_state_dict(manifest):
    from pgtformer_amd.manifest import manifest_to_json

    ref = json.load(open(os.path.join(GOLD, "state_dict_manifest.json")))
    mine = manifest_to_json(manifest)
    assert list(mine) == list(ref)  # same keys, same order
    assert mine == ref
    assert len(mine) == 961


def test_weightgen_is_deterministic(manifest, cfg):
    from pgtformer_amd.weightgen import generate_tensor

    for name in ("encoder.conv_in.weight", "ft_layers.3.self_attn.in_proj_weight",
                 "conditionnet.cp.resnet.bn1.running_var", "quantizer.codebooks.0.weight"):
        shape, dt = manifest[name]
        a = generate_tensor(name, shape, dt, cfg, 0)
        b = generate_tensor(name, shape, dt, cfg, 0)
        assert a.shape == tuple(shape) and np.array_equal(a, b)
    w = generate_tensor("quantizer.codebooks.0.weight", (1025, 512), "float32", cfg, 0)
    e = generate_tensor("quantizer.codebooks.0.embed_ema", (1024, 512), "float32", cfg, 0)
    assert np.array_equal(w[:-1], e) and not w[-1].any()
    # a known-answer pin so that a numpy RNG change is noticed
    v = generate_tensor("encoder.conv_in.bias", (64,), "float32", cfg, 0)
    assert abs(float(v[:4].sum()) - float(np.float32(v[0] + v[1] + v[2] + v[3]))) < 1e-6


@pytest.mark.parametrize("name", list(cases.CASES))
def test_oracle_function_matches_reference(name):
    gold = np.load(os.path.join(GOLD, "ops_golden.npz"))
    outs = cases.run_oracle(name)
    for i, o in enumerate(outs):
        ref = torch.from_numpy(gold[f"{name}.{i}"])
        assert o.shape == ref.shape
        if ref.dtype == torch.int64:
            assert torch.equal(o, ref)
        else:
            # same ATen CPU kernels, same op order -> equal to fp32 round-off
            assert (o - ref).abs().max().item() <= 2e-6 * max(1.0, ref.abs().max().item())


def test_window_policy_and_u8():
    from oracle import pgt_oracle as O

    assert O.window_triples(0) == []
    assert O.window_triples(1) == [(0, 0, 0)]
    assert O.window_triples(2) == [(0, 0, 1), (0, 1, 1)]
    assert O.window_triples(3) == [(0, 0, 1), (0, 1, 2), (1, 2, 2)]
    t7 = O.window_triples(7)
    assert t7[0] == (0, 0, 1) and t7[-1] == (5, 6, 6) and t7[3] == (2, 3, 4) and len(t7) == 7
    f = torch.tensor([[[-0.2, 0.0, 0.5, 0.999, 1.0, 1.7]]]).expand(3, 1, 6)
    u = O.frame_to_u8(f)
    assert u[0, :, 0].tolist() == [0, 0, 127, 254, 255, 255]  # truncation, not rounding


@pytest.mark.slow
def test_oracle_whole_model_matches_reference(cfg, full_sd, golden_window):
    from oracle import pgt_oracle as O

    g = np.load(os.path.join(GOLD, "full_golden.npz"))
    x, _, _ = golden_window
    taps = {}
    out, logits, lq = O.pgtformer_forward(full_sd, cfg, x, w=1.0, taps=taps)
    assert out.shape == (3, 3, 512, 512) and logits.shape == (3, 32, 32, 1, 1024)
    assert lq.shape == (3, 32, 32, 512)
    assert np.array_equal(taps["codes"].numpy().astype(np.int16), g["codes"])
    crop = out[1, :, 192:320, 192:320].numpy()
    assert np.abs(crop - g["out_mid_crop"]).max() <= 1e-4
    assert np.abs(lq[:, 12:20, 12:20, :].numpy() - g["lq_feat_crop"]).max() <= 1e-5
    assert np.abs(logits[:, :2, :2].numpy() - g["logits_tok0"]).max() <= 1e-4
    assert np.abs(taps["cond"].numpy() - g["cond_f16"].astype(np.float32)).max() <= 4e-3
    st = np.array([[o.mean().item(), o.std().item(), o.min().item(), o.max().item()] for o in out])
    assert np.abs(st - g["out_stats"]).max() <= 1e-3


def test_weightgen_sft_gains(manifest, cfg):
    """The last convs of the SFT scale / shift branches are drawn with the calibrated gains of weightgen.SFT_GAINS (the
    multiplicative fusions otherwise square the decoder trunk's magnitude four times: rms 4.5e5 at 256x256)."""
    from pgtformer_amd.weightgen import SFT_GAINS, _generate_plain, generate_tensor

    for size, (gs, gh) in SFT_GAINS.items():
        for branch, gain in (("scale", gs), ("shift", gh)):
            for leaf in ("weight", "bias"):
                name = f"fuse_convs_dict.{size}.{branch}.2.{leaf}"
                shape, dt = manifest[name]
                assert np.array_equal(generate_tensor(name, shape, dt, cfg, 0),
                                      _generate_plain(name, shape, cfg, 0) * np.float32(gain))
    name = "fuse_convs_dict.64.scale.0.weight"      # every other tensor is untouched
    assert np.array_equal(generate_tensor(name, *manifest[name], cfg, 0), _generate_plain(name, manifest[name][0], cfg, 0))


@pytest.mark.slow
def test_oracle_at_the_fitted_tail_operating_point(cfg, full_sd):
    """tests/golden/r3_golden.npz (made by the imported reference with the fitted-tail weight scheme, make_golden_r3.py):
    the oracle reproduces the reference's restored middle frame bit for bit, and the operating point is the one the PSNR
    contract needs - frames inside [0, 1] up to a few percent of the pixels, PSNR(reference, GT) >= 25 dB."""
    from oracle import pgt_oracle as O
    from pgtformer_amd.synth import make_clip, window_from_clip
    from tests.golden.r3_scheme import fitted_tail_state_dict

    g = np.load(os.path.join(GOLD, "r3_golden.npz"))
    sd = fitted_tail_state_dict(full_sd)
    lq_u8, gt = make_clip(4, 512, seed=1234)
    x = torch.from_numpy(window_from_clip(lq_u8, 2).astype(np.float32) / np.float32(255.0)).permute(0, 3, 1, 2).contiguous()
    out, logits, _ = O.pgtformer_forward(sd, cfg, x, w=1.0)            # the held-out window (frames 1, 2, 3)
    assert np.array_equal(out[1, :, ::8, :].numpy(), g["w2.out_mid_rows"])
    assert np.array_equal(logits.argmax(-1).numpy().astype(np.int16), g["w2.codes"])
    gt_t = torch.from_numpy(gt[[1, 2, 3]]).permute(0, 3, 1, 2)
    psnr = float(-10 * torch.log10(((out.double() - gt_t.double()) ** 2).mean()))
    assert abs(psnr - float(g["w2.psnr_ref_vs_gt_db"][0])) < 1e-6
    for tag in ("w1", "w2"):
        assert float(g[f"{tag}.psnr_ref_vs_gt_db"][0]) >= 25.0
        lo, hi = g[f"{tag}.out_stats"][:, 2].min(), g[f"{tag}.out_stats"][:, 3].max()
        assert lo > -0.5 and hi < 1.5


@pytest.mark.slow
def test_oracle_on_a_window_of_another_clip(cfg, full_sd):
    """tests/golden/r3_golden_more.npz (`make_golden_r3.py --more`: the reference on windows of other synthetic clips at the
    fitted-tail operating point - the windows the GPU contract test uses beyond the golden clip): the oracle reproduces the
    reference's middle frame and codes on one of them bit for bit."""
    from oracle import pgt_oracle as O
    from pgtformer_amd.synth import make_clip, window_from_clip
    from tests.golden.r3_scheme import fitted_tail_state_dict

    g = np.load(os.path.join(GOLD, "r3_golden_more.npz"))
    assert sorted({k.split(".")[0] for k in g.files}) == ["c1077w4", "c2077w1", "c3077w3"]
    assert all(float(g[f"{t}.psnr_ref_vs_gt_db"][0]) >= 25.0 for t in ("c1077w4", "c2077w1", "c3077w3"))
    lq_u8, _ = make_clip(3, 512, seed=2077)
    x = torch.from_numpy(window_from_clip(lq_u8, 1).astype(np.float32) / np.float32(255.0)).permute(0, 3, 1, 2).contiguous()
    out, logits, _ = O.pgtformer_forward(fitted_tail_state_dict(full_sd), cfg, x, w=1.0)
    assert np.array_equal(out[1, :, ::8, :].numpy(), g["c2077w1.out_mid_rows"])
    assert np.array_equal(logits.argmax(-1).numpy().astype(np.int16), g["c2077w1.codes"])


@pytest.mark.slow
@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5])
def test_oracle_at_the_later_operating_points(cfg, manifest, seed):
    """tests/golden/r5_golden_s1 / _s2, r6_golden_s3 / _s4 / _s5.npz (make_golden_r5.py: the imported reference at weight seeds 1 .. 5, each with
    re-calibrated SFT gains and its own fitted tail; seeds 4 and 5 with the damped code transformer): on a HELD-OUT window of every point - the
    first window of the point's second clip; the generator itself compares on the fitted one - the oracle reproduces the reference's codes
    and its restored middle frame bit for bit (at the thread count the fixture was generated with) and its top-2 logit margins to 1e-5.  The GPU contract tests of these points compare against the same files."""
    from oracle import pgt_oracle as O
    from pgtformer_amd.synth import make_clip, window_from_clip
    from pgtformer_amd.weightgen import generate_state_dict
    from tests.golden.r5_scheme import POINTS, point_state_dict

    pt = POINTS[seed]
    g = np.load(os.path.join(GOLD, pt["golden"]))
    clip, i = pt["windows"][3]
    assert (clip, i) != pt["train"]
    tag = f"c{clip}w{i}"
    sd = point_state_dict(generate_state_dict(manifest, cfg, seed=seed), seed)
    lq_u8, _ = make_clip(pt["clip_frames"][clip], 512, seed=clip)
    x = torch.from_numpy(window_from_clip(lq_u8, i).astype(np.float32) / np.float32(255.0)).permute(0, 3, 1, 2).contiguous()
    # CPU fp32 convolutions split their sums by the number of threads: reference and oracle agree bit for bit at the SAME count, and to
    # ~4e-7 (two ulps at 0.5) across counts.  The fixtures of seed 3 were generated with 5 threads, the others with 8 (r5_scheme.POINTS).
    nt = torch.get_num_threads()
    torch.set_num_threads(pt.get("gen_threads", 8))
    try:
        out, logits, _ = O.pgtformer_forward(sd, cfg, x, w=1.0)
    finally:
        torch.set_num_threads(nt)
    lg = logits.reshape(-1, logits.shape[-1])
    assert np.array_equal(out[1, :, ::8, :].numpy(), g[f"{tag}.out_mid_rows"])
    assert np.array_equal(lg.argmax(-1).numpy().astype(np.int16), g[f"{tag}.codes"].reshape(-1))
    top2 = lg.topk(2, dim=-1).values
    # (the oracle's token-major Linear sums in another order than the reference's permuted one: logits agree to ~3e-6, not bit for bit)
    assert np.abs((top2[:, 0] - top2[:, 1]).numpy().astype(np.float32) - g[f"{tag}.top2_margin"].reshape(-1)).max() < 1e-5
