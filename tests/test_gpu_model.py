"""GPU parity tests, module and whole-model level, against
 (a) the committed reference-derived goldens (tests/golden/*.npz, made by the imported reference), and
 (b) the CPU oracle run on the GPU box's host cores on the same seeded inputs.
Metrics are also written to gpurun_out/parity_model.json.

Tolerances:  f32 path: max|d| <= 1e-3*max|ref| per module, whole-model PSNR(build, reference) >= 80 dB on
the clamped middle frame and 100 % code agreement except tokens whose reference top-2 logit margin is
< 1e-3.  x3f16 (the default / benchmarked mode: split-half code branch, IEEE-half decoder): the SAME code criterion
(every code equal to the reference's except where the reference's own top-2 margin is < 1e-3), logits within 2e-3, and
north_star's PSNR contract at the fitted-tail operating point (tests/golden/make_golden_r3.py: reference frames inside
[0, 1], PSNR(reference, GT) = 28.7 dB): |PSNR(build, GT) - PSNR(reference, GT)| <= 1e-3 dB with UNCLAMPED
PSNR(build, reference) >= 70 dB.  bf16x3 / mixed (bf16 decoder): same code criterion, PSNR(build, reference) >= 35 dB;
their PSNR-contract figure is reported (5e-3 dB: the reason they are not the default).  Pure bf16 (opt-in speed mode): reported, plus a teacher-forced decoder check (reference codes fed
in: PSNR >= 30 dB) and bit-equality of the in-place concat path with the copying path.
"""
import json
import os

import numpy as np
import pytest
import torch

from tests.golden import cases

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
DEV = "cuda"
_LOG = {}


@pytest.fixture(scope="module", autouse=True)
def _dump_log():
    yield
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/parity_model.json", "w") as f:
        json.dump(_LOG, f, indent=1)


def nhwc(x5):  # (B,D,C,H,W) or (N,C,H,W) -> (N,H,W,C)
    if x5.dim() == 5:
        x5 = x5.reshape(-1, *x5.shape[2:])
    return x5.permute(0, 2, 3, 1).contiguous()


def back(y, like):  # (N,H,W,C) -> shape of the golden (B,D,C,H,W) / (N,C,H,W)
    return y.float().cpu().permute(0, 3, 1, 2).reshape(like.shape)


def psnr(a, b):
    mse = float(((a.double() - b.double()) ** 2).mean())
    return 200.0 if mse == 0 else -10.0 * np.log10(mse)


@pytest.fixture(scope="module")
def models(cfg, full_sd):
    from pgtformer_amd import PGTFormer

    out = {}
    for prec in ("fp32", "bf16", "mixed", "bf16x3", "x3f16"):
        m = PGTFormer(**cfg)
        m.load_state_dict(full_sd, strict=True)
        out[prec] = m.prepare(DEV, prec)
    return out


def _run_case(model, name, dtype):
    kind, prefix, shape, seed = cases.CASES[name]
    x = [t.to(DEV) for t in cases.case_inputs(name)]
    mod = model.get_submodule(prefix) if prefix else None
    from pgtformer_amd import ops
    if kind in ("resblock", "downsample", "upsample", "enclayer"):
        y = mod(nhwc(x[0]).to(dtype))
        return [y]
    if kind == "salayer":
        L, b, e = x[0].shape
        y = mod(x[0].reshape(L, e).to(dtype), b, L, query_pos=x[1].reshape(L, e).to(dtype))
        return [y.reshape(L, b, e)]
    if kind == "fuse":
        return [mod(nhwc(x[0]).to(dtype), nhwc(x[1]).to(dtype), w=1.0)]
    if kind == "adain":
        from pgtformer_amd.archs.codeformer_arch import adaptive_instance_normalization as adain
        return [adain(nhwc(x[0]).to(dtype), nhwc(x[1]).to(dtype))]
    if kind == "embed":
        return [model.quantizer.embed_code(x[0], dtype)[:, ::4, ::4]]
    if kind == "rq":
        agg, codes = model.quantizer.quantize(x[0].to(dtype))
        return [agg, codes.long()]
    raise KeyError(kind)


@pytest.mark.parametrize("name", list(cases.CASES))
def test_module_matches_reference_golden_f32(models, name):
    gold = np.load(os.path.join(GOLD, "ops_golden.npz"))
    outs = _run_case(models["fp32"], name, torch.float32)
    kind = cases.CASES[name][0]
    for i, o in enumerate(outs):
        ref = torch.from_numpy(gold[f"{name}.{i}"])
        if ref.dtype == torch.int64:
            assert torch.equal(o.cpu(), ref)
            continue
        got = o.float().cpu() if kind in ("salayer", "embed", "rq") else back(o, ref)
        err = (got - ref).abs().max().item()
        scale = max(1.0, ref.abs().max().item())
        _LOG[f"f32/{name}.{i}"] = {"max_abs_err": err, "ref_absmax": scale}
        assert err <= 1e-3 * scale, f"{name}: {err:.3e} vs scale {scale:.3f}"


@pytest.mark.parametrize("name", [n for n in cases.CASES if cases.CASES[n][0] not in ("embed", "rq")])
def test_module_matches_reference_golden_bf16(models, name):
    gold = np.load(os.path.join(GOLD, "ops_golden.npz"))
    outs = _run_case(models["bf16"], name, torch.bfloat16)
    kind = cases.CASES[name][0]
    ref = torch.from_numpy(gold[f"{name}.0"])
    got = outs[0].float().cpu() if kind == "salayer" else back(outs[0], ref)
    err = (got - ref).abs().max().item()
    rel = ((got - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()
    _LOG[f"bf16/{name}"] = {"max_abs_err": err, "rel_rms": rel}
    assert rel <= 5e-2, f"{name}: relative RMS error {rel:.3e}"


def _full(models, prec, x):
    m = models[prec]
    out, logits, lq = m(x.to(DEV), w=1.0)
    torch.cuda.synchronize()
    return out.cpu(), logits.cpu(), lq.cpu(), m.last_codes.cpu().numpy().astype(np.int16)


def test_whole_model_f32_matches_reference(models, golden_window):
    g = np.load(os.path.join(GOLD, "full_golden.npz"))
    x, win_u8, gt = golden_window
    out, logits, lq, codes = _full(models, "fp32", x)
    assert out.shape == (3, 3, 512, 512) and logits.shape == (3, 32, 32, 1, 1024) and lq.shape == (3, 32, 32, 512)
    margin = g["logit_margin"].reshape(codes.shape)
    mism = codes != g["codes"]
    rec = {"code_agreement": float(1 - mism.mean()), "n_mismatch": int(mism.sum()),
           "mismatch_margins": [float(v) for v in margin[mism][:16]]}
    ref_crop = torch.from_numpy(g["out_mid_crop"])
    crop = out[1, :, 192:320, 192:320]
    rec["psnr_mid_crop_clamped_db"] = psnr(crop.clamp(0, 1), ref_crop.clamp(0, 1))
    rec["max_abs_err_crop"] = float((crop - ref_crop).abs().max())
    rec["lq_feat_err"] = float(np.abs(lq[:, 12:20, 12:20, :].numpy() - g["lq_feat_crop"]).max())
    rec["logits_err"] = float(np.abs(logits[:, :2, :2].numpy() - g["logits_tok0"]).max())
    f16 = torch.from_numpy(g["out_f16"].astype(np.float32))
    rec["psnr_full_sub4_clamped_db"] = psnr(out[:, :, ::4, ::4].clamp(0, 1), f16.clamp(0, 1))
    _LOG["whole/fp32"] = rec
    assert rec["lq_feat_err"] < 1e-3
    assert (margin[mism] < 1e-3).all(), "code flips only where the reference's own top-2 margin is < 1e-3"
    if rec["n_mismatch"] == 0:
        assert rec["psnr_mid_crop_clamped_db"] >= 80.0, rec
        assert rec["max_abs_err_crop"] < 2e-2
    # run-to-run determinism
    out2, _, _, codes2 = _full(models, "fp32", x)
    assert torch.equal(out, out2) and np.array_equal(codes, codes2)
    # uint8 driver ingest == float ingest
    u8 = models["fp32"].restore_middle_u8(torch.from_numpy(win_u8).to(DEV), w=1.0).cpu()
    want = (out[1].clamp(0, 1).permute(1, 2, 0) * 255).to(torch.uint8)
    assert (u8.int() - want.int()).abs().max() <= 1


def _reduced_record(out, logits, lq, codes, g, gt):
    ref_crop = torch.from_numpy(g["out_mid_crop"])
    crop = out[1, :, 192:320, 192:320]
    f16 = torch.from_numpy(g["out_f16"].astype(np.float32))
    gt_t = torch.from_numpy(gt[:3]).permute(0, 3, 1, 2)[:, :, ::4, ::4]
    mism = codes != g["codes"]
    margin = g["logit_margin"].reshape(codes.shape)
    return {"code_agreement": float(1 - mism.mean()), "n_mismatch": int(mism.sum()),
            "mismatch_margins": [float(v) for v in margin[mism][:16]],
            "max_mismatch_margin": float(margin[mism].max()) if mism.any() else 0.0,
            "logits_err": float(np.abs(logits[:, :2, :2].numpy() - g["logits_tok0"]).max()),
            "psnr_mid_crop_clamped_db": psnr(crop.clamp(0, 1), ref_crop.clamp(0, 1)),
            "psnr_full_sub4_clamped_db": psnr(out[:, :, ::4, ::4].clamp(0, 1), f16.clamp(0, 1)),
            "psnr_build_vs_gt_db": psnr(out[:, :, ::4, ::4].clamp(0, 1), gt_t),
            "psnr_ref_vs_gt_db": psnr(f16.clamp(0, 1), gt_t),
            "lq_feat_err": float(np.abs(lq[:, 12:20, 12:20, :].numpy() - g["lq_feat_crop"]).max())}


@pytest.mark.parametrize("prec", ["x3f16", "bf16x3", "mixed"])
def test_whole_model_default_mode_matches_reference(models, golden_window, prec):
    """The modes with a split-half (or fp32) code branch reproduce the reference: every arg-max code
    (archs/pgtformer_arch.py:663) equals the fp32 reference's, logits / lq_feat to fp32-class error, and the restored frames to >= 55 dB with the IEEE-half decoder (x3f16, the
    default), >= 35 dB with the bf16 decoder.  (Random-tail weights: the frames are noise against the GT - the PSNR contract
    is asserted at the fitted-tail operating point, test_psnr_contract_at_the_operating_point.)"""
    g = np.load(os.path.join(GOLD, "full_golden.npz"))
    x, _, gt = golden_window
    out, logits, lq, codes = _full(models, prec, x)
    assert out.shape == (3, 3, 512, 512) and logits.shape == (3, 32, 32, 1, 1024) and lq.shape == (3, 32, 32, 512)
    assert torch.isfinite(out).all()
    rec = _reduced_record(out, logits, lq, codes, g, gt)
    _LOG[f"whole/{prec}"] = rec
    # round 5: the gate is what is measured - every code of the golden window equal (kernel selection is static: the same bits on
    # every box), logits within 1e-4 (measured 1.24e-5 / 5e-6); rounds 1-4 tolerated 0.3 % differing codes below a margin of 1e-3
    assert rec["n_mismatch"] == 0, rec
    assert rec["logits_err"] < 1e-4 and rec["lq_feat_err"] < 1e-3, rec
    floor = 55.0 if prec == "x3f16" else 35.0
    assert rec["psnr_full_sub4_clamped_db"] >= floor and rec["psnr_mid_crop_clamped_db"] >= floor - 0.5, rec
    rec["psnr_vs_gt_abs_diff_db"] = abs(rec["psnr_build_vs_gt_db"] - rec["psnr_ref_vs_gt_db"])   # reported only: noise vs GT here
    out2, _, _, codes2 = _full(models, prec, x)                      # run-to-run determinism
    assert torch.equal(out, out2) and np.array_equal(codes, codes2)


@pytest.fixture(scope="module")
def tail_models(cfg, full_sd):
    """the fitted-tail weight scheme (tests/golden/r3_scheme.py) in the default mode, the bf16-decoder mode and fp32"""
    from pgtformer_amd import PGTFormer
    from tests.golden.r3_scheme import fitted_tail_state_dict

    sd = fitted_tail_state_dict(full_sd)
    out = {}
    for prec in ("x3f16", "bf16x3", "fp32"):
        m = PGTFormer(**cfg)
        m.load_state_dict(sd, strict=True)
        out[prec] = m.prepare(DEV, prec)
    return out


def _op_point_record(out, codes, g, tag, gt_frames):
    """out (3,3,512,512) fp32 (unclamped) against the reference fixtures of window `tag` at the fitted-tail operating point"""
    ref_rows = torch.from_numpy(g[f"{tag}.out_mid_rows"]).double()              # every 8th row of the middle frame, fp32
    rows = out[1][:, ::8, :].double()
    gt_rows = torch.from_numpy(gt_frames[1]).permute(2, 0, 1)[:, ::8, :].double()
    p_ref, p_build = psnr(ref_rows, gt_rows), psnr(rows, gt_rows)
    sub = torch.from_numpy(g[f"{tag}.out_f16_sub2"].astype(np.float32))
    mine = out[:, :, ::2, ::2] if sub.shape[0] == 3 else out[1:2, :, ::2, ::2]
    d = (mine.double() - sub.double())
    return {"psnr_ref_vs_gt_db": p_ref, "psnr_build_vs_gt_db": p_build, "dpsnr_db": p_build - p_ref,
            "psnr_build_vs_ref_unclamped_db": psnr(rows, ref_rows),
            "rel_rms_unclamped": float(d.pow(2).mean().sqrt() / sub.double().pow(2).mean().sqrt()),
            "psnr_build_vs_ref_f16_sub2_db": psnr(mine, sub), "out_min": float(out.min()), "out_max": float(out.max()),
            "saturated_fraction": float(((out < 0) | (out > 1)).float().mean()),
            "code_agreement": float((codes == g[f"{tag}.codes"]).mean())}


@pytest.mark.parametrize("tag", ["w1", "w2"])
def test_psnr_contract_at_the_operating_point(tail_models, tag):
    """north_star: "outputs match the reference within 1e-3 PSNR on identical 512x512 inputs".  Operating point: the
    fitted-tail weight scheme (tests/golden/make_golden_r3.py) - the REFERENCE's restored frames sit inside [0, 1] (1.6 % of
    the pixels outside) at PSNR(reference, GT) = 28.7 dB on the middle frame, the frame the driver keeps
    (inference.py:15).  w1 = the golden window (the tail was fitted on it), w2 = the next window of the clip (held out).
    Default mode (x3f16): |PSNR(build, GT) - PSNR(reference, GT)| <= 1e-3 dB on the exact fp32 rows of the middle frame,
    every code equal, unclamped PSNR(build, reference) >= 70 dB.  The bf16-decoder mode is measured on the same footing
    and reported (CPU emulation: 57.6 dB / 4.8e-3 dB - why it is not the default); fp32: round-off."""
    from pgtformer_amd.synth import make_clip, window_from_clip

    g = np.load(os.path.join(GOLD, "r3_golden.npz"))
    i = 1 if tag == "w1" else 2
    lq_u8, gt = make_clip(4, 512, seed=1234)
    win = window_from_clip(lq_u8, i)
    x = torch.from_numpy(win.astype(np.float32) / np.float32(255.0)).permute(0, 3, 1, 2).contiguous()
    gt_frames = gt[[i - 1, i, i + 1]]
    for prec in ("x3f16", "bf16x3", "fp32"):
        m = tail_models[prec]
        out = m(x.to(DEV), w=1.0)[0].float().cpu()
        codes = m.last_codes.cpu().numpy().astype(np.int16)
        rec = _op_point_record(out, codes, g, tag, gt_frames)
        _LOG[f"operating_point/{tag}/{prec}"] = rec
        assert rec["psnr_ref_vs_gt_db"] >= 25.0 and rec["saturated_fraction"] < 0.05, rec
        assert rec["code_agreement"] == 1.0, rec
        if prec == "x3f16":
            assert abs(rec["dpsnr_db"]) <= 1e-3, rec
            assert rec["psnr_build_vs_ref_unclamped_db"] >= 70.0, rec
            # the u8 frame the driver writes: floor(clamp(x, 0, 1) * 255) of the middle frame (inference.py:16-18)
            u8 = m.restore_middle_u8(torch.from_numpy(win).to(DEV), w=1.0).cpu()
            ref_u8 = (torch.from_numpy(g[f"{tag}.out_mid_rows"]).clamp(0, 1) * 255).to(torch.uint8).permute(1, 2, 0)
            du = (u8[::8].int() - ref_u8.int()).abs()
            _LOG[f"operating_point/{tag}/{prec}"].update(u8_max_diff=int(du.max()), u8_equal_fraction=float((du == 0).float().mean()))
            assert du.max() <= 1 and (du == 0).float().mean() >= 0.97
        elif prec == "fp32":
            assert abs(rec["dpsnr_db"]) <= 1e-4 and rec["psnr_build_vs_ref_unclamped_db"] >= 90.0, rec
        else:
            assert rec["psnr_build_vs_ref_unclamped_db"] >= 50.0, rec


def test_psnr_contract_on_the_benchmarked_path(tail_models):
    """The path bench.py times - unique uint8 frames in, overlap-aware windows (per-frame work once per frame), middle-only
    decoder tail, several windows per forward - at the operating point: the clip's windows w1 and w2 in ONE forward, each
    restored middle frame against the reference's (fp32 rows of r3_golden.npz): |dPSNR vs GT| <= 1e-3 dB, unclamped
    PSNR(build, reference) >= 70 dB, and the driver's uint8 frames within +-1 of the reference's."""
    from pgtformer_amd.driver import WindowRunner
    from pgtformer_amd.synth import make_clip

    g = np.load(os.path.join(GOLD, "r3_golden.npz"))
    lq_u8, gt = make_clip(4, 512, seed=1234)
    m = tail_models["x3f16"]
    frames = torch.from_numpy(lq_u8).to(DEV)
    out, _, _ = m.forward_nhwc(frames, w=1.0, win=m.window_index(2, 3, DEV), middle_only=True)     # (2,512,512,3) fp32
    assert out.shape == (2, 512, 512, 3) and out.dtype == torch.float32
    out = out.cpu()
    runner = WindowRunner(m, 1.0, use_graph=True, batch=2, lanes=2)
    padded = torch.cat([frames[:1], frames, frames[-1:]], 0)             # replicate-padded clip: outputs 0..3, windows 1, 2 inside
    u8 = runner.run_clip(padded, torch.empty_like(frames)).cpu()
    for j, tag in ((0, "w1"), (1, "w2")):
        rows = out[j].permute(2, 0, 1)[:, ::8, :].double()
        ref = torch.from_numpy(g[f"{tag}.out_mid_rows"]).double()
        gt_rows = torch.from_numpy(gt[j + 1]).permute(2, 0, 1)[:, ::8, :].double()
        rec = {"dpsnr_db": psnr(rows, gt_rows) - psnr(ref, gt_rows), "psnr_build_vs_ref_unclamped_db": psnr(rows, ref)}
        ref_u8 = (ref.float().clamp(0, 1) * 255).to(torch.uint8).permute(1, 2, 0)
        du = (u8[j + 1][::8].int() - ref_u8.int()).abs()
        rec.update(u8_max_diff=int(du.max()), u8_equal_fraction=float((du == 0).float().mean()))
        _LOG[f"operating_point_benchmarked_path/{tag}/x3f16"] = rec
        assert abs(rec["dpsnr_db"]) <= 1e-3 and rec["psnr_build_vs_ref_unclamped_db"] >= 70.0, rec
        assert rec["u8_max_diff"] <= 1 and rec["u8_equal_fraction"] >= 0.97, rec


@pytest.mark.parametrize("tag", ["c2077w1", "c1077w4", "c3077w3"])
def test_psnr_contract_on_windows_of_other_clips(tail_models, tag):
    """The contract beyond the clip the tail was fitted on: windows of three other synthetic clips (reference rows in
    r3_golden_more.npz, `make_golden_r3.py --more`), chosen where a half decoder WITHOUT the weight-rounding compensation is
    furthest from the reference (profiles/r3_psnr_sweep.md: +1.1e-3, -1.3e-3, +1.0e-3 dB).  Through the benchmarked path
    (uint8 frames, overlap-aware window, middle-only tail): |dPSNR vs GT| <= 1e-3 dB, every code equal to the reference's,
    unclamped PSNR(build, reference) >= 70 dB."""
    from pgtformer_amd.synth import make_clip

    g = np.load(os.path.join(GOLD, "r3_golden_more.npz"))
    seed, i = int(tag[1:5]), int(tag[6:])
    lq_u8, gt = make_clip(i + 2, 512, seed=seed)
    m = tail_models["x3f16"]
    frames = torch.from_numpy(lq_u8[i - 1:i + 2]).to(DEV)
    out, _, _ = m.forward_nhwc(frames, w=1.0, win=m.window_index(1, 3, DEV), middle_only=True)        # (1,512,512,3) fp32
    rows = out[0].float().cpu().permute(2, 0, 1)[:, ::8, :].double()
    ref = torch.from_numpy(g[f"{tag}.out_mid_rows"]).double()
    gt_rows = torch.from_numpy(gt[i]).permute(2, 0, 1)[:, ::8, :].double()
    codes = m.last_codes.cpu().numpy().astype(np.int16)
    rec = {"psnr_ref_vs_gt_db": psnr(ref, gt_rows), "dpsnr_db": psnr(rows, gt_rows) - psnr(ref, gt_rows),
           "psnr_build_vs_ref_unclamped_db": psnr(rows, ref), "code_agreement": float((codes == g[f"{tag}.codes"]).mean())}
    _LOG[f"operating_point_other_clips/{tag}/x3f16"] = rec
    assert rec["psnr_ref_vs_gt_db"] >= 25.0 and rec["code_agreement"] == 1.0, rec
    assert abs(rec["dpsnr_db"]) <= 1e-3 and rec["psnr_build_vs_ref_unclamped_db"] >= 70.0, rec


def test_psnr_contract_sweep_against_reference_fixtures(tail_models):
    """The 12 windows of the regression sweep (windows 1..6 of the 8-frame clips 4077 and 5077) through the benchmarked path,
    default mode, against REFERENCE fixtures (tests/golden/r4_golden_sweep.npz, make_golden_r3.py --sweep: the reference's codes,
    its own top-2 logit margins and every 8th fp32 row of the middle frame).  Gate:
      * every code equals the reference's, except at tokens where the REFERENCE's top-2 margin is below 1e-4 - a near-tie the
        reference itself resolves only up to its fp32 summation order (its logit noise is ~5e-6; the build's 1.25e-5);
      * on every window without a differing token: |PSNR(build, GT) - PSNR(reference, GT)| <= 1e-3 dB on the fixture rows and
        PSNR(build, reference) >= 75 dB.
    Round 3 compared with the fp32 BUILD here and tolerated one differing window without knowing who was right: on that
    window (clip 4077, window 6) the reference's margin at the token is 1.4e-6."""
    from pgtformer_amd.synth import make_clip

    g = np.load(os.path.join(GOLD, "r4_golden_sweep.npz"))
    m = tail_models["x3f16"]
    recs, near_ties = [], 0
    for seed in (4077, 5077):
        lq_u8, gt = make_clip(8, 512, seed=seed)
        fr = torch.from_numpy(lq_u8).to(DEV)
        outs, codes = [], []
        for s0 in range(0, 6, 2):                       # two windows per forward
            y, _, _ = m.forward_nhwc(fr[s0:s0 + 4], w=1.0, win=m.window_index(2, 3, DEV), middle_only=True)
            outs.append(y.float().cpu())
            codes.append(m.last_codes.cpu().clone().reshape(2, -1))
        outs, codes = torch.cat(outs, 0), torch.cat(codes, 0)
        for j in range(6):
            tag = f"c{seed}w{j + 1}"
            ref_codes = torch.from_numpy(g[f"{tag}.codes"].astype(np.int64)).reshape(-1)
            margin = torch.from_numpy(g[f"{tag}.top2_margin"]).reshape(-1)
            diff = (codes[j].long() != ref_codes).nonzero().reshape(-1)
            ref_rows = torch.from_numpy(g[f"{tag}.out_mid_rows"]).double()
            rows = outs[j].permute(2, 0, 1)[:, ::8, :].double()
            gt_rows = torch.from_numpy(gt[j + 1]).permute(2, 0, 1)[:, ::8, :].double()
            rec = {"clip_seed": seed, "window": j + 1, "differing_tokens": int(diff.numel()),
                   "reference_margin_at_differing_tokens": [float(margin[i]) for i in diff],
                   "smallest_reference_margin": float(margin.min()), "psnr_ref_vs_gt_db": psnr(ref_rows, gt_rows),
                   "dpsnr_db": psnr(rows, gt_rows) - psnr(ref_rows, gt_rows), "psnr_build_vs_ref_db": psnr(rows, ref_rows)}
            recs.append(rec)
            assert all(mg < 1e-4 for mg in rec["reference_margin_at_differing_tokens"]), rec     # only the reference's near-ties
            assert rec["psnr_ref_vs_gt_db"] >= 25.0, rec
            if diff.numel() == 0:
                assert abs(rec["dpsnr_db"]) <= 1e-3 and rec["psnr_build_vs_ref_db"] >= 75.0, rec
            else:
                # an explicit allow-list (round 5): a NEW window with a differently resolved near-tie fails instead of being
                # absorbed by a count; (4077, 6): ONE token, reference margin 1.4e-6 (a quarter of the reference's own fp32
                # summation-order noise), the fp32 build takes the reference's code there
                near_ties += 1
                assert (seed, j + 1) in NEAR_TIE_WINDOWS and diff.numel() == 1, rec
    _LOG["operating_point_sweep/x3f16_vs_reference"] = {"windows": recs, "windows_with_a_near_tie_resolved_differently": near_ties}


NEAR_TIE_WINDOWS = {(4077, 6)}


def _point_models(cfg, manifest, seed):
    from pgtformer_amd import PGTFormer
    from pgtformer_amd.weightgen import generate_state_dict
    from tests.golden.r5_scheme import POINTS, point_state_dict

    sd = point_state_dict(generate_state_dict(manifest, cfg, seed=seed), seed)
    out = {"point": POINTS[seed], "seed": seed}
    for prec in ("x3f16", "fp32"):
        m = PGTFormer(**cfg)
        m.load_state_dict(sd, strict=True)
        out[prec] = m.prepare(DEV, prec)
    return out


@pytest.fixture(scope="module")
def tail_models_s1(cfg, manifest):
    """the SECOND operating point (tests/golden/r5_scheme.py POINTS[1]: weight seed 1, re-calibrated SFT gains, its own fitted tail) in
    the default mode and fp32"""
    return _point_models(cfg, manifest, 1)


def test_psnr_contract_at_a_second_operating_point(tail_models_s1):
    _psnr_contract_at_point(tail_models_s1)


def _psnr_contract_at_point(tail_models_s1, gate_db=1e-3):
    """the contract at an INDEPENDENT draw of everything it depends on - all 961 tensors from weight seed 1
    (other rounding defects D = W - half(W), other activation ranges for the half decoder), SFT gains re-calibrated on the
    reference for that draw, a decoder tail fitted on a window of another clip (7077 w2) - against REFERENCE fixtures
    (tests/golden/r5_golden_s1.npz, make_golden_r5.py: 8 windows of 3 clips, one fitted, seven held out; the reference's codes,
    its top-2 margins, every 8th fp32 row of the middle frame; PSNR(reference, GT) 26.3 - 28.9 dB).  Benchmarked path (uint8
    frames, overlap-aware windows, middle-only tail), default mode:
      * every code equals the reference's (the smallest reference margin over the 8 windows is 1.1e-4: no near-tie to excuse);
      * |PSNR(build, GT) - PSNR(reference, GT)| <= 1e-3 dB and PSNR(build, reference) >= 75 dB on every window;
      * no half store of the forward sits at the saturation limit (check_range);
    fp32 mode: every code equal, <= 1e-4 dB.
    (test_psnr_contract_at_a_third / _fourth_operating_point: the same at weight seeds 2 and 3.)"""
    from pgtformer_amd.synth import make_clip

    pt = tail_models_s1["point"]
    g = np.load(os.path.join(GOLD, pt["golden"]))
    tags = sorted({k.split(".")[0] for k in g.files})
    assert len(tags) == 8
    recs = []
    clips = {}
    for tag in tags:
        seed, i = (int(v) for v in tag[1:].split("w"))
        if seed not in clips:
            clips[seed] = make_clip(pt["clip_frames"][seed], 512, seed=seed)
        lq_u8, gt = clips[seed]
        frames = torch.from_numpy(lq_u8[i - 1:i + 2]).to(DEV)
        ref = torch.from_numpy(g[f"{tag}.out_mid_rows"]).double()
        gt_rows = torch.from_numpy(gt[i]).permute(2, 0, 1)[:, ::8, :].double()
        ref_codes = g[f"{tag}.codes"].astype(np.int64).reshape(-1)
        rec = {"window": tag, "psnr_ref_vs_gt_db": psnr(ref, gt_rows), "smallest_reference_margin": float(g[f"{tag}.top2_margin"].min())}
        for prec in ("x3f16", "fp32"):
            m = tail_models_s1[prec]
            out, _, _ = m.forward_nhwc(frames, w=1.0, win=m.window_index(1, 3, DEV), middle_only=True)
            rows = out[0].float().cpu().permute(2, 0, 1)[:, ::8, :].double()
            codes = m.last_codes.cpu().numpy().astype(np.int64).reshape(-1)
            rec[prec] = {"dpsnr_db": psnr(rows, gt_rows) - psnr(ref, gt_rows), "psnr_build_vs_ref_db": psnr(rows, ref),
                         "differing_tokens": int((codes != ref_codes).sum())}
        recs.append(rec)
        assert rec["psnr_ref_vs_gt_db"] >= pt["min_psnr_ref_gt_db"], rec      # (the fitted tail restores: the contract is not about noise)
        assert rec["x3f16"]["differing_tokens"] == 0 and rec["fp32"]["differing_tokens"] == 0, rec
        assert abs(rec["x3f16"]["dpsnr_db"]) <= gate_db and rec["x3f16"]["psnr_build_vs_ref_db"] >= 75.0, rec
        assert abs(rec["fp32"]["dpsnr_db"]) <= 1e-4 and rec["fp32"]["psnr_build_vs_ref_db"] >= 90.0, rec
    m = tail_models_s1["x3f16"]
    fr = torch.from_numpy(clips[pt["train"][0]][0][:4]).to(DEV)
    bad = m.check_range(fr, w=1.0, win=m.window_index(2, 3, DEV))
    _LOG[f"operating_point_{tail_models_s1['seed'] + 1}/x3f16_vs_reference"] = {"windows": recs, "tensors_range_checked": m.last_range_launches,
                                                                                "saturating": [list(map(str, r)) for r in bad]}
    assert m.last_range_launches > 300 and bad == [], bad


def test_psnr_contract_at_a_third_operating_point(cfg, manifest):
    """The contract at a THIRD independent draw (tests/golden/r5_scheme.py POINTS[2]: weight seed 2, SFT gains re-calibrated on the
    reference for it, tail fitted on clip 10077 w2; 8 windows of clips 10077 / 11077 / 12077, r5_golden_s2.npz), whose fixtures were
    generated AFTER every constant of the build - bands of the mean field, sample sizes - was fixed: a held-out point.  Same asserts
    as the second point: every code equal, <= 1e-3 dB and >= 75 dB from the reference in the default mode, <= 1e-4 dB in fp32,
    nothing saturated.  Measured: worst window 9.5e-4 dB (clip 11077 w3) - inside the contract without headroom.  Round 6 found what
    that window is made of: this draw's code prediction COLLAPSES (1 - 4 distinct codes per window, >= 98.9 %
    of the tokens on one code), the decoder's input is a constant field per channel, the half decoder's rounding errors are then
    spatially coherent (a third of the error of a 32 x 32 feature map is a per-channel constant) and the reference's own restoration
    error on that window is a per-channel DC offset (+0.048 in R): a mean error of 1.2e-5 of the build's R output IS 1e-3 dB.  Exact
    weights in every decoder stage leave the window at 1.03e-3 (profiles/r6_b_all_exact_spread.jsonl): not a weight-rounding effect."""
    _psnr_contract_at_point(_point_models(cfg, manifest, 2))


def test_psnr_contract_at_a_fourth_operating_point(cfg, manifest):
    """A FOURTH independent draw (POINTS[3]: weight seed 3, gains re-calibrated x0.91 .. x1.26, tail fitted on clip 13077 w2; 8
    windows of clips 13077 / 14077 / 15077, tests/golden/r6_golden_s3.npz - fixtures from the reference alone; the build ran on them
    for the first time with every constant frozen, profiles/r6_d_spread.jsonl).  This draw's code prediction does NOT collapse (10 -
    13 distinct codes per window, 42 - 64 % of the tokens on the most frequent one; PSNR(reference, GT) 26.1 - 30.9 dB; the
    reference's smallest top-2 margins 3.3e-6 / 5.7e-6): every code equal, 81.5 - 81.7 dB from the reference, worst window 2.6e-4 dB -
    asserted <= 5e-4.  The regime of a trained checkpoint (hundreds of distinct codes) is further on this side."""
    _psnr_contract_at_point(_point_models(cfg, manifest, 3), gate_db=5e-4)


def test_psnr_contract_at_a_fifth_operating_point_with_diverse_codes(cfg, manifest):
    """The regime a trained checkpoint is in: DIVERSE codes.  POINTS[4] (weight seed 4): the random-init code transformer's residual
    branches are scaled by 0.1 (tests/golden/r5_scheme.py: the tokens keep their identity instead of all becoming alike), which gives
    ~100 distinct codes per window instead of 1 - 13; gains re-calibrated on the reference, tail fitted on clip 16077 w2, 8 windows of clips
    16077 / 17077 / 18077 from the REFERENCE (tests/golden/r6_golden_s4.npz), the build run on them with every constant frozen.  Every
    code equal, worst window asserted <= 5e-4 dB (measured: see DESIGN.md section 6)."""
    _psnr_contract_at_point(_point_models(cfg, manifest, 4), gate_db=5e-4)


def test_psnr_contract_at_a_sixth_operating_point_with_diverse_codes(cfg, manifest):
    """The fifth point's scheme at ANOTHER weight seed (POINTS[5]: seed 5, residual branches of the code transformer x0.1, gains re-calibrated on
    the reference, tail fitted on clip 19077 w2; 8 windows of clips 19077 / 20077 / 21077 from the reference, tests/golden/r6_golden_s5.npz;
    65 - 81 distinct codes per window, the most frequent one on 18 - 24 % of the tokens): is the diverse-code figure a property of the regime or of
    one draw?  A gate of 5e-4 dB was written down BEFORE the build first ran on these fixtures, and the first run MISSED it on one window: the
    fitted one, -5.5e-4 dB; the other seven <= 3.8e-4, every code equal, all eight windows with the same sign (mean -2.5e-4: a systematic term
    the compensation leaves at this draw; profiles/r6_q_sixth_point_spread.jsonl).  So the diverse-code points sit at 2.6e-4 / 3.3e-4 / 5.5e-4:
    inside the contract by a factor of two to four, not by the factor the fifth point alone suggested.  Asserted here: the contract (1e-3)."""
    _psnr_contract_at_point(_point_models(cfg, manifest, 5), gate_db=1e-3)


def test_exact_weight_mode_end_to_end(cfg, manifest, tail_models_s1, monkeypatch):
    """The opt-in precision mode (`PGT_EXACT_W` / `ops.EXACT_W_STAGES`: decoder weights on two half planes) through the
    whole model, second operating point, every decoder stage exact, three windows of the reference fixtures: the marked layers really take
    the two-plane launches (counted at `ops.conv2d` / `ops.linear`, which set `pgt_conv_desc::w2`), every code equals the reference's, the contract holds, and the frames
    sit CLOSER to the reference's than the default mode's on every window (measured on 8 windows: 81.1 against 80.7 dB worst,
    profiles/r6_b_all_exact_spread.jsonl)."""
    from pgtformer_amd import PGTFormer, ops
    from pgtformer_amd.synth import make_clip
    from pgtformer_amd.weightgen import generate_state_dict
    from tests.golden.r5_scheme import POINTS, point_state_dict

    monkeypatch.setattr(ops, "EXACT_W_STAGES", ("512", "256", "128", "64", "32"))
    m = PGTFormer(**cfg)
    m.load_state_dict(point_state_dict(generate_state_dict(manifest, cfg, seed=1), 1), strict=True)
    m.prepare(DEV, "x3f16")
    marked = [mod for mod in m.modules() if getattr(mod, "pw2", None) is not None]
    assert len(marked) >= 40, len(marked)
    calls = {"w2": 0, "all": 0}

    def counting(real):
        def f(*a, **k):
            calls["all"] += 1
            calls["w2"] += int(bool(k.get("w2")))      # ops.conv2d / ops.linear put it into pgt_conv_desc::w2
            return real(*a, **k)
        return f

    monkeypatch.setattr(ops, "conv2d", counting(ops.conv2d))
    monkeypatch.setattr(ops, "linear", counting(ops.linear))
    pt = POINTS[1]
    g = np.load(os.path.join(GOLD, pt["golden"]))
    tags = sorted({k.split(".")[0] for k in g.files})[:3]
    default = tail_models_s1["x3f16"]
    recs = []
    for tag in tags:
        seed, i = (int(v) for v in tag[1:].split("w"))
        lq_u8, gt = make_clip(pt["clip_frames"][seed], 512, seed=seed)
        frames = torch.from_numpy(lq_u8[i - 1:i + 2]).to(DEV)
        ref = torch.from_numpy(g[f"{tag}.out_mid_rows"]).double()
        gt_rows = torch.from_numpy(gt[i]).permute(2, 0, 1)[:, ::8, :].double()
        rec = {"window": tag}
        for name, mod in (("exact", m), ("default", default)):
            out, _, _ = mod.forward_nhwc(frames, w=1.0, win=mod.window_index(1, 3, DEV), middle_only=True)
            rows = out[0].float().cpu().permute(2, 0, 1)[:, ::8, :].double()
            codes = mod.last_codes.cpu().numpy().astype(np.int64).reshape(-1)
            rec[name] = {"dpsnr_db": psnr(rows, gt_rows) - psnr(ref, gt_rows), "psnr_build_vs_ref_db": psnr(rows, ref),
                         "differing_tokens": int((codes != g[f"{tag}.codes"].astype(np.int64).reshape(-1)).sum())}
        recs.append(rec)
        assert rec["exact"]["differing_tokens"] == 0, rec
        assert abs(rec["exact"]["dpsnr_db"]) <= 1e-3, rec
        assert rec["exact"]["psnr_build_vs_ref_db"] > rec["default"]["psnr_build_vs_ref_db"], rec
    _LOG["exact_weight_mode/second_point"] = {"windows": recs, "two_plane_launches": calls["w2"], "conv_launches": calls["all"],
                                              "modules_with_two_planes": len(marked)}
    assert calls["w2"] >= 3 * 40, calls


def test_graph_replay_after_the_allocator_returned_memory_to_the_driver(models):
    """a graph replay right after torch.cuda.empty_cache() died ONCE inside the HIP runtime in a long test
    process (the driver then made that release opt-in).  What a captured graph of this build points at: its lane's static input /
    output (held by the runner), tensors allocated DURING capture (the graph's private pool: the allocator never returns the blocks
    of a live graph) and a few tensors created in the eager warm-up that the modules / ops keep for good (ops._FB_COUNTERS, the
    frame-index tables of the fusion blocks, the grouped sub-pixel defects).  This test drives exactly the incident's sequence - capture
    two lanes, eager range-check pass, empty_cache(), replay - 10 times, with a second runner created and destroyed in between (its
    pools become releasable), and checks the replayed frames bit for bit.
    (Later in round 6 the crash became reproducible and was traced to something else - hipGraphLaunch on a graph with a captured fork,
    test_captured_forward_has_no_fork; the empty_cache() call was a bystander.  The sequence stays tested.)"""
    import gc

    from pgtformer_amd.driver import WindowRunner
    from pgtformer_amd.synth import make_clip

    m = models["x3f16"]
    lq, _ = make_clip(6, 512, seed=77)
    frames = torch.from_numpy(lq).to(DEV)
    r = WindowRunner(m, 1.0, True, 512, 512, batch=4, lanes=2, check_range=True)
    want = r.run(frames).clone()
    for i in range(10):
        other = WindowRunner(m, 1.0, True, 512, 512, batch=2, lanes=1, check_range=False)
        other.run(frames[:4])
        del other
        gc.collect()
        bad = m.check_range(frames, w=1.0, win=r.win)          # the eager pass of the incident
        assert bad == []
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        got = r.run(frames)
        torch.cuda.synchronize()
        assert torch.equal(got, want), f"replay {i} after empty_cache() differs"
    _LOG["graph_replay_after_empty_cache"] = {"replays": 10, "bit_equal": True}


def test_captured_forward_has_no_fork(models):
    """The root cause of the segfault inside hipGraphLaunch: ROCm 7.0's hip::Graph::UpdateStreams searches, for every
    root of a graph beyond the first, the executable graph's internal streams for one on another hardware queue than the launch stream - without
    bounding the search (rocgdb: both internal streams on the launch stream's queue, index past the end, null dereference).  A graph with one root
    never enters the loop, so a forward that is being captured must not fork its condition branch onto the side stream; eager forwards keep the
    fork; frames are bit-identical either way."""
    from pgtformer_amd.driver import WindowRunner
    from pgtformer_amd.synth import make_clip

    m = models["x3f16"]
    lq, _ = make_clip(4, 512, seed=78)
    frames = torch.from_numpy(lq).to(DEV)
    eager = m.restore_middle_u8(frames, w=1.0, win=m.window_index(2, 3, DEV)).clone()
    from pgtformer_amd.archs import pgtformer_arch
    assert m.last_forked is pgtformer_arch.SIDE_STREAM      # (PGT_SIDE_STREAM=0 switches the eager fork off as well)
    r = WindowRunner(m, 1.0, True, 512, 512, batch=2, lanes=1, check_range=False)
    assert m.last_forked is False          # (the runner's last forward was the captured one)
    got = r.run(frames)
    torch.cuda.synchronize()
    assert torch.equal(got, eager.reshape(got.shape))


def test_whole_model_pure_bf16_report(models, golden_window):
    """Pure bf16 (opt-in speed mode) is reported, not a parity mode: with random-init weights ~2 % of the codes flip."""
    g = np.load(os.path.join(GOLD, "full_golden.npz"))
    x, _, gt = golden_window
    out, logits, lq, codes = _full(models, "bf16", x)
    assert torch.isfinite(out).all()
    rec = _reduced_record(out, logits, lq, codes, g, gt)
    _LOG["whole/bf16"] = rec
    assert rec["lq_feat_err"] < 0.5 and rec["code_agreement"] > 0.9


@pytest.mark.parametrize("prec", ["bf16", "bf16x3", "x3f16"])
def test_teacher_forced_decoder_and_concat_paths(models, golden_window, prec):
    """Decoder-side arithmetic separated from code flips: the reference's codes are fed in (forward_nhwc(codes=...)), so
    quantiser gather -> AdaIN -> decoder -> SFT fusion (incl. the in-place [enc | dec | fut] concat writes of the bf16
    decoder, archs/pgtformer_arch.py:460-484, :684-710) must reproduce the reference output; and the in-place concat path
    must equal the copying path bit for bit (same kernels, same values, different destinations)."""
    g = np.load(os.path.join(GOLD, "full_golden.npz"))
    x, win_u8, _ = golden_window
    m = models[prec]
    codes = torch.from_numpy(g["codes"].astype(np.int64))
    frames = torch.from_numpy(win_u8).to(DEV)
    out_d, _, _ = m.forward_nhwc(frames, w=1.0, codes=codes)
    out_c, _, _ = m.forward_nhwc(frames, w=1.0, codes=codes, direct=False)
    torch.cuda.synchronize()
    for _ in range(4):      # (round 5: a race between the two streams of a forward showed as a RARE difference here - repeat)
        assert torch.equal(m.forward_nhwc(frames, w=1.0, codes=codes)[0], out_d)
    assert torch.equal(m.last_codes.cpu().reshape(-1), codes.reshape(-1).to(torch.int32))
    same = torch.equal(out_d, out_c)
    dmax = float((out_d.float() - out_c.float()).abs().max())
    out = out_d.float().cpu().permute(0, 3, 1, 2)
    ref_crop = torch.from_numpy(g["out_mid_crop"])
    f16 = torch.from_numpy(g["out_f16"].astype(np.float32))
    rec = {"direct_equals_copying_path": bool(same), "direct_vs_copy_max_abs": dmax,
           "psnr_mid_crop_clamped_db": psnr(out[1, :, 192:320, 192:320].clamp(0, 1), ref_crop.clamp(0, 1)),
           "psnr_full_sub4_clamped_db": psnr(out[:, :, ::4, ::4].clamp(0, 1), f16.clamp(0, 1))}
    _LOG[f"teacher_forced/{prec}"] = rec
    assert same, rec
    floor = {"x3f16": 55.0, "bf16x3": 35.0, "bf16": 30.0}[prec]
    assert rec["psnr_full_sub4_clamped_db"] >= floor, rec


def test_overlap_aware_windows_equal_stacked_windows(models):
    """Per-frame work once per unique frame (forward_nhwc(win=...)) == the stacked-windows forward (the per-frame operators
    act on each frame independently; reference inference.py:47-74)."""
    from pgtformer_amd.synth import make_clip

    lq, _ = make_clip(4, 512, seed=5)
    frames = torch.from_numpy(lq).to(DEV)
    for prec in ("fp32", "bf16x3", "x3f16"):
        m = models[prec]
        win = m.window_index(2, 3, DEV)
        a, la, _ = m.forward_nhwc(frames, w=1.0, win=win)
        b, lb, _ = m.forward_nhwc(frames[win.long()].contiguous(), w=1.0)
        torch.cuda.synchronize()
        d_out, d_log = float((a.float() - b.float()).abs().max()), float((la - lb).abs().max())
        p_db = psnr(a.float().clamp(0, 1).cpu(), b.float().clamp(0, 1).cpu())
        same_codes = bool(torch.equal(la.argmax(-1), lb.argmax(-1)))
        _LOG[f"overlap_vs_stacked/{prec}"] = {"out_max_abs": d_out, "logits_max_abs": d_log, "psnr_db": p_db,
                                              "same_codes": same_codes}
        # the two calls see different frame counts, so tile / split-K choices (hence fp32 summation orders) may differ:
        # fp32 agrees to round-off.  Default mode: identical codes and logits to split-half round-off; the bf16 decoder
        # of this random-init network amplifies flipped bf16 roundings of its inputs to its own noise floor (PSNR(build,
        # reference) is 35.4 dB for the same reason), so the two outputs are held to that floor, not to bit equality
        assert d_log <= 2e-4 and same_codes, (prec, d_log)
        assert (d_out <= 2e-3) if prec == "fp32" else (p_db >= (50.0 if prec == "x3f16" else 33.0)), (prec, d_out, p_db)


def test_parsing_map_and_public_paths_match_host_oracle(models, cfg, full_sd, golden_window):
    """BiSeNet parsing map (the `cond` tap) against the reference golden; forward(code_only=True); w=0 (no fusion) with
    adain=False against the CPU oracle run with the same switches (reference: pgtformer_arch.py:651-653, :670, :699)."""
    from oracle import pgt_oracle as O

    g = np.load(os.path.join(GOLD, "full_golden.npz"))
    x, _, _ = golden_window
    m = models["fp32"]
    logits, lq = m(x.to(DEV), code_only=True)
    par = m.last_parsing.float().cpu()[..., :57].permute(0, 3, 1, 2)
    cond_err = float((par - torch.from_numpy(g["cond_f16"].astype(np.float32))).abs().max())
    assert cond_err <= 2e-3 * max(1.0, float(np.abs(g["cond_f16"]).max())), cond_err
    assert float(np.abs(logits.cpu()[:, :2, :2].numpy() - g["logits_tok0"]).max()) < 5e-3
    out, logits2, _ = m(x.to(DEV), w=0, adain=False)
    o_out, o_logits, _ = O.pgtformer_forward(full_sd, cfg, x, w=0, adain_on=False)
    rec = {"cond_max_abs_err": cond_err, "w0_noadain_out_err": float((out.cpu() - o_out).abs().max()),
           "code_only_logits_equal": bool(torch.equal(logits, logits2))}
    _LOG["public_paths/fp32"] = rec
    assert rec["code_only_logits_equal"]
    assert rec["w0_noadain_out_err"] < 2e-3 * max(1.0, float(o_out.abs().max())), rec
    # the default mode's parsing map: BiSeNet convs on split-half arithmetic, fp32 storage
    mx = models["bf16x3"]
    mx(x.to(DEV), code_only=True)
    parx = mx.last_parsing.float().cpu()[..., :57].permute(0, 3, 1, 2)
    rec["cond_max_abs_err_bf16x3"] = float((parx - torch.from_numpy(g["cond_f16"].astype(np.float32))).abs().max())
    assert rec["cond_max_abs_err_bf16x3"] <= 2e-3 * max(1.0, float(np.abs(g["cond_f16"]).max())), rec


def test_oracle_on_this_host_matches_build_f32(models, cfg, full_sd, golden_window):
    """The CPU oracle, run here on the GPU box's host cores, against the HIP path: checks every
    intermediate the goldens do not carry (BiSeNet map, all encoder feature maps, query embedding)."""
    from oracle import pgt_oracle as O

    x, _, _ = golden_window
    taps = {}
    o_out, o_logits, o_lq = O.pgtformer_forward(full_sd, cfg, x, w=1.0, taps=taps)
    m = models["fp32"]
    out, logits, lq, codes = _full(models, "fp32", x)
    rec = {"out_max_abs_err": float((out - o_out).abs().max()), "logits_max_abs_err": float((logits - o_logits).abs().max()),
           "lq_max_abs_err": float((lq - o_lq).abs().max()),
           "codes_agree": float((codes == taps["codes"].numpy().astype(np.int16)).mean())}
    _LOG["whole/fp32_vs_host_oracle"] = rec
    assert rec["lq_max_abs_err"] < 1e-3 and rec["logits_max_abs_err"] < 5e-3


@pytest.mark.parametrize("seed", [1, 2])
def test_default_mode_codes_with_other_weights_and_inputs(cfg, seed):
    """The 100 % code agreement of the default mode is not a property of weight seed 0 / the golden window: other
    random-init weights and another window of the synthetic clip, against the CPU oracle run here on the GPU box's host.
    Every code must equal the oracle's except where the oracle's own top-2 logit margin is below 1e-3."""
    from oracle import pgt_oracle as O
    from pgtformer_amd import PGTFormer
    from pgtformer_amd.manifest import pgtformer_manifest
    from pgtformer_amd.synth import make_clip, window_from_clip
    from pgtformer_amd.weightgen import generate_state_dict

    sd = generate_state_dict(pgtformer_manifest(cfg), cfg, seed=seed)
    lq_u8, _ = make_clip(5, 512, seed=4321 + seed)
    win = window_from_clip(lq_u8, 2 + seed % 2)
    x = torch.from_numpy(win.astype(np.float32) / np.float32(255.0)).permute(0, 3, 1, 2).contiguous()
    taps = {}
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    o_out, o_logits, o_lq = O.pgtformer_forward(sd, cfg, x, w=1.0, taps=taps)
    m = PGTFormer(**cfg)
    m.load_state_dict(sd, strict=True)
    m.prepare(DEV)                       # the default mode
    out, logits, lq = m(x.to(DEV))
    codes = m.last_codes.cpu().numpy().reshape(-1)
    want = taps["codes"].numpy().reshape(-1)
    top2 = o_logits.reshape(-1, o_logits.shape[-1]).topk(2, -1).values
    margin = (top2[:, 0] - top2[:, 1]).numpy()
    bad = codes != want
    d = (out.float().cpu().clamp(0, 1) - o_out.clamp(0, 1))
    rec = {"code_agreement": float(1.0 - bad.mean()), "mismatches": int(bad.sum()),
           "max_mismatch_margin": float(margin[bad].max()) if bad.any() else 0.0, "min_margin": float(margin.min()),
           "logits_err": float((logits.cpu() - o_logits).abs().max()),
           "psnr_db": float(-10.0 * torch.log10(d.pow(2).mean()))}
    _LOG[f"whole/x3f16_seed{seed}_vs_host_oracle"] = rec
    assert rec["max_mismatch_margin"] < 1e-3 and rec["code_agreement"] >= 0.997, rec
    assert rec["logits_err"] < 2e-3, rec
    if not bad.any():
        assert rec["psnr_db"] >= 34.0, rec


def test_stage1_rqvae_matches_reference(models, golden_window):
    from pgtformer_amd.archs.tdcrqvae3_arch import TDCRQVAE3

    g = np.load(os.path.join(GOLD, "full_golden.npz"))
    x, _, _ = golden_window
    out, _, codes = TDCRQVAE3.forward(models["fp32"], x.to(DEV))
    codes = codes.cpu().numpy().astype(np.int16)
    agree = float((codes == g["stage1_codes"]).mean())
    err = float(np.abs(out[1, :, 192:320, 192:320].cpu().numpy() - g["stage1_out_mid_crop"]).max())
    _LOG["stage1/fp32"] = {"code_agreement": agree, "max_abs_err_crop": err}
    assert agree >= 0.999
    if agree == 1.0:
        assert err < 2e-2


def test_batched_windows_equal_separate_forwards(models):
    """B windows per forward == B separate forwards (the reference only accepts B=1, rstt_layers.py:904):
    f32 to accumulation-order round-off; and the clip runner (batch 2, ragged tail, replicate-padded halos)
    reproduces the per-window driver policy."""
    from pgtformer_amd.driver import WindowRunner
    from pgtformer_amd import parallel
    from pgtformer_amd.synth import make_clip

    m = models["fp32"]
    lq, _ = make_clip(3, 512, seed=77)
    clip = torch.from_numpy(lq).to(DEV)
    padded = parallel.padded_local_clip(clip, 0, 1)          # [f0, f0, f1, f2, f2]
    wins = [padded[j:j + 3] for j in range(3)]
    sep = [m.forward_nhwc(w, w=1.0)[0].float() for w in wins]
    both = m.forward_nhwc(torch.cat(wins[:2], 0).contiguous(), w=1.0)[0].float()
    err = max((both[:3] - sep[0]).abs().max().item(), (both[3:] - sep[1]).abs().max().item())
    _LOG["batch2_vs_separate_f32"] = {"max_abs_err": err}
    assert err < 2e-3
    runner = WindowRunner(m, 1.0, use_graph=False, batch=2)
    out = runner.run_clip(padded, torch.empty_like(clip))
    for j in range(3):
        want = (sep[j][1].clamp(0, 1) * 255).to(torch.uint8)
        assert (out[j].int() - want.int()).abs().max().item() <= 1, j


def test_large_batch_chunks_inputs_over_2gib(models):
    """16 windows per forward: the 128-channel 512x512 tensors exceed 2 GiB (32-bit gather offsets), ops.conv2d then runs
    frame chunks; first / middle / last window equal their single-window forwards (fp32: round-off only)."""
    from pgtformer_amd.synth import make_clip

    m = models["fp32"]
    lq, _ = make_clip(18, 512, seed=78)
    clip = torch.from_numpy(lq).to(DEV)
    wins = [clip[j:j + 3] for j in range(16)]
    both = m.forward_nhwc(torch.cat(wins, 0).contiguous(), w=1.0)[0].float()
    err = 0.0
    for j in (0, 9, 15):
        sep = m.forward_nhwc(wins[j].contiguous(), w=1.0)[0].float()
        err = max(err, (both[3 * j:3 * j + 3] - sep).abs().max().item())
    _LOG["batch16_vs_separate_f32"] = {"max_abs_err": err}
    assert err < 2e-3


def test_middle_only_tail_equals_full_forward(models):
    """The driver keeps the middle frame of every window (inference.py:15).  forward_nhwc(middle_only=True) stops computing
    the two outer frames after the decoder's last temporal operation (the 256x256 fusion block's temporal mix; with w=0 the
    128x128 EncoderLayers): its frames equal the middle frames of the full forward - fp32 to round-off, default mode to the
    bf16 decoder's noise floor with identical codes."""
    from pgtformer_amd.synth import make_clip

    lq, _ = make_clip(4, 512, seed=9)
    frames = torch.from_numpy(lq).to(DEV)
    for prec, wv in (("fp32", 1.0), ("fp32", 0), ("bf16x3", 1.0), ("x3f16", 1.0)):
        m = models[prec]
        win = m.window_index(2, 3, DEV)
        full, lf, _ = m.forward_nhwc(frames, w=wv, win=win)
        mid, lm, _ = m.forward_nhwc(frames, w=wv, win=win, middle_only=True)
        torch.cuda.synchronize()
        assert mid.shape == (2, 512, 512, 3) and full.shape == (6, 512, 512, 3)
        want = full[1::3].float()
        d = float((mid.float() - want).abs().max())
        p_db = psnr(mid.float().clamp(0, 1).cpu(), want.clamp(0, 1).cpu())
        _LOG[f"middle_only/{prec}/w{wv}"] = {"max_abs": d, "psnr_db": p_db, "logits_equal": bool(torch.equal(lf, lm))}
        assert torch.equal(lf, lm)
        assert (d <= 2e-3) if prec == "fp32" else (p_db >= (50.0 if prec == "x3f16" else 33.0)), (prec, wv, d, p_db)


def test_driver_119_frame_clip_through_the_real_model(models):
    """BASELINE configs[0] geometry (119 frames of 512x512 rgb24) through the REAL model and the pipelined host driver
    (pinned clip, 16 windows per forward, ragged tail batch, HIP graph): 119 frames in -> 119 out, and frames of the first,
    an interior and the last window equal their single-window forwards (fp32; u8 +-1)."""
    from pgtformer_amd.driver import WindowRunner, restore_clip_host
    from pgtformer_amd.synth import make_clip

    m = models["fp32"]
    base, _ = make_clip(7, 512, seed=11)
    clip = np.concatenate([base] * 17, 0)[:119]
    padded = torch.empty((121, 512, 512, 3), dtype=torch.uint8).pin_memory()
    padded[1:120].copy_(torch.from_numpy(clip))
    out = torch.empty((119, 512, 512, 3), dtype=torch.uint8).pin_memory()
    runner = WindowRunner(m, 1.0, use_graph=True, batch=16)
    restore_clip_host(runner, padded, out)
    torch.cuda.synchronize()
    assert np.array_equal(padded[0].numpy(), clip[0]) and np.array_equal(padded[120].numpy(), clip[118])   # replicate halos
    worst = 0
    for j in (0, 57, 118):
        tri = [max(j - 1, 0), j, min(j + 1, 118)]
        single = m.restore_middle_u8(torch.from_numpy(clip[tri]).to(DEV), w=1.0).cpu()
        worst = max(worst, int((out[j].int() - single.int()).abs().max()))
    _LOG["driver_119_frames/fp32"] = {"max_u8_diff_vs_single_window": worst}
    assert worst <= 1
    # two and three forwards in flight (one HIP graph per lane, batches dealt round-robin): the same frames, bit for bit,
    # from the pinned-host pipeline and from a device-resident clip
    for lanes in (2, 3):
        r2 = WindowRunner(m, 1.0, use_graph=True, batch=16, lanes=lanes)
        out2 = torch.empty((119, 512, 512, 3), dtype=torch.uint8).pin_memory()
        restore_clip_host(r2, padded, out2)
        torch.cuda.synchronize()
        assert torch.equal(out, out2), lanes
        out3 = torch.zeros((119, 512, 512, 3), dtype=torch.uint8, device=DEV)
        r2.run_clip(padded.to(DEV), out3)
        torch.cuda.synchronize()
        assert torch.equal(out, out3.cpu()), lanes
        del r2


def test_cli_streams_a_raw_clip_end_to_end(models, tmp_path):
    """`python -m pgtformer_amd.driver -i clip.rgb -o out.rgb --synthetic` (the counterpart of `python inference.py -i .. -o ..`):
    decode in chunks -> segments with 1-frame halos -> pipelined restore -> sink; 21 frames in segments of 16, 8 windows per
    forward, against one pass of the in-process driver over the whole clip (same synthetic weights; u8 within +-1).
    Kernel selection is deterministic (static heuristic; timing-based tuning is opt-in), so two PROCESSES write bit-identical
    frames - in fp32 and in the default mode - as two runs of the reference do (make_golden.py: repeat diff 0)."""
    import subprocess
    import sys

    from pgtformer_amd.driver import WindowRunner, restore_clip_host
    from pgtformer_amd.synth import make_clip

    base, _ = make_clip(7, 512, seed=21)
    clip = np.concatenate([base] * 3, 0)
    src = tmp_path / "clip.rgb"
    src.write_bytes(clip.tobytes())
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("PGT_AUTOTUNE", "PGT_AUTOTUNE_CACHE")}

    def cli(prec, tag):
        dst = tmp_path / f"out_{prec}_{tag}" / "restored.rgb"
        r = subprocess.run([sys.executable, "-m", "pgtformer_amd.driver", "-i", str(src), "-o", str(dst), "--synthetic", "--batch", "8",
                            "--segment", "16", "--precision", prec], cwd=repo, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
        assert r.returncode == 0, r.stdout.decode()[-2000:]
        return np.frombuffer(dst.read_bytes(), np.uint8).reshape(21, 512, 512, 3)

    for prec in ("fp32", "x3f16"):
        got, again = cli(prec, "a"), cli(prec, "b")
        assert np.array_equal(got, again), f"{prec}: two processes wrote different frames"
        m = models[prec]
        padded = torch.empty((23, 512, 512, 3), dtype=torch.uint8).pin_memory()
        padded[1:22].copy_(torch.from_numpy(clip))
        want = torch.empty((21, 512, 512, 3), dtype=torch.uint8).pin_memory()
        restore_clip_host(WindowRunner(m, 1.0, use_graph=True, batch=8, lanes=2), padded, want)
        torch.cuda.synchronize()
        d = np.abs(got.astype(np.int16) - want.numpy().astype(np.int16))
        _LOG[f"cli_stream/{prec}"] = {"max_u8_diff": int(d.max()), "equal_fraction": float((d == 0).mean()),
                                      "two_processes_bit_equal": True}
        assert d.max() <= 1 and (d == 0).mean() > 0.99, _LOG[f"cli_stream/{prec}"]


def test_range_telemetry_reports_saturated_half_stores(tail_models):
    """The IEEE-half modes clamp at +-65504 instead of producing inf: PGTFormer.check_range counts, per
    operator, the half outputs that sit at the limit.  On the operating-point windows NOTHING saturates (asserted: the PSNR
    contract figures above are not figures of a clamped decoder); a decoder whose conv_in weights are scaled by 1e5 is reported,
    and the driver refuses to run it."""
    from pgtformer_amd import hip
    from pgtformer_amd.driver import WindowRunner
    from pgtformer_amd.synth import make_clip

    m = tail_models["x3f16"]
    lq_u8, _ = make_clip(4, 512, seed=1234)
    fr = torch.from_numpy(lq_u8).to(DEV)
    bad = m.check_range(fr, w=1.0, win=m.window_index(2, 3, DEV))
    _LOG["range_telemetry/x3f16"] = {"tensors_checked": m.last_range_launches, "saturating": [list(map(str, b)) for b in bad]}
    assert m.last_range_launches > 300 and bad == [], bad
    # a deliberately out-of-range decoder
    conv = m.decoder.conv_in
    keep = conv.weight.detach().clone()
    try:
        with torch.no_grad():
            conv.weight.mul_(1e5)
        conv._pack(m.dev, conv.dt)
        bad = m.check_range(fr, w=1.0, win=m.window_index(2, 3, DEV))
        assert bad and all(c > 0 for _, _, c in bad), bad
        runner = WindowRunner(m, use_graph=False, batch=2, check_range=True)
        with pytest.raises(hip.PgtError, match="half range"):
            runner.run(fr)
    finally:
        with torch.no_grad():
            conv.weight.copy_(keep)
        conv._pack(m.dev, conv.dt)
    assert m.check_range(fr, w=1.0, win=m.window_index(2, 3, DEV)) == []


def test_configs2_clip_through_rccl_at_one_gpu(tmp_path):
    """BASELINE.json configs[2] (256-frame clip, frame-range shard, ONE all_gather of boundary frames, restored frames gathered to
    rank 0) at N = 1 through the REAL collective library: bench.py --clip-frames 256 under torchrun --nproc-per-node 1 with the
    `nccl` backend (= RCCL) and PGT_FORCE_COLLECTIVE=1, so that the all_gather / gather / barrier / all_reduce of the path run on a
    1-rank RCCL communicator - against the same job in a plain process (no process group): the restored clips are bit-equal (sha256
    over the 256 uint8 frames).  What this cannot show: more than one rank (one GPU per box; the gloo world-2 / world-8 tests of
    tests/test_abi_and_parallel.py cover the sharding logic)."""
    import subprocess
    import sys

    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ["bench.py", "--gpus", "1", "--clip-frames", "256", "--steps", "1", "--warmup", "1"]
    env = dict(os.environ, PGT_RANGE_CHECK="0")
    a, b = str(tmp_path / "rccl.json"), str(tmp_path / "plain.json")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", "29541"] + common + ["--dump-clip", a], cwd=repo, env=dict(env, PGT_FORCE_COLLECTIVE="1"),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["scaling"] == "strong" and line["config"]["clip_frames"] == 256 and line["n_gpus"] == 1
    r2 = subprocess.run([sys.executable] + common + ["--dump-clip", b], cwd=repo, env=env, capture_output=True, text=True, timeout=600)
    assert r2.returncode == 0, r2.stderr[-2000:]
    ja, jb = json.load(open(a)), json.load(open(b))
    _LOG["configs2_n1_rccl"] = {"rccl": ja, "plain": jb, "frames_per_s_rccl": line["value"],
                                "frames_per_s_plain": json.loads(r2.stdout.strip().splitlines()[-1])["value"]}
    assert ja["collectives"] == "RCCL" and jb["collectives"] == "none" and ja["frames"] == jb["frames"] == 256
    assert ja["sha256"] == jb["sha256"]


@pytest.mark.parametrize("prec", ["fp32", "x3f16"])
def test_exported_program_replays_bit_equal_from_python_and_from_c(models, prec, tmp_path):
    """The whole-graph entry (pgt_program_load / _run / _destroy; SURVEY section 8b `pgt_forward_window`): one forward of the
    prepared model on 2 sliding windows is recorded as a tape of C-ABI calls (pgtformer_amd/export.py), then replayed
      (a) through the library from this process on FRESH buffers (another workspace address, other frames than the recorded ones),
      (b) by a C host that links libpgt_hip.so and the HIP runtime only (tests/c/program_smoke.c),
    and both equal the Python host's restore_middle_u8 BIT FOR BIT - in fp32 and in the default mode (same kernels, same order,
    fixed-order reductions)."""
    import subprocess

    from pgtformer_amd.export import export_program, run_program
    from pgtformer_amd.synth import make_clip

    m = models[prec]
    path = str(tmp_path / f"pgt_{prec}.prog")
    info = export_program(m, 2, path)
    assert info["calls"] > 400 and info["workspace_bytes"] > 0 and os.path.getsize(path) > info["persistent_bytes"]
    # (a) the recorded input reproduces the recorded output; other frames reproduce the Python host's result for them
    got = run_program(path, info["input"])
    assert torch.equal(got.reshape(info["output"].shape), info["output"])
    lq_u8, _ = make_clip(4, 512, seed=4321)
    frames = torch.from_numpy(lq_u8).to(DEV)
    want = m.restore_middle_u8(frames, w=1.0, win=m.window_index(2, 3, DEV))
    pad = torch.empty(12345, dtype=torch.uint8, device=DEV)      # (moves the next allocations: the replay must not depend on addresses)
    got = run_program(path, frames).reshape(want.shape)
    del pad
    assert torch.equal(got, want), int((got.int() - want.int()).abs().max())
    # (b) a C host
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "program_smoke")
    lib_dir = os.path.join(repo, "pgtformer_amd", "lib")
    cc = subprocess.run(["gcc", "-D__HIP_PLATFORM_AMD__", os.path.join(repo, "tests", "c", "program_smoke.c"), "-I", os.path.join(repo, "include"),
                         "-I", "/opt/rocm/include", "-L", lib_dir, "-lpgt_hip", "-L", "/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + lib_dir,
                         "-Wl,-rpath,/opt/rocm/lib", "-o", exe], capture_output=True, text=True, timeout=300)
    assert cc.returncode == 0, cc.stderr[-2000:]
    fin, fout = str(tmp_path / "in.u8"), str(tmp_path / "out.u8")
    lq_u8.tofile(fin)
    r = subprocess.run([exe, path, fin, fout, "3"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-2000:])
    c_out = torch.from_numpy(np.fromfile(fout, dtype=np.uint8).reshape(tuple(want.shape)))
    _LOG[f"exported_program/{prec}"] = {"calls": info["calls"], "persistent_mb": round(info["persistent_bytes"] / 1e6, 1),
                                        "workspace_mb": round(info["workspace_bytes"] / 1e6, 1), "c_host_stdout": r.stdout.strip().splitlines()[-3:]}
    assert torch.equal(c_out, want.cpu()), int((c_out.int() - want.cpu().int()).abs().max())
