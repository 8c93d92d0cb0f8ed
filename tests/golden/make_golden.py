"""Generate golden fixtures by running the REFERENCE (imported from /root/reference) in the build
container. Usage:  python tests/golden/make_golden.py

Writes (all small, committed):
  tests/golden/state_dict_manifest.json   961 key -> [shape, dtype] of reference PGTFormer.state_dict()
  tests/golden/ops_golden.npz             outputs of reference sub-modules for tests/golden/cases.py
  tests/golden/full_golden.npz            whole-model outputs on the synthetic window (codes, crops,
                                          statistics, BiSeNet condition map) + TDCRQVAE3 stage-I outputs
The reference cannot travel to the GPU box; only these data files and this script are committed.
Import stubs for the absent basicsr/timm/torchvision live in tests/golden/_refshim (build-owned).
"""
import hashlib
import json
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(HERE, "_refshim"))
sys.path.insert(0, HERE)
REF = "/root/reference"


def main():
    os.chdir(REF)
    sys.path.insert(0, REF)
    from archs.pgtformer_arch import PGTFormer  # noqa: E402  (reference)
    from archs.tdcrqvae3_arch import TDCRQVAE3  # noqa: E402
    from archs.codeformer_arch import adaptive_instance_normalization  # noqa: E402

    import cases  # noqa: E402
    from pgtformer_amd.config import default_config
    from pgtformer_amd.manifest import pgtformer_manifest
    from pgtformer_amd.synth import make_clip, window_from_clip
    from pgtformer_amd.weightgen import generate_state_dict
    from oracle import pgt_oracle as O

    torch.manual_seed(0)
    cfg = default_config()
    model = PGTFormer(**cfg)
    model.eval()
    ref_sd = model.state_dict()
    with open(os.path.join(HERE, "state_dict_manifest.json"), "w") as f:
        json.dump({k: [list(v.shape), str(v.dtype)] for k, v in ref_sd.items()}, f)
    manifest = pgtformer_manifest(cfg)
    sd = generate_state_dict(manifest, cfg, seed=0)
    # buffers the reference computes itself must agree with the restated generator
    for k, v in ref_sd.items():
        if k.endswith("relative_position_index"):
            assert torch.equal(v, sd[k]), k
    missing = model.load_state_dict(sd, strict=True)
    print("load_state_dict:", missing)

    # ---------------- per-function cases ----------------
    ops = {}
    with torch.no_grad():
        for name, (kind, prefix, shape, seed) in cases.CASES.items():
            x = cases.case_inputs(name)
            mod = model.get_submodule(prefix) if prefix else None
            if kind == "resblock":
                outs = [mod(x[0].clone(), None)]
            elif kind in ("downsample", "upsample", "enclayer"):
                outs = [mod(x[0].clone())]
            elif kind == "salayer":
                outs = [mod(x[0].clone(), query_pos=x[1])]
            elif kind == "fuse":
                outs = [mod(x[0].clone(), x[1].clone(), temb=None, w=1.0)]
            elif kind == "adain":
                outs = [adaptive_instance_normalization(x[0], x[1])]
            elif kind == "embed":
                outs = [mod.embed_code(x[0])[:, ::4, ::4]]
            elif kind == "rq":
                ql, codes = mod.quantize(x[0])
                outs = [ql[-1], codes]
            ora = cases.run_oracle(name, sd)
            for i, (a, b) in enumerate(zip(outs, ora)):
                err = (a.double() - b.double()).abs().max().item()
                print(f"case {name}[{i}] shape {tuple(a.shape)} ref-vs-oracle max|d| = {err:.3e}")
                ops[f"{name}.{i}"] = a.numpy()
    np.savez_compressed(os.path.join(HERE, "ops_golden.npz"), **ops)

    # ---------------- whole model ----------------
    lq_u8, gt = make_clip(4, 512, seed=1234)
    win = window_from_clip(lq_u8, 1)
    x = torch.from_numpy(win.astype(np.float32) / np.float32(255.0)).permute(0, 3, 1, 2).contiguous()
    cond_holder = {}
    hk = model.conditionnet.register_forward_hook(lambda m, i, o: cond_holder.__setitem__("cond", o))
    with torch.no_grad():
        t0 = time.time()
        out, logits, lq_feat = model(x.clone(), w=1.0)
        t_ref = time.time() - t0
        t0 = time.time()
        out2 = model(x.clone(), w=1.0)[0]
        t_ref2 = time.time() - t0
    hk.remove()
    print(f"reference forward: {t_ref:.1f}s first, {t_ref2:.1f}s second; repeat diff "
          f"{(out - out2).abs().max().item():.3e}")
    taps = {}
    t0 = time.time()
    o_out, o_logits, o_lq = O.pgtformer_forward(sd, cfg, x, w=1.0, taps=taps)
    print(f"oracle forward: {time.time() - t0:.1f}s")
    for nm, a, b in (("out", out, o_out), ("logits", logits, o_logits), ("lq_feat", lq_feat, o_lq),
                     ("cond", cond_holder["cond"], taps["cond"])):
        print(f"full {nm}: shape {tuple(a.shape)} ref-vs-oracle max|d| = "
              f"{(a.double() - b.double()).abs().max().item():.3e}  (ref absmax {a.abs().max().item():.3f})")
    codes = logits.argmax(-1)
    print("codes equal:", torch.equal(codes, taps["codes"]))
    top2 = logits.reshape(-1, logits.shape[-1]).topk(2, -1).values
    margin = (top2[:, 0] - top2[:, 1])
    print("top-2 logit margin: min %.3e  median %.3e" % (margin.min().item(), margin.median().item()))
    outc = out.clamp(0, 1)
    mse = ((outc - torch.from_numpy(gt[[0, 1, 2]]).permute(0, 3, 1, 2)) ** 2).mean().item()
    print("PSNR(ref out, GT) = %.3f dB ; out range [%.3f, %.3f]" % (
        -10 * np.log10(mse), out.min().item(), out.max().item()))

    # stage-I path (config 4) on the same input
    with torch.no_grad():
        t0 = time.time()
        s_out, s_loss, s_codes = TDCRQVAE3.forward(model, x.clone())
        print(f"reference TDCRQVAE3 forward: {time.time() - t0:.1f}s")
    so_out, so_loss, so_codes = O.tdcrqvae3_forward(sd, cfg, x)
    print("stage-I out max|d| = %.3e, loss d = %.3e, codes equal: %s" % (
        (s_out - so_out).abs().max().item(), abs(s_loss.item() - so_loss.item()),
        torch.equal(s_codes, so_codes)))

    full = {
        "out_mid_crop": out[1, :, 192:320, 192:320].numpy(),
        "out_stats": np.array([[o.mean().item(), o.std().item(), o.min().item(), o.max().item()]
                               for o in out], np.float64),
        "out_sha256": np.frombuffer(hashlib.sha256(out.numpy().tobytes()).digest(), np.uint8),
        "out_f16": out.numpy().astype(np.float16)[:, :, ::4, ::4],
        "codes": codes.numpy().astype(np.int16),
        "logit_margin": margin.numpy().astype(np.float32),
        "logits_tok0": logits[:, :2, :2].numpy(),
        "lq_feat_crop": lq_feat[:, 12:20, 12:20, :].numpy(),
        "cond_f16": cond_holder["cond"].numpy().astype(np.float16),
        "stage1_codes": s_codes.numpy().astype(np.int16),
        "stage1_out_mid_crop": s_out[1, :, 192:320, 192:320].numpy(),
        "stage1_loss": np.array([s_loss.item()], np.float64),
        "ref_forward_seconds": np.array([t_ref2]),
    }
    np.savez_compressed(os.path.join(HERE, "full_golden.npz"), **full)
    for fn in ("ops_golden.npz", "full_golden.npz", "state_dict_manifest.json"):
        print(fn, os.path.getsize(os.path.join(HERE, fn)) // 1024, "KiB")


if __name__ == "__main__":
    main()
