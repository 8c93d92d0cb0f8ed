"""Fixtures of SURVEY §8 row (f)4 from the REFERENCE (imported from /root/reference in the build container):
    python tests/golden/make_golden_r2b.py  ->  tests/golden/r2b_golden.npz

  * archs/tdcrqvae3_arch.py: VQEmbedding in TRAINING mode (forward :188-199 = find_nearest_embedding, _update_buffers,
    embed, _update_embedding) for cases_r2b.EMA, several consecutive steps.  The random draws the reference makes for the
    restart of unused codes (torch.rand_like in _tile_with_noise, torch.randperm) are recorded by replaying the same
    generator state, so that the oracle and the HIP path can be fed the same permutation / noise.
  * modules/swin.py: BasicLayer.forward for cases_r2b.LAYER.
Import stubs for the absent basicsr / timm / torchvision / mmcv live in tests/golden/_refshim (build-owned).
The script also checks the oracle restatements against the reference outputs and prints the differences."""
import os
import sys

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
    from modules import swin as RS                                   # reference
    from archs.tdcrqvae3_arch import VQEmbedding                     # reference

    import cases_r2b as C
    from oracle import pgt_oracle as O

    out = {}
    with torch.no_grad():
        for name, (k, d, n, decay, restart, steps, seed) in C.EMA.items():
            w0, batches = C.ema_case(name)
            vq = VQEmbedding(k, d, ema=True, decay=decay, restart_unused_codes=restart)
            vq.weight.data.copy_(w0)
            vq.embed_ema.copy_(w0[:-1])
            vq.cluster_size_ema.zero_()
            vq.train()
            ow, ocs, oem = w0.clone(), torch.zeros(k), w0[:-1].clone()
            for s, x in enumerate(batches):
                torch.manual_seed(seed + 100 + s)
                state = torch.get_rng_state()
                embeds, idxs = vq(x)                                  # reference training step
                # replay the draws of this step: rand_like (only when n < K) then randperm
                torch.set_rng_state(state)
                noise = None
                nv = n
                if restart:
                    if n < k:
                        rep = (k + n - 1) // n
                        noise = torch.rand_like(x.repeat(rep, 1))
                        nv = rep * n
                    perm = torch.randperm(nv)
                else:
                    perm = None
                # oracle on the same assignments and draws
                d2 = O._distances({"quantizer.codebooks.0.weight": ow}, x, 0)
                oidx = d2.argmin(-1)
                assert torch.equal(oidx, idxs), (name, s)
                oemb = ow[oidx]
                ow, ocs, oem = O.vq_ema_step(ow, ocs, oem, x, oidx, decay, 1e-5, restart, perm, noise)
                print(f"{name} step {s}: embeds max|d| = {(embeds - oemb).abs().max().item():.2e}  weight "
                      f"{(vq.weight - ow).abs().max().item():.2e}  cluster_size_ema {(vq.cluster_size_ema - ocs).abs().max().item():.2e}  "
                      f"embed_ema {(vq.embed_ema - oem).abs().max().item():.2e}  restarted {(int((ocs == 1).sum()))}")
                out[f"{name}.{s}.idxs"] = idxs.numpy().astype(np.int32)
                out[f"{name}.{s}.weight"] = vq.weight.detach().numpy().copy()
                out[f"{name}.{s}.cluster_size_ema"] = vq.cluster_size_ema.numpy().copy()
                out[f"{name}.{s}.embed_ema"] = vq.embed_ema.numpy().copy()
                if perm is not None:
                    out[f"{name}.{s}.perm"] = perm[:k].numpy().astype(np.int32)
                if noise is not None:
                    out[f"{name}.{s}.noise"] = noise.numpy()
        for name, (dim, depth, heads, ws, fmap, mlp_ratio, qkv_bias, seed) in C.LAYER.items():
            layer = RS.BasicLayer(dim, depth, heads, window_size=ws, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias)
            layer.eval()
            p = C.layer_params(name)
            sd = layer.state_dict()
            for kk, v in p.items():
                assert sd[kk].shape == v.shape, (kk, sd[kk].shape, v.shape)
                sd[kk] = v
            layer.load_state_dict(sd, strict=True)
            x = C.layer_input(name)
            y = layer(x)
            yo = O.swin_basic_layer(p, x, depth, heads, ws)
            print(f"{name}: ref-vs-oracle max|d| = {(y - yo).abs().max().item():.3e}  (absmax {y.abs().max().item():.3f})")
            out[f"{name}.out"] = y[:, :C.KEEP[name]].numpy()                  # leading channels only (fixture size)
    np.savez_compressed(os.path.join(HERE, "r2b_golden.npz"), **out)
    print("r2b_golden.npz", os.path.getsize(os.path.join(HERE, "r2b_golden.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
