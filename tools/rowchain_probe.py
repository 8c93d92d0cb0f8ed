"""Where the time of the fused token-row chains goes: builds pgtformer_amd/csrc/rowchain.hip with -DRC_PROBE=<bits> (parts of the
kernel switched off, see the probe hooks at the top of that file) into throw-away libraries and times both chains at the
model's shape.  Results of probe builds are WRONG by construction; only their timing is read.
Usage (GPU box): python tools/rowchain_probe.py [--rows 1572864] [--probes 0 1 2 ...] [--define NAME=VAL ...]"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import tempfile

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, "pgtformer_amd", "csrc")


def build(bits, defines, tmp):
    lib = os.path.join(tmp, "librc_%d_%s.so" % (bits, "_".join(defines).replace("=", "")))
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-I", os.path.join(REPO, "include"),
           "-I", CSRC, "-ffp-contract=on", "-Wno-unused-result", "-DRC_PROBE=%d" % bits] + ["-D" + d for d in defines] + \
          ["-x", "hip", os.path.join(CSRC, "rowchain.hip"), os.path.join(CSRC, "capi.cpp"), "-o", lib]
    subprocess.check_call(cmd)
    return C.CDLL(lib)


def timeit(fn, iters):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1572864)
    ap.add_argument("--iters", type=int, default=8)
    ap.add_argument("--probes", type=int, nargs="*", default=[0, 1, 2, 3, 4, 8, 16, 32, 64, 128])
    ap.add_argument("--define", nargs="*", default=[])
    ap.add_argument("--ln", default=None, help="PGT_RC_LN variant")
    ap.add_argument("--mlp", default=None, help="PGT_RC_MLP variant")
    a = ap.parse_args()
    if a.ln:
        os.environ["PGT_RC_LN"] = a.ln
    if a.mlp:
        os.environ["PGT_RC_MLP"] = a.mlp
    dev, H = "cuda", torch.float16
    rows = a.rows
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.randn((rows, 256), device=dev, dtype=H, generator=g)
    sc = torch.randn((rows, 256), device=dev, dtype=H, generator=g)
    wq = (0.06 * torch.randn((768, 256), device=dev, generator=g)).to(H)
    b768 = torch.randn((768,), device=dev, generator=g)
    b1, b2 = torch.randn((256,), device=dev, generator=g), torch.randn((256,), device=dev, generator=g)
    qkv = torch.empty((rows, 768), device=dev, dtype=H)
    out = torch.empty((rows, 256), device=dev, dtype=H)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    vp = lambda t: C.c_void_p(t.data_ptr())   # noqa: E731
    with tempfile.TemporaryDirectory() as tmp:
        for bits in a.probes:
            L = build(bits, a.define, tmp)

            def ln():
                rc = L.pgt_ln_linear(3, vp(x), 256, rows, 256, C.c_float(1e-5), vp(wq), vp(b768), 0, 768, vp(qkv), 768, st)
                assert rc == 0, rc

            def mlp():
                rc = L.pgt_attn_proj_mlp(3, vp(x), 256, vp(sc), 256, rows, 256, vp(wq), vp(b768), 0, vp(b1), vp(b2), C.c_float(1e-5),
                                         vp(out), 256, st)
                assert rc == 0, rc
            rec = {"probe": bits, "defines": a.define, "ln": a.ln or "auto", "mlp": a.mlp or "w8", "rows": rows,
                   "ln_linear_us": round(timeit(ln, a.iters), 1), "proj_mlp_us": round(timeit(mlp, a.iters), 1)}
            if bits & 256:      # effective shader clock of each kernel: s_memtime ticks / (100 MHz ticks * 10 ns)
                for name, fn, buf in (("ln_linear", ln, qkv), ("proj_mlp", mlp, out)):
                    for _ in range(3):
                        fn()
                    torch.cuda.synchronize()
                    t = buf.view(-1)[:8].view(torch.int64).cpu().tolist()
                    rec[name + "_ghz"] = round(t[0] / (t[1] * 10.0), 3)
                    rec[name + "_wg0_us"] = round(t[1] / 100.0, 1)
            print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
