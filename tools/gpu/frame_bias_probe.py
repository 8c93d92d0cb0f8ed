"""pgt_frame_bias at the decoder's shapes: time and a checksum of the bias rows (run under two builds of the library through PGT_LIB_PATH to
compare them bit for bit)."""
import sys, os, json, hashlib, torch
sys.path.insert(0, os.getcwd())
from pgtformer_amd import ops
from tools.bench_micro import timeit
torch.manual_seed(0)
lib = os.environ.get("PGT_LIB_PATH", "shipping/x").split("/")[-2]
for (n, hw, k, cout, aff, groups) in ((1536, 64, 512, 512, False, 1), (1536, 64, 512, 512, True, 1), (1536, 64, 1056, 512, False, 1), (1536, 256, 256, 256, True, 1),
                                     (1536, 256, 544, 256, False, 1), (1536, 1024, 256, 256, True, 1), (1536, 4096, 128, 128, True, 1), (512, 16384, 64, 64, True, 1),
                                     (1536, 64, 512, 1024, False, 4), (96, 1024, 512, 512, False, 1), (36, 64, 512, 512, False, 1)):
    x = torch.randn((n, 1, hw, k), device="cuda").half()
    d = torch.randn((k, cout), device="cuda") * 1e-4
    b = torch.randn(cout, device="cuda")
    a = (torch.rand((n, k), device="cuda") + 0.5, torch.randn((n, k), device="cuda") * 0.1, ops.ACT_SILU) if aff else None
    f = lambda: ops.frame_bias(x, d, b, affine_in=a, groups=groups)
    y = f()
    sha = hashlib.sha256(y.cpu().numpy().tobytes()).hexdigest()[:12]
    us = timeit(f, 20)
    print(json.dumps({"lib": lib, "rows": n, "pixels": hw, "K": k, "Cout": cout, "fused_apply": aff, "groups": groups, "us": round(us, 1), "sha": sha}), flush=True)
