"""Experiment: cache-policy bits (nt / sc0 / sc1) on igemm4's A-operand LDS-DMA loads - same arithmetic, only where the lines sit in L2.
Run once per library build (PGT_LIB_PATH); prints time and a checksum of the output per shape."""
import sys, os, json, hashlib, torch
sys.path.insert(0, os.getcwd())
from pgtformer_amd import ops
from tools.bench_micro import timeit
dt = torch.float16
torch.manual_seed(0)
for (n, h, cin, cout) in ((96, 256, 128, 128), (48, 256, 256, 128), (32, 256, 320, 128), (96, 128, 256, 256), (96, 64, 256, 256), (96, 32, 512, 512)):
    w = ops.pack_conv_weight(torch.randn((cout, cin, 3, 3), device="cuda") / (cin * 9) ** 0.5, dt)
    b = torch.zeros(cout, device="cuda")
    x = torch.randn((n, h, h, cin), device="cuda").to(dt)
    y = ops.conv2d(x, w, b, kh=3, kw=3, pad=(1, 1, 1, 1))
    sha = hashlib.sha256(y.cpu().numpy().tobytes()).hexdigest()[:12]
    us = timeit(lambda: ops.conv2d(x, w, b, kh=3, kw=3, pad=(1, 1, 1, 1)), 20)
    fl = 2.0 * n * h * h * cin * cout * 9
    print(json.dumps({"lib": os.environ.get("PGT_LIB_PATH", "shipping").split("/")[-2] if os.environ.get("PGT_LIB_PATH") else "shipping",
                      "shape": [n, h, h, cin, cout], "us": round(us, 1), "tflops": round(fl / us / 1e6, 1), "sha": sha}), flush=True)
    del x, y
