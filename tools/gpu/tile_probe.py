"""Experiment: is the 512 x 128 tile (Cout <= 128) slow because of its shape or because of the layers' short K?  The same maps and Cin with
Cout = 128 (tile 512 x 128) and Cout = 256 (tile 256 x 256); plain launches, event-timed."""
import sys, os, json, torch
sys.path.insert(0, os.getcwd())
from pgtformer_amd import ops
from tools.bench_micro import timeit
dt = torch.float16
torch.manual_seed(0)
for (n, h, cin) in ((96, 256, 128), (48, 256, 256), (96, 128, 128), (96, 128, 256), (96, 64, 128)):
    x = torch.randn((n, h, h, cin), device="cuda").to(dt)
    for cout in (128, 256):
        for k in (3, 1):
            w = ops.pack_conv_weight(torch.randn((cout, cin, k, k), device="cuda") / (cin * k * k) ** 0.5, dt)
            b = torch.zeros(cout, device="cuda")
            pad = (1, 1, 1, 1) if k == 3 else (0, 0, 0, 0)
            us = timeit(lambda: ops.conv2d(x, w, b, kh=k, kw=k, pad=pad), 20)
            fl = 2.0 * n * h * h * cin * cout * k * k
            print(json.dumps({"shape": [n, h, h, cin, cout, k], "K": cin * k * k, "us": round(us, 1), "tflops": round(fl / us / 1e6, 1)}), flush=True)
    del x
