import sys, os, json, torch
sys.path.insert(0, os.getcwd())
from pgtformer_amd import ops
from tools.bench_micro import timeit
dt = torch.float16
for (h, cin, cout) in ((256, 128, 128), (256, 256, 128), (128, 256, 256)):
    w = ops.pack_conv_weight(torch.randn((cout, cin, 3, 3), device="cuda") / (cin * 9) ** 0.5, dt)
    b = torch.zeros(cout, device="cuda")
    for n in (4, 8, 16, 32, 96):
        x = torch.randn((n, h, h, cin), device="cuda").to(dt)
        us = timeit(lambda: ops.conv2d(x, w, b, kh=3, kw=3, pad=(1, 1, 1, 1)), 10)
        fl = 2.0 * n * h * h * cin * cout * 9
        print(json.dumps({"shape": [n, h, h, cin, cout], "input_MB": round(x.numel() * 2 / 1e6), "us": round(us, 1), "tflops": round(fl / us / 1e6, 1)}), flush=True)
        del x
