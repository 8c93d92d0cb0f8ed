#!/usr/bin/env python
"""Debug aid for pgtformer_amd/export.py: identity replay (same addresses) and placed replay of a recorded forward."""
import ctypes as C
import gc
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pgtformer_amd import PGTFormer, default_config, export, hip, ops  # noqa: E402
from pgtformer_amd.manifest import pgtformer_manifest  # noqa: E402
from pgtformer_amd.weightgen import generate_state_dict  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
dev = torch.device("cuda", 0)
cfg = default_config()
m = PGTFormer(**cfg)
m.load_state_dict(generate_state_dict(pgtformer_manifest(cfg), cfg, seed=0), strict=True)
m.prepare(dev, prec)
nw = 2
frames = torch.randint(0, 256, (nw + 2, 512, 512, 3), dtype=torch.uint8).to(dev)
out = torch.zeros((nw, 512, 512, 3), dtype=torch.uint8, device=dev)
win = m.window_index(nw, 3, dev)
fwd = lambda: m.restore_middle_u8(frames, w=1.0, out=out, win=win)      # noqa: E731
with torch.no_grad():
    fwd()
    torch.cuda.synchronize()
    gc.collect()
    persistent = export._live_cuda_storages(dev)
    calls, _ = export.record(fwd, dev)
want = out.cpu()
print("calls", len(calls), "want mean", float(want.float().mean()))
with torch.no_grad():
    out.zero_()
    fwd()
    torch.cuda.synchronize()
d = (out.cpu().int() - want.int()).abs()
print("plain repeat: differing bytes", int((d > 0).sum()), "max", int(d.max()))
# identity replay: same raw arguments, right away
L = hip.lib()
out.zero_()
torch.cuda.synchronize()
stream = ops._stream()
for name, args in calls:
    if name in export.QUERIES:
        continue
    raw = []
    keep = []
    for k, a in enumerate(args):
        if k == len(args) - 1:
            raw.append(stream)
        elif a[0] == "ptr":
            raw.append(C.c_void_p(a[1]) if a[1] else None)
        elif a[0] == "blob":
            d = hip.ConvDesc.from_buffer_copy(a[1])
            keep.append(d)
            raw.append(C.byref(d))
        else:
            raw.append(a[1])
    rc = getattr(L, name)(*raw)
    assert rc == 0, (name, rc, L.pgt_last_error())
torch.cuda.synchronize()
d = (out.cpu().int() - want.int()).abs()
print("identity replay equal:", bool(torch.equal(out.cpu(), want)), "differing bytes", int((d > 0).sum()), "max", int(d.max()), "rows with a difference", sorted(set((d > 0).nonzero()[:, 1].tolist()))[:20], "frames", sorted(set((d > 0).nonzero()[:, 0].tolist())))
tape, playout, work = export.build_program(calls, persistent, frames, out)
from collections import Counter
cnt = Counter()
for fid, recs in tape:
    for kind, aux, val in recs:
        if kind == export.K_PTR:
            cnt[aux] += 1
print("pointer args by region (0 persist, 1 work, 2 in, 3 out):", dict(cnt), "persistent storages", len(playout),
      "persist MB", sum(n for _, n in playout.values()) / 1e6, "work MB", work / 1e6)
path = "/tmp/dbg.prog"
export.write_program(path, tape, playout, work, frames.numel(), out.numel(), b"dbg", storages=persistent)
got = export.run_program(path, frames).reshape(want.shape)
d = (got.cpu().int() - want.int()).abs()
print("placed replay equal:", bool(torch.equal(got.cpu(), want)), "differing bytes", int((d > 0).sum()), "max", int(d.max()))

# the packaged path, as the test uses it
info = export.export_program(m, 2, "/tmp/dbg2.prog")
got = export.run_program("/tmp/dbg2.prog", info["input"]).reshape(info["output"].shape)
d = (got.cpu().int() - info["output"].cpu().int()).abs()
print("export_program + run_program: differing bytes", int((d > 0).sum()), "max", int(d.max()), "got mean", float(got.float().mean()), "want mean", float(info["output"].float().mean()))
