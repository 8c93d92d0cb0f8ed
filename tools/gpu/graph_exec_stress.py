"""What kills hipGraphLaunch in a long process?  (rocgdb: hip::Graph::UpdateStreams dereferences parallel_streams[2] == nullptr.)
  python tools/gpu/graph_exec_stress.py keep   N    - create N runners (one captured graph each, with parallel branches), KEEP them alive, replay each
  python tools/gpu/graph_exec_stress.py drop   N    - the same, destroying every runner (and collecting) before the next is made
  python tools/gpu/graph_exec_stress.py linear N    - `keep` with the side stream of the forward switched off (PGT_SIDE_STREAM=0 must be set by the caller)
  python tools/gpu/graph_exec_stress.py pipe   N [lanes] [batch] - N runners one after the other, each streaming a pinned-host clip through its lanes (the failing test's path)
Prints one line per runner; the last line printed before a crash is the count the process survived."""
import gc, os, sys, faulthandler
import numpy as np, torch
faulthandler.enable()
sys.path.insert(0, os.getcwd())
from pgtformer_amd import PGTFormer, default_config
from pgtformer_amd.driver import WindowRunner
from pgtformer_amd.manifest import pgtformer_manifest
from pgtformer_amd.synth import make_clip
from pgtformer_amd.weightgen import generate_state_dict

mode, n = sys.argv[1], int(sys.argv[2])
dev = torch.device("cuda", 0)
cfg = default_config()
m = PGTFormer(**cfg)
m.load_state_dict(generate_state_dict(pgtformer_manifest(cfg), cfg, seed=0), strict=True)
m.prepare(dev, "x3f16")
lq, _ = make_clip(3, 512, seed=5)
frames = torch.from_numpy(lq).to(dev)
alive, want = [], None
if mode == "pipe":
    from pgtformer_amd.driver import restore_clip_host
    lanes = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    batch = int(sys.argv[4]) if len(sys.argv) > 4 else 8
    base, _ = make_clip(7, 512, seed=21)
    clip = np.concatenate([base] * 3, 0)
    for i in range(1, n + 1):
        padded = torch.empty((23, 512, 512, 3), dtype=torch.uint8).pin_memory()
        padded[1:22].copy_(torch.from_numpy(clip))
        got = torch.empty((21, 512, 512, 3), dtype=torch.uint8).pin_memory()
        restore_clip_host(WindowRunner(m, 1.0, use_graph=True, batch=batch, lanes=lanes), padded, got)
        torch.cuda.synchronize()
        want = got.clone() if want is None else want
        assert torch.equal(got, want)
        print(f"pipe: runner {i} ok (lanes {lanes}, batch {batch}); GPU memory {torch.cuda.memory_reserved() / 2**30:.1f} GiB", flush=True)
    print("done", flush=True)
    sys.exit(0)
for i in range(1, n + 1):
    r = WindowRunner(m, 1.0, True, 512, 512, batch=1, lanes=1, check_range=False)
    got = r.run(frames).clone()
    torch.cuda.synchronize()
    want = got if want is None else want
    assert torch.equal(got, want)
    if mode == "drop":
        del r
        gc.collect()
    else:
        alive.append(r)
        if i % 8 == 0:           # every live graph still replays
            for q in alive:
                q.run(frames)
            torch.cuda.synchronize()
    print(f"{mode}: runner {i} ok; live graphs {len(alive)}; GPU memory {torch.cuda.memory_reserved() / 2**30:.1f} GiB", flush=True)
print("done", flush=True)
