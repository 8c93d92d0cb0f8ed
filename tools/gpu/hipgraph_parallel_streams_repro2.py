"""Deterministic trigger for the hipGraphLaunch segfault on a two-root graph?  Model: the runtime maps a new stream to the least-loaded of its hardware
queues, and hip::Graph::UpdateStreams skips internal streams on the launch stream's queue without bounding the search.  So: create 16 raw streams (queues
0 1 2 3 0 1 2 3 ... under that model), destroy two that share a queue (k and k + 4), capture a forked graph (its two internal streams should now land on
that queue), replay it on every remaining raw stream.  One child process per (k, launch stream); prints which combinations die.
    python tools/gpu/hipgraph_parallel_streams_repro2.py            # the sweep
    python tools/gpu/hipgraph_parallel_streams_repro2.py k j        # one child: destroy streams k, k + 4; launch on stream j"""
import ctypes, os, subprocess, sys

if len(sys.argv) == 3:
    import torch
    k, j = int(sys.argv[1]), int(sys.argv[2])
    hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
    dev = torch.device("cuda", 0)
    x = torch.zeros(1 << 16, device=dev)
    y = torch.zeros(1 << 16, device=dev)
    side = torch.cuda.Stream(device=dev)
    torch.cuda.synchronize()
    raw = []
    for _ in range(16):
        h = ctypes.c_void_p()
        assert hip.hipStreamCreateWithFlags(ctypes.byref(h), 1) == 0
        raw.append(h)
    for d in (k, k + 4):
        assert hip.hipStreamDestroy(raw[d]) == 0
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        cur = torch.cuda.current_stream()
        x.add_(1)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            y.add_(1)
        cur.wait_stream(side)
        x.add_(y)
    ext = torch.cuda.ExternalStream(raw[j].value, device=dev)
    with torch.cuda.stream(ext):
        g.replay()
    torch.cuda.synchronize()
    print("ok")
    sys.exit(0)

dead = []
for k in range(4):
    row = []
    for j in range(16):
        if j in (k, k + 4):
            row.append(" ")
            continue
        r = subprocess.run([sys.executable, __file__, str(k), str(j)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
        row.append("." if r.returncode == 0 else "X")
        if r.returncode != 0:
            dead.append((k, j, r.returncode))
    print(f"destroyed streams {k} and {k + 4}; replay on raw stream 0..15: {''.join(row)}", flush=True)
print("died (k, launch stream, return code):", dead)
