"""Random search for a plain-torch trigger of the hipGraphLaunch fault on a two-root graph: raw HIP streams created and destroyed at
random, forked graphs captured, kept alive or dropped at random, replayed on random streams.  One child per seed; the parent reports the seeds that died.
    python tools/gpu/hipgraph_parallel_streams_repro3.py [n_seeds] [trials]          python tools/gpu/hipgraph_parallel_streams_repro3.py child <seed> <trials>"""
import ctypes, os, random, subprocess, sys

if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    seed, trials = int(sys.argv[2]), int(sys.argv[3])
    rng = random.Random(seed)
    hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
    dev = torch.device("cuda", 0)
    x = torch.zeros(1 << 14, device=dev)
    y = torch.zeros(1 << 14, device=dev)
    pool = [torch.cuda.Stream(device=dev) for _ in range(34)]
    raw, graphs = [], []
    for t in range(trials):
        for _ in range(rng.randint(0, 6)):
            h = ctypes.c_void_p()
            assert hip.hipStreamCreateWithFlags(ctypes.byref(h), 1) == 0
            raw.append(h)
        for _ in range(rng.randint(0, 6)):
            if raw:
                assert hip.hipStreamDestroy(raw.pop(rng.randrange(len(raw)))) == 0
        side = rng.choice(pool)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            cur = torch.cuda.current_stream()
            x.add_(1)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                y.add_(1)
            cur.wait_stream(side)
            x.add_(y)
        graphs.append(g)
        if len(graphs) > 8:
            graphs.pop(rng.randrange(len(graphs)))
        for _ in range(4):
            q = rng.choice(graphs)
            if raw and rng.random() < 0.5:
                s = torch.cuda.ExternalStream(rng.choice(raw).value, device=dev)
            else:
                s = rng.choice(pool)
            with torch.cuda.stream(s):
                q.replay()
        torch.cuda.synchronize()
        if t % 50 == 49:
            print(f"seed {seed}: {t + 1} trials ok", flush=True)
    print("done", flush=True)
    sys.exit(0)

n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 8
trials = int(sys.argv[2]) if len(sys.argv) > 2 else 400
dead = []
for seed in range(n_seeds):
    r = subprocess.run([sys.executable, __file__, "child", str(seed), str(trials)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    last = [l for l in r.stdout.decode().splitlines() if l][-1:] or [""]
    print(f"seed {seed}: return code {r.returncode}; last line: {last[0]}", flush=True)
    if r.returncode != 0:
        dead.append((seed, r.returncode, last[0]))
print("died:", dead)
