"""Is the hipGraphLaunch crash (hip::Graph::UpdateStreams reads parallel_streams[i] == nullptr) reproducible WITHOUT this library?  Plain torch:
a captured graph with two forked branches, replayed on every stream of torch's pool, with 0 .. 7 raw HIP streams created (and kept) before each
capture so that the runtime's stream -> hardware-queue assignment shifts.  Prints progress; a segfault ends the process."""
import ctypes, faulthandler, os, sys
import torch
faulthandler.enable()
hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
dev = torch.device("cuda", 0)
x = torch.zeros(1 << 20, device=dev)
y = torch.zeros(1 << 20, device=dev)
z = torch.zeros(1 << 20, device=dev)
nbranch = int(sys.argv[1]) if len(sys.argv) > 1 else 2
pool = [torch.cuda.Stream(device=dev) for _ in range(40)]       # more than the pool holds: all of its streams exist from here on
side = pool[:nbranch]
raw = []
trial = 0
for extra in list(range(8)) * 6:
    for _ in range(extra):                 # shift the runtime's round-robin of streams over hardware queues
        h = ctypes.c_void_p()
        assert hip.hipStreamCreateWithFlags(ctypes.byref(h), 1) == 0
        raw.append(h)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        cur = torch.cuda.current_stream()
        x.add_(1)
        for i, s in enumerate(side):
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                (y if i % 2 == 0 else z).add_(1)
        for s in side:
            cur.wait_stream(s)
        x.add_(y)
    for s in pool[8:40]:
        with torch.cuda.stream(s):
            g.replay()
    torch.cuda.synchronize()
    trial += 1
    print(f"trial {trial}: {extra} raw streams added ({len(raw)} alive), graph with {nbranch} forked branches replayed on 32 streams: ok", flush=True)
    if len(raw) > 64:
        for h in raw:
            hip.hipStreamDestroy(h)
        raw = []
print("done", flush=True)
