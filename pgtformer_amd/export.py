"""Whole-graph export: the launch schedule of one forward as a TAPE of C-ABI calls that `libpgt_hip.so` replays itself
(`pgt_program_load / pgt_program_run / pgt_program_destroy`, include/pgt_hip.h) - the whole-graph entry a non-Python host needs
(SURVEY.md section 8b "Call" / "Ownership": `pgt_forward_window`, an opaque model handle owning immutable repacked weights).

The reference's graph lives in Python (`PGTFormer.forward`, archs/pgtformer_arch.py:598-714) and so does this build's: ~600
launches per forward whose order, shapes, precision choices, BatchNorm / LayerNorm folds and compensation set-up are decided by
the host modules.  Instead of restating that orchestration in a second language, the host records it ONCE: every call the forward
makes through the C-ABI (function, scalars, descriptor structs, pointers) is written down with its pointers resolved to
  * a PERSISTENT region  - storages that outlive the forward: repacked weights, bias tables, index tensors, arrival counters; their
                           bytes go into the program file and are uploaded once by pgt_program_load;
  * the WORKSPACE        - every temporary of the forward.  The caching allocator's addresses are kept RELATIVE inside merged
                           address ranges, so two temporaries that shared memory at different times share it in the replay too:
                           the replay runs the same launches in the same order, any reuse that was safe then is safe now;
  * the INPUT / OUTPUT   - the uint8 frames in, the restored uint8 frames out (caller-owned buffers at replay time).
The tape is what a HIP-graph capture of the same forward holds, in a portable form: no torch, no Python at replay time, every
launch on the caller's stream, no allocation and no synchronisation in pgt_program_run (a C host may capture it into a hipGraph).
A program is specific to what was recorded: the checkpoint, the precision mode, the window batch and the frame size.

    from pgtformer_amd.export import export_program
    export_program(model, n_windows=2, path="pgt_b2.prog")          # model.prepare(...)d, on the GPU
"""
import ctypes as C
import gc
import os
import struct

import torch

from . import hip, ops

MAGIC = b"PGTPROG1"
ALIGN = 256
R_PERSIST, R_WORK, R_IN, R_OUT = 0, 1, 2, 3
K_INT, K_F32, K_NULL, K_PTR, K_DESC, K_STREAM = 0, 1, 2, 3, 4, 5
# functions that only answer questions (no launch, no stream): never part of a tape
QUERIES = {"pgt_version", "pgt_last_error", "pgt_conv2d_workspace_bytes", "pgt_conv_gn_workspace_bytes", "pgt_conv2d_affine_in_ok",
           "pgt_groupnorm_workspace_bytes", "pgt_sampled_pixel", "pgt_sampled_pixel_cells", "pgt_frame_bias_workspace_bytes", "pgt_sampled_rownorm_workspace_bytes",
           "pgt_attn_proj_mlp_sample_workspace_bytes", "pgt_packed_weight_bytes", "pgt_commit_loss_workspace_bytes",
           "pgt_program_load", "pgt_program_destroy", "pgt_program_run", "pgt_program_workspace_bytes", "pgt_program_io_bytes", "pgt_program_info"}


def tape_functions():
    """the functions a tape may hold, in the order of their ids in the dispatch table (csrc/program_dispatch.inc)"""
    return sorted(n for n in hip.SIGNATURES if n not in QUERIES)


def _align(n):
    return (n + ALIGN - 1) // ALIGN * ALIGN


def _live_cuda_storages(device):
    """{storage address: (bytes, storage)} of every CUDA tensor alive on `device` right now"""
    device = torch.device(device)
    if device.index is None:          # "cuda" names the current device; tensors report "cuda:<index>"
        device = torch.device("cuda", torch.cuda.current_device())
    out = {}
    for o in gc.get_objects():
        try:
            if torch.is_tensor(o) and o.is_cuda and o.device == device:
                st = o.untyped_storage()
                if st.data_ptr() not in out or out[st.data_ptr()][0] < st.nbytes():
                    out[st.data_ptr()] = (st.nbytes(), st)
        except Exception:      # noqa: BLE001  (objects half torn down by gc)
            pass
    return out


def record(fn, device):
    """run fn() once with every C-ABI call recorded -> [(name, [resolved args])] (hip._TracedLib)"""
    from .archs import pgtformer_arch
    keep_side, pgtformer_arch.SIDE_STREAM = pgtformer_arch.SIDE_STREAM, False      # one stream: tape order = program order
    hip.TRACE, hip.TRACE_TENSORS = [], {}
    try:
        res = fn()
        torch.cuda.synchronize(device)
        return hip.TRACE, res
    finally:
        hip.TRACE, hip.TRACE_TENSORS = None, None
        pgtformer_arch.SIDE_STREAM = keep_side


def pack_by_liveness(bufs):
    """bufs: [(bytes, first, last)] - buffers live over the closed call ranges [first, last] -> ([offset per buffer], total bytes) such
    that buffers whose ranges intersect do not overlap in space.  Largest first, each at the lowest ALIGN-ed offset free of the
    already placed buffers it is ever live with (offline storage allocation by first fit: a few hundred buffers, quadratic is fine)."""
    order = sorted(range(len(bufs)), key=lambda i: (-bufs[i][0], bufs[i][1]))
    off = [0] * len(bufs)
    placed = []
    total = 0
    for i in order:
        nb, a, b = bufs[i]
        size = _align(max(1, nb))
        busy = sorted((off[j], off[j] + _align(max(1, bufs[j][0]))) for j in placed if not (bufs[j][2] < a or b < bufs[j][1]))
        o = 0
        for lo, hi in busy:
            if o + size <= lo:
                break
            o = max(o, hi)
        off[i] = o
        placed.append(i)
        total = max(total, o + size)
    return off, total


def build_program(calls, persistent, in_t, out_t):
    """calls: record()'s list; persistent: {storage address: bytes} of everything that outlives the forward; in_t / out_t: the
    input / output tensors.  Returns (tape records, persistent layout {addr: (offset, bytes)}, workspace bytes)."""
    in_lo, in_hi = in_t.data_ptr(), in_t.data_ptr() + in_t.numel() * in_t.element_size()
    out_lo, out_hi = out_t.data_ptr(), out_t.data_ptr() + out_t.numel() * out_t.element_size()
    names = {n: i for i, n in enumerate(tape_functions())}
    used_persist, temps = {}, {}
    # pass 1: classify every pointer; a temporary = one storage (base, bytes) with the first and last call that touches it
    idx = -1
    for name, args in calls:
        if name in QUERIES:
            continue
        idx += 1
        assert name in names, f"{name}: not a tape function"
        for a in args[:-1]:
            if a[0] != "ptr" or a[1] == 0:
                continue
            v, base, nb = a[1], a[2], a[3]
            if in_lo <= v < in_hi or out_lo <= v < out_hi:
                continue
            assert base is not None, f"{name}: pointer {v:#x} was not handed out by ops._p (cannot be placed)"
            if base in persistent:
                used_persist[base] = max(persistent[base][0], nb)
            else:
                # (storage address, bytes, storage object): a block the allocator hands out twice during the forward is two temporaries
                t = temps.setdefault((base, nb, a[4] if len(a) > 4 else 0), [idx, idx])
                t[1] = idx
    # the recorded forward ran on ONE stream in tape order: two temporaries whose [first, last] call ranges do not intersect are
    # never live together and may share bytes (the torch allocator's address ranges - what the tape used to keep - are its whole
    # high-water footprint: 2.0 GB for two windows in the default mode; packed by liveness: what is live at the worst moment)
    keys = list(temps)
    woff, work_bytes = pack_by_liveness([(k[1], temps[k][0], temps[k][1]) for k in keys])
    woff = dict(zip(keys, woff))
    playout, off = {}, 0
    for base in sorted(used_persist):
        playout[base] = (off, used_persist[base])
        off += _align(used_persist[base])

    def place(v, base, nb, sid=0):
        if in_lo <= v < in_hi:
            return R_IN, v - in_lo
        if out_lo <= v < out_hi:
            return R_OUT, v - out_lo
        if base in playout:
            return R_PERSIST, playout[base][0] + (v - base)
        assert base <= v < base + nb, hex(v)
        return R_WORK, woff[(base, nb, sid)] + (v - base)

    tape = []
    for name, args in calls:
        if name in QUERIES:
            continue
        sig = hip.SIGNATURES[name]
        assert len(sig) == len(args), (name, len(sig), len(args))
        recs = []
        for k, (a, ty) in enumerate(zip(args, sig)):
            if k == len(args) - 1:
                recs.append((K_STREAM, 0, 0))
            elif a[0] == "blob":
                recs.append((K_DESC, len(a[1]), a[1]))
            elif a[0] == "ptr":
                if a[1] == 0:
                    recs.append((K_NULL, 0, 0))
                else:
                    region, o = place(a[1], a[2], a[3], a[4] if len(a) > 4 else 0)
                    recs.append((K_PTR, region, o))
            elif ty is hip.f32:
                recs.append((K_F32, 0, struct.unpack("<I", struct.pack("<f", float(a[1])))[0]))
            else:
                recs.append((K_INT, 0, int(a[1]) & 0xFFFFFFFFFFFFFFFF))
        tape.append((names[name], recs))
    return tape, playout, work_bytes


def write_program(path, tape, playout, work_bytes, in_bytes, out_bytes, meta=b"", storages=None):
    """file: MAGIC | u32 version | u32 n_functions | names (u16 len + bytes each) | u64 persist_bytes | u64 work_bytes | u64 in_bytes
    | u64 out_bytes | u32 n_calls | u32 meta_len | meta | calls | persistent bytes.  A call: u16 function, u16 n_args, then per
    argument u32 kind, u32 aux, u64 value (K_DESC: aux = byte count, value = offset into the descriptor pool that follows the calls:
    u64 pool_bytes | pool)."""
    names = tape_functions()
    persist_bytes = max([o + _align(n) for o, n in playout.values()], default=0)
    pool = bytearray()
    body = bytearray()
    for fid, recs in tape:
        body += struct.pack("<HH", fid, len(recs))
        for kind, aux, val in recs:
            if kind == K_DESC:
                off = len(pool)
                pool += val
                pool += b"\0" * (-len(pool) % 8)
                body += struct.pack("<IIQ", kind, aux, off)
            else:
                body += struct.pack("<IIQ", kind, aux, val)
    with open(path, "wb") as f:
        f.write(MAGIC)
        f.write(struct.pack("<II", 1, len(names)))
        for n in names:
            b = n.encode()
            f.write(struct.pack("<H", len(b)) + b)
        f.write(struct.pack("<QQQQII", persist_bytes, work_bytes, in_bytes, out_bytes, len(tape), len(meta)))
        f.write(meta)
        f.write(body)
        f.write(struct.pack("<Q", len(pool)))
        f.write(pool)
        # persistent bytes, region by region (device -> host copies of the live storages)
        pos = 0
        for base in sorted(playout, key=lambda b: playout[b][0]):
            off, nb = playout[base]
            assert off == pos
            st = storages[base][1]
            raw = torch.empty(0, dtype=torch.uint8, device=st.device).set_(st, 0, (st.nbytes(),))
            f.write(raw.cpu().numpy().tobytes()[:nb].ljust(nb, b"\0"))
            f.write(b"\0" * (_align(nb) - nb))
            pos += _align(nb)
    return persist_bytes


@torch.no_grad()
def export_program(model, n_windows, path, w=1.0, height=512, width=512, overlap=True, full_tail=False, verify=True):
    """Record `model.restore_middle_u8` on `n_windows` sliding 3-frame windows (uint8 frames in -> restored uint8 middle frames out,
    reference inference.py:12-19) and write the program file.  Returns a dict with the sizes and the example input / output
    (device tensors) of the recorded forward - what pgt_program_run must reproduce bit for bit."""
    dev = torch.device(model.dev)
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    t = model.t
    n_in = n_windows + t - 1 if overlap else n_windows * t
    g = torch.Generator().manual_seed(1234)
    frames = torch.randint(0, 256, (n_in, height, width, 3), dtype=torch.uint8, generator=g).to(dev)
    out = torch.zeros((n_windows, height, width, 3), dtype=torch.uint8, device=dev)
    win = model.window_index(n_windows, t, dev) if overlap else None
    kw = {"win": win} if overlap else {}
    if full_tail:
        kw["full_tail"] = True

    def fwd():
        return model.restore_middle_u8(frames, w=w, out=out, **kw)
    fwd()                                  # warm-up: every lazily created persistent tensor (index caches, counters) now exists
    torch.cuda.synchronize(dev)
    gc.collect()
    persistent = _live_cuda_storages(dev)
    calls, _ = record(fwd, dev)
    want = out.clone()
    tape, playout, work_bytes = build_program(calls, persistent, frames, out)
    if not playout:
        raise hip.PgtError("export: the recorded forward touched no persistent storage (no weights found on %s)" % dev)
    meta = (f"precision={getattr(model, 'precision', '?')} windows={n_windows} frames_in={n_in} size={height}x{width} overlap={int(overlap)} "
            f"full_tail={int(full_tail)} w={w} lib={hip.lib().pgt_version().decode()}").encode()
    persist_bytes = write_program(path, tape, playout, work_bytes, frames.numel(), out.numel(), meta, storages=persistent)
    if verify:
        # the tape holds C-ABI calls only: an operation of the forward that did not go through the library (a torch-native op) would be
        # missing from the replay - the file is only kept if the library reproduces the recorded frames bit for bit
        got = run_program(path, frames).reshape(want.shape)
        if not torch.equal(got, want):
            os.remove(path)
            raise hip.PgtError("export: the replayed program differs from the recorded forward (%d of %d bytes): an operation outside the "
                               "C-ABI took part in it" % (int((got != want).sum()), want.numel()))
    return {"calls": len(tape), "persistent_bytes": persist_bytes, "workspace_bytes": work_bytes, "input": frames, "output": want,
            "input_bytes": frames.numel(), "output_bytes": out.numel()}


def run_program(path, frames_u8, out_u8=None):
    """Replay a program through the library (pgt_program_load / _run / _destroy) from Python: the check that the file alone
    reproduces the recorded forward (the C host does the same three calls, tests/c/program_smoke.c)."""
    L = hip.lib()
    h = C.c_void_p()
    hip.check(L.pgt_program_load(path.encode(), C.byref(h)), "pgt_program_load")
    try:
        nin, nout = C.c_size_t(), C.c_size_t()
        hip.check(L.pgt_program_io_bytes(h, C.byref(nin), C.byref(nout)), "pgt_program_io_bytes")
        assert frames_u8.numel() == nin.value and frames_u8.dtype == torch.uint8 and frames_u8.is_contiguous(), (frames_u8.shape, nin.value)
        if out_u8 is None:
            out_u8 = torch.empty(nout.value, dtype=torch.uint8, device=frames_u8.device)
        ws = torch.empty(max(1, L.pgt_program_workspace_bytes(h)), dtype=torch.uint8, device=frames_u8.device)
        hip.check(L.pgt_program_run(h, ops._p(frames_u8), ops._p(out_u8), ops._p(ws), ws.numel(), ops._stream()), "pgt_program_run")
        torch.cuda.synchronize()
        return out_u8
    finally:
        L.pgt_program_destroy(h)
