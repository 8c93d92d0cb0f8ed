"""Operator layer: thin Python wrappers that launch the HIP kernels of libpgt_hip.so on torch device
tensors (torch is used for memory, streams and views only — all arithmetic happens in the kernels).

Tensors are channels-last: images (N,H,W,C), token matrices (rows,C); the last dim must be contiguous
and the pixel/row stride may exceed C (channel slices of a wider buffer are valid inputs/outputs).
dtype torch.float32 selects the exact-f32 kernels, torch.bfloat16 / torch.float16 the 16-bit MFMA kernels (bf16: 8
significand bits; IEEE half: 11 - the decoder-side type of the default precision mode).
"""
import ctypes as C

import torch

from . import hip
from .hip import (ACT_GELU, ACT_LEAKY02, ACT_NONE, ACT_RELU, ACT_SIGMOID, ACT_SILU,  # noqa: F401
                  EPI_PLAIN, EPI_SFT, PGT_BF16, PGT_F16X3, PGT_F32)

# "dtype" tag of split modules / tensors (include/pgt_hip.h: PGT_F16X3).  A split tensor with C logical channels is an
# X3_PLANE (torch.float16) tensor whose last dim is 2C: [hi (C) | lo (C)], value = hi + lo: two IEEE-half planes, 22
# significand bits, three MFMAs per product (two bf16 planes - 16 bits - until round 3's sweep of the code flips,
# profiles/r3_psnr_sweep.md).  The ops below take `x3=True` for such operands; reshapes over the leading dims work unchanged,
# the hi plane `t[..., :C]` is a valid half view of the rounded tensor.
X3 = "f16x3"
X3F = "f16x3/f32"       # split arithmetic, fp32 storage: the module splits its fp32 input on the way in, writes fp32
X3_PLANE = torch.float16
# GroupNorm statistics from the producing conv's epilogue (PGT_EPILOGUE_GN=0: always the separate statistics pass)
import os as _os
USE_EPILOGUE_GN = _os.environ.get("PGT_EPILOGUE_GN", "1") != "0"
# GroupNorm apply + SiLU inside the consuming conv's operand load where the library has that form (pgt_conv2d_affine_in:
# the 64-channel 3x3 layers of the 512x512 level); PGT_FUSE_GN_APPLY=0: always the separate apply pass
USE_FUSED_GN_APPLY = _os.environ.get("PGT_FUSE_GN_APPLY", "1") != "0"
# sampled mean + mean-field bias in one launch (pgt_frame_bias); PGT_FRAME_BIAS=0: the two launches of round 3
USE_FRAME_BIAS = _os.environ.get("PGT_FRAME_BIAS", "1") != "0"


class GnStats:
    """GroupNorm statistics a producing conv left behind (pgt_conv2d_gn): attached to the conv's output tensor object as
    `._pgt_gn`; the next GroupNorm over exactly that tensor finalises them instead of re-reading the tensor."""
    __slots__ = ("ws", "n", "nsub", "hw", "c", "groups", "ptr", "ld")

    def __init__(self, n, nsub, hw, c, groups, device):
        nbytes = hip.lib().pgt_conv_gn_workspace_bytes(n, nsub, hw, groups)
        self.ws = torch.empty(nbytes // 4, dtype=torch.float32, device=device)
        self.n, self.nsub, self.hw, self.c, self.groups = n, nsub, hw, c, groups
        self.ptr = self.ld = None

    def bind(self, t, c_stored):
        """record the tensor the statistics belong to and attach them to it"""
        self.ptr, self.ld = t.data_ptr(), (tuple(t.shape), c_stored)
        t._pgt_gn = self
        return t


def gn_ok(n, hw, cout, groups=32, cin=None, k=None):
    """can a conv producing (n, hw pixels, cout) leave GroupNorm statistics? (tile rows up to 512 must divide hw).
    The 64 -> 64 3x3 layers are left to the separate statistics pass: their streaming kernel (igemm6: 144 weight registers
    held across tiles) has no registers to spare for the reduction - measured on MI355X, its statistics variant lost more
    than the statistics pass costs (profiles/r2_gn_epilogue_ab.md)."""
    if cin == 64 and cout <= 64 and k == 3:
        return False
    return groups > 0 and cout % groups == 0 and cout % 8 == 0 and hw % 512 == 0


def pack_x3_weight(w3):
    """w3: fp32 (Cout, taps, Cin), Cin % 64 == 0 -> bf16 (Cout, taps*3*Cin): per tap and per 64-channel block
    [w_hi | w_hi | w_lo] (64 each) - the B operand matching the kernels' K order [x_hi | x_lo | x_hi] per block
    (x*w ~= x_hi*w_hi + x_lo*w_hi + x_hi*w_lo)."""
    w3 = w3.float()
    cout, taps, cin = w3.shape
    assert cin % 64 == 0, "split-half operands come in 64-channel K blocks"
    hi = w3.to(X3_PLANE)
    lo = (w3 - hi.float()).to(X3_PLANE)
    hi4, lo4 = hi.reshape(cout, taps, cin // 64, 1, 64), lo.reshape(cout, taps, cin // 64, 1, 64)
    return torch.cat([hi4, hi4, lo4], 3).reshape(cout, -1).contiguous()


def pack_x3_fold_weight(w3):
    """Folded form for 64-output-channel layers (pgt_conv_desc.x3_fold): w3 fp32 (64, taps, Cin) -> bf16 (128, taps*2*Cin):
    rows 0..63 hold [w_hi | w_hi] per tap and 64-channel block, rows 64..127 [w_lo | 0]; the kernel visits the input as
    [x_hi | x_lo] per block and adds the two halves of its 128-column tile: x_hi*w_hi + x_lo*w_hi + x_hi*w_lo in two K
    segments instead of three, on a tile that is full instead of half idle."""
    w3 = w3.float()
    cout, taps, cin = w3.shape
    assert cout == 64 and cin % 64 == 0
    hi = w3.to(X3_PLANE)
    lo = (w3 - hi.float()).to(X3_PLANE)
    hi4, lo4 = hi.reshape(cout, taps, cin // 64, 1, 64), lo.reshape(cout, taps, cin // 64, 1, 64)
    top = torch.cat([hi4, hi4], 3).reshape(cout, -1)
    bot = torch.cat([lo4, torch.zeros_like(lo4)], 3).reshape(cout, -1)
    return torch.cat([top, bot], 0).contiguous()


def _dtype_code(dtype):
    """model-side dtype tag -> (C-ABI dtype, torch storage dtype)"""
    if isinstance(dtype, str):
        assert dtype in (X3, X3F), dtype
        return PGT_F16X3, X3_PLANE
    return {torch.float32: PGT_F32, torch.bfloat16: PGT_BF16, torch.float16: hip.PGT_F16}[dtype], dtype


def w2_rows(cout):
    """rows of the exact-weight operand of a layer with `cout` output channels: [32 rows w_hi | 32 rows w_lo * 2048] per 32 channels"""
    return (cout + 31) // 32 * 64


def pack_conv_weight(w, dtype, cin_pad=None, scale=None, fold=False, w2=False):
    """Reference weight -> the conv / linear kernels' operand, on the device (pgt_pack_conv_weight).  w: fp32 device tensor
    (Cout, Cin, KH, KW) (nn.Conv2d) or (Cout, Cin) (nn.Linear, or any K-major matrix); dtype: torch.float32 / bfloat16 /
    float16, or X3 / X3F (split-half: [w_hi | w_hi | w_lo] per 64-channel block; fold=True: the 64-output-channel folded form);
    cin_pad: zero-pad the input channels; scale: optional fp32 (Cout,) factor applied before rounding (BatchNorm fold).
    w2=True (half / bf16): the EXACT-WEIGHT operand (pgt_conv_desc::w2) - (w2_rows(Cout), K), two 16-bit planes per filter row;
    conv2d / linear take it with w2=Cout."""
    assert w.dtype == torch.float32 and w.dim() in (2, 4)
    w = w.contiguous()
    cout, cin = w.shape[0], w.shape[1]
    kh, kw = (w.shape[2], w.shape[3]) if w.dim() == 4 else (1, 1)
    cp = cin if cin_pad is None or cin_pad < cin else cin_pad
    code, st = _dtype_code(dtype)
    assert not (w2 and (fold or isinstance(dtype, str) or dtype == torch.float32)), "exact weights: single-plane 16-bit layers"
    L = hip.lib()
    form = 2 if w2 else int(bool(fold))
    nbytes = L.pgt_packed_weight_bytes(code, cout, cp, kh, kw, form)
    rows = w2_rows(cout) if w2 else (128 if fold else cout)
    out = torch.empty((rows, nbytes // (rows * torch.empty((), dtype=st).element_size())), device=w.device, dtype=st)
    hip.check(L.pgt_pack_conv_weight(code, _p(w), cout, cin, kh, kw, cp, _p(scale), form, _p(out), _stream()),
              "pgt_pack_conv_weight")
    return out


# Exact-weight layers: decoder stages can run with two-plane weights (two MFMAs per product, pgt_conv_desc::w2)
# instead of the mean-field compensation.  PGT_EXACT_W = comma-separated decoder stages ("512,32", "512,256,128,64,32", ...).
# DEFAULT: none.  Measured in round 6 (profiles/r6_a_exact_weight_spread.jsonl, r6_b_*): with stages 512 + 32 exact - the ones the
# oracle ablation of round 5 pointed at - the worst window of the third operating point goes from 9.5e-4 to 1.38e-3 dB, with EVERY
# decoder stage exact it is 1.03e-3, at -4.3 % frames/s: the residual of that figure is not weight rounding (DESIGN.md section 2.3 has
# what it is), so the form is an opt-in precision feature, not the default.
# PGT_WCOMP_STAGES = decoder stages whose layers keep the mean-field compensation (unset: all of them)
WCOMP_STAGES = (tuple(t for t in _os.environ["PGT_WCOMP_STAGES"].split(",") if t) if "PGT_WCOMP_STAGES" in _os.environ else None)
EXACT_W_STAGES = tuple(t for t in _os.environ.get("PGT_EXACT_W", "").split(",") if t)


def _w2_gn_ok(hw, cout, groups):
    """epilogue GroupNorm statistics in the exact-weight form of the phased kernel: whole channel groups per tile of 64 / 128 OUTPUT
    channels (half the tile's weight rows) and whole 512 / 256-row tiles per image (csrc/igemm.hip)"""
    wide = w2_rows(cout) > 128
    return cout % groups == 0 and (128 if wide else 64) % (cout // groups) == 0 and hw % (256 if wide else 512) == 0


def w2_ok(x, cout, cin, kh, kw, stride, pad, *, ups=False, act=ACT_NONE, res=None, post_relu=False, sft=None, out=None, out_f32=False,
          tile=(0, 0), scalar_epi=False, kernel=0, splitk=0, out_parity=None, out_rows=None, x3=False, out_x3=False, bias=None,
          gn=None, **_other):
    """Does the library have the exact-weight form (pgt_conv_desc::w2) for this launch?  Mirror of the dispatch in csrc/igemm.hip:
    the 64-channel ring kernel where it is legal, else the phased LDS-DMA kernel (IEEE half, Cin % 64 == 0, 16-byte epilogue)."""
    if x3 or out_x3 or x.dtype != torch.float16 or x.dim() != 4 or ups or splitk not in (0, 1) or scalar_epi or kernel not in (0, 4, 8):
        return False
    al = lambda t: t is None or (t.data_ptr() % 16 == 0 and (_ld_img(t) if t.dim() == 4 else t.stride(-2)) % 8 == 0)      # noqa: E731
    aligned = cout % 8 == 0 and al(out) and al(res) and (sft is None or (al(sft[0]) and al(sft[1])))
    if kernel != 4 and cin == 64 and (aligned or cout <= 16) and ring_covers(
            x, cout, kh, kw, stride, pad, ups=ups, act=act, res=res, post_relu=post_relu, sft=sft, out=out, out_f32=out_f32, tile=tile,
            scalar_epi=scalar_epi, kernel=kernel, splitk=splitk, out_parity=out_parity, out_rows=out_rows, bias=bias):
        return True       # the 64-channel ring kernel (two weight planes in registers)
    if cin % 64 or not aligned or kh * kw > 30 or _ld_img(x) % 8 or x.data_ptr() % 16 or tuple(tile)[0]:
        return False
    return w2_rows(cout) * kh * kw * cin * 2 < (1 << 31)


def fold_batchnorm(gamma, beta, mean, var, eps, bias=None):
    """eval-BatchNorm2d folded into the preceding conv: (scale, bias') fp32 (C,) on the device (pgt_fold_batchnorm)."""
    c = gamma.numel()
    scale = torch.empty((c,), dtype=torch.float32, device=gamma.device)
    out = torch.empty((c,), dtype=torch.float32, device=gamma.device)
    hip.check(hip.lib().pgt_fold_batchnorm(_p(gamma), _p(beta), _p(mean), _p(var), float(eps), _p(bias), c, _p(scale), _p(out), _stream()),
              "pgt_fold_batchnorm")
    return scale, out


def to_x3(x, out=None):
    """fp32 (..., C) -> split-half (..., 2C)."""
    assert x.dtype == torch.float32 and x.stride(-1) == 1
    c = x.shape[-1]
    x2 = x.reshape(-1, c) if x.is_contiguous() else None
    if x2 is None:
        assert x.dim() == 4
        rows, lds = x.shape[0] * x.shape[1] * x.shape[2], _ld_img(x)
    else:
        rows, lds = x2.shape[0], c
    if out is None:
        out = torch.empty(tuple(x.shape[:-1]) + (2 * c,), device=x.device, dtype=X3_PLANE)
    assert out.is_contiguous() and out.shape[-1] == 2 * c
    with _Prof("x3_convert", 0, _nb(x, out)):
        hip.check(hip.lib().pgt_x3_split(_p(x), lds, _p(out), 2 * c, c, rows, c, _stream()), "pgt_x3_split")
    return out


def x3_to_half(x, out=None):
    """split-half (..., 2C) -> IEEE half (..., C) (hi + lo rounded once: 11 significand bits; the hi plane alone has 8)."""
    assert x.dtype == X3_PLANE and x.is_contiguous() and x.shape[-1] % 2 == 0
    c = x.shape[-1] // 2
    if out is None:
        out = torch.empty(tuple(x.shape[:-1]) + (c,), device=x.device, dtype=torch.float16)
    assert out.dtype == torch.float16 and out.is_contiguous() and out.shape[-1] == c
    with _Prof("x3_convert", 0, _nb(x, out)):
        hip.check(hip.lib().pgt_x3_to_half(_p(x), 2 * c, c, _p(out), c, x.numel() // (2 * c), c, _stream()), "pgt_x3_to_half")
    return out


def from_x3(x):
    """split-half (..., 2C) -> fp32 (..., C)."""
    assert x.dtype == X3_PLANE and x.is_contiguous() and x.shape[-1] % 2 == 0
    c = x.shape[-1] // 2
    out = torch.empty(tuple(x.shape[:-1]) + (c,), device=x.device, dtype=torch.float32)
    with _Prof("x3_convert", 0, _nb(x, out)):
        hip.check(hip.lib().pgt_x3_merge(_p(x), 2 * c, c, _p(out), c, x.numel() // (2 * c), c, _stream()), "pgt_x3_merge")
    return out


# When set to a list, every op brackets its launch with events on the launch stream and appends
# {"kernel", "flops", "bytes", "events", ...} records (bench.py's live per-kernel roofline measurement): `flops` counts every
# reference product once (2 per multiply-add), `bytes` the algorithmic HBM traffic (operands once, results once).
PROFILE = None


class _Prof:
    """with _Prof("layernorm", flops, bytes): <launch>   - no-op unless PROFILE is a list"""
    __slots__ = ("rec", "e0")

    def __init__(self, kernel, flops, nbytes, **extra):
        self.rec = None
        if PROFILE is not None:
            self.rec = dict(kernel=kernel, flops=float(flops), bytes=float(nbytes), **extra)

    def __enter__(self):
        if self.rec is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *exc):
        if self.rec is not None and exc[0] is None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            self.rec["events"] = (self.e0, e1)
            PROFILE.append(self.rec)
        return False


def _nb(*ts):
    """bytes of the logical contents of tensors / views (None skipped)"""
    return sum(t.numel() * t.element_size() for t in ts if t is not None)


def _dt(t):
    if t.dtype == torch.float32:
        return PGT_F32
    if t.dtype == torch.bfloat16:
        return PGT_BF16
    if t.dtype == torch.float16:
        return hip.PGT_F16
    raise TypeError(f"unsupported activation dtype {t.dtype}")


def _dev(t):
    if not t.is_cuda:
        raise hip.PgtError("pgtformer_amd ops run on the GPU only (tensor is on %s); no CPU fallback" % t.device)
    if t.device.index != torch.cuda.current_device():
        # kernels are launched on the current device's stream: a tensor of another device would be a foreign pointer there
        raise hip.PgtError(f"tensor on {t.device} but the current device is cuda:{torch.cuda.current_device()}: "
                           "call torch.cuda.set_device(model.dev) (one process per GPU)")
    return t


def _p(t):
    if t is None:
        return None
    if hip.TRACE_TENSORS is not None:      # export: which storage a pointer argument belongs to (pgtformer_amd.export)
        st = t.untyped_storage()
        # (the storage object's address tells two tensors apart that the allocator placed at the same address one after the other)
        hip.TRACE_TENSORS[t.data_ptr()] = (st.data_ptr(), st.nbytes(), st._cdata)
    return C.c_void_p(_dev(t).data_ptr())


def _stream(t=None):
    """HIP stream the launch goes to: the current stream of the tensor's device (ops assert elsewhere that the tensors of
    one call live on one device); without a tensor, of the current device."""
    dev = None if t is None else t.device
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def _ld_img(x):
    """pixel stride of an (N,H,W,C) view; validates that N,H,W are densely packed over that stride."""
    n, h, w, c = x.shape
    assert c == 1 or x.stride(3) == 1, "channel dim must be contiguous"
    if w > 1:
        ld = x.stride(2)
    elif h > 1:
        ld = x.stride(1)
    elif n > 1:
        ld = x.stride(0)
    else:
        ld = c
    assert ld >= c, (x.shape, x.stride())
    assert h == 1 or x.stride(1) == w * ld, f"rows not dense: {x.stride()} {tuple(x.shape)}"
    assert n == 1 or x.stride(0) == h * w * ld, f"images not dense: {x.stride()} {tuple(x.shape)}"
    return ld


def _ld_rows(x):
    assert x.dim() == 2 and (x.stride(1) == 1 or x.shape[1] == 1)
    return x.stride(0) if x.shape[0] > 1 else max(x.shape[1], x.stride(0))


# Per-shape kernel selection ("find" mode): with enable_autotune() every new bf16 conv/linear signature is timed once
# (outside graph capture) over the built kernel/tile variants and the fastest one is cached; the library's static
# heuristic stays one of the candidates, so tuning never loses to it.  fp32 (parity) launches are never tuned.
AUTOTUNE = None
_TUNE_CANDIDATES = ((0, (0, 0), 0), (1, (64, 64), 0), (1, (64, 128), 0), (1, (128, 64), 0), (1, (128, 128), 0),
                    (4, (0, 256), 0), (4, (0, 128), 0), (5, (0, 0), 0), (6, (0, 0), 0))


def enable_autotune(flag=True):
    global AUTOTUNE
    AUTOTUNE = ({} if AUTOTUNE is None else AUTOTUNE) if flag else None


_TUNE_KEY_LEN = 25      # fields of an AUTOTUNE key (conv2d); a cache file with another arity is stale


def _tune_tag():
    """library version + device architecture a tuned table is valid for (kernel variants / legality change with both)."""
    arch = ""
    if torch.cuda.is_available():
        arch = getattr(torch.cuda.get_device_properties(torch.cuda.current_device()), "gcnArchName", "")
    return {"pgt_version": hip.lib().pgt_version().decode(), "arch": arch}


def save_autotune(path):
    """Write the tuned (signature -> kernel variant) table as JSON (reload with load_autotune to skip re-tuning)."""
    import json
    rows = [[list(k[:9]) + [list(k[9])] + list(k[10:]), [v[0], list(v[1]), v[2]]] for k, v in (AUTOTUNE or {}).items()]
    with open(path, "w") as f:
        json.dump({"tag": _tune_tag(), "table": rows}, f)


def load_autotune(path):
    """Load a saved table; returns False (and leaves tuning to start afresh) when the file was written by another library
    version or for another GPU architecture."""
    import json
    blob = json.load(open(path))
    if not isinstance(blob, dict) or blob.get("tag") != _tune_tag():
        return False
    if any(len(k) != _TUNE_KEY_LEN for k, _ in blob["table"]):      # written with another key layout: start afresh (and re-save)
        return False
    enable_autotune()
    for k, v in blob["table"]:
        AUTOTUNE[tuple(k[:9]) + (tuple(k[9]),) + tuple(k[10:])] = (v[0], tuple(v[1]), v[2])
    return True


def _tune_conv(d, args, device, iters=4, gn_ws=None):
    L = hip.lib()
    best, best_t = _TUNE_CANDIDATES[0], None
    for cand in _TUNE_CANDIDATES:
        if gn_ws is not None and cand[0] in (2, 5, 6):
            continue       # statistics epilogue: kernels 1 and 4 (the heuristic picks among them)
        d.kernel, (d.force_bm, d.force_bn), d.stages = cand
        ws_bytes = L.pgt_conv2d_workspace_bytes(C.byref(d))
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=device) if ws_bytes else None
        if gn_ws is not None:
            call = lambda: L.pgt_conv2d_gn(C.byref(d), *args, _p(gn_ws), _p(ws), ws_bytes, _stream())  # noqa: E731
        else:
            call = lambda: L.pgt_conv2d_ws(C.byref(d), *args, _p(ws), ws_bytes, _stream())  # noqa: E731
        if call() != 0:
            continue   # variant not legal for this shape
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            call()
        e1.record()
        e1.synchronize()
        t = e0.elapsed_time(e1)
        if best_t is None or t < best_t * (0.97 if cand[0] else 1.0):
            best, best_t = cand, t
    return best


def conv2d(x, w, bias=None, *, kh=1, kw=1, stride=1, pad=(0, 0, 0, 0), ups=False, act=ACT_NONE, res=None,
           post_relu=False, sft=None, out=None, out_f32=False, tile=(0, 0), scalar_epi=False, kernel=0, splitk=0, stages=0,
           out_parity=None, out_rows=None, x3=False, gn=None, x3_fold=False, out_x3=False, affine_in=None, w2=None):
    """Implicit-GEMM conv. x: (N,H,W,Cin); w: (Cout, kh*kw*Cin) packed; pad=(top,bottom,left,right).
    w2: None, or Cout: w is the EXACT-WEIGHT operand pack_conv_weight(..., w2=True) wrote ((w2_rows(Cout), kh*kw*Cin): two 16-bit
    planes per filter row, two MFMAs per product; launches w2_ok accepts).
    sft=(dec, shift, w_scalar) selects the SFT epilogue. Returns (N,Ho,Wo,Cout).
    out_parity=(py, px): write the (N,Ho,Wo,Cout) result to out[:, py::2, px::2, :] of a required (N,2Ho,2Wo,Cout) `out`
    (sub-pixel convolutions; plain epilogue only).
    out_rows=(mul, xmul, off): general form - output pixel m is written to row mul*m + xmul*(m % Wo) + off of `out`
    (any (..., Cout) view; its pixel stride is the row pitch).
    gn: None, or a number of groups: the epilogue also reduces the GroupNorm statistics of the output, attached to the
    returned tensor as `._pgt_gn` (ops.GnStats) for the GroupNorm that follows; or (GnStats, sub) when several launches
    write one tensor (the caller binds the statistics to the tensor after the last launch).
    out_x3: fp32 x / w, result stored as split-half planes (N,Ho,Wo,2*Cout) (pgt_conv_desc::out_split).
    affine_in=(scale, shift, act): the conv reads act(x * scale[n, c] + shift[n, c]) - the GroupNorm apply + SiLU of the
    Normalize in front of it; fused into the operand load where the library has that form (affine_in_fuses), else one
    affine_act pass."""
    n, h, wd, cin = x.shape
    cout = w.shape[0]
    if w2 is not None:
        cout = int(w2)
        assert w.shape[0] == w2_rows(cout) and not x3 and not x3_fold and not out_x3 and x.dtype == torch.float16, (w.shape, cout, x.dtype)
    if affine_in is not None and not affine_in_fuses(x, cout, kh, kw, stride, pad, ups=ups, act=act, res=res, post_relu=post_relu, sft=sft,
                                                    out=out, out_f32=out_f32, tile=tile, scalar_epi=scalar_epi, kernel=kernel, splitk=splitk,
                                                    out_parity=out_parity, out_rows=out_rows, x3=x3, out_x3=out_x3, bias=bias, force=w2 is not None):
        x = affine_act(x, affine_in[0], affine_in[1], affine_in[2], x3=x3)
        affine_in = None
    if out_x3:
        assert x.dtype == torch.float32 and not x3 and not out_f32 and sft is None and out_rows is None and out_parity is None and res is None
    if x3:   # split-half operands: x (N,H,W,2*Cin) = [hi | lo], w (Cout, kh*kw*3*Cin), y (N,Ho,Wo,2*Cout) unless out_f32
        assert x.dtype == X3_PLANE and cin % 2 == 0 and sft is None and not ups and out_rows is None and out_parity is None
        cin //= 2
        if x3_fold:   # 64 output channels, w = pack_x3_fold_weight(...): (128, kh*kw*2*Cin)
            assert cout == 128 and w.shape[1] == kh * kw * 2 * cin and gn is None, (w.shape, kh, kw, cin)
            cout = 64
        else:
            assert w.shape[1] == kh * kw * 3 * cin, (w.shape, kh, kw, cin)
    else:
        assert w.shape[1] == kh * kw * cin, (w.shape, kh, kw, cin)
    assert w.dtype == x.dtype and w.is_contiguous()
    cst = 2 * cout if (out_x3 or (x3 and not out_f32)) else cout      # stored output channels
    if n > 1 and n * h * wd * _ld_img(x) * x.element_size() >= (1 << 31) and out_rows is None and out_parity is None:
        # the kernels take 32-bit byte offsets: run a >= 2 GiB input as frame chunks (frames are independent)
        hv0, wv0 = (h * 2, wd * 2) if ups else (h, wd)
        ho0 = (hv0 + pad[0] + pad[1] - kh) // stride + 1
        wo0 = (wv0 + pad[2] + pad[3] - kw) // stride + 1
        if out is None:
            out = torch.empty((n, ho0, wo0, cst), device=x.device, dtype=torch.float32 if out_f32 else (X3_PLANE if out_x3 else x.dtype))
        per = max(1, ((1 << 31) - 1) // (h * wd * _ld_img(x) * x.element_size()))
        st = None
        if gn is not None and not isinstance(gn, tuple) and USE_EPILOGUE_GN and gn_ok(n, ho0 * wo0, cout, gn, cin, kh):
            # the statistics workspace records ONE tile height for the whole tensor (gn_finalize_conv_kernel), and the
            # kernel / tile choice depends on the launch's frame count: every chunk must therefore be the same launch
            # shape.  Equal chunks (the largest divisor of n that fits); if n has no useful divisor the GroupNorm that
            # follows takes its separate statistics pass instead.
            eq = max(d for d in range(1, per + 1) if n % d == 0)
            if 2 * eq > per or eq == n:
                per = eq
                st = GnStats(n, 1, ho0 * wo0, cout, gn, x.device)
        for i in range(0, n, per):
            sl = slice(i, min(n, i + per))
            fpi = bias.shape[0] // n if (bias is not None and bias.dim() == 2) else 0      # bias vectors (frames) per image
            conv2d(x[sl], w, bias if fpi == 0 else bias[sl.start * fpi:sl.stop * fpi], kh=kh, kw=kw, stride=stride, pad=pad, ups=ups, act=act,
                   res=None if res is None else res[sl], post_relu=post_relu,
                   sft=None if sft is None else (sft[0][sl], sft[1][sl], sft[2]), out=out[sl], out_f32=out_f32,
                   tile=tile, scalar_epi=scalar_epi, kernel=kernel, splitk=splitk, stages=stages, x3=x3,
                   gn=None if st is None else (st, 0, i), x3_fold=x3_fold, out_x3=out_x3, w2=w2,
                   affine_in=None if affine_in is None else (affine_in[0][sl], affine_in[1][sl], affine_in[2]))
        return out if st is None else st.bind(out, cst)
    hv, wv = (h * 2, wd * 2) if ups else (h, wd)
    ho = (hv + pad[0] + pad[1] - kh) // stride + 1
    wo = (wv + pad[2] + pad[3] - kw) // stride + 1
    if out_parity is not None:
        assert out is not None and tuple(out.shape) == (n, 2 * ho, 2 * wo, cout) and res is None and sft is None
        out_rows = (4, -2, out_parity[0] * 2 * wo + out_parity[1])
    elif out_rows is not None:
        assert out is not None and out.shape[-1] == cout and res is None and sft is None
    else:
        if out is None:
            out = torch.empty((n, ho, wo, cst), device=x.device, dtype=torch.float32 if out_f32 else (X3_PLANE if out_x3 else x.dtype))
        assert tuple(out.shape) == (n, ho, wo, cst), (out.shape, (n, ho, wo, cst))
    d = hip.ConvDesc()
    d.dtype = PGT_F16X3 if x3 else _dt(x)
    d.N, d.H, d.W, d.Cin, d.ldx, d.ups = n, h, wd, cin, _ld_img(x), int(ups)
    d.KH, d.KW, d.stride, d.pad_t, d.pad_l = kh, kw, stride, pad[0], pad[2]
    d.Ho, d.Wo, d.Cout, d.ldy = ho, wo, cout, _ld_img(out)
    d.act, d.post_relu = act, int(post_relu)
    d.ldr = _ld_img(res) if res is not None else 0
    d.out_f32 = int(out_f32)
    d.force_bm, d.force_bn = tile
    d.scalar_epilogue = int(scalar_epi)
    d.kernel = int(kernel)
    d.splitk = int(splitk)
    d.stages = int(stages)
    d.x3_fold = int(bool(x3_fold))
    d.out_split = int(bool(out_x3))
    d.w2 = int(w2 is not None)
    if bias is not None and bias.dim() == 2:      # one bias vector per frame (mean_field_bias): (frames, Cout) fp32
        assert bias.shape[1] == cout and bias.is_contiguous() and (n * ho * wo) % bias.shape[0] == 0 and bias.shape[0] % n == 0, (bias.shape, n, cout)
        d.bias_rows = (n * ho * wo) // bias.shape[0]
    if out_rows is not None:
        d.orow_mul, d.orow_xmul, d.orow_off = out_rows
    st = None        # epilogue GroupNorm statistics
    if gn is not None:
        if isinstance(gn, tuple):
            st, d.gn_sub = gn[0], gn[1]
            d.gn_img0, d.gn_nimg = (gn[2], st.n) if len(gn) > 2 else (0, 0)
            assert st.hw == ho * wo and st.c == cout
        elif (USE_EPILOGUE_GN and gn_ok(n, ho * wo, cout, gn, cin, kh) and not out_f32 and not scalar_epi
              and kernel in (0, 1, 4) and splitk in (0, 1) and (w2 is None or _w2_gn_ok(ho * wo, cout, gn))):
            st = GnStats(n, 1, ho * wo, cout, gn, x.device)
        if st is not None:
            d.gn_groups, d.gn_nsub = st.groups, st.nsub
    dec = shift = None
    if sft is not None:
        dec, shift, sw = sft
        d.epi, d.ld_dec, d.ld_shift, d.sft_w = EPI_SFT, _ld_img(dec), _ld_img(shift), float(sw)
        assert dec.dtype == x.dtype and shift.dtype == x.dtype
    if res is not None and x3 and res.dtype == torch.float32:
        assert out_f32 and tuple(res.shape) == tuple(out.shape)      # split-half arithmetic on fp32-stored tensors
        d.res_f32 = 1
    elif res is not None:
        assert res.dtype == x.dtype and tuple(res.shape[:-1]) == tuple(out.shape[:-1]) and res.shape[-1] == (2 * cout if x3 else cout)
    prof = PROFILE
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    L = hip.lib()
    if (AUTOTUNE is not None and kernel == 0 and tile == (0, 0) and splitk == 0 and not scalar_epi
            and x.dtype == torch.bfloat16 and not x3 and affine_in is None):
        key = (n, h, wd, cin, d.ldx, d.ups, kh, kw, stride, tuple(pad), cout, d.ldy, act, d.post_relu, d.ldr, d.epi,
               d.ld_dec, d.ld_shift, d.out_f32, bias is None, d.orow_mul, d.orow_xmul, d.orow_off, d.gn_groups, d.bias_rows)
        cfg = AUTOTUNE.get(key)
        if cfg is None and not torch.cuda.is_current_stream_capturing():
            if L.pgt_conv2d_workspace_bytes(C.byref(d)):
                cfg = AUTOTUNE[key] = (0, (0, 0), 0)   # split-K layer: every candidate would time the same split-K path
            else:
                cfg = AUTOTUNE[key] = _tune_conv(d, (_p(x), _p(w), _p(bias), _p(res), _p(dec), _p(shift), _p(out)), x.device,
                                                 gn_ws=None if st is None else st.ws)
        d.kernel, (d.force_bm, d.force_bn), d.stages = cfg if cfg is not None else (0, (0, 0), 0)
    ws_bytes = L.pgt_conv2d_workspace_bytes(C.byref(d))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device) if ws_bytes else None   # split-K scratch
    if affine_in is not None:
        a_sc, a_sh, a_act = affine_in
        assert st is None and sft is None and a_sc.shape == (n, cin) and a_sh.shape == (n, cin) and a_sc.is_contiguous() and a_sh.is_contiguous()
        if not L.pgt_conv2d_affine_in_ok(C.byref(d)):
            raise hip.PgtError("conv2d: ops.affine_in_fuses and pgt_conv2d_affine_in_ok disagree on this launch")
        hip.check(L.pgt_conv2d_affine_in(C.byref(d), _p(x), _p(a_sc), _p(a_sh), int(a_act), _p(w), _p(bias), _p(res), _p(out), _stream()),
                  "pgt_conv2d_affine_in")
    elif st is not None:
        hip.check(L.pgt_conv2d_gn(C.byref(d), _p(x), _p(w), _p(bias), _p(res), _p(dec), _p(shift), _p(out), _p(st.ws), _p(ws),
                                  ws_bytes, _stream()), "pgt_conv2d_gn")
        if not isinstance(gn, tuple):
            st.bind(out, cst)
    else:
        hip.check(L.pgt_conv2d_ws(C.byref(d), _p(x), _p(w), _p(bias), _p(res), _p(dec), _p(shift), _p(out), _p(ws),
                                  ws_bytes, _stream()), "pgt_conv2d")
    if prof is not None:
        e1.record()
        m = n * ho * wo
        es = x.element_size()
        prof.append({"kernel": "igemm", "flops": 2.0 * m * cout * kh * kw * cin,
                     "bytes": float(n * h * wd * x.shape[-1] * es + m * out.shape[-1] * out.element_size() + w.numel() * es
                                    + (m * res.shape[-1] * es if res is not None else 0)),
                     "shape": (n, h, wd, cin, cout, kh, stride, int(ups)), "events": (e0, e1),
                     "cfg": (d.kernel, d.force_bm, d.force_bn), "x3": bool(x3), "dt": str(x.dtype).replace("torch.", ""),
                     "w2": w2 is not None})
    return out


def ring_covers(x, cout, kh, kw, stride, pad, *, ups=False, act=ACT_NONE, res=None, post_relu=False, sft=None, out=None,
                out_f32=False, tile=(0, 0), scalar_epi=False, kernel=0, splitk=0, out_parity=None, out_rows=None, x3=False,
                out_x3=False, bias=None, **_other):
    """Does the 64-channel ring kernel (csrc/igemm8.hip) cover this launch?  Mirror of ring_legal (csrc/igemm.hip)."""
    if x3 or out_x3 or x.dtype not in (torch.float16, torch.bfloat16) or x.dim() != 4:
        return False
    n, h, wd, cin = x.shape
    p2 = lambda v: v > 0 and (v & (v - 1)) == 0      # noqa: E731
    if not (cin == 64 and 1 <= cout <= 64 and kh == 3 and kw == 3 and stride == 1 and tuple(pad) == (1, 1, 1, 1) and not ups):
        return False
    if not (p2(wd) and p2(h) and wd >= 128 and h >= 4 and _ld_img(x) % 8 == 0):
        return False
    if act != ACT_NONE or post_relu or sft is not None or out_parity is not None or out_rows is not None or scalar_epi:
        return False
    if kernel not in (0, 8) or tuple(tile) != (0, 0) or splitk not in (0, 1):
        return False
    if bias is not None and bias.dim() == 2 and ((n * h * wd) // bias.shape[0]) % (4 * wd):
        return False      # a strip of the ring kernel (>= 4 rows x 128 columns) takes one bias vector: bias bands must be whole strips
    if n * h * wd * _ld_img(x) * x.element_size() >= (1 << 31):
        return True if n > 1 else False       # (conv2d runs frame chunks: each chunk is asked again)
    aligned = cout % 8 == 0 and (out is None or (_ld_img(out) % 8 == 0 and out.data_ptr() % 16 == 0)) and \
        (res is None or (_ld_img(res) % 8 == 0 and res.data_ptr() % 16 == 0))
    if res is not None and n * h * wd * _ld_img(res) * res.element_size() >= (1 << 32):
        return False
    return aligned or cout <= 32


def affine_in_fuses(x, cout, kh, kw, stride, pad, *, force=False, **kw_):
    """Does the library have the fused-operand form (pgt_conv2d_affine_in) for this conv?  The launches the ring kernel covers -
    the library re-checks (pgt_conv2d_affine_in_ok) and conv2d raises if the two ever disagree.  force: whatever the
    PGT_FUSE_GN_APPLY switch says (exact-weight launches on the ring kernel keep their fused operand)."""
    return (USE_FUSED_GN_APPLY or force) and ring_covers(x, cout, kh, kw, stride, pad, **kw_)


def linear(x, w, bias=None, *, act=ACT_NONE, res=None, out=None, out_f32=False, x3=False, gn=None, w2=None):
    """x: (rows, Cin) -> (rows, Cout); x3: split-half rows (rows, 2*Cin) -> (rows, 2*Cout) (or fp32 (rows, Cout)).
    gn=(groups, n_images): the rows are n_images images of rows / n_images tokens each; the epilogue leaves the GroupNorm
    statistics of the output per image (attached to the returned tensor, see conv2d).  w2: see conv2d."""
    rows, cin = x.shape
    cst = (w.shape[0] if w2 is None else int(w2)) * (2 if x3 and not out_f32 else 1)
    if out is None:
        out = torch.empty((rows, cst), device=x.device, dtype=torch.float32 if out_f32 else x.dtype)
    nimg = gn[1] if gn is not None else 1
    if gn is None and rows * _ld_rows(x) * x.element_size() >= (1 << 31):
        # the kernels take 32-bit byte offsets: present a >= 2 GiB row matrix as several equal "images", which conv2d runs
        # as chunks (whole frames per image when the bias is per frame)
        per = ((1 << 31) - 1) // (_ld_rows(x) * x.element_size())
        frames = bias.shape[0] if (bias is not None and bias.dim() == 2) else 0
        nimg = next((d for d in range(2, 65537) if rows % d == 0 and rows // d <= per and (frames == 0 or frames % d == 0)), None)
        assert nimg is not None, f"linear: cannot split {rows} rows into < 2 GiB chunks"
    assert rows % nimg == 0
    hw = rows // nimg
    x4 = x.as_strided((nimg, 1, hw, cin), (hw * _ld_rows(x), 0, _ld_rows(x), 1))
    o4 = out.as_strided((nimg, 1, hw, cst), (hw * _ld_rows(out), 0, _ld_rows(out), 1))
    r4 = None if res is None else res.as_strided((nimg, 1, hw, cst), (hw * _ld_rows(res), 0, _ld_rows(res), 1))
    o4 = conv2d(x4, w, bias, act=act, res=r4, out=o4, out_f32=out_f32, x3=x3, gn=None if gn is None else gn[0], w2=w2)
    st = getattr(o4, "_pgt_gn", None)
    return out if st is None else st.bind(out, cst)


def groupnorm_affine(x, gamma, beta, groups=32, eps=1e-6, x3=False):
    """GroupNorm statistics of x (N,H,W,C) folded with gamma/beta -> (scale, shift) fp32 (N,C).  When the conv that produced
    x left its statistics (x._pgt_gn, ops.GnStats) they are finalised instead of re-reading x."""
    n, h, w, c = x.shape
    L = hip.lib()
    st = getattr(x, "_pgt_gn", None)
    if st is not None and USE_EPILOGUE_GN:
        cl = c // 2 if x3 else c
        if (st.ptr == x.data_ptr() and st.n == n and st.nsub * st.hw == h * w and st.c == cl and st.groups == groups
                and st.ld[1] == c):
            scale = torch.empty((n, cl), dtype=torch.float32, device=x.device)
            shift = torch.empty((n, cl), dtype=torch.float32, device=x.device)
            hip.check(L.pgt_groupnorm_from_partials(_p(st.ws), n, st.nsub, st.hw, cl, groups, eps, _p(gamma), _p(beta), _p(scale),
                                                    _p(shift), _stream()), "pgt_groupnorm_from_partials")
            return scale, shift
    if x3:
        c //= 2
        nbytes = L.pgt_groupnorm_workspace_bytes(n, h * w, c, groups)
        ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=x.device)
        scale = torch.empty((n, c), dtype=torch.float32, device=x.device)
        shift = torch.empty((n, c), dtype=torch.float32, device=x.device)
        with _Prof("groupnorm_stats", 0, _nb(x)):
            hip.check(L.pgt_groupnorm_affine_x3(_p(x), _ld_img(x), c, n, h * w, c, groups, eps, _p(gamma), _p(beta),
                                                _p(scale), _p(shift), _p(ws), nbytes, _stream()), "pgt_groupnorm_affine_x3")
        return scale, shift
    nbytes = L.pgt_groupnorm_workspace_bytes(n, h * w, c, groups)
    ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=x.device)
    scale = torch.empty((n, c), dtype=torch.float32, device=x.device)
    shift = torch.empty((n, c), dtype=torch.float32, device=x.device)
    with _Prof("groupnorm_stats", 0, _nb(x)):
        hip.check(L.pgt_groupnorm_affine(_dt(x), _p(x), _ld_img(x), n, h * w, c, groups, eps, _p(gamma), _p(beta),
                                         _p(scale), _p(shift), _p(ws), nbytes, _stream()), "pgt_groupnorm_affine")
    return scale, shift


def affine_act(x, scale, shift, act=ACT_NONE, out=None, x3=False):
    """y = act(x*scale[n,c] + shift[n,c]); x (N,H,W,C)."""
    n, h, w, c = x.shape
    if out is None:
        out = torch.empty((n, h, w, c), device=x.device, dtype=x.dtype)
    with _Prof("norm_apply_act", 0, _nb(x, out)):
        if x3:
            hip.check(hip.lib().pgt_affine_act_x3(_p(x), _ld_img(x), c // 2, _p(out), _ld_img(out), c // 2, n, h * w, c // 2,
                                                  _p(scale), _p(shift), act, _stream()), "pgt_affine_act_x3")
        else:
            hip.check(hip.lib().pgt_affine_act(_dt(x), _p(x), _ld_img(x), _p(out), _ld_img(out), n, h * w, c, _p(scale),
                                               _p(shift), act, _stream()), "pgt_affine_act")
    return out


def groupnorm_act(x, gamma, beta, act=ACT_SILU, groups=32, eps=1e-6, out=None, x3=False):
    scale, shift = groupnorm_affine(x, gamma, beta, groups, eps, x3=x3)
    return affine_act(x, scale, shift, act, out=out, x3=x3)


def layernorm(x, gamma, beta, eps=1e-5, pos=None, x3=False):
    """x (rows,C). Returns LN(x), or (LN(x), LN(x)+pos) when pos is given."""
    rows, c = x.shape
    if x3:
        c //= 2
        y = torch.empty((rows, 2 * c), device=x.device, dtype=x.dtype)
        y2 = torch.empty((rows, 2 * c), device=x.device, dtype=x.dtype) if pos is not None else None
        with _Prof("layernorm", 0, _nb(x, y, pos, y2)):
            hip.check(hip.lib().pgt_layernorm_x3(_p(x), _ld_rows(x), c, rows, c, _p(gamma), _p(beta), eps, _p(y), 2 * c, c,
                                                 _p(pos), _ld_rows(pos) if pos is not None else 0, c, _p(y2), 2 * c, c,
                                                 _stream()), "pgt_layernorm_x3")
        return y if pos is None else (y, y2)
    y = torch.empty((rows, c), device=x.device, dtype=x.dtype)
    y2 = torch.empty((rows, c), device=x.device, dtype=x.dtype) if pos is not None else None
    with _Prof("layernorm", 0, _nb(x, y, pos, y2)):
        hip.check(hip.lib().pgt_layernorm(_dt(x), _p(x), _ld_rows(x), rows, c, _p(gamma), _p(beta), eps, _p(y), c,
                                          _p(pos), _ld_rows(pos) if pos is not None else 0, _p(y2), c, _stream()),
                  "pgt_layernorm")
    return y if pos is None else (y, y2)


def channel_stats(x, want_var=True):
    """per-(n,c) mean and unbiased variance over pixels of x (N,H,W,C) -> fp32 (N,C)."""
    n, h, w, c = x.shape
    mean = torch.empty((n, c), dtype=torch.float32, device=x.device)
    var = torch.empty((n, c), dtype=torch.float32, device=x.device) if want_var else None
    hip.check(hip.lib().pgt_channel_stats(_dt(x), _p(x), _ld_img(x), n, h * w, c, _p(mean), _p(var), _stream()),
              "pgt_channel_stats")
    return mean, var


def sampled_channel_mean(x):
    """fp32 (N, C): mean of x (N,H,W,C) - or (N, HW, C) - per frame and channel over the library's fixed pixel sample (<= 1024
    pixels of a frame: pgt_sampled_channel_mean)."""
    if x.dim() == 3:
        x = x.unsqueeze(1)
    n, h, w, c = x.shape
    mean = torch.empty((n, c), dtype=torch.float32, device=x.device)
    with _Prof("mean_field", 0, float(n * min(h * w, 1024) * c * x.element_size())):
        hip.check(hip.lib().pgt_sampled_channel_mean(_dt(x), _p(x), _ld_img(x), n, h * w, c, _p(mean), _stream()), "pgt_sampled_channel_mean")
    return mean


def mean_field_bias(mean, defect_t, bias=None):
    """(R, Cout) fp32 per-frame bias = bias + mean (R, K) @ defect_t (K, Cout): the frame-constant part of the error a layer makes
    with its weights rounded to 16 bits, put back (pgt_mean_field_bias; conv2d / linear take the result as `bias`)."""
    r, k = mean.shape
    assert defect_t.shape[0] == k and defect_t.is_contiguous() and mean.is_contiguous() and mean.dtype == defect_t.dtype == torch.float32
    cout = defect_t.shape[1]
    out = torch.empty((r, cout), dtype=torch.float32, device=mean.device)
    with _Prof("mean_field", 2.0 * r * k * cout, float(defect_t.numel() * 4)):
        hip.check(hip.lib().pgt_mean_field_bias(_p(mean), _p(defect_t), _p(bias), r, k, cout, _p(out), _stream()), "pgt_mean_field_bias")
    return out


# pgt_frame_bias: per-frame arrival counters (zero between calls).  Consecutive calls on one stream share them; launches that run
# CONCURRENTLY need their own: the driver's lanes set LANE around their forwards (driver.WindowRunner), the model sets BRANCH around the
# sub-network it runs on a second stream.  Allocated once per
# (device, lane) OUTSIDE any graph capture (the lanes' eager warm-up passes come first) and never freed.
LANE = 0
BRANCH = 0          # 1 inside the condition branch that PGTFormer forks onto a second stream (archs/pgtformer_arch.py): in the pure-bf16 mode
#                     BiSeNet's convolutions are compensated too, and its frame_bias launches run CONCURRENTLY with the encoder's
_FB_COUNTERS = {}
_FB_MAX_FRAMES = 4096
_LANE_IDS = [0]


def new_lane_id():
    """a process-unique counter-set id for something that launches forwards CONCURRENTLY with others on the same device (every lane
    of every driver.WindowRunner takes one: two runners, two models or two threads never share arrival counters; lane 0 is the
    plain eager caller)"""
    _LANE_IDS[0] += 1
    return _LANE_IDS[0]


def _fb_counters(device, n):
    key = (str(device), LANE, BRANCH)
    c = _FB_COUNTERS.get(key)
    if c is None:
        if torch.cuda.is_current_stream_capturing():
            raise hip.PgtError("frame_bias: first use for lane %d / branch %d inside a graph capture (run one eager forward first)" % (LANE, BRANCH))
        c = _FB_COUNTERS[key] = torch.zeros(_FB_MAX_FRAMES, dtype=torch.int32, device=device)
    if n > _FB_MAX_FRAMES:
        raise hip.PgtError("frame_bias: %d bias rows in one launch (limit %d): run the batch as chunks" % (n, _FB_MAX_FRAMES))
    return c


# The mean field of the weight-rounding compensation is taken per horizontal BAND of a frame: WCOMP_BANDS bias vectors per frame, each
# for hw / bands consecutive output rows of the raster (pgt_conv_desc::bias_rows) - the part of (W - W16) x that varies slowly down
# the image is put back as well.  Measured on the two operating points (profiles/r5_i_bands.jsonl): max |dPSNR| 7.2e-4 / 1.02e-3 with
# one vector per frame, 7.2e-4 / 7.3e-4 with 16 bands.  PGT_WCOMP_BANDS=1: one vector per frame (rounds 3-4).
WCOMP_BANDS = int(_os.environ.get("PGT_WCOMP_BANDS", "16"))


def banded(x, bands=None):
    """(view, b): x (N,H,W,C) image batch as (N*b, H*W/b, C) - b horizontal bands per frame, the largest power-of-two divisor of
    `bands` for which a band is a whole number of 512-pixel tiles (b = 1: the frames themselves)"""
    n, h, w, c = x.shape
    hw = h * w
    b = WCOMP_BANDS if bands is None else bands
    while b > 1 and (hw % b or (hw // b) % 512):
        b //= 2
    b = max(1, b)
    ld = _ld_img(x)
    return x.as_strided((n * b, hw // b, c), ((hw // b) * ld, ld, 1), x.storage_offset()), b


_BAND_CELLS_MIN = int(_os.environ.get("PGT_WCOMP_CELLS", "16"))      # (A/B: smallest sample of a band, in cells of 16 pixels)


def band_sample_cells(b):
    """cells of the pixel sample of one band when a frame is cut in b bands: 64 (x 16 pixels) for whole frames, never fewer than 16"""
    return 0 if b <= 1 else max(_BAND_CELLS_MIN, 64 // b)


def frame_bias(x, defect_t, bias=None, affine_in=None, groups=1, scale_div=1, sample_cells=0):
    """(N, Cout) fp32 per-frame bias of a compensated 16-bit layer in ONE launch (pgt_frame_bias): bias + mean_n @ defect_t with
    mean_n the sampled channel mean of frame n of x (N,H,W,K) / (N,HW,K) - or of act(x * scale + shift) rounded to x.dtype when
    the layer reads its operand through the fused GroupNorm apply (affine_in=(scale, shift, act)).
    groups = G > 1: defect_t (K, G * Csub) and bias (G * Csub) hold G layers that read the same operand side by side; returns
    (G, N, Csub) - out[g] is layer g's contiguous (N, Csub) bias matrix.
    scale_div: x's frames are bands of images (ops.banded), scale_div of them per image (1, 2, 4, 8 or 16): one workgroup serves the
    bands of an image; scale / shift hold one row per IMAGE.
    sample_cells: 0 = the library's 64 cells x 16 pixels per frame; bands pass band_sample_cells(b)."""
    if x.dim() == 3:
        x = x.unsqueeze(1)
    n, h, w, k = x.shape
    cout = defect_t.shape[1]
    assert defect_t.shape[0] == k and defect_t.is_contiguous() and defect_t.dtype == torch.float32
    assert cout % groups == 0
    out = torch.empty((n, cout) if groups == 1 else (groups, n, cout // groups), dtype=torch.float32, device=x.device)
    L = hip.lib()
    nbytes = L.pgt_frame_bias_workspace_bytes(n, k, cout)
    ws = torch.empty(nbytes // 4, dtype=torch.float32, device=x.device)
    sc, sh, act = affine_in if affine_in is not None else (None, None, ACT_NONE)
    with _Prof("mean_field", 2.0 * n * k * cout, float(n * min(h * w, 16 * (sample_cells or 64)) * k * x.element_size() + defect_t.numel() * 4)):
        hip.check(L.pgt_frame_bias(_dt(x), _p(x), _ld_img(x), n, h * w, k, _p(sc), _p(sh), int(act), _p(defect_t), _p(bias), cout,
                                   int(groups), int(scale_div), int(sample_cells), _p(out), _p(ws), nbytes, _p(_fb_counters(x.device, n)), _stream()), "pgt_frame_bias")
    return out


def sampled_rownorm_mean(x, eps=1e-5):
    """fp32 (N, C): per frame and channel, the mean over the library's pixel sample of the NORMALISED rows of x (N, HW, C) - each
    row minus its mean, times its rstd, rounded to x.dtype: the operand pgt_ln_linear multiplies (pgt_sampled_rownorm_mean)."""
    n, hw, c = x.shape
    mean = torch.empty((n, c), dtype=torch.float32, device=x.device)
    ld = x.stride(1) if hw > 1 else c
    assert x.stride(2) == 1 and (n == 1 or x.stride(0) == hw * ld)
    ws = torch.empty(max(4, hip.lib().pgt_sampled_rownorm_workspace_bytes(n, hw, c) // 4), dtype=torch.float32, device=x.device)
    with _Prof("mean_field", 0, float(n * min(hw, 1024) * c * x.element_size())):
        hip.check(hip.lib().pgt_sampled_rownorm_mean(_dt(x), _p(x), ld, n, hw, c, float(eps), _p(mean), _p(ws), _stream()), "pgt_sampled_rownorm_mean")
    return mean


def fold_layernorm(w, gamma, beta, bias=None):
    """LayerNorm's affine part folded into the Linear that follows: returns (W diag(gamma), bias + W beta) fp32 on the device
    (pgt_fold_layernorm); w (Cout, Cin) fp32."""
    assert w.dtype == torch.float32 and w.dim() == 2 and w.is_contiguous()
    cout, cin = w.shape
    w_out = torch.empty_like(w)
    b_out = torch.empty((cout,), dtype=torch.float32, device=w.device)
    hip.check(hip.lib().pgt_fold_layernorm(_p(w), _p(gamma), _p(beta), _p(bias), cout, cin, _p(w_out), _p(b_out), _stream()), "pgt_fold_layernorm")
    return w_out, b_out


def ln_linear(x, w, bias, eps=1e-5, out=None, x3=False):
    """y = half((x - mean) * rstd) @ w.T + bias: LayerNorm (affine part folded into w / bias by fold_layernorm) and the Linear that
    follows it in one launch.  x (rows, 256) half; w (Cout, 256) packed half; bias (Cout,) or per frame (frames, Cout).
    x3: split rows (rows, 512) -> (rows, 2 Cout), w in the split packed form (Cout, 768), the normalised row split as well."""
    rows = x.shape[0]
    cout = w.shape[0]
    cin = x.shape[1] // 2 if x3 else x.shape[1]
    assert x.dtype == torch.float16 and w.dtype == torch.float16 and w.is_contiguous() and w.shape[1] == (3 * cin if x3 else cin)
    if out is None:
        out = torch.empty((rows, 2 * cout if x3 else cout), device=x.device, dtype=x.dtype)
    brows = 0
    if bias is not None and bias.dim() == 2:
        assert not x3 and bias.shape[1] == cout and bias.is_contiguous() and rows % bias.shape[0] == 0
        brows = rows // bias.shape[0]
    prof = PROFILE
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    if x3:
        hip.check(hip.lib().pgt_ln_linear_x3(_p(x), _ld_rows(x), cin, rows, cin, float(eps), _p(w), _p(bias), cout, _p(out),
                                             _ld_rows(out), cout, _stream()), "pgt_ln_linear_x3")
    else:
        hip.check(hip.lib().pgt_ln_linear(_dt(x), _p(x), _ld_rows(x), rows, cin, float(eps), _p(w), _p(bias), brows, cout, _p(out),
                                          _ld_rows(out), _stream()), "pgt_ln_linear")
    if prof is not None:
        e1.record()
        prof.append({"kernel": "igemm", "flops": 2.0 * rows * cin * cout, "bytes": float(_nb(x, out, w)),
                     "shape": (1, 1, rows, cin, cout, 1, 1, 0), "events": (e0, e1), "cfg": ("ln_linear", 0, 0), "x3": bool(x3),
                     "dt": "float16", "chain": "ln_linear"})
    return out


def ln_mlp(x, w2, b_fc1, b_fc2, eps=1e-5, out=None, x3=True):
    """y = x + fc2(GELU(fc1(LN(x)))) on split rows (rows, 512) in one launch (pgt_ln_mlp_x3): w2 = [Wfc1 diag(gamma); Wfc2] in the
    split packed form (512, 768), b_fc1 carrying W1 beta (fold_layernorm)."""
    assert x3, "the half blocks take attn_proj_mlp"
    rows, c2 = x.shape
    c = c2 // 2
    assert x.dtype == torch.float16 and w2.dtype == torch.float16 and tuple(w2.shape) == (2 * c, 3 * c) and w2.is_contiguous()
    if out is None:
        out = torch.empty((rows, c2), device=x.device, dtype=x.dtype)
    assert out.data_ptr() != x.data_ptr()
    prof = PROFILE
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    hip.check(hip.lib().pgt_ln_mlp_x3(_p(x), _ld_rows(x), c, rows, c, float(eps), _p(w2), _p(b_fc1), _p(b_fc2), _p(out),
                                      _ld_rows(out), c, _stream()), "pgt_ln_mlp_x3")
    if prof is not None:
        e1.record()
        prof.append({"kernel": "igemm", "flops": 4.0 * rows * c * c, "bytes": float(_nb(x, x, out, w2)),
                     "shape": (1, 1, rows, c, 2 * c, 1, 1, 0), "events": (e0, e1), "cfg": ("ln_mlp", 0, 0), "x3": True,
                     "dt": "float16", "chain": "ln_mlp"})
    return out


def attn_proj_mlp(ao, shortcut, w3, b_proj, b_fc1, b_fc2, eps=1e-5, out=None):
    """The tail of a window-attention block in one launch: x1 = ao @ Wproj.T + b_proj + shortcut; y = x1 + fc2(GELU(fc1(LN(x1)))).
    w3 = [Wproj; Wfc1 diag(gamma2); Wfc2] (768, 256) packed half; the three biases are (256,) vectors, or all three per frame
    (frames, 256) - the compensated form (mean_field_bias)."""
    rows, c = ao.shape
    assert ao.dtype == torch.float16 and shortcut.dtype == torch.float16 and tuple(shortcut.shape) == (rows, c)
    assert w3.dtype == torch.float16 and tuple(w3.shape) == (3 * c, c) and w3.is_contiguous()
    if out is None:
        out = torch.empty((rows, c), device=ao.device, dtype=ao.dtype)
    assert out.data_ptr() != ao.data_ptr() and out.data_ptr() != shortcut.data_ptr()
    brows = 0
    if b_proj.dim() == 2:
        assert all(b.dim() == 2 and b.shape == b_proj.shape and b.is_contiguous() for b in (b_proj, b_fc1, b_fc2)) and rows % b_proj.shape[0] == 0
        brows = rows // b_proj.shape[0]
    else:
        assert b_fc1.dim() == 1 and b_fc2.dim() == 1
    prof = PROFILE
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    hip.check(hip.lib().pgt_attn_proj_mlp(_dt(ao), _p(ao), _ld_rows(ao), _p(shortcut), _ld_rows(shortcut), rows, c, _p(w3),
                                          _p(b_proj), _p(b_fc1), _p(b_fc2), brows, float(eps), _p(out), _ld_rows(out), _stream()),
              "pgt_attn_proj_mlp")
    if prof is not None:
        e1.record()
        prof.append({"kernel": "igemm", "flops": 6.0 * rows * c * c, "bytes": float(_nb(ao, shortcut, out, w3)),
                     "shape": (1, 1, rows, c, 3 * c, 1, 1, 0), "events": (e0, e1), "cfg": ("proj_mlp", 0, 0), "x3": False,
                     "dt": "float16", "chain": "proj_mlp"})
    return out


def attn_proj_mlp_sample(ao, shortcut, w3, b_proj, b_fc1, frames, eps=1e-5):
    """The sampled pass of attn_proj_mlp (pgt_attn_proj_mlp_sample): per frame, the channel means of fc1's operand (the normalised
    x1) and of fc2's operand (the GELU'd hidden row) over the library's pixel sample -> (mean_ln, mean_hid), fp32 (frames, 256).
    ao / shortcut: (frames * HW, 256) rows; b_proj (256,) or (frames, 256)."""
    rows, c = ao.shape
    hw = rows // frames
    assert rows == frames * hw and ao.dtype == torch.float16 and shortcut.dtype == torch.float16
    L = hip.lib()
    ws = torch.empty(L.pgt_attn_proj_mlp_sample_workspace_bytes(frames, hw) // 4, dtype=torch.float32, device=ao.device)
    means = torch.empty((2, frames, c), dtype=torch.float32, device=ao.device)
    with _Prof("mean_field", 4.0 * frames * min(hw, 1024) * c * c, float(2 * frames * min(hw, 1024) * c * 2 + w3.numel() * 2)):
        hip.check(L.pgt_attn_proj_mlp_sample(_dt(ao), _p(ao), _ld_rows(ao), _p(shortcut), _ld_rows(shortcut), frames, hw, c, _p(w3),
                                             _p(b_proj), int(b_proj.dim() == 2), _p(b_fc1), float(eps), _p(ws), _p(means[0]), _p(means[1]),
                                             _stream()), "pgt_attn_proj_mlp_sample")
    return means[0], means[1]


def weight_defect(w, packed, scale=None, sum_taps=True):
    """(K, Cout) fp32 rounding defect of a packed 16-bit weight - the operand of mean_field_bias (pgt_weight_defect): w fp32
    (Cout, Cin[, KH, KW]) on the device, packed = pack_conv_weight(w, half / bf16, ...) of it (Cin possibly zero-padded)."""
    assert w.dtype == torch.float32 and w.dim() in (2, 4) and packed.dim() == 2 and packed.is_contiguous()
    w = w.contiguous()
    cout, cin = w.shape[0], w.shape[1]
    kh, kw = (w.shape[2], w.shape[3]) if w.dim() == 4 else (1, 1)
    cp = packed.shape[1] // (kh * kw)
    assert packed.shape == (cout, kh * kw * cp) and cp >= cin, (packed.shape, w.shape)
    out = torch.empty((cp if sum_taps else kh * kw * cp, cout), dtype=torch.float32, device=w.device)
    hip.check(hip.lib().pgt_weight_defect(_dt(packed), _p(w), cout, cin, kh, kw, cp, _p(scale), _p(packed), int(bool(sum_taps)),
                                          _p(out), _stream()), "pgt_weight_defect")
    return out


def sampled_pixels(hw, cells=0):
    """pixel indices of the library's sample of an hw-pixel frame (host-side mirror for tests / emulation); cells: see frame_bias"""
    L = hip.lib()
    out, i = [], 0
    while True:
        p = L.pgt_sampled_pixel_cells(hw, cells, i) if cells else L.pgt_sampled_pixel(hw, i)
        if p < 0:
            return out
        out.append(p)
        i += 1


def adain_affine(mean_c, var_c, mean_s, var_s, eps=1e-5):
    scale = torch.empty_like(mean_c)
    shift = torch.empty_like(mean_c)
    hip.check(hip.lib().pgt_adain_affine(_p(mean_c), _p(var_c), _p(mean_s), _p(var_s), eps, _p(scale), _p(shift),
                                         mean_c.numel(), _stream()), "pgt_adain_affine")
    return scale, shift


def window_attention(qkv, bias, B, T, H, W, C_, heads, win, shift, x3=False):
    """qkv (B*T*H*W, 3C) -> (B*T*H*W, C); x3: split-half rows (.., 6C) = [hi q k v | lo q k v] -> (.., 2C)."""
    rows = B * T * H * W
    if x3:
        assert tuple(qkv.shape) == (rows, 6 * C_) and qkv.dtype == X3_PLANE
        out = torch.empty((rows, 2 * C_), device=qkv.device, dtype=qkv.dtype)
        with _Prof("window_attention", 4.0 * rows * (T * win[0] * win[1]) * C_, _nb(qkv, out), x3=True):
            hip.check(hip.lib().pgt_window_attention_x3(_p(qkv), _ld_rows(qkv), 3 * C_, _p(out), 2 * C_, C_, _p(bias), B, T, H,
                                                        W, C_, heads, win[0], win[1], shift[0], shift[1], _stream()),
                      "pgt_window_attention_x3")
        return out
    assert tuple(qkv.shape) == (rows, 3 * C_)
    out = torch.empty((rows, C_), device=qkv.device, dtype=qkv.dtype)
    with _Prof("window_attention", 4.0 * rows * (T * win[0] * win[1]) * C_, _nb(qkv, out)):
        hip.check(hip.lib().pgt_window_attention(_dt(qkv), _p(qkv), _ld_rows(qkv), _p(out), C_, _p(bias), B, T, H, W,
                                                 C_, heads, win[0], win[1], shift[0], shift[1], _stream()),
                  "pgt_window_attention")
    return out


def window_attention3d(qkv, bias, B, D, H, W, C_, heads, win, shift, pad_row=None):
    """Video-Swin window attention (modules/swin.py): qkv (B*D*H*W, 3C) -> (B*D*H*W, C); win = (wd, wh, ww), shift =
    (sd, sh, sw).  bf16 / fp16 with wd*wh*ww a multiple of 48 (<= 192) on whole windows: the MFMA kernel; fp32 storage,
    other window sizes (<= 256 tokens) and feature maps that are not multiples of the window (padded as the reference
    does; pad_row = the qkv row of a padding token, i.e. the qkv bias, or None = zeros): the general kernel."""
    rows = B * D * H * W
    assert tuple(qkv.shape) == (rows, 3 * C_) and qkv.dtype in (torch.bfloat16, torch.float16, torch.float32)
    out = torch.empty((rows, C_), device=qkv.device, dtype=qkv.dtype)
    dt = _dt(qkv)
    if pad_row is not None:
        pad_row = pad_row.to(device=qkv.device, dtype=qkv.dtype).contiguous()
        assert pad_row.numel() == 3 * C_
    hip.check(hip.lib().pgt_window_attention3d(dt, _p(qkv), _ld_rows(qkv), _p(out), C_, _p(bias), _p(pad_row), B, D, H, W, C_,
                                               heads, win[0], win[1], win[2], shift[0], shift[1], shift[2], _stream()),
              "pgt_window_attention3d")
    return out


def mha(q, k, v, B, L, heads, hd, scale, x3=None):
    """q,k,v (B*L, heads*hd) views -> (B*L, heads*hd).  x3=(q_lo, k_lo, v_lo): q, k, v are the hi planes of split-half
    rows whose lo planes start that many elements further; returns split rows (B*L, 2*heads*hd)."""
    if x3 is not None:
        e = heads * hd
        out = torch.empty((B * L, 2 * e), device=q.device, dtype=q.dtype)
        with _Prof("mha", 4.0 * B * L * L * e, 2 * 3 * B * L * e * q.element_size() + _nb(out), x3=True):
            hip.check(hip.lib().pgt_mha_x3(_p(q), _ld_rows(q), x3[0], _p(k), _ld_rows(k), x3[1], _p(v), _ld_rows(v), x3[2],
                                           _p(out), 2 * e, e, B, L, heads, hd, scale, _stream()), "pgt_mha_x3")
        return out
    out = torch.empty((B * L, heads * hd), device=q.device, dtype=q.dtype)
    with _Prof("mha", 4.0 * B * L * L * heads * hd, 3 * B * L * heads * hd * q.element_size() + _nb(out)):
        hip.check(hip.lib().pgt_mha(_dt(q), _p(q), _ld_rows(q), _p(k), _ld_rows(k), _p(v), _ld_rows(v), _p(out),
                                    heads * hd, B, L, heads, hd, scale, _stream()), "pgt_mha")
    return out


def argmax_rows(logits):
    rows, k = logits.shape
    assert logits.dtype == torch.float32
    codes = torch.empty((rows,), dtype=torch.int32, device=logits.device)
    hip.check(hip.lib().pgt_argmax_rows(_p(logits), _ld_rows(logits), rows, k, _p(codes), _stream()),
              "pgt_argmax_rows")
    return codes


def rq_argmin(dot, xnorm, enorm):
    rows, k = dot.shape
    codes = torch.empty((rows,), dtype=torch.int32, device=dot.device)
    hip.check(hip.lib().pgt_rq_argmin(_p(dot), _ld_rows(dot), _p(xnorm), _p(enorm), rows, k, _p(codes), _stream()),
              "pgt_rq_argmin")
    return codes


def rq_nearest(x, book, xnorm, enorm):
    """Fused nearest-code look-up (distance GEMM + arg-min in one kernel, bf16): x (rows, D), book (K, D) -> int32 codes."""
    rows, d = x.shape
    assert x.dtype == torch.bfloat16 and book.dtype == torch.bfloat16 and book.is_contiguous() and book.shape[1] == d
    codes = torch.empty((rows,), dtype=torch.int32, device=x.device)
    hip.check(hip.lib().pgt_rq_nearest(PGT_BF16, _p(x), _ld_rows(x), _p(book), _p(xnorm), _p(enorm), rows, book.shape[0], d,
                                       _p(codes), _stream()), "pgt_rq_nearest")
    return codes


def rq_soft_codes(dot, xnorm, enorm, temp=1.0):
    rows, k = dot.shape
    soft = torch.empty((rows, k), dtype=torch.float32, device=dot.device)
    codes = torch.empty((rows,), dtype=torch.int32, device=dot.device)
    hip.check(hip.lib().pgt_rq_soft_codes(_p(dot), _ld_rows(dot), _p(xnorm), _p(enorm), rows, k, float(temp), _p(soft),
                                          _p(codes), _stream()), "pgt_rq_soft_codes")
    return soft, codes


def sample_rows(prob, u):
    """one categorical draw per row of prob (rows, K) fp32 by inverse CDF with the caller's uniforms u (rows,) in [0, 1)"""
    rows, k = prob.shape
    assert prob.dtype == torch.float32 and u.dtype == torch.float32 and u.numel() == rows and u.is_contiguous()
    codes = torch.empty((rows,), dtype=torch.int32, device=prob.device)
    hip.check(hip.lib().pgt_sample_rows(_p(prob), _ld_rows(prob), rows, k, _p(u), _p(codes), _stream()), "pgt_sample_rows")
    return codes


def commit_loss(x, q, out=None, scale=1.0):
    """out[0] (+)= scale * mean((x - q)^2) over two (rows, C) matrices; out: fp32 device scalar (1,) (None: new, overwritten)."""
    rows, c = x.shape
    assert q.shape == x.shape and q.dtype == x.dtype
    L = hip.lib()
    nbytes = L.pgt_commit_loss_workspace_bytes()
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    acc = out is not None
    if out is None:
        out = torch.empty((1,), dtype=torch.float32, device=x.device)
    hip.check(L.pgt_commit_loss(_dt(x), _p(x), _ld_rows(x), _p(q), _ld_rows(q), rows, c, _p(out), float(scale), int(acc),
                                _p(ws), nbytes, _stream()), "pgt_commit_loss")
    return out


def straight_through(x, q):
    """x + (q - x) element-wise over (rows, C) matrices (the reference's straight-through value, same fp32 order)."""
    rows, c = x.shape
    out = torch.empty((rows, c), device=x.device, dtype=x.dtype)
    hip.check(hip.lib().pgt_straight_through(_dt(x), _p(x), _ld_rows(x), _p(q), _ld_rows(q), _p(out), c, rows, c, _stream()),
              "pgt_straight_through")
    return out


def embed_rows(codebook, codes, dtype, out=None, accumulate=False, resid=None):
    rows, d = codes.numel(), codebook.shape[1]
    if out is None:
        out = torch.empty((rows, d), device=codes.device, dtype=dtype)
    hip.check(hip.lib().pgt_embed_rows(_dt(out), _p(codebook), d, _p(codes), rows, _p(out), _ld_rows(out),
                                       int(accumulate), _p(resid), _ld_rows(resid) if resid is not None else 0,
                                       _stream()), "pgt_embed_rows")
    return out


def vq_cluster_stats(x, codes, k):
    """One batch's per-code statistics for the EMA codebook update: flat fp32 [K*D sums | K counts] (reference:
    VQEmbedding._update_buffers, tdcrqvae3_arch.py:139-158).  x (rows, D) fp32, codes int32 (rows,)."""
    rows, d = x.shape
    assert x.dtype == torch.float32 and codes.dtype == torch.int32 and codes.numel() == rows
    stats = torch.empty((k * d + k,), dtype=torch.float32, device=x.device)
    hip.check(hip.lib().pgt_vq_cluster_stats(_p(x), _ld_rows(x), _p(codes), rows, k, d, _p(stats), _stream()),
              "pgt_vq_cluster_stats")
    return stats


def vq_ema_update(cluster_size_ema, embed_ema, stats, restart, weight, decay, eps):
    """In place: EMA buffers, restart of dead codes (restart (K, D) or None), codebook rows weight[:K] (reference: :160-186)."""
    k, d = embed_ema.shape
    assert weight.shape[0] >= k and weight.shape[1] == d and weight.is_contiguous() and embed_ema.is_contiguous()
    hip.check(hip.lib().pgt_vq_ema_update(_p(cluster_size_ema), _p(embed_ema), _p(stats), _p(restart), _p(weight), d, k, d,
                                          float(decay), float(1.0 - decay), float(eps), _stream()), "pgt_vq_ema_update")


def row_sumsq(x):
    rows, c = x.shape
    out = torch.empty((rows,), dtype=torch.float32, device=x.device)
    hip.check(hip.lib().pgt_row_sumsq(_dt(x), _p(x), _ld_rows(x), rows, c, _p(out), _stream()), "pgt_row_sumsq")
    return out


def maxpool3x3s2(x):
    n, h, w, c = x.shape
    assert x.is_contiguous()
    out = torch.empty((n, (h - 1) // 2 + 1, (w - 1) // 2 + 1, c), device=x.device, dtype=x.dtype)
    hip.check(hip.lib().pgt_maxpool3x3s2(_dt(x), _p(x), n, h, w, c, _p(out), _stream()), "pgt_maxpool3x3s2")
    return out


def gate_add(x, gate=None, addvec=None, addt=None, out=None):
    """y = x*gate[n,c] + addvec[n,c] + addt; x (N,H,W,C), gate/addvec (N,C) of x.dtype."""
    n, h, w, c = x.shape
    if out is None:
        out = torch.empty((n, h, w, c), device=x.device, dtype=x.dtype)
    for t in (gate, addvec):
        assert t is None or (t.dtype == x.dtype and t.is_contiguous() and t.numel() == n * c)
    hip.check(hip.lib().pgt_gate_add(_dt(x), _p(x), _ld_img(x), n, h * w, c, _p(gate), _p(addvec), _p(addt),
                                     _ld_img(addt) if addt is not None else 0, _p(out), _ld_img(out), _stream()),
              "pgt_gate_add")
    return out


def resize_bilinear_ac(x, ho, wo, out=None):
    n, h, w, c = x.shape
    if out is None:
        out = torch.empty((n, ho, wo, c), device=x.device, dtype=x.dtype)
    hip.check(hip.lib().pgt_resize_bilinear_ac(_dt(x), _p(x), _ld_img(x), n, h, w, c, _p(out), _ld_img(out), ho, wo,
                                               _stream()), "pgt_resize_bilinear_ac")
    return out


def copy_into(src, dst):
    """dst[...] = src with dtype conversion; both (..., C) views with the same leading shape, dense rows."""
    c = src.shape[-1]
    rows = src.numel() // c
    lds = _ld_img(src) if src.dim() == 4 else _ld_rows(src)
    ldd = _ld_img(dst) if dst.dim() == 4 else _ld_rows(dst)
    assert dst.shape == src.shape
    with _Prof("copy_gather", 0, _nb(src, dst)):
        hip.check(hip.lib().pgt_copy2d(_dt(src), _p(src), lds, _dt(dst), _p(dst), ldd, rows, c, _stream()), "pgt_copy2d")
    return dst


def cast(x, dtype):
    if x.dtype == dtype:
        return x
    out = torch.empty(x.shape, device=x.device, dtype=dtype)
    return copy_into(x, out)


def gather_frames(src, idx, out=None):
    """out[i] = src[idx[i]] over the leading (frame) axis.  src: (F, ..., C) with a contiguous last dim and densely packed
    leading dims over its row stride (a channel slice of a wider buffer is fine); idx: int32 device tensor (n,).
    Returns (n, ..., C)."""
    assert idx.dtype == torch.int32 and idx.is_contiguous()
    n = idx.numel()
    if out is None:
        out = torch.empty((n,) + tuple(src.shape[1:]), device=src.device, dtype=src.dtype)
    assert tuple(out.shape[1:]) == tuple(src.shape[1:]) and out.shape[0] == n and out.dtype == src.dtype
    es = src.element_size()
    if src.is_contiguous() and out.is_contiguous():
        rows, row_bytes = 1, src[0].numel() * es      # whole frames (e.g. uint8 (F,H,W,3) clips)
        while row_bytes >= (1 << 30):                 # keep the 32-bit row size: split a frame into equal rows
            rows, row_bytes = rows * 2, row_bytes // 2
        lds = ldd = row_bytes
    else:
        assert src.dim() == 4
        rows, row_bytes = src.shape[1] * src.shape[2], src.shape[3] * es
        lds, ldd = _ld_img(src) * es, _ld_img(out) * es
    with _Prof("copy_gather", 0, 2 * _nb(out)):
        hip.check(hip.lib().pgt_gather_frames(_p(src), lds, _p(out), ldd, _p(idx), n, rows, row_bytes, _stream()),
                  "pgt_gather_frames")
    return out


def zero_(t):
    """t[...] = 0 for a (..., C) view with a contiguous last dim whose C * element_size is a multiple of 16 bytes (pad channel
    slices of wider buffers, or whole buffers)."""
    c = t.shape[-1]
    rows = t.numel() // c
    ld = (_ld_img(t) if t.dim() == 4 else _ld_rows(t)) if rows > 1 else c
    es = t.element_size()
    with _Prof("copy_gather", 0, _nb(t)):
        hip.check(hip.lib().pgt_zero2d(_p(t), ld * es, rows, c * es, _stream()), "pgt_zero2d")
    return t


def input_channels(dtype):
    """Channels the 3-channel input frames are padded to: one 16-byte chunk (the K granularity of pgt_conv2d's gather) - 4 in
    fp32 (K = 36 / 196 for the 3x3 / 7x7 input layers instead of 72 / 392), 8 in bf16 / half."""
    return 16 // torch.empty((), dtype=dtype).element_size()


def prep_input(src, dtype, want_raw=True, want_norm=True):
    """src: uint8 (N,H,W,3) or float32 (N,3,H,W) -> raw, norm as (N,H,W,input_channels(dtype)) channel-padded tensors
    (one 16-byte chunk per pixel: 4 channels in fp32, 8 in a 16-bit dtype)."""
    if src.dtype == torch.uint8:
        n, h, w, _ = src.shape
        kind = 0
    else:
        assert src.dtype == torch.float32
        n, _, h, w = src.shape
        kind = 1
    assert src.is_contiguous()
    cp = input_channels(dtype)
    raw = torch.empty((n, h, w, cp), device=src.device, dtype=dtype) if want_raw else None
    norm = torch.empty((n, h, w, cp), device=src.device, dtype=dtype) if want_norm else None
    ref = raw if raw is not None else norm
    hip.check(hip.lib().pgt_prep_input(_dt(ref), _p(src), kind, n, h, w, _p(raw), _p(norm), _stream()),
              "pgt_prep_input")
    return raw, norm


def nhwc_to_nchw_f32(x):
    n, h, w, c = x.shape
    out = torch.empty((n, c, h, w), device=x.device, dtype=torch.float32)
    hip.check(hip.lib().pgt_nhwc_to_nchw_f32(_dt(x), _p(x), _ld_img(x), n, h, w, c, _p(out), _stream()),
              "pgt_nhwc_to_nchw_f32")
    return out


def frame_to_u8(x, out=None):
    """x (H,W,3) activation view -> uint8 (H,W,3): floor(clamp(x,0,1)*255)."""
    h, w, c = x.shape
    assert c == 3
    if out is None:
        out = torch.empty((h, w, 3), device=x.device, dtype=torch.uint8)
    assert out.is_contiguous() and tuple(out.shape) == (h, w, 3)
    hip.check(hip.lib().pgt_frame_to_u8(_dt(x), _p(x), x.stride(1), h, w, _p(out), _stream()), "pgt_frame_to_u8")
    return out


# ---- range telemetry (include/pgt_hip.h: pgt_count_saturated) ------------------------------------------------------------------
# fp32 -> half stores saturate at +-65504 (PGT_F16 / the planes of PGT_F16X3).  With RANGE_CHECK set to a list every operator
# below counts the elements of its IEEE-half outputs that sit at the limit (or are not finite) and appends
# (operator, output shape, int32 device counter): PGTFormer.check_range / WindowRunner run one such pass on the first frames
# they see and refuse to continue when a layer saturates (a checkpoint whose activations leave the half range needs
# precision="bf16x3").
RANGE_CHECK = None
RANGE_CTX = []          # names of the modules whose forward() is executing (PGTFormer.check_range registers the hooks)


def _note_range(name, out):
    if RANGE_CHECK is None:
        return
    for t in (out if isinstance(out, (tuple, list)) else (out,)):
        if not torch.is_tensor(t) or t.dtype != torch.float16 or not t.is_cuda or t.numel() == 0 or t.shape[-1] % 8:
            continue
        if t.dim() == 4:
            rows, ld = t.shape[0] * t.shape[1] * t.shape[2], _ld_img(t)
        elif t.dim() == 2:
            rows, ld = t.shape[0], _ld_rows(t)
        else:
            continue
        if ld % 8 or t.data_ptr() % 16:
            continue
        cnt = torch.zeros((1,), dtype=torch.int32, device=t.device)
        hip.check(hip.lib().pgt_count_saturated(_p(t), ld, rows, t.shape[-1], _p(cnt), _stream()), "pgt_count_saturated")
        RANGE_CHECK.append(((RANGE_CTX[-1] + ":" if RANGE_CTX else "") + name, tuple(t.shape), cnt))


def _range_wrap(fn):
    import functools

    @functools.wraps(fn)
    def wrapped(*a, **k):
        out = fn(*a, **k)
        # (placed outputs - out_rows / out_parity - cover a view that several launches fill: its other rows are not written yet)
        if RANGE_CHECK is not None and k.get("out_rows") is None and k.get("out_parity") is None:
            _note_range(fn.__name__, out)
        return out
    return wrapped


for _n in ("conv2d", "affine_act", "layernorm", "ln_linear", "ln_mlp", "attn_proj_mlp", "window_attention", "window_attention3d", "mha",
           "to_x3", "x3_to_half"):
    globals()[_n] = _range_wrap(globals()[_n])


def range_report(records):
    """[(operator, shape, saturated count)] of the records a RANGE_CHECK pass collected (synchronises)."""
    return [(n, s, int(c.item())) for n, s, c in records]
