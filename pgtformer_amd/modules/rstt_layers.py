"""Temporal-coherent transformer layers and residual conv blocks, HIP-backed.

Host-side mirror of the symbols the reference path uses from modules/rstt_layers.py — same class
names, constructor arguments and state-dict keys — whose forward() launches the gfx950 kernels via
`pgtformer_amd.ops` instead of ATen.  Activations are channels-last: a (B,D,C,H,W) reference tensor is
held here as (B*D, H, W, C); the (D,Wh,Ww) window partition / roll / reverse of the reference
(rstt_layers.py:55-88, 307-327) never materialise: they are address arithmetic inside the attention
kernel.

torch.nn modules are used as PARAMETER CONTAINERS only (so checkpoints load unchanged); their
forward() is never called.
"""
import os

import torch
import torch.nn as nn

from .. import ops
from ..ops import ACT_GELU, ACT_SILU, X3, X3F

USE_X3_FOLD = os.environ.get("PGT_X3_FOLD", "1") != "0"   # A/B switch of the folded 64-channel split-half convs
USE_X3_C64 = os.environ.get("PGT_X3_C64", "1") != "0"     # A/B switch of the register-weight split-half 3x3 kernel (igemm6x3.hip)


USE_WCOMP = os.environ.get("PGT_WCOMP", "1") != "0"       # A/B switch of the mean-field weight-rounding compensation
USE_WCOMP_LINEAR = os.environ.get("PGT_WCOMP_LINEAR", "1") != "0"   # ... of the token-row linears (window-attention blocks)
USE_WCOMP_MLP = os.environ.get("PGT_WCOMP_MLP", "1") != "0"         # ... of fc1 / fc2 only (layer-by-layer path; the fused chains cannot)
USE_ROWCHAIN = os.environ.get("PGT_ROWCHAIN", "1") != "0"           # A/B switch of the fused token-row chains (rowchain.hip)
USE_ROWCHAIN_X3 = os.environ.get("PGT_ROWCHAIN_X3", "1") != "0"     # ... of their split-half forms (encoder-side blocks)


class HipModule(nn.Module):
    """Base: `prepare(device, dtype)` repacks this module's weights for the kernels (recursively)."""

    def prepare(self, device, dtype):
        prepare_tree(self, device, dtype)
        return self

    def _pack(self, device, dtype):
        pass


def prepare_tree(m, device, dtype):
    """Walk a module tree (through plain nn containers too) and repack every HipModule."""
    if isinstance(m, HipModule):
        m.dev, m.dt = device, dtype
        m._pack(device, dtype)
    for child in m.children():
        prepare_tree(child, device, dtype)


def _is_x3(dtype):
    return isinstance(dtype, str) and dtype == X3


def _is_x3f(dtype):
    return isinstance(dtype, str) and dtype == X3F


def _pack_matrix(w, device, dtype, cin_pad=None, scale=None, fold=False, w2=False):
    """reference weight - (Cout, Cin, KH, KW), or any K-major (Cout, K) matrix - -> kernel operand on `device`: the repack is
    the library's (pgt_pack_conv_weight: K-major rows, channel padding, rounding to the compute type, the split-half forms;
    w2: the exact-weight operand of a half layer, two planes per filter row)"""
    w = w.detach().to(device=device, dtype=torch.float32)
    return ops.pack_conv_weight(w, dtype, cin_pad=cin_pad, scale=scale, fold=fold, w2=w2)


def mark_exact_weights(m, flag=True):
    """Every layer of the module tree `m` is to run with EXACT weights where its type allows (IEEE-half layers: two weight planes,
    pgt_conv_desc::w2) - call before prepare(); DESIGN.md section 2.3"""
    for sub in m.modules():
        sub.exact_w = bool(flag)


def _exact(m, dtype, k_channels):
    """does layer `m` keep an exact-weight operand next to its single-plane one? (half layers whose K comes in 64-channel blocks)"""
    return bool(getattr(m, "exact_w", False)) and dtype == torch.float16 and k_channels % 64 == 0


def _wants_wcomp(dtype, module=None):
    """single-plane 16-bit operands (half / bf16 layers): their weight rounding is compensated to first order - unless the module
    belongs to a decoder stage the compensation is switched off for (mark_uncompensated; PGT_WCOMP_STAGES)"""
    return USE_WCOMP and dtype in (torch.float16, torch.bfloat16) and not getattr(module, "no_wcomp", False)


def mark_uncompensated(m, flag=True):
    """no mean-field compensation for the layers of the module tree `m` (call before prepare())"""
    for sub in m.modules():
        sub.no_wcomp = bool(flag)


def _defect_t(w, pw, scale=None, sum_taps=True):
    """Rounding defect of a packed 16-bit weight, as the (K, Cout) fp32 operand of ops.mean_field_bias:
    D[k][o] = sum over the filter taps of (w * scale - packed)[o][k][tap] (sum_taps=False: one row per (tap, k), for a conv
    whose taps read different frames).  w: the fp32 reference weight (Cout, Cin[, KH, KW]); pw = _pack_matrix(w, ...) of it.
    DESIGN.md section 2.2: y = W x + b run with W16 leaves (W - W16) x, whose frame-constant part (W - W16) mean(x) is
    put back as a per-frame bias."""
    return ops.weight_defect(w.detach().to(device=pw.device, dtype=torch.float32), pw, scale=scale, sum_taps=sum_taps)   # pgt_weight_defect


def _frame_bias(x, pdef, pb, frames=None, affine_in=None):
    """bias operand of a layer: the (frames, Cout) per-frame bias of a compensated layer - x (N,H,W,C) image batch, or
    (rows, C) tokens of `frames` frames - or the plain (Cout,) bias `pb` when the layer is not compensated or its frames are
    not whole 512-row tiles (pgt_conv_desc::bias_rows)"""
    if pdef is None or pdef.shape[0] > 3840:      # (pgt_mean_field_bias holds K <= 3840 means in LDS: wider layers keep the plain bias)
        return pb
    if x.dim() == 2:
        if not USE_WCOMP_LINEAR:
            return pb
        if not frames or x.shape[0] % frames or (x.shape[0] // frames) % 512:
            return pb
        x = x.as_strided((frames, x.shape[0] // frames, x.shape[1]), (x.stride(0) * (x.shape[0] // frames), x.stride(0), 1))
    elif (x.shape[1] * x.shape[2]) % 512:
        return pb
    div = 1
    if x.dim() == 4 and ops.USE_FRAME_BIAS and x.dtype in (torch.float16, torch.bfloat16):
        x, div = ops.banded(x)        # convolutions: ops.WCOMP_BANDS bias vectors per frame (the token-row chains keep one per frame)
    if ops.USE_FRAME_BIAS and x.dtype in (torch.float16, torch.bfloat16):
        return ops.frame_bias(x, pdef, pb, affine_in=affine_in, scale_div=div, sample_cells=ops.band_sample_cells(div))        # one launch
    if affine_in is not None:
        x = ops.affine_act(x, *affine_in)
    return ops.mean_field_bias(ops.sampled_channel_mean(x), pdef, pb)


def _f32(t, device):
    return None if t is None else t.detach().to(device=device, dtype=torch.float32).contiguous()


class Conv2d(nn.Conv2d, HipModule):
    """nn.Conv2d parameter container + implicit-GEMM launch. Weight (Cout,Cin,KH,KW) is repacked to
    (Cout, KH*KW*Cin) K-major; `bn` folds an eval-mode BatchNorm2d; `cin_pad` zero-pads input channels
    (57->64; "chunk": 3 -> one 16-byte chunk of the layer's dtype, 4 in fp32 / 8 in 16 bits) to the 16-byte gather granule."""

    def __init__(self, cin, cout, k, stride=1, padding=0, bias=True, pad4=None, cin_pad=None):
        nn.Conv2d.__init__(self, cin, cout, k, stride=stride, padding=padding, bias=bias)
        self.pad4 = pad4 if pad4 is not None else (padding, padding, padding, padding)
        self.cin_pad = cin_pad
        self._bn_ref = ()  # (BatchNorm2d,) to fold; a tuple so it is not registered as a sub-module

    def _pack(self, device, dtype):
        b = _f32(self.bias, device)
        scale = None
        if self._bn_ref:      # eval-mode BatchNorm2d folded into this conv (pgt_fold_batchnorm)
            bn = self._bn_ref[0]
            scale, b = ops.fold_batchnorm(_f32(bn.weight, device), _f32(bn.bias, device), _f32(bn.running_mean, device),
                                          _f32(bn.running_var, device), bn.eps, b)
        cout, cin, kh, kw = self.weight.shape
        cpad = self.cin_pad
        if cpad == "chunk":      # the 3-channel input layers: one 16-byte chunk per pixel (ops.prep_input writes exactly that)
            cpad = ops.input_channels(dtype) if isinstance(dtype, torch.dtype) else 8
        cin_k = self._fold_cin = cpad if (cpad is not None and cpad > cin) else cin
        self.pw = _pack_matrix(self.weight, device, dtype, cin_pad=cin_k, scale=scale)
        self.pb = b
        self.pdef = _defect_t(self.weight, self.pw, scale=scale) if _wants_wcomp(dtype, self) else None
        # exact-weight layers keep the two-plane operand as well; run() takes it wherever the library has the form for the launch
        self.pw2 = _pack_matrix(self.weight, device, dtype, cin_pad=cin_k, scale=scale, w2=True) if _exact(self, dtype, cin_k) else None
        # split-half layers with 64 output channels: the folded form fills the 128-column tile (pgt_conv_desc.x3_fold)
        self.pw_fold = None
        if (_is_x3(dtype) or _is_x3f(dtype)) and cout == 64 and cin_k % 64 == 0 and USE_X3_FOLD:
            self.pw_fold = _pack_matrix(self.weight, device, dtype, cin_pad=cin_k, scale=scale, fold=True)

    def run(self, x, **kw):
        fold = self.pw_fold is not None
        if (fold and USE_X3_C64 and (_is_x3(self.dt) or _is_x3f(self.dt)) and self._fold_cin == 64 and self.kernel_size == (3, 3) and self.stride == (1, 1)
                and self.pad4 == (1, 1, 1, 1) and x.shape[2] >= 32 and not (x.shape[1] & (x.shape[1] - 1)) and not (x.shape[2] & (x.shape[2] - 1))):
            fold = False      # the register-weight kernel of these layers takes the standard split matrix (igemm6x3.hip)
            kw = {k: v for k, v in kw.items() if k != "gn"}     # (no epilogue statistics on these layers: ops.gn_ok)
        if fold and kw.get("gn") is not None:
            if self._fold_cin == 64 and self.kernel_size[0] == 3:
                kw = {k: v for k, v in kw.items() if k != "gn"}     # these layers leave no epilogue statistics (ops.gn_ok)
            else:
                fold = False
        w = self.pw_fold if fold else self.pw
        if fold:
            kw = dict(kw, x3_fold=True)
        if _is_x3f(self.dt):     # fp32 in / fp32 out, split-half MFMA arithmetic in between
            assert kw.get("affine_in") is None
            return ops.conv2d(ops.to_x3(x), w, self.pb, kh=self.kernel_size[0], kw=self.kernel_size[1],
                              stride=self.stride[0], pad=self.pad4, x3=True, out_f32=True, **kw)
        b = self.pb
        a_in = kw.get("affine_in")
        if a_in is not None and not ops.affine_in_fuses(x, self.out_channels, self.kernel_size[0], self.kernel_size[1], self.stride[0], self.pad4,
                                                        x3=_is_x3(self.dt), bias=b, **{k: v for k, v in kw.items() if k not in ("affine_in", "gn")}):
            x = ops.affine_act(x, a_in[0], a_in[1], a_in[2], x3=_is_x3(self.dt))      # no fused-operand form for this launch: one apply pass
            kw = {k: v for k, v in kw.items() if k != "affine_in"}
            a_in = None
        if getattr(self, "pw2", None) is not None and ops.w2_ok(x, self.out_channels, self._fold_cin, self.kernel_size[0], self.kernel_size[1],
                                                                self.stride[0], self.pad4, bias=b, **{k: v for k, v in kw.items() if k != "affine_in"}):
            # exact weights (two planes, two MFMAs per product): nothing to compensate
            return ops.conv2d(x, self.pw2, b, kh=self.kernel_size[0], kw=self.kernel_size[1], stride=self.stride[0], pad=self.pad4,
                              w2=self.out_channels, **kw)
        if self.pdef is not None and self.stride == (1, 1) and not kw.get("ups"):     # (same-size layers: frames of H*W output pixels)
            b = _frame_bias(x, self.pdef, self.pb, affine_in=a_in)
        return ops.conv2d(x, w, b, kh=self.kernel_size[0], kw=self.kernel_size[1],
                          stride=self.stride[0], pad=self.pad4, x3=_is_x3(self.dt), **kw)


class Linear(nn.Linear, HipModule):
    def _pack(self, device, dtype):
        self.pw = _pack_matrix(self.weight, device, dtype)
        self.pb = _f32(self.bias, device)
        self.pdef = _defect_t(self.weight, self.pw) if _wants_wcomp(dtype, self) else None
        self.pw2 = _pack_matrix(self.weight, device, dtype, w2=True) if _exact(self, dtype, self.in_features) and self.out_features % 8 == 0 else None

    def run(self, x, frames=None, **kw):
        """frames: the rows of x are the tokens of that many frames (equal counts): enables the per-frame compensation"""
        if getattr(self, "pw2", None) is not None and _rows_w2_ok(x, kw.get("res"), kw.get("out")):
            return ops.linear(x, self.pw2, self.pb, w2=self.out_features, **kw)      # exact weights: nothing to compensate
        return ops.linear(x, self.pw, _frame_bias(x, self.pdef, self.pb, frames), x3=_is_x3(self.dt), **kw)


def _rows_w2_ok(x, *others):
    """token-row operands the exact-weight form takes: IEEE half rows, 16-byte aligned with 16-byte row pitch"""
    if x.dtype != torch.float16:
        return False
    return all(t is None or (t.data_ptr() % 16 == 0 and ops._ld_rows(t) % 8 == 0) for t in (x,) + others)


class GroupNorm(nn.GroupNorm, HipModule):
    def _pack(self, device, dtype):
        self.pg, self.pbeta = _f32(self.weight, device), _f32(self.bias, device)

    def run(self, x, act=ACT_SILU, out=None):
        return ops.groupnorm_act(x, self.pg, self.pbeta, act, self.num_groups, self.eps, out=out, x3=_is_x3(self.dt))

    def coeffs(self, x, act=ACT_SILU):
        """the (scale, shift, act) of this GroupNorm + activation over x for a consumer that applies them itself
        (Conv2d.run(..., affine_in=...)): statistics only, the normalised tensor is not written"""
        scale, shift = ops.groupnorm_affine(x, self.pg, self.pbeta, self.num_groups, self.eps, x3=_is_x3(self.dt))
        return scale, shift, act


class LayerNorm(nn.LayerNorm, HipModule):
    def _pack(self, device, dtype):
        self.pg, self.pbeta = _f32(self.weight, device), _f32(self.bias, device)

    def run(self, x, pos=None):
        return ops.layernorm(x, self.pg, self.pbeta, self.eps, pos, x3=_is_x3(self.dt))


def Normalize(in_channels):
    """GroupNorm(32, C, eps=1e-6) (reference: rstt_layers.py:754-755)."""
    return GroupNorm(num_groups=32, num_channels=in_channels, eps=1e-6, affine=True)


def get_window_size(x_size, window_size, shift_size=None):
    """Clamp window (and zero the shift) where the feature map is not larger than the window
    (reference: rstt_layers.py:90-114)."""
    use_w = list(window_size)
    use_s = list(shift_size) if shift_size is not None else None
    for i in range(len(x_size)):
        if x_size[i] <= window_size[i]:
            use_w[i] = x_size[i]
            if use_s is not None:
                use_s[i] = 0
    return tuple(use_w) if use_s is None else (tuple(use_w), tuple(use_s))


class TDResnetBlock(HipModule):
    """GN-SiLU-conv3x3-GN-SiLU-conv3x3 + (identity | 1x1 nin_shortcut) (reference: rstt_layers.py:835-904).
    x: (N,H,W,Cin) -> (N,H,W,Cout); the residual add is the second conv's epilogue."""

    def __init__(self, *, in_channels, out_channels=None, conv_shortcut=False, dropout=0.0, temb_channels=0):
        super().__init__()
        out_channels = in_channels if out_channels is None else out_channels
        self.in_channels, self.out_channels = in_channels, out_channels
        assert not conv_shortcut and temb_channels == 0, "not used on the PGTFormer path"
        self.checkpointing = False
        self.norm1 = Normalize(in_channels)
        self.conv1 = Conv2d(in_channels, out_channels, 3, padding=1)
        self.norm2 = Normalize(out_channels)
        self.conv2 = Conv2d(out_channels, out_channels, 3, padding=1)
        if in_channels != out_channels:
            self.nin_shortcut = Conv2d(in_channels, out_channels, 1)

    def forward(self, x, temb=None, out=None, gn_next=False):
        """out: optional (N,H,W,Cout) view (e.g. a channel slice of a concat buffer) that receives the result.
        gn_next: a GroupNorm follows this block - its statistics come out of conv2's epilogue (as norm2's come out of
        conv1's: SURVEY K1/K7 "statistics from the producer")."""
        # GroupNorm apply + SiLU ride in the consuming conv's operand load where the library has that form (the 64-channel
        # 512x512 layers: Conv2d.run falls back to one apply pass elsewhere)
        h = self.conv1.run(x, affine_in=self.norm1.coeffs(x, ACT_SILU), gn=self.norm2.num_groups)
        sc = self.nin_shortcut.run(x) if self.in_channels != self.out_channels else x
        return self.conv2.run(h, affine_in=self.norm2.coeffs(h, ACT_SILU), res=sc, out=out, gn=32 if gn_next else None)


class Mlp(HipModule):
    def __init__(self, in_features, hidden_features=None, out_features=None):
        super().__init__()
        self.fc1 = Linear(in_features, hidden_features or in_features)
        self.fc2 = Linear(hidden_features or in_features, out_features or in_features)


class WindowAttention3D(HipModule):
    """Parameters of the (D,Wh,Ww)-window attention (reference: rstt_layers.py:134-193)."""

    def __init__(self, dim, num_frames_q, num_frames_kv, window_size, num_heads, qkv_bias=True):
        super().__init__()
        assert num_frames_q == num_frames_kv, "self-attention only on the PGTFormer path"
        self.dim, self.num_frames, self.window_size, self.num_heads = dim, num_frames_q, tuple(window_size), num_heads
        d, wh, ww = num_frames_q, window_size[0], window_size[1]
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * d - 1) * (2 * wh - 1) * (2 * ww - 1), num_heads))
        from ..weightgen import relative_position_index
        self.register_buffer("relative_position_index", torch.from_numpy(relative_position_index(d, (wh, ww))))
        self.q = Linear(dim, dim, bias=qkv_bias)
        self.kv = Linear(dim, dim * 2, bias=qkv_bias)
        self.proj = Linear(dim, dim)
        nn.init.trunc_normal_(self.relative_position_bias_table, std=.02)

    def _pack(self, device, dtype):
        # one fused (3C,C) projection [q | k | v]; dense per-head bias gathered once from the table
        wcat = torch.cat([self.q.weight.detach(), self.kv.weight.detach()], 0)
        self.w_qkv = _pack_matrix(wcat, device, dtype)
        self.d_qkv = _defect_t(wcat, self.w_qkv) if _wants_wcomp(dtype, self) else None
        self.w_qkv2 = _pack_matrix(wcat, device, dtype, w2=True) if _exact(self, dtype, self.dim) else None
        if self.q.bias is not None:
            self.b_qkv = _f32(torch.cat([self.q.bias.detach(), self.kv.bias.detach()], 0), device)
        else:
            self.b_qkv = None
        n = self.relative_position_index.shape[0]
        tbl = self.relative_position_bias_table.detach().float().cpu()
        idx = self.relative_position_index.cpu().reshape(-1)
        self.bias_dense = tbl[idx].reshape(n, n, -1).permute(2, 0, 1).contiguous().to(device)


class VSTSREncoderTransformerBlock(HipModule):
    """LN -> window attention -> +shortcut -> LN -> MLP(GELU) -> +residual (reference: rstt_layers.py:236-338)."""

    def __init__(self, dim, num_heads, num_frames=4, window_size=(8, 8), shift_size=(0, 0), mlp_ratio=4.0, qkv_bias=True):
        super().__init__()
        self.dim, self.num_heads, self.num_frames = dim, num_heads, num_frames
        self.window_size, self.shift_size = tuple(window_size), tuple(shift_size)
        self.norm1 = LayerNorm(dim)
        self.attn = WindowAttention3D(dim, num_frames, num_frames, self.window_size, num_heads, qkv_bias)
        self.norm2 = LayerNorm(dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))

    def _pack(self, device, dtype):
        """Operands of the fused token-row chains (ops.ln_linear, ops.attn_proj_mlp: half layers of 256 channels): norm1 folded
        into the [q | k | v] projection, norm2 into fc1 (pgt_fold_layernorm), proj / fc1 / fc2 stacked into one matrix."""
        self.fused = False
        if _exact(self, dtype, self.dim):
            return      # exact-weight blocks run layer by layer on the two-plane linears (the chains hold single-plane weights)
        # the chains are built for the shipping blocks: 256 channels, Mlp hidden width == dim (mlp_ratio 1, the reference's
        # EncoderLayer call sites: archs/tdcrqvae3_arch.py:499), biased q / kv; anything else runs layer by layer
        if not (USE_ROWCHAIN and self.dim == 256 and self.attn.q.bias is not None and self.attn.kv.bias is not None and
                self.mlp.fc1.out_features == self.dim == self.mlp.fc2.in_features):
            return
        f32 = lambda t: t.detach().to(device=device, dtype=torch.float32).contiguous()   # noqa: E731
        a, m = self.attn, self.mlp
        if dtype == torch.float16 or _is_x3(dtype):
            wq, self.f_bqkv = ops.fold_layernorm(f32(torch.cat([a.q.weight.detach(), a.kv.weight.detach()], 0)), f32(self.norm1.weight),
                                                 f32(self.norm1.bias), f32(torch.cat([a.q.bias.detach(), a.kv.bias.detach()], 0)))
            self.f_wqkv = _pack_matrix(wq, device, dtype)
            w1, self.f_b1 = ops.fold_layernorm(f32(m.fc1.weight), f32(self.norm2.weight), f32(self.norm2.bias), f32(m.fc1.bias))
            self.f_b2 = f32(m.fc2.bias)
        if dtype == torch.float16:
            self.f_dqkv = _defect_t(wq, self.f_wqkv) if _wants_wcomp(dtype, self) else None
            self.f_w3 = torch.cat([_pack_matrix(f32(a.proj.weight), device, dtype), _pack_matrix(w1, device, dtype),
                                   _pack_matrix(f32(m.fc2.weight), device, dtype)], 0).contiguous()
            self.f_bp = f32(a.proj.bias)
            self.f_dproj = self.f_dfc1 = self.f_dfc2 = None
            if _wants_wcomp(dtype, self):
                d = self.dim
                self.f_dproj = _defect_t(a.proj.weight, self.f_w3[:d])
                self.f_dfc1 = _defect_t(w1, self.f_w3[d:2 * d])
                self.f_dfc2 = _defect_t(m.fc2.weight, self.f_w3[2 * d:])
            self.fused = True
        elif _is_x3(dtype) and USE_ROWCHAIN_X3:
            # split blocks (encoder side): LN1 -> q|k|v and LN2 -> Mlp -> residual fused; proj + shortcut stays a linear launch
            self.f_w2 = torch.cat([_pack_matrix(w1, device, dtype), _pack_matrix(f32(m.fc2.weight), device, dtype)], 0).contiguous()
            self.fused = True

    def forward(self, xt, B, H, W, out=None, gn_images=None):
        """xt: (B*D*H*W, C) tokens in (b,d,y,x) order; out: optional (rows, C) view receiving the result.
        gn_images: number of images (B*D) when a GroupNorm follows: fc2's epilogue leaves its statistics (layer-by-layer path
        only: the fused chains have no statistics epilogue, the GroupNorm that follows then takes its own statistics pass -
        +0.2 ms per such block at 128 x 128, counted under "groupnorm statistics pass" in bench.py)."""
        C = self.dim
        win, shift = get_window_size((H, W), self.window_size, self.shift_size)
        x3 = _is_x3(self.dt)
        nf = B * self.num_frames                      # frames of H*W tokens (rows in (b, d, y, x) order)
        if getattr(self, "fused", False) and x3 and xt.shape[0] % 128 == 0:
            qkv = ops.ln_linear(xt, self.f_wqkv, self.f_bqkv, self.norm1.eps, x3=True)
            ao = ops.window_attention(qkv, self.attn.bias_dense, B, self.num_frames, H, W, C, self.num_heads, win, shift, x3=True)
            x1 = self.attn.proj.run(ao, frames=nf, res=xt)
            return ops.ln_mlp(x1, self.f_w2, self.f_b1, self.f_b2, self.norm2.eps, out=out, x3=True)
        if getattr(self, "fused", False) and not x3 and xt.dtype == torch.float16 and xt.shape[0] % 128 == 0:
            # two launches either side of the attention (rowchain.hip); the compensated per-frame bias of the q|k|v projection is
            # taken from the sampled mean of the NORMALISED rows, the proj one from the attention output; fc1 / fc2 run on rows
            # that never reach HBM and keep their plain bias
            bq = self.f_bqkv
            hw = xt.shape[0] // nf
            if self.f_dqkv is not None and USE_WCOMP_LINEAR and hw % 512 == 0:
                x3d = xt.as_strided((nf, hw, C), (xt.stride(0) * hw, xt.stride(0), 1))
                bq = ops.mean_field_bias(ops.sampled_rownorm_mean(x3d, self.norm1.eps), self.f_dqkv, self.f_bqkv)
            qkv = ops.ln_linear(xt, self.f_wqkv, bq, self.norm1.eps)
            ao = ops.window_attention(qkv, self.attn.bias_dense, B, self.num_frames, H, W, C, self.num_heads, win, shift)
            bp, b1, b2 = _frame_bias(ao, self.f_dproj, self.f_bp, nf), self.f_b1, self.f_b2
            if bp.dim() == 2:
                # fc1 / fc2 read rows that never reach HBM: the per-frame means of their operands come from a sampled pass of
                # the same chain (ops.attn_proj_mlp_sample, <= 1024 rows per frame)
                m_ln, m_hid = ops.attn_proj_mlp_sample(ao, xt, self.f_w3, bp, self.f_b1, nf, self.norm2.eps)
                b1, b2 = ops.mean_field_bias(m_ln, self.f_dfc1, self.f_b1), ops.mean_field_bias(m_hid, self.f_dfc2, self.f_b2)
            return ops.attn_proj_mlp(ao, xt, self.f_w3, bp, b1, b2, self.norm2.eps, out=out)
        ln = self.norm1.run(xt)
        if getattr(self.attn, "w_qkv2", None) is not None and _rows_w2_ok(ln):
            qkv = ops.linear(ln, self.attn.w_qkv2, self.attn.b_qkv, w2=3 * C)
        else:
            qkv = ops.linear(ln, self.attn.w_qkv, _frame_bias(ln, self.attn.d_qkv, self.attn.b_qkv, nf), x3=x3)
        ao = ops.window_attention(qkv, self.attn.bias_dense, B, self.num_frames, H, W, C, self.num_heads, win, shift, x3=x3)
        x1 = self.attn.proj.run(ao, frames=nf, res=xt)
        nfm = nf if USE_WCOMP_MLP else None
        m = self.mlp.fc1.run(self.norm2.run(x1), frames=nfm, act=ACT_GELU)
        return self.mlp.fc2.run(m, frames=nfm, res=x1, out=out, gn=None if gn_images is None else (32, gn_images))


class EncoderLayer(HipModule):
    """depth blocks alternating un-shifted / shifted windows (reference: rstt_layers.py:499-575)."""

    def __init__(self, dim, depth, num_heads, num_frames, window_size=(8, 8), mlp_ratio=4.0, qkv_bias=True):
        super().__init__()
        self.window_size = tuple(window_size)
        self.shift_size = tuple(i // 2 for i in window_size)
        self.depth, self.num_frames, self.dim = depth, num_frames, dim
        self.blocks = nn.ModuleList([
            VSTSREncoderTransformerBlock(dim, num_heads, num_frames, self.window_size,
                                         (0, 0) if i % 2 == 0 else self.shift_size, mlp_ratio, qkv_bias)
            for i in range(depth)])

    def forward(self, x, out=None, gn_next=False):
        """x: (B*D, H, W, C) -> same; out: optional (B*D, H, W, C) view (channel slice of a wider buffer) for the result.
        gn_next: a GroupNorm follows (its statistics come out of the last MLP epilogue)."""
        n, h, w, c = x.shape
        assert h % self.window_size[0] == 0 or h <= self.window_size[0]
        xt = x.reshape(n * h * w, c)
        for i, blk in enumerate(self.blocks):
            fin = i == len(self.blocks) - 1
            last = out is not None and fin
            xt = blk(xt, n // self.num_frames, h, w,
                     out=out.as_strided((n * h * w, c), (out.stride(2), 1), out.storage_offset()) if last else None,
                     gn_images=n if (gn_next and fin) else None)
        y = out if out is not None else xt.reshape(n, h, w, c)
        st = getattr(xt, "_pgt_gn", None)
        return y if st is None else st.bind(y, c)
