"""TEST INFRASTRUCTURE: torch-CPU emulation of the operator contracts of `pgtformer_amd.ops`.

Used only by tests: (a) on CPU, monkeypatched over `pgtformer_amd.ops` to check the HOST logic (module
graph, weight repack, channels-last index math) of the product against the oracle without a GPU;
(b) on the GPU box, as the per-operator expected value for the HIP kernels at arbitrary shapes.
It is never imported by the package; the product has no CPU path.
"""
import math

import torch
import torch.nn.functional as F

ACT_NONE, ACT_RELU, ACT_GELU, ACT_SILU, ACT_LEAKY02, ACT_SIGMOID = range(6)


def _act(v, act):
    if act == ACT_RELU:
        return F.relu(v)
    if act == ACT_GELU:
        return F.gelu(v)
    if act == ACT_SILU:
        return F.silu(v)
    if act == ACT_LEAKY02:
        return F.leaky_relu(v, 0.2)
    if act == ACT_SIGMOID:
        return torch.sigmoid(v)
    return v


def _merge(x):
    """split (..., 2C) -> fp32 (..., C)"""
    c = x.shape[-1] // 2
    return x[..., :c].float() + x[..., c:].float()


X3_PLANE = torch.float16      # plane type of split tensors (pgtformer_amd.ops.X3_PLANE)


def _plane(v):
    """fp32 -> one plane: IEEE half, saturating at +-65504 like the kernels' conversions"""
    return v.clamp(-65504.0, 65504.0).to(X3_PLANE)


def _split(v):
    """fp32 (..., C) -> split (..., 2C) on two half planes: hi = half(v), lo = half(v - hi)"""
    v = v.clamp(-65504.0, 65504.0)          # one clamp per value (common.h split8): hi cannot overflow, |v - hi| <= ulp / 2
    hi = v.to(X3_PLANE)
    lo = (v - hi.float()).to(X3_PLANE)
    return torch.cat([hi, lo], -1).contiguous()


def _store_x3(val, out):
    s = _split(val)
    if out is None:
        return s
    out.copy_(s)
    return out


def _unpack_x3_weight(w, taps):
    """(Cout, taps*3*Cin): per tap and 64-channel block [w_hi | w_hi | w_lo] -> fp32 (Cout, taps*Cin) = w_hi + w_lo"""
    cout = w.shape[0]
    w5 = w.float().reshape(cout, taps, -1, 3, 64)
    assert torch.equal(w5[:, :, :, 0], w5[:, :, :, 1])
    return (w5[:, :, :, 0] + w5[:, :, :, 2]).reshape(cout, -1)


W2_SCALE = 2048.0       # csrc/igemm_common.h kW2Inv: the lo plane of an exact-weight operand is stored scaled by 2^11


def _unpack_w2_weight(w, cout):
    """exact-weight operand (2 * ceil32(cout) rows: per 32 channels [32 rows hi | 32 rows lo * 2048]) -> fp32 (cout, K) = hi + lo / 2048"""
    g = w.float().reshape(-1, 2, 32, w.shape[1])
    return (g[:, 0] + g[:, 1] / W2_SCALE).reshape(-1, w.shape[1])[:cout].contiguous()


def pack_conv_weight(w, dtype, cin_pad=None, scale=None, fold=False, w2=False):
    """torch restatement of pgt_pack_conv_weight (K-major rows, zero-padded input channels, optional per-channel factor)"""
    w4 = w.float() if w.dim() == 4 else w.float()[:, :, None, None]
    if scale is not None:
        w4 = w4 * scale.float().view(-1, 1, 1, 1)
    cout, cin, kh, kw = w4.shape
    if cin_pad is not None and cin_pad > cin:
        w4 = F.pad(w4, (0, 0, 0, 0, 0, cin_pad - cin))
    k3 = w4.permute(0, 2, 3, 1).reshape(cout, kh * kw, -1)              # (Cout, taps, Cin_pad)
    if isinstance(dtype, str):                                          # split-half
        hi = _plane(k3)
        lo = _plane(k3 - hi.float())
        hi4, lo4 = hi.reshape(cout, kh * kw, -1, 1, 64), lo.reshape(cout, kh * kw, -1, 1, 64)
        if fold:
            top = torch.cat([hi4, hi4], 3).reshape(cout, -1)
            bot = torch.cat([lo4, torch.zeros_like(lo4)], 3).reshape(cout, -1)
            return torch.cat([top, bot], 0).contiguous()
        return torch.cat([hi4, hi4, lo4], 3).reshape(cout, -1).contiguous()
    if w2:                                                              # exact-weight form: two planes per filter row
        k2 = k3.reshape(cout, -1)
        c32 = (cout + 31) // 32 * 32
        k2 = F.pad(k2, (0, 0, 0, c32 - cout))
        hi = k2.to(dtype)
        lo = ((k2 - hi.float()) * W2_SCALE).to(dtype)
        return torch.stack([hi.reshape(-1, 32, k2.shape[1]), lo.reshape(-1, 32, k2.shape[1])], 1).reshape(2 * c32, -1).contiguous()
    return k3.reshape(cout, -1).to(dtype).contiguous()


def fold_batchnorm(gamma, beta, mean, var, eps, bias=None):
    s = gamma.float() / torch.sqrt(var.float() + eps)
    b0 = bias.float() if bias is not None else torch.zeros_like(s)
    return s, (b0 - mean.float()) * s + beta.float()


def to_x3(x, out=None):
    return _store_x3(x.float(), out)


def from_x3(x):
    return _merge(x).contiguous()


def x3_to_half(x, out=None):
    return _store(_merge(x), out, torch.float16)


def gather_frames(src, idx, out=None):
    r = src[idx.long()]
    if out is None:
        return r.contiguous()
    out.copy_(r)
    return out


def _store(val, out, dtype):
    if out is None:
        return val.to(dtype).contiguous()
    out.copy_(val.to(out.dtype))
    return out


def conv2d(x, w, bias=None, *, kh=1, kw=1, stride=1, pad=(0, 0, 0, 0), ups=False, act=ACT_NONE, res=None,
           post_relu=False, sft=None, out=None, out_f32=False, tile=(0, 0), scalar_epi=False, kernel=0, splitk=0, stages=0, out_x3=False,
           out_parity=None, out_rows=None, x3=False, gn=None, x3_fold=False, affine_in=None, w2=None):
    if w2 is not None:         # exact-weight operand: the layer multiplies by hi + lo / 2048 (fp32 below)
        w = _unpack_w2_weight(w, int(w2))
    if affine_in is not None:      # the fused operand (pgt_conv2d_affine_in) = the apply pass's result, rounded to the tensor's type
        x = affine_act(x, affine_in[0], affine_in[1], affine_in[2], x3=x3)
    n, h, wd, cin = x.shape
    cout = w.shape[0]
    if x3:
        assert x.dtype == X3_PLANE and w.dtype == X3_PLANE and sft is None and not ups
        cin //= 2
        x = _merge(x)
        if x3_fold:   # (128, taps*2*Cin): rows 0..63 [w_hi | w_hi], rows 64..127 [w_lo | 0] per tap and 64-channel block
            assert cout == 128
            w5 = w.float().reshape(128, kh * kw, -1, 2, 64)
            assert torch.equal(w5[:64, :, :, 0], w5[:64, :, :, 1]) and not w5[64:, :, :, 1].any()
            w = (w5[:64, :, :, 0] + w5[64:, :, :, 0]).reshape(64, -1)
            cout = 64
        else:
            w = _unpack_x3_weight(w, kh * kw)
        if res is not None:
            res = res if res.dtype == torch.float32 else _merge(res)      # fp32 residual: fp32-stored tensors (BiSeNet)
    assert w.shape[1] == kh * kw * cin and (w.dtype == x.dtype or w2 is not None)
    xi = x.float().permute(0, 3, 1, 2)
    if ups:
        xi = F.interpolate(xi, scale_factor=2.0, mode="nearest")
    xi = F.pad(xi, (pad[2], pad[3], pad[0], pad[1]))
    wt = w.float().reshape(cout, kh, kw, cin).permute(0, 3, 1, 2)
    y = F.conv2d(xi, wt, bias.float() if (bias is not None and bias.dim() == 1) else None, stride=stride).permute(0, 2, 3, 1)
    if bias is not None and bias.dim() == 2:       # one bias vector per frame: bias_rows = N*Ho*Wo / frames consecutive output pixels
        fr = bias.shape[0]
        y = (y.reshape(fr, -1, cout) + bias.float()[:, None, :]).reshape(y.shape)
    y = _act(y, act)
    if sft is not None:
        dec, shift, sw = sft
        y = dec.float() + sw * (dec.float() * y + shift.float())
    else:
        if res is not None:
            y = y + res.float()
        if post_relu:
            y = F.relu(y)
    if out_parity is not None:
        out[:, out_parity[0]::2, out_parity[1]::2, :] = y.to(out.dtype)
        return out
    if out_rows is not None:
        mul, xmul, off = out_rows
        wo = y.shape[2]
        m = torch.arange(y.shape[0] * y.shape[1] * wo)
        rows = mul * m + xmul * (m % wo) + off
        flat = out.as_strided((int(rows.max()) + 1, cout), (out.stride(-2), 1))
        flat[rows] = y.reshape(-1, cout).to(out.dtype)
        return out
    if out_x3 or (x3 and not out_f32):
        return _store_x3(y, out)
    return _store(y, out, torch.float32 if out_f32 else x.dtype)


def linear(x, w, bias=None, *, act=ACT_NONE, res=None, out=None, out_f32=False, x3=False, gn=None, w2=None):
    if w2 is not None:
        w = _unpack_w2_weight(w, int(w2))
    if x3:
        x, w = _merge(x), _unpack_x3_weight(w, 1)
        res = None if res is None else _merge(res)
    y = F.linear(x.float(), w.float(), bias.float() if (bias is not None and bias.dim() == 1) else None)
    if bias is not None and bias.dim() == 2:
        y = (y.reshape(bias.shape[0], -1, y.shape[-1]) + bias.float()[:, None, :]).reshape(y.shape)
    y = _act(y, act)
    if res is not None:
        y = y + res.float()
    if x3 and not out_f32:
        return _store_x3(y, out)
    return _store(y, out, torch.float32 if out_f32 else x.dtype)


def groupnorm_affine(x, gamma, beta, groups=32, eps=1e-6, x3=False):
    if x3:
        x = _merge(x)
    n, h, w, c = x.shape
    xf = x.float().reshape(n, h * w, groups, c // groups)
    mean = xf.mean(dim=(1, 3))
    var = xf.var(dim=(1, 3), unbiased=False)
    rstd = (var + eps).rsqrt()
    rs = rstd.repeat_interleave(c // groups, 1)
    mu = mean.repeat_interleave(c // groups, 1)
    scale = rs * gamma
    shift = beta - mu * scale
    return scale.contiguous(), shift.contiguous()


def affine_act(x, scale, shift, act=ACT_NONE, out=None, x3=False):
    n = x.shape[0]
    xf = _merge(x) if x3 else x.float()
    y = _act(xf * scale.reshape(n, 1, 1, -1) + shift.reshape(n, 1, 1, -1), act)
    return _store_x3(y, out) if x3 else _store(y, out, x.dtype)


def groupnorm_act(x, gamma, beta, act=ACT_SILU, groups=32, eps=1e-6, out=None, x3=False):
    s, b = groupnorm_affine(x, gamma, beta, groups, eps, x3=x3)
    return affine_act(x, s, b, act, out=out, x3=x3)


def layernorm(x, gamma, beta, eps=1e-5, pos=None, x3=False):
    if x3:
        xf = _merge(x)
        y = F.layer_norm(xf, (xf.shape[-1],), gamma, beta, eps)
        return _split(y) if pos is None else (_split(y), _split(y + _merge(pos)))
    y = F.layer_norm(x.float(), (x.shape[-1],), gamma, beta, eps)
    if pos is None:
        return y.to(x.dtype)
    return y.to(x.dtype), (y + pos.float()).to(x.dtype)


def channel_stats(x, want_var=True):
    n, h, w, c = x.shape
    xf = x.float().reshape(n, h * w, c)
    return xf.mean(1).contiguous(), (xf.var(1, unbiased=True).contiguous() if want_var else None)


_SAMPLES = {}


def sampled_pixels(hw, cells=0):
    """the library's pixel sample of an hw-pixel frame (pgt_sampled_pixel is host code: callable without a GPU)"""
    if (hw, cells) not in _SAMPLES:
        from pgtformer_amd import ops as real_ops
        _SAMPLES[(hw, cells)] = torch.tensor(real_ops.sampled_pixels(hw, cells), dtype=torch.long)
    return _SAMPLES[(hw, cells)]


def band_sample_cells(b):
    return 0 if b <= 1 else max(16, 64 // b)


def sampled_channel_mean(x, cells=0):
    if x.dim() == 4:
        x = x.reshape(x.shape[0], x.shape[1] * x.shape[2], x.shape[3])
    return x[:, sampled_pixels(x.shape[1], cells), :].float().mean(1)


def mean_field_bias(mean, defect_t, bias=None):
    y = mean.float() @ defect_t.float()
    return y if bias is None else y + bias.float()


WCOMP_BANDS = 16


def banded(x, bands=None):
    n, h, w, c = x.shape
    hw = h * w
    b = WCOMP_BANDS if bands is None else bands
    while b > 1 and (hw % b or (hw // b) % 512):
        b //= 2
    b = max(1, b)
    return x.reshape(n * b, hw // b, c), b


def frame_bias(x, defect_t, bias=None, affine_in=None, groups=1, scale_div=1, sample_cells=0):
    """pgt_frame_bias: sampled mean (of the fused operand when affine_in is given) + mean-field bias in one call; groups > 1:
    (G, N, Csub) - the G layers' bias matrices"""
    if affine_in is not None:
        sc, sh = (t.repeat_interleave(scale_div, 0) for t in affine_in[:2])
        x = affine_act(x if x.dim() == 4 else x.unsqueeze(1), sc, sh, affine_in[2])
    y = mean_field_bias(sampled_channel_mean(x, sample_cells), defect_t, bias)
    return y if groups == 1 else y.reshape(y.shape[0], groups, -1).permute(1, 0, 2).contiguous()


def affine_in_fuses(x, cout, kh, kw, stride, pad, **_kw):
    """the emulation has no kernels to choose from: the fused-operand form is always 'available' (conv2d applies it itself)"""
    return True


def _rownorm(x, eps):
    """(x - mean) * rstd per row, rounded to x.dtype: the operand of the fused LayerNorm -> Linear kernels (rowchain.hip)"""
    xf = x.float()
    mu = xf.mean(-1, keepdim=True)
    d = xf - mu
    return (d * (d.pow(2).mean(-1, keepdim=True) + eps).rsqrt()).to(x.dtype)


def sampled_rownorm_mean(x, eps=1e-5):
    return _rownorm(x[:, sampled_pixels(x.shape[1]), :], eps).float().mean(1)


def fold_layernorm(w, gamma, beta, bias=None):
    w = w.float()
    b = w @ beta.float()
    return (w * gamma.float()[None, :]).contiguous(), (b if bias is None else b + bias.float())


def _rownorm_x3(x, eps):
    """split rows -> normalised rows, split again (the operand of the split-half chain kernels)"""
    xf = _merge(x)
    mu = xf.mean(-1, keepdim=True)
    d = xf - mu
    return _split(d * (d.pow(2).mean(-1, keepdim=True) + eps).rsqrt())


def ln_linear(x, w, bias, eps=1e-5, out=None, x3=False):
    if x3:
        return linear(_rownorm_x3(x, eps), w, bias, out=out, x3=True)
    return linear(_rownorm(x, eps), w, bias, out=out)


def ln_mlp(x, w2, b_fc1, b_fc2, eps=1e-5, out=None, x3=True):
    c = x.shape[1] // 2
    h = linear(_rownorm_x3(x, eps), w2[:c], b_fc1, act=ACT_GELU, x3=True)
    return linear(h, w2[c:], b_fc2, res=x, out=out, x3=True)


def attn_proj_mlp(ao, shortcut, w3, b_proj, b_fc1, b_fc2, eps=1e-5, out=None):
    c = ao.shape[1]
    x1 = linear(ao, w3[:c], b_proj, res=shortcut)
    h = linear(_rownorm(x1, eps), w3[c:2 * c], b_fc1, act=ACT_GELU)
    return linear(h, w3[2 * c:], b_fc2, res=x1, out=out)


def attn_proj_mlp_sample(ao, shortcut, w3, b_proj, b_fc1, frames, eps=1e-5):
    rows, c = ao.shape
    hw = rows // frames
    idx = sampled_pixels(hw)
    a3, s3 = ao.reshape(frames, hw, c)[:, idx, :], shortcut.reshape(frames, hw, c)[:, idx, :]
    ns = len(idx)
    x1 = linear(a3.reshape(-1, c), w3[:c], b_proj, res=s3.reshape(-1, c))
    xh = _rownorm(x1, eps)
    h = linear(xh, w3[c:2 * c], b_fc1, act=ACT_GELU)
    return xh.float().reshape(frames, ns, c).mean(1), h.float().reshape(frames, ns, c).mean(1)


def weight_defect(w, packed, scale=None, sum_taps=True):
    w32 = w.float()
    cout, cin = w32.shape[0], w32.shape[1]
    if scale is not None:
        w32 = w32 * scale.view(-1, *([1] * (w32.dim() - 1)))
    taps = w32.shape[2] * w32.shape[3] if w32.dim() == 4 else 1
    cp = packed.shape[1] // taps
    d = torch.zeros((cout, taps, cp), dtype=torch.float64)
    d[:, :, :cin] = w32.reshape(cout, cin, taps).permute(0, 2, 1).double() - packed.view(cout, taps, cp)[:, :, :cin].double()
    d[:, :, cin:] = -packed.view(cout, taps, cp)[:, :, cin:].double()
    d = d.sum(1) if sum_taps else d.reshape(cout, taps * cp)
    return d.float().t().contiguous()


def adain_affine(mean_c, var_c, mean_s, var_s, eps=1e-5):
    scale = (var_s + eps).sqrt() / (var_c + eps).sqrt()
    return scale, mean_s - mean_c * scale


def window_attention(qkv, bias, B, T, H, W, C_, heads, win, shift, x3=False):
    """Independent formulation: roll / partition with torch ops, dense mask from region labels."""
    if x3:
        return _split(window_attention(_merge(qkv), bias, B, T, H, W, C_, heads, win, shift))
    wh, ww = win
    sh, sw = shift
    hd = C_ // heads
    x = qkv.float().reshape(B, T, H, W, 3 * C_)
    if sh or sw:
        x = torch.roll(x, shifts=(-sh, -sw), dims=(2, 3))
    xw = x.reshape(B, T, H // wh, wh, W // ww, ww, 3 * C_).permute(0, 2, 4, 1, 3, 5, 6)
    nW = (H // wh) * (W // ww)
    N = T * wh * ww
    xw = xw.reshape(B * nW, N, 3, heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = xw[0] * (hd ** -0.5), xw[1], xw[2]
    attn = q @ k.transpose(-2, -1) + bias.unsqueeze(0)
    if sh or sw:
        img = torch.zeros(1, T, H, W, 1)
        cnt = 0
        for hs in (slice(0, -wh), slice(-wh, -sh), slice(-sh, None)):
            for ws_ in (slice(0, -ww), slice(-ww, -sw), slice(-sw, None)):
                img[:, :, hs, ws_, :] = cnt
                cnt += 1
        mw = img.reshape(1, T, H // wh, wh, W // ww, ww, 1).permute(0, 2, 4, 1, 3, 5, 6).reshape(nW, N)
        am = mw.unsqueeze(1) - mw.unsqueeze(2)
        am = torch.where(am != 0, torch.full_like(am, -100.0), torch.zeros_like(am))
        attn = (attn.reshape(B, nW, heads, N, N) + am.unsqueeze(1).unsqueeze(0)).reshape(-1, heads, N, N)
    o = (attn.softmax(-1) @ v).transpose(1, 2).reshape(B, H // wh, W // ww, T, wh, ww, C_)
    o = o.permute(0, 3, 1, 4, 2, 5, 6).reshape(B, T, H, W, C_)
    if sh or sw:
        o = torch.roll(o, shifts=(sh, sw), dims=(2, 3))
    return o.reshape(B * T * H * W, C_).to(qkv.dtype)


def window_attention3d(qkv, bias, B, D, H, W, C_, heads, win, shift, pad_row=None):
    """Video-Swin form: (wd,wh,ww) windows, 3-axis roll, 27-region mask from region labels (independent formulation);
    feature maps that are not multiples of the window are padded at the far end with `pad_row` tokens (zeros if None)."""
    wd, wh, ww = win
    sd, sh, sw = shift
    hd = C_ // heads
    x = qkv.float().reshape(B, D, H, W, 3 * C_)
    D0, H0, W0 = D, H, W
    D, H, W = -(-D // wd) * wd, -(-H // wh) * wh, -(-W // ww) * ww
    if (D, H, W) != (D0, H0, W0):
        fill = torch.zeros(3 * C_) if pad_row is None else pad_row.float().reshape(-1)
        xp = fill.reshape(1, 1, 1, 1, -1).expand(B, D, H, W, 3 * C_).clone()
        xp[:, :D0, :H0, :W0] = x
        x = xp
    shifted = bool(sd or sh or sw)
    if shifted:
        x = torch.roll(x, shifts=(-sd, -sh, -sw), dims=(1, 2, 3))

    def part(t):   # (B,D,H,W,ch) -> (B*nW, N, ch)
        ch = t.shape[-1]
        t = t.reshape(B, D // wd, wd, H // wh, wh, W // ww, ww, ch).permute(0, 1, 3, 5, 2, 4, 6, 7)
        return t.reshape(-1, wd * wh * ww, ch)
    nW = (D // wd) * (H // wh) * (W // ww)
    N = wd * wh * ww
    xw = part(x).reshape(B * nW, N, 3, heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = xw[0] * (hd ** -0.5), xw[1], xw[2]
    attn = q @ k.transpose(-2, -1) + bias.unsqueeze(0)
    if shifted:
        img = torch.zeros(1, D, H, W, 1)
        cnt = 0
        for ds in (slice(0, D - wd), slice(D - wd, D - sd), slice(D - sd, D)):
            for hs in (slice(0, H - wh), slice(H - wh, H - sh), slice(H - sh, H)):
                for ws_ in (slice(0, W - ww), slice(W - ww, W - sw), slice(W - sw, W)):
                    img[:, ds, hs, ws_, :] = cnt
                    cnt += 1
        mw = part(img.expand(B, D, H, W, 1))[:nW, :, 0]
        am = mw.unsqueeze(1) - mw.unsqueeze(2)
        am = torch.where(am != 0, torch.full_like(am, -100.0), torch.zeros_like(am))
        attn = (attn.reshape(B, nW, heads, N, N) + am.unsqueeze(1).unsqueeze(0)).reshape(-1, heads, N, N)
    o = (attn.softmax(-1) @ v).transpose(1, 2).reshape(B, D // wd, H // wh, W // ww, wd, wh, ww, C_)
    o = o.permute(0, 1, 4, 2, 5, 3, 6, 7).reshape(B, D, H, W, C_)
    if shifted:
        o = torch.roll(o, shifts=(sd, sh, sw), dims=(1, 2, 3))
    o = o[:, :D0, :H0, :W0]
    return o.reshape(B * D0 * H0 * W0, C_).to(qkv.dtype)


def mha(q, k, v, B, L, heads, hd, scale, x3=None):
    if x3 is not None:   # q, k, v are hi-plane views of split rows; the lo planes start x3[i] elements further
        e = heads * hd

        def full(t, lo):
            base = t.as_strided((t.shape[0], lo + e), (t.stride(0), 1), t.storage_offset())
            return base[:, :e].float() + base[:, lo:lo + e].float()
        return _split(mha(full(q, x3[0]), full(k, x3[1]), full(v, x3[2]), B, L, heads, hd, scale))

    def split(t):
        return t.float().reshape(B, L, heads, hd).permute(0, 2, 1, 3)
    attn = ((split(q) * scale) @ split(k).transpose(-2, -1)).softmax(-1)
    return (attn @ split(v)).permute(0, 2, 1, 3).reshape(B * L, heads * hd).to(q.dtype)


def argmax_rows(logits):
    return logits.argmax(-1).to(torch.int32)


def rq_argmin(dot, xnorm, enorm):
    return ((xnorm.unsqueeze(1) + enorm.unsqueeze(0)) - 2.0 * dot).argmin(-1).to(torch.int32)


def rq_nearest(x, book, xnorm, enorm):
    return rq_argmin(linear(x, book, None, out_f32=True), xnorm, enorm)


def rq_soft_codes(dot, xnorm, enorm, temp=1.0):
    dist = (xnorm.unsqueeze(1) + enorm.unsqueeze(0)) - 2.0 * dot
    return F.softmax(-dist / temp, dim=-1), dist.argmin(-1).to(torch.int32)


def sample_rows(prob, u):
    c = prob.float().cumsum(-1)
    t = (u.float() * c[:, -1]).unsqueeze(1)
    return (c > t).float().argmax(-1).to(torch.int32)


def commit_loss(x, q, out=None, scale=1.0):
    v = scale * (x.float() - q.float()).pow(2.0).mean().reshape(1)
    return v if out is None else out + v


def straight_through(x, q):
    return (x.float() + (q.float() - x.float())).to(x.dtype)


def vq_cluster_stats(x, codes, k):
    """flat [K*D sums | K counts]; sums in row order (sequential index_add on CPU)"""
    rows, d = x.shape
    sums = torch.zeros((k, d), dtype=torch.float32).index_add_(0, codes.long(), x.float())
    cnt = torch.bincount(codes.long(), minlength=k).float()
    return torch.cat([sums.reshape(-1), cnt])


def vq_ema_update(cluster_size_ema, embed_ema, stats, restart, weight, decay, eps):
    k, d = embed_ema.shape
    alpha = float(torch.tensor(1.0 - decay, dtype=torch.float32))
    cs = cluster_size_ema * float(torch.tensor(decay, dtype=torch.float32)) + alpha * stats[k * d:]
    em = embed_ema * float(torch.tensor(decay, dtype=torch.float32)) + alpha * stats[:k * d].reshape(k, d)
    if restart is not None:
        dead = ~(cs >= 1)
        em = torch.where(dead.unsqueeze(1), restart, em)
        cs = torch.where(dead, torch.ones_like(cs), cs)
    n = cs.sum()
    norm = n * (cs + eps) / (n + k * eps)
    cluster_size_ema.copy_(cs)
    embed_ema.copy_(em)
    weight[:k].copy_(em / norm.unsqueeze(1))


def embed_rows(codebook, codes, dtype, out=None, accumulate=False, resid=None):
    e = codebook[codes.long()]
    if out is None:
        out = torch.zeros((codes.numel(), codebook.shape[1]), dtype=dtype)
        accumulate = False
    out.copy_(((out.float() if accumulate else 0) + e).to(out.dtype))
    if resid is not None:
        resid.copy_((resid.float() - e).to(resid.dtype))
    return out


def row_sumsq(x):
    return x.float().pow(2).sum(1)


def maxpool3x3s2(x):
    return F.max_pool2d(x.float().permute(0, 3, 1, 2), 3, 2, 1).permute(0, 2, 3, 1).to(x.dtype).contiguous()


def gate_add(x, gate=None, addvec=None, addt=None, out=None):
    n, c = x.shape[0], x.shape[3]
    y = x.float()
    if gate is not None:
        y = y * gate.float().reshape(n, 1, 1, c)
    if addvec is not None:
        y = y + addvec.float().reshape(n, 1, 1, c)
    if addt is not None:
        y = y + addt.float()
    return _store(y, out, x.dtype)


def resize_bilinear_ac(x, ho, wo, out=None):
    y = F.interpolate(x.float().permute(0, 3, 1, 2), (ho, wo), mode="bilinear", align_corners=True)
    return _store(y.permute(0, 2, 3, 1), out, x.dtype)


def copy_into(src, dst):
    dst.copy_(src.to(dst.dtype))
    return dst


def cast(x, dtype):
    return x if x.dtype == dtype else x.to(dtype)


def zero_(t):
    t.zero_()
    return t


def input_channels(dtype):
    return 16 // torch.empty((), dtype=dtype).element_size()


def prep_input(src, dtype, want_raw=True, want_norm=True):
    if src.dtype == torch.uint8:
        r = src.float() / 255.0
    else:
        r = src.permute(0, 2, 3, 1)
    mean = torch.tensor([0.485, 0.456, 0.406])
    std = torch.tensor([0.229, 0.224, 0.225])
    nn_ = (r - mean) / std
    pad = lambda t: F.pad(t, (0, input_channels(dtype) - 3)).to(dtype).contiguous()  # noqa: E731
    return (pad(r) if want_raw else None), (pad(nn_) if want_norm else None)


def nhwc_to_nchw_f32(x):
    return x.float().permute(0, 3, 1, 2).contiguous()


def frame_to_u8(x, out=None):
    r = (x.float().clamp(0, 1) * 255).to(torch.uint8)
    if out is not None:
        out.copy_(r)
        return out
    return r


ALL = ["conv2d", "linear", "groupnorm_affine", "affine_act", "groupnorm_act", "layernorm", "channel_stats",
       "adain_affine", "window_attention", "mha", "argmax_rows", "rq_argmin", "embed_rows", "row_sumsq",
       "maxpool3x3s2", "gate_add", "resize_bilinear_ac", "copy_into", "cast", "prep_input", "nhwc_to_nchw_f32",
       "frame_to_u8", "to_x3", "from_x3", "x3_to_half", "pack_conv_weight", "fold_batchnorm", "sample_rows", "gather_frames", "window_attention3d", "rq_nearest", "rq_soft_codes", "commit_loss",
       "straight_through", "zero_", "vq_cluster_stats", "vq_ema_update", "sampled_channel_mean", "mean_field_bias",
       "frame_bias", "banded", "band_sample_cells", "affine_in_fuses", "sampled_rownorm_mean", "weight_defect", "fold_layernorm", "ln_linear", "ln_mlp", "attn_proj_mlp", "attn_proj_mlp_sample"]


def install(monkeypatch):
    """Replace every operator of pgtformer_amd.ops by its CPU emulation (tests only)."""
    import sys

    import pgtformer_amd.ops as real
    me = sys.modules[__name__]
    for name in ALL:
        monkeypatch.setattr(real, name, getattr(me, name))
    monkeypatch.setattr(real, "USE_EPILOGUE_GN", False)   # the emulated convs leave no epilogue statistics
