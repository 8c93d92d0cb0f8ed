"""PGTFormer top-level model, BiSeNet face-parsing condition net and the SFT fusion block, HIP-backed.

Host-side mirror of the reference's archs/pgtformer_arch.py: same class names, constructor kwargs,
state-dict keys (961 tensors) and `forward(x, w, detach_16, code_only, adain)` return tuple, so a
reference checkpoint loads with `load_state_dict(strict=True)`; every arithmetic op is a gfx950 kernel
launched through `pgtformer_amd.ops` (no ATen compute, no CPU fallback).

Layout: a reference (B*T, C, H, W) tensor lives here as channels-last (B*T, H, W, C).  In that layout
every token re-ordering of the reference forward (pgtformer_arch.py:614, :640, :646) is the identity:
(t, y, x)-major rows of a (B*T*H*W, C) matrix.
"""
import torch
import torch.nn as nn

from .. import ops
from ..modules.rstt_layers import (Conv2d, HipModule, LayerNorm, Linear, TDResnetBlock, _defect_t, _exact, _f32, _frame_bias, _is_x3,  # noqa: F401
                                    _pack_matrix, _wants_wcomp)
from ..ops import ACT_LEAKY02, ACT_RELU, ACT_SIGMOID, ACT_SILU
from ..registry import ARCH_REGISTRY
from .codeformer_arch import TransformerSALayer, adaptive_instance_normalization
from .tdcrqvae3_arch import TDCRQVAE3

import os as _os
SIDE_STREAM = _os.environ.get("PGT_SIDE_STREAM", "1") != "0"


def _range_pop(module, args, output):
    """forward hook of PGTFormer.check_range: leaves the module's record context; returns None explicitly (a hook's non-None
    return value REPLACES the module's output - the root module's name is '', which `pop() and None` would hand back)"""
    ops.RANGE_CTX.pop()
    return None


# ----------------------------------------------------------------------------------------------
# BiSeNet (reference: pgtformer_arch.py:34-397).  Eval-mode BatchNorm is folded into the preceding
# conv at pack time; ReLU / sigmoid / residual adds are conv epilogues.
# ----------------------------------------------------------------------------------------------
def _bn(c):
    return nn.BatchNorm2d(c)


class BasicBlock(HipModule):
    def __init__(self, in_chan, out_chan, stride=1):
        super().__init__()
        self.conv1 = Conv2d(in_chan, out_chan, 3, stride=stride, padding=1, bias=False)
        self.bn1 = _bn(out_chan)
        self.conv2 = Conv2d(out_chan, out_chan, 3, padding=1, bias=False)
        self.bn2 = _bn(out_chan)
        self.downsample = None
        if in_chan != out_chan or stride != 1:
            self.downsample = nn.Sequential(Conv2d(in_chan, out_chan, 1, stride=stride, bias=False), _bn(out_chan))
            self.downsample[0]._bn_ref = (self.downsample[1],)
        self.conv1._bn_ref = (self.bn1,)
        self.conv2._bn_ref = (self.bn2,)

    def forward(self, x):
        r = self.conv1.run(x, act=ACT_RELU)
        sc = x if self.downsample is None else self.downsample[0].run(x)
        return self.conv2.run(r, res=sc, post_relu=True)   # relu(shortcut + bn2(conv2(.)))


def create_layer_basic(in_chan, out_chan, bnum, stride=1):
    layers = [BasicBlock(in_chan, out_chan, stride=stride)]
    for _ in range(bnum - 1):
        layers.append(BasicBlock(out_chan, out_chan, stride=1))
    return nn.Sequential(*layers)


class Resnet18(HipModule):
    def __init__(self):
        super().__init__()
        self.conv1 = Conv2d(3, 64, 7, stride=2, padding=3, bias=False, cin_pad="chunk")
        self.bn1 = _bn(64)
        self.conv1._bn_ref = (self.bn1,)
        self.layer1 = create_layer_basic(64, 64, bnum=2, stride=1)
        self.layer2 = create_layer_basic(64, 128, bnum=2, stride=2)
        self.layer3 = create_layer_basic(128, 256, bnum=2, stride=2)
        self.layer4 = create_layer_basic(256, 512, bnum=2, stride=2)

    def forward(self, x):
        x = ops.maxpool3x3s2(self.conv1.run(x, act=ACT_RELU))
        for blk in self.layer1:
            x = blk(x)
        feat8 = x
        for blk in self.layer2:
            feat8 = blk(feat8)
        feat16 = feat8
        for blk in self.layer3:
            feat16 = blk(feat16)
        feat32 = feat16
        for blk in self.layer4:
            feat32 = blk(feat32)
        return feat8, feat16, feat32


class ConvBNReLU(HipModule):
    def __init__(self, in_chan, out_chan, ks=3, stride=1, padding=1):
        super().__init__()
        self.conv = Conv2d(in_chan, out_chan, ks, stride=stride, padding=padding, bias=False)
        self.bn = _bn(out_chan)
        self.conv._bn_ref = (self.bn,)

    def forward(self, x, **kw):
        return self.conv.run(x, act=ACT_RELU, **kw)


class BiSeNetOutput(HipModule):
    def __init__(self, in_chan, mid_chan, n_classes):
        super().__init__()
        self.conv = ConvBNReLU(in_chan, mid_chan, ks=3, stride=1, padding=1)
        self.conv_out = Conv2d(mid_chan, n_classes, 1, bias=False)

    def forward(self, x, out=None):
        return self.conv_out.run(self.conv(x), out=out)


def _global_avg(x):
    """(N,H,W,C) -> (N,1,1,C) in x.dtype (F.avg_pool2d over the full map)."""
    mean, _ = ops.channel_stats(x, want_var=False)
    return ops.cast(mean, x.dtype).reshape(x.shape[0], 1, 1, x.shape[3])


class AttentionRefinementModule(HipModule):
    def __init__(self, in_chan, out_chan):
        super().__init__()
        self.conv = ConvBNReLU(in_chan, out_chan, ks=3, stride=1, padding=1)
        self.conv_atten = Conv2d(out_chan, out_chan, 1, bias=False)
        self.bn_atten = _bn(out_chan)
        self.conv_atten._bn_ref = (self.bn_atten,)

    def forward(self, x):
        """returns (feat, atten) — the product is fused with the following add by the caller."""
        feat = self.conv(x)
        atten = self.conv_atten.run(_global_avg(feat), act=ACT_SIGMOID)
        return feat, atten.reshape(feat.shape[0], feat.shape[3])


class ContextPath(HipModule):
    def __init__(self):
        super().__init__()
        self.resnet = Resnet18()
        self.arm16 = AttentionRefinementModule(256, 128)
        self.arm32 = AttentionRefinementModule(512, 128)
        self.conv_head32 = ConvBNReLU(128, 128, ks=3, stride=1, padding=1)
        self.conv_head16 = ConvBNReLU(128, 128, ks=3, stride=1, padding=1)
        self.conv_avg = ConvBNReLU(512, 128, ks=1, stride=1, padding=0)

    def forward(self, x):
        feat8, feat16, feat32 = self.resnet(x)
        n = x.shape[0]
        avg = self.conv_avg(_global_avg(feat32)).reshape(n, 128)     # broadcast add == nearest up of 1x1
        f32, a32 = self.arm32(feat32)
        feat32_sum = ops.gate_add(f32, gate=a32, addvec=avg)
        feat32_up = self.conv_head32(feat32_sum, ups=True)              # nearest x2 fused in the conv gather
        f16, a16 = self.arm16(feat16)
        feat16_sum = ops.gate_add(f16, gate=a16, addt=feat32_up)
        feat16_up = self.conv_head16(feat16_sum, ups=True)
        return feat8, feat16_up, feat32_up


class FeatureFusionModule(HipModule):
    def __init__(self, in_chan, out_chan):
        super().__init__()
        self.convblk = ConvBNReLU(in_chan, out_chan, ks=1, stride=1, padding=0)
        self.conv1 = Conv2d(out_chan, out_chan // 4, 1, bias=False)
        self.conv2 = Conv2d(out_chan // 4, out_chan, 1, bias=False)

    def forward(self, fsp, fcp):
        n, h, w, c1 = fsp.shape
        fcat = torch.empty((n, h, w, c1 + fcp.shape[3]), device=fsp.device, dtype=fsp.dtype)
        ops.copy_into(fsp, fcat[..., :c1])
        ops.copy_into(fcp, fcat[..., c1:])
        feat = self.convblk(fcat)
        atten = self.conv2.run(self.conv1.run(_global_avg(feat), act=ACT_RELU), act=ACT_SIGMOID)
        return ops.gate_add(feat, gate=atten.reshape(n, feat.shape[3]), addt=feat)   # feat*atten + feat


class BiSeNet(HipModule):
    def __init__(self, n_classes):
        super().__init__()
        self.n_classes = n_classes
        self.cp = ContextPath()
        self.ffm = FeatureFusionModule(256, 256)
        self.conv_out = BiSeNetOutput(256, 256, n_classes)
        self.conv_out16 = BiSeNetOutput(128, 64, n_classes)
        self.conv_out32 = BiSeNetOutput(128, 64, n_classes)

    def prepare_x3f(self, device):
        """bf16x3 mode: tensors stay fp32 (the glue kernels - max-pool, gates, resizes - are fp32 kernels) but every conv
        whose shape fits the split-half LDS-DMA kernel (Cin % 64 == 0, Cout % 8 == 0, no fused up-sampling, a dense fp32
        output) multiplies on 3 f16 MFMAs per product instead of the fp32 MFMA (1/16 of the 16-bit rate)."""
        from ..modules.rstt_layers import prepare_tree
        from ..ops import X3F
        prepare_tree(self, device, torch.float32)
        skip = {id(self.cp.conv_head32.conv), id(self.cp.conv_head16.conv),                 # nearest x2 fused in the gather
                id(self.conv_out.conv_out), id(self.conv_out16.conv_out), id(self.conv_out32.conv_out)}   # 19-channel slices
        for m in self.modules():
            if isinstance(m, Conv2d) and id(m) not in skip and m.in_channels % 64 == 0 and m.out_channels % 8 == 0:
                m.dt = X3F
                m._pack(device, X3F)

    def forward(self, x):
        """x: (N,512,512,ops.input_channels(dtype)) ImageNet-normalised -> (N,32,32,64): 3*n_classes parsing logits + zero pad."""
        nc = self.n_classes
        feat_res8, feat_cp8, feat_cp16 = self.cp(x)
        feat_fuse = self.ffm(feat_res8, feat_cp8)
        n = x.shape[0]
        cpad = (3 * nc + 7) // 8 * 8
        outf = ops.zero_(torch.empty((n, 32, 32, cpad), device=x.device, dtype=x.dtype))
        ops.resize_bilinear_ac(self.conv_out(feat_fuse), 32, 32, out=outf[..., 0:nc])
        ops.resize_bilinear_ac(self.conv_out16(feat_cp8), 32, 32, out=outf[..., nc:2 * nc])
        f32 = self.conv_out32(feat_cp16)
        assert f32.shape[1] == 32 and f32.shape[2] == 32, "the reference concatenates this head un-resized"
        ops.copy_into(f32, outf[..., 2 * nc:3 * nc])
        return outf


# ----------------------------------------------------------------------------------------------
# SFT fusion (reference: pgtformer_arch.py:402-484)
# ----------------------------------------------------------------------------------------------
def normalize(in_channels):
    from ..modules.rstt_layers import GroupNorm
    return GroupNorm(num_groups=32, num_channels=in_channels, eps=1e-6, affine=True)


class ResBlock(HipModule):
    def __init__(self, in_channels, out_channels=None):
        super().__init__()
        self.in_channels = in_channels
        self.out_channels = in_channels if out_channels is None else out_channels
        self.norm1 = normalize(in_channels)
        self.conv1 = Conv2d(in_channels, self.out_channels, 3, padding=1)
        self.norm2 = normalize(self.out_channels)
        self.conv2 = Conv2d(self.out_channels, self.out_channels, 3, padding=1)
        if self.in_channels != self.out_channels:
            self.conv_out = Conv2d(in_channels, self.out_channels, 1)

    def _pack(self, device, dtype):
        # bf16: a channel count that is not a multiple of 64 (the [enc|dec|fut] concats: 288, 544, 1056) is padded with
        # zero channels / zero filter taps so that the large-tile LDS-DMA kernels (64-channel K blocks) apply
        cin = self.in_channels
        self.cpad = (cin + 63) // 64 * 64 if (dtype != torch.float32 and cin % 64 and cin > 64) else None
        self.conv1.cin_pad = self.cpad
        if cin != self.out_channels:
            self.conv_out.cin_pad = self.cpad

    def forward(self, x_in):
        """x_in: (n,h,w,Cin); with channel padding active the caller passes the (n,h,w,cpad) buffer (zero pad channels)."""
        if self.cpad is not None:
            assert x_in.shape[-1] == self.cpad, (x_in.shape, self.cpad)
            n, hh, ww, _ = x_in.shape
            hbuf = torch.empty((n, hh, ww, self.cpad), device=x_in.device, dtype=x_in.dtype)
            ops.zero_(hbuf[..., self.in_channels:])                  # only the pad channels need the zeros
            self.norm1.run(x_in[..., :self.in_channels], ACT_SILU, out=hbuf[..., :self.in_channels])
            h = self.conv1.run(hbuf, gn=32)          # norm2's statistics from conv1's epilogue
        else:
            h = self.conv1.run(self.norm1.run(x_in, ACT_SILU), gn=32)
        h = self.norm2.run(h, ACT_SILU)
        sc = self.conv_out.run(x_in) if self.in_channels != self.out_channels else x_in
        return self.conv2.run(h, res=sc)


class Fuse_sft_block(HipModule):
    """Controllable feature fusion with the temporal 1x1 mix over T-stacked channels."""

    def __init__(self, in_ch, out_ch, t=3):
        super().__init__()
        self.tcc = 32
        self.encode_enc = ResBlock(2 * in_ch + self.tcc, out_ch)
        self.scale = nn.Sequential(Conv2d(in_ch, out_ch, 3, padding=1), nn.LeakyReLU(0.2, True),
                                   Conv2d(out_ch, out_ch, 3, padding=1))
        self.shift = nn.Sequential(Conv2d(in_ch, out_ch, 3, padding=1), nn.LeakyReLU(0.2, True),
                                   Conv2d(out_ch, out_ch, 3, padding=1))
        self.t = t
        self.tconvenc = Conv2d(in_ch, self.tcc, 1)
        self.tconvdec = Conv2d(in_ch, self.tcc, 1)
        self.tfusion0 = Conv2d(2 * t * self.tcc, self.tcc * self.t, 1)
        self.tfusion1 = Conv2d(self.tcc, self.tcc, 1)
        self.in_ch, self.out_ch = in_ch, out_ch

    def _pack(self, device, dtype):
        # scale.0 and shift.0 read the same tensor: one conv with the two filter banks stacked on Cout
        wss = torch.cat([self.scale[0].weight.detach(), self.shift[0].weight.detach()], 0)
        self.w_ss0 = _pack_matrix(wss, device, dtype)
        self.d_ss0 = _defect_t(wss, self.w_ss0) if _wants_wcomp(dtype, self) else None
        # exact-weight block: two-plane operands of the stacked scale.0 | shift.0 conv and of the temporal mix
        exact = _exact(self, dtype, self.in_ch)
        self.w_ss0_2 = _pack_matrix(wss, device, dtype, w2=True) if exact else None
        self.b_ss0 = _f32(torch.cat([self.scale[0].bias.detach(), self.shift[0].bias.detach()], 0), device)
        # bf16: tconvenc/tconvdec -> stack over T -> tfusion0 -> tfusion1 are all 1x1 and linear (reference :467-473), so
        # fut of output frame `to` is ONE linear map of the window's T [enc|dec] pixels: a (T x 1)-tap conv over the
        # concat buffer viewed as (windows, T, h*w, channels), K = T * 2C, composed here in fp32.
        self.w_mix = None
        if dtype != torch.float32:
            t, tcc, c = self.t, self.tcc, self.in_ch
            we, be = self.tconvenc.weight.detach().float().view(tcc, c), self.tconvenc.bias.detach().float()
            wd_, bd = self.tconvdec.weight.detach().float().view(tcc, c), self.tconvdec.bias.detach().float()
            w0, b0 = self.tfusion0.weight.detach().float().view(t * tcc, 2 * t * tcc), self.tfusion0.bias.detach().float()
            w1, b1 = self.tfusion1.weight.detach().float().view(tcc, tcc), self.tfusion1.bias.detach().float()
            bcat = torch.cat([be] * t + [bd] * t)
            self.w_mix, self.b_mix, self.d_mix, self.w_mix2 = [], [], [], []
            for to in range(t):
                rows = w0[to * tcc:(to + 1) * tcc]                                           # (tcc, 2*t*tcc)
                taps = [torch.cat([w1 @ rows[:, ti * tcc:(ti + 1) * tcc] @ we,
                                   w1 @ rows[:, (t + ti) * tcc:(t + ti + 1) * tcc] @ wd_], 1) for ti in range(t)]
                wm = torch.stack(taps, 1).reshape(tcc, -1)
                self.w_mix.append(_pack_matrix(wm, device, dtype))     # K-major (tcc, T*2C)
                # every tap reads its own frame of the window: the defect keeps one row per (frame, channel)
                self.d_mix.append(_defect_t(wm, self.w_mix[-1]) if _wants_wcomp(dtype, self) else None)
                self.w_mix2.append(_pack_matrix(wm, device, dtype, w2=True) if exact and (2 * c) % 64 == 0 else None)
                self.b_mix.append(_f32(w1 @ (rows @ bcat + b0[to * tcc:(to + 1) * tcc]) + b1, device))

    def concat_width(self):
        """(real, padded) channel count of the [enc | dec | fut] concat buffer."""
        ct = 2 * self.in_ch + self.tcc
        return ct, (self.encode_enc.cpad if self.encode_enc.cpad is not None else ct)

    def new_concat(self, n, h, wd, device, dtype):
        """Concat buffer for a (n,h,wd,C) level (bf16 modes): producers may write its enc / dec slices directly."""
        ct, ctp = self.concat_width()
        cat = torch.empty((n, h, wd, ctp), device=device, dtype=dtype)
        if ctp != ct:
            ops.zero_(cat[..., ct:])
        return cat

    def _frame_index(self, b, k, device):
        """int32 device tensor [k, T+k, 2T+k, ...]: frame k of each of the b windows (cached: created once, before graph capture)."""
        key = (b, k, str(device))
        cache = self.__dict__.setdefault("_fidx", {})
        if key not in cache:
            cache[key] = (torch.arange(b, dtype=torch.int32) * self.t + k).to(device)
        return cache[key]

    def forward(self, enc_feat, dec_feat, temb=None, w=1, cat=None, keep=None):
        """enc_feat, dec_feat: (B*T, h, w, C) (reference: :460-484).  cat: a new_concat() buffer whose enc and/or dec
        slices were already written by the producers (then enc_feat / dec_feat ARE those slices and are not copied).
        keep: None, or a frame index k: only frame k of every window is wanted from here on (the driver keeps the middle
        frame, inference.py:15, and everything after this block's temporal mix is per-frame): the mix is evaluated for
        output frame k only and the per-frame tail (ResBlock, scale / shift, modulation) runs on B frames instead of
        B*T; returns (B, h, w, C)."""
        n, h, wd, c = dec_feat.shape
        t, tcc = self.t, self.tcc
        b = n // t
        dev, dt = dec_feat.device, dec_feat.dtype
        if self.w_mix is not None:
            ct, ctp = self.concat_width()
            if cat is None:
                cat = self.new_concat(n, h, wd, dev, dt)               # [enc | dec | fut | 0]
            if enc_feat.data_ptr() != cat.data_ptr():
                ops.copy_into(enc_feat, cat[..., :c])
            if dec_feat.data_ptr() != cat[..., c:2 * c].data_ptr():
                ops.copy_into(dec_feat, cat[..., c:2 * c])
            src = cat.view(b, t, h * wd, ctp)[..., :2 * c]            # windows x T frames x pixels x [enc|dec]
            dst = cat.view(n, 1, h * wd, ctp)[..., 2 * c:ct]          # fut channels, rows = frame * h*w + pixel
            # the kernels take 32-bit byte offsets: more windows than fit 2 GiB of concat buffer run as window chunks
            per = max(1, ((1 << 31) - 1) // (t * h * wd * ctp * cat.element_size()))
            # per-window means of the T [enc | dec] frames, (b, T*2C), for the weight-rounding compensation of the mix
            wmean = None
            mix2 = self.w_mix2[0] is not None and src.dtype == torch.float16 and ctp % 8 == 0
            if self.d_mix[0] is not None and (h * wd) % 512 == 0 and not mix2:
                wmean = ops.sampled_channel_mean(cat.view(n, h * wd, ctp)[..., :2 * c]).view(b, t * 2 * c)
            for i0 in range(0, b, per):
                i1 = min(b, i0 + per)
                for to in (range(t) if keep is None else (keep,)):
                    # output pixel m = window*h*w + pix  ->  row (window*T + to)*h*w + pix
                    if mix2:      # exact weights: nothing to compensate
                        ops.conv2d(src[i0:i1], self.w_mix2[to], self.b_mix[to], kh=t, kw=1, out=dst[i0 * t:i1 * t],
                                   out_rows=(t, 1 - t, to * h * wd), w2=tcc)
                        continue
                    bm = self.b_mix[to] if wmean is None else ops.mean_field_bias(wmean[i0:i1], self.d_mix[to], self.b_mix[to])
                    ops.conv2d(src[i0:i1], self.w_mix[to], bm, kh=t, kw=1, out=dst[i0 * t:i1 * t],
                               out_rows=(t, 1 - t, to * h * wd))
            if keep is not None:     # frame `keep` of every window: [enc | dec | fut | 0] rows of B frames
                cat = ops.gather_frames(cat, self._frame_index(b, keep, dev))
                dec_feat = cat[..., c:2 * c]
            e = self.encode_enc(cat if self.encode_enc.cpad is not None else cat[..., :ct])
            co = self.out_ch
            if self.w_ss0_2 is not None and ops.w2_ok(e, 2 * co, e.shape[-1], 3, 3, 1, (1, 1, 1, 1), act=ACT_LEAKY02):
                ss = ops.conv2d(e, self.w_ss0_2, self.b_ss0, kh=3, kw=3, pad=(1, 1, 1, 1), act=ACT_LEAKY02, w2=2 * co)
            else:
                ss = ops.conv2d(e, self.w_ss0, _frame_bias(e, self.d_ss0, self.b_ss0), kh=3, kw=3, pad=(1, 1, 1, 1), act=ACT_LEAKY02)
            shift = self.shift[2].run(ss[..., co:])
            return self.scale[2].run(ss[..., :co], sft=(dec_feat, shift, w))
        # per-frame 1x1 -> T frames stacked on channels: [enc t0..t2 | dec t0..t2]
        stacked = torch.empty((b, h, wd, 2 * t * tcc), device=dev, dtype=dt)
        for bi in range(b):
            for ti in range(t):
                f = bi * t + ti
                self.tconvenc.run(enc_feat[f:f + 1], out=stacked[bi:bi + 1, :, :, ti * tcc:(ti + 1) * tcc])
                self.tconvdec.run(dec_feat[f:f + 1], out=stacked[bi:bi + 1, :, :, (t + ti) * tcc:(t + ti + 1) * tcc])
        fut0 = self.tfusion0.run(stacked)                                   # (b,h,w,T*32)  temporal fusion
        cat = torch.empty((n, h, wd, 2 * c + tcc), device=dev, dtype=dt)    # [enc | dec | fut]
        ops.copy_into(enc_feat, cat[..., :c])
        ops.copy_into(dec_feat, cat[..., c:2 * c])
        for bi in range(b):
            for ti in range(t):
                f = bi * t + ti
                self.tfusion1.run(fut0[bi:bi + 1, :, :, ti * tcc:(ti + 1) * tcc], out=cat[f:f + 1, :, :, 2 * c:])
        e = self.encode_enc(cat)
        ss = ops.conv2d(e, self.w_ss0, self.b_ss0, kh=3, kw=3, pad=(1, 1, 1, 1), act=ACT_LEAKY02)  # (n,h,w,2C)
        co = self.out_ch
        shift = self.shift[2].run(ss[..., co:])
        # out = dec + w*(dec*scale + shift) as the epilogue of the last scale conv
        out = self.scale[2].run(ss[..., :co], sft=(dec_feat, shift, w))
        return out if keep is None else ops.gather_frames(out, self._frame_index(b, keep, dev))


# ----------------------------------------------------------------------------------------------
# PGTFormer (reference: pgtformer_arch.py:490-714)
# ----------------------------------------------------------------------------------------------
@ARCH_REGISTRY.register()
class PGTFormer(TDCRQVAE3):
    def __init__(self, ddconfig, dim_embd=512, n_head=8, n_layers=9, connect_list=("32", "64", "128", "256"),
                 fix_modules=("quantizer", "decoder", "conditionnet"), w=0, detach_16=True, adain=False, tf=3,
                 droprate=0.0, **kwargs):
        super().__init__(ddconfig=ddconfig, tf=tf, **kwargs)
        self.fix_modules = list(fix_modules) if fix_modules is not None else None
        self.t, self.w, self.detach_16, self.adain = tf, w, detach_16, adain
        self.connect_list = list(connect_list)
        self.n_layers, self.dim_embd, self.dim_mlp, self.n_head = n_layers, dim_embd, dim_embd * 2, n_head
        self.conditionnet = BiSeNet(19)
        self.convpos = Conv2d(57, 512, 1, cin_pad=64)
        self.feat_emb = Linear(512, dim_embd)
        self.ft_layers = nn.Sequential(*[TransformerSALayer(embed_dim=dim_embd, nhead=n_head, dim_mlp=self.dim_mlp,
                                                            dropout=droprate) for _ in range(n_layers)])
        self.codebook_size = self.quantizer.n_embed[-1]
        self.quantizer_depth = self.quantizer.code_shape[-1]
        self.idx_pred_layer = nn.Sequential(LayerNorm(dim_embd),
                                            Linear(dim_embd, self.quantizer_depth * self.codebook_size, bias=False))
        self.channels = {"16": 512, "32": 512, "64": 256, "128": 256, "256": 128, "512": 64}
        self.fuse_encoder_indices = {"512": 0, "256": 1, "128": 2, "64": 3, "32": 4, "16": 5}
        self.fuse_convs_dict = nn.ModuleDict()
        for f_size in self.connect_list:
            in_ch = self.channels[f_size]
            self.fuse_convs_dict[f_size] = Fuse_sft_block(in_ch, in_ch, t=tf)
        self.requires_grad_(False)   # inference-only build
        nn.Module.eval(self)

    def eval(self):
        # the reference's train() override returns None, so `m = m.eval()` breaks there
        # (pgtformer_arch.py:577-581); here eval() returns self as nn.Module promises.
        return nn.Module.eval(self)

    @torch.no_grad()
    def forward(self, x, w=None, detach_16=True, code_only=None, adain=None):
        """x: (B*T,3,512,512) fp32 in [0,1] (or uint8 (B*T,512,512,3)).  Returns the reference's tuple
        (out (B*T,3,512,512) fp32, logits (B*T,32,32,1,1024) fp32, lq_feat (B*T,32,32,512) fp32)."""
        out_nhwc, logits, lq = self.forward_nhwc(x, w=w, code_only=code_only, adain=adain)
        lq32 = ops.from_x3(lq) if _is_x3(self.enc_dt) else ops.cast(lq, torch.float32)
        if code_only:
            return logits, lq32
        return ops.nhwc_to_nchw_f32(out_nhwc), logits, lq32

    @staticmethod
    def window_index(n_windows, t=3, device="cuda"):
        """Frame indices of `n_windows` consecutive sliding windows over n_windows + t - 1 frames: window i = frames
        i .. i+t-1 (the reference driver's window policy, inference.py:47-74)."""
        return (torch.arange(n_windows, dtype=torch.int32)[:, None] + torch.arange(t, dtype=torch.int32)[None, :]).reshape(-1).to(device)

    @torch.no_grad()
    def forward_nhwc(self, x, w=None, code_only=None, adain=None, win=None, codes=None, direct=None, middle_only=False):
        """Same computation, channels-last results and no layout conversion: out (B*T,512,512,3) in the
        decoder dtype, logits fp32, lq_feat (B*T,32,32,512) in the encoder dtype ((B*T,32,32,1024) split-half planes
        [hi | lo] in bf16x3 mode).

        win: None (x holds the B*T frames of B windows back to back), or an int32 device tensor (B*T,) of indices into
        the frames of x: the windows' UNIQUE frames are given once and everything per-frame (BiSeNet, convpos, the encoder
        up to its first temporal attention) is computed once per frame, then gathered to window order.  Results equal the
        win=None call on x[win] (the per-frame operators act on each frame independently).
        codes: optional (B*T,32,32,depth) integer tensor that REPLACES the predicted codes (teacher forcing: tests feed the
        reference's codes to separate decoder arithmetic from code flips).
        direct: None = automatic; False forces the copying (non in-place) concat path in bf16 (tests).
        middle_only: the caller keeps only the middle frame of every window (the reference driver: `[0][1]`,
        inference.py:15).  Everything after the decoder's last temporal operation (with the shipping config: the temporal
        mix of the 256x256 fusion block; attention stops at 128x128) is per-frame, so from there on only the middle frames
        are computed: out is (B,512,512,3).  The middle frame is the same as in the full computation."""
        self._check_ready()
        w = self.w if w is None else w
        adain = self.adain if adain is None else adain
        t = self.t
        raw, nx = self._ingest(x)
        bt = raw.shape[0] if win is None else win.numel()
        b = bt // t
        # condition branch: BiSeNet parsing map -> positional embedding of the code transformer (per frame).  Its many small
        # launches (18 frames, maps down to 16x16: a fraction of the chip each) can run on a second stream next to the
        # encoder's large HBM-bound launches (PGT_SIDE_STREAM=0 disables; the fork / join is captured into the HIP graph).
        x3 = _is_x3(self.enc_dt)

        def condition_branch():
            self.last_parsing = self.conditionnet(nx)                       # (F,32,32,64): 3 x 19 parsing logits + pad
            c = self.convpos.run(self.last_parsing)                         # (F,32,32,512)
            if win is not None:
                c = ops.gather_frames(c, win)                               # (bt,32,32,512)
            p = c.reshape(bt * c.shape[1] * c.shape[2], c.shape[3])          # rows (b,t,y,x) == (T*H*W, B) order
            return c, (ops.to_x3(p) if x3 else p)                           # fp32 BiSeNet map -> split-half operand

        # NOT inside a stream capture: a captured fork gives the graph a second root, and ROCm 7.0's hipGraphLaunch
        # (hip::Graph::UpdateStreams) then looks among the executable graph's internal streams for one that sits on another hardware
        # queue than the launch stream WITHOUT bounding the search - when both internal streams share the launch stream's queue it
        # reads past the end of the vector and dereferences what it finds (the segfault "inside the HIP runtime's graph launch" of
        # rounds 5 / 6: DESIGN.md section 3.4; which queue a stream gets depends on every stream the process created before).  A
        # graph without forks has one root and never enters that loop; with two forwards in flight the fork is worth nothing
        # (172.9 against 172.9 frames/s), with one +1.4 %.
        side = None
        self.last_forked = False
        if SIDE_STREAM and raw.is_cuda and not torch.cuda.is_current_stream_capturing():
            self.last_forked = True
            main = torch.cuda.current_stream(raw.device)
            side = self.__dict__.setdefault("_side_stream", None) or torch.cuda.Stream(device=raw.device)
            self.__dict__["_side_stream"] = side
            side.wait_stream(main)
            with torch.cuda.stream(side):
                keep_branch, ops.BRANCH = ops.BRANCH, 1      # concurrent with the encoder: its own frame_bias arrival counters
                try:
                    cond, pos = condition_branch()
                finally:
                    ops.BRANCH = keep_branch
        else:
            cond, pos = condition_branch()
        self.last_cond = cond
        th, tw = cond.shape[1], cond.shape[2]
        # encoder.  bf16: the fusion blocks' [enc | dec | fut] concat buffers exist up front and the encoder levels / the
        # decoder levels write their feature maps straight into the enc / dec slices (no concat copies)
        cats, feat_out = {}, None
        can_direct = (self.dec_dt in (torch.bfloat16, torch.float16) and w > 0 and not code_only)
        direct = can_direct if direct is None else (direct and can_direct)
        if direct:
            feat_out = {}
            for f_size in self.connect_list:
                blk = self.fuse_convs_dict[f_size]
                if blk.w_mix is None:
                    continue
                res = int(f_size)
                cats[f_size] = blk.new_concat(bt, res, res, raw.device, self.dec_dt)
                feat_out[self.fuse_encoder_indices[f_size]] = cats[f_size][..., :blk.in_ch]
        want = {self.fuse_encoder_indices[f] for f in self.connect_list}
        z, feats = self.encoder(raw, return_multi_res_feats=True, feat_out=feat_out, win=win, want_feats=want,
                                feat_dtype=self.dec_dt if self.dec_dt in (torch.float16, torch.bfloat16) else None)
        enc_feat = {}
        for f_size in self.connect_list:
            f = feats[self.fuse_encoder_indices[f_size]]
            enc_feat[str(f.shape[2])] = f
        lq_feat = self.quant_conv.run(z)                                     # (bt,32,32,512); x3: (bt,32,32,1024)
        lq_style = lq_feat       # AdaIN style statistics (fp32, from the merged hi + lo planes of a split tensor: below)
        # code-prediction transformer over the T*32*32 tokens of each window
        if side is not None:
            torch.cuda.current_stream(raw.device).wait_stream(side)          # join: pos is first used here
        L = t * th * tw
        q = self.feat_emb.run(lq_feat.reshape(bt * th * tw, lq_feat.shape[3]))
        for layer in self.ft_layers:
            q = layer(q, b, L, query_pos=pos)
        ln = self.idx_pred_layer[0].run(q)
        logits2d = ops.linear(ln, self.idx_pred_layer[1].pw, None, out_f32=True, x3=x3)   # (bt*32*32, depth*K) fp32
        logits = logits2d.reshape(bt, *self.quantizer.code_shape, self.codebook_size)
        if code_only:
            return None, logits, lq_feat
        # quantisation: first-max code per token, codebook gather, AdaIN against the LQ features
        depth = self.quantizer_depth
        if codes is None:
            codes = ops.argmax_rows(logits2d.reshape(bt * th * tw * depth, self.codebook_size))
        else:
            codes = codes.to(device=raw.device, dtype=torch.int32).contiguous()
        self.last_codes = codes.reshape(bt, th, tw, depth)
        quant = self.quantizer.embed_code(self.last_codes, self.dec_dt)      # (bt,32,32,512)
        if adain:
            quant = adaptive_instance_normalization(quant, ops.from_x3(lq_style) if x3 else lq_style)
        z_q = self.post_quant_conv.run(quant)

        # the decoder's last temporal operation: a fusion block's temporal mix (w > 0) or an EncoderLayer
        mid = None
        if middle_only:
            sizes = [self.decoder.resolution >> i for i in reversed(range(self.decoder.num_resolutions))]   # 32 .. 512
            temporal = [(str(sz), "fuse" if (str(sz) in self.connect_list and w > 0) else "attn")
                        for sz, lvl in zip(sizes, reversed(list(self.decoder.up)))
                        if (str(sz) in self.connect_list and w > 0) or len(lvl.attn) > 0]
            mid = (t // 2,) + (temporal[-1] if temporal else (str(sizes[0]), "start"))

        def fuse(f_size, h):
            if f_size in self.connect_list and w > 0:
                keep = mid[0] if (mid is not None and mid[1:] == (f_size, "fuse")) else None
                return self.fuse_convs_dict[f_size](ops.cast(enc_feat[f_size], self.dec_dt), h, temb=None, w=w,
                                                    cat=cats.get(f_size), keep=keep)
            return h

        def fuse_dst(f_size):
            if f_size in cats:
                c = self.fuse_convs_dict[f_size].in_ch
                return cats[f_size][..., c:2 * c]
            return None

        out = self.decoder(z_q, fuse=fuse, fuse_dst=fuse_dst if cats else None, mid=mid)   # (bt | b, 512,512,3)
        return out, logits, lq_feat

    @torch.no_grad()
    def check_range(self, window_u8, w=1.0, win=None, full_tail=False):
        """Range telemetry of the 16-bit modes: ONE eager forward of `window_u8` (as restore_middle_u8 takes it) in which every
        operator counts the elements of its IEEE-half outputs that sit at the saturation limit +-65504 or are not finite
        (ops.RANGE_CHECK, pgt_count_saturated).  Returns the list of (operator, shape, count) with count > 0 - empty when the
        checkpoint's activations fit the half range.  The half decoder of the default mode clamps silently otherwise:
        callers fall back to prepare(device, "bf16x3") (bf16 decoder: no range limit).  Rows that stay on chip inside the fused
        token-row chains (x1 and the hidden row of a block tail; the normalised rows are bounded by 16) are not tensors and
        are not counted: a block whose x1 saturated shows up in its output and in the layers that follow."""
        recs, hooks = [], []
        for name, mod in self.named_modules():          # records carry the innermost module whose forward() is running
            hooks.append(mod.register_forward_pre_hook(lambda m, a, n=name: ops.RANGE_CTX.append(n)))
            hooks.append(mod.register_forward_hook(_range_pop))
        ops.RANGE_CHECK = recs
        try:
            self.forward_nhwc(window_u8, w=w, win=win, middle_only=not full_tail)
        finally:
            ops.RANGE_CHECK = None
            del ops.RANGE_CTX[:]
            for h in hooks:
                h.remove()
        self.last_range_launches = len(recs)
        return [r for r in ops.range_report(recs) if r[2] > 0]

    @torch.no_grad()
    def restore_middle_u8(self, window_u8, w=1.0, win=None, out=None, full_tail=False):
        """Driver fast path (reference: inference.py:12-19): uint8 (3,H,W,3) window -> restored middle
        frame as uint8 (H,W,3) with floor(clamp(x,0,1)*255), without leaving the device.
        B windows stacked on the frame axis, (B*3,H,W,3), give (B,H,W,3): B independent windows per forward
        (the reference accepts only B=1, modules/rstt_layers.py:904; here B>1 == B separate calls).
        win: see forward_nhwc - window_u8 then holds the windows' unique frames.
        full_tail: compute all T frames through the per-frame tail of the decoder as the reference does before it discards
        two of them (`[0][1]`); default: only the middle frames after the last temporal operation (same result)."""
        res, _, _ = self.forward_nhwc(window_u8, w=w, win=win, middle_only=not full_tail)   # (B,H,W,3): the middle frames
        if full_tail:
            res = res[self.t // 2::self.t]
        b = res.shape[0]
        if b == 1 and out is None:
            return ops.frame_to_u8(res[0])
        if out is None:
            out = torch.empty((b,) + tuple(res.shape[1:3]) + (3,), device=res.device, dtype=torch.uint8)
        for i in range(b):
            ops.frame_to_u8(res[i], out=out[i])
        return out
