"""Video-Swin layers of the reference's modules/swin.py, HIP-backed.

Host-side mirror of `WindowAttention3D` (:85-167), `Mlp` (:14-35), `SwinTransformerBlock3D` (:170-275) and `BasicLayer`
(:326-409, the `tdswin_pre` / `tdswin_post` stages of TDRQVAE, archs/tdrqvae_arch.py:834-854): same class names, constructor
arguments and state-dict keys, so a reference checkpoint loads unchanged.  torch.nn modules are parameter containers; the
forward() launches the gfx950 kernels.  Tokens stay channels-last (B*D*H*W, C); roll on three axes, (Wd, Wh, Ww) window
partition / reverse, the relative-position bias and the 27-region shift mask (`compute_mask` :311-323) are address arithmetic
and a region test inside ONE attention kernel (`pgt_window_attention3d`, csrc/window_attn_mfma.hip) - no mask tensor, no
permuted copies.

Precision: `prepare(device, torch.bfloat16)` runs everything in bf16 storage (fp32 statistics, softmax, accumulation);
`prepare(device, torch.float32)` keeps LayerNorm and the Linears in exact fp32 and rounds only the attention kernel's qkv
input / output to `attn_dtype` (default torch.float16: BASELINE.json configs[4] quotes the fp16 kernel; the conv / linear
family has bf16 and fp32 forms only).  Inference only (drop / attn_drop / drop_path must be 0)."""
import torch
import torch.nn as nn

from .. import ops
from ..ops import ACT_GELU
from .rstt_layers import HipModule, LayerNorm, Linear, get_window_size


def relative_position_index(window_size):
    """(Wd*Wh*Ww, Wd*Wh*Ww) int64 index into the bias table (reference: modules/swin.py:102-117)."""
    wd, wh, ww = window_size
    coords = torch.stack(torch.meshgrid(torch.arange(wd), torch.arange(wh), torch.arange(ww), indexing="ij"))
    flat = torch.flatten(coords, 1)
    rel = (flat[:, :, None] - flat[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += wd - 1
    rel[:, :, 1] += wh - 1
    rel[:, :, 2] += ww - 1
    rel[:, :, 0] *= (2 * wh - 1) * (2 * ww - 1)
    rel[:, :, 1] *= 2 * ww - 1
    return rel.sum(-1)


class Mlp(HipModule):
    """fc1 -> exact-erf GELU -> fc2 (reference: modules/swin.py:14-35)."""

    def __init__(self, in_features, hidden_features=None, out_features=None, drop=0.0):
        super().__init__()
        assert drop == 0.0
        self.fc1 = Linear(in_features, hidden_features or in_features)
        self.fc2 = Linear(hidden_features or in_features, out_features or in_features)


class WindowAttention3D(HipModule):
    """Window attention with a fused qkv Linear and a (2Wd-1)(2Wh-1)(2Ww-1) x heads bias table (reference: :85-167)."""

    def __init__(self, dim, window_size, num_heads, qkv_bias=False, qk_scale=None, attn_drop=0.0, proj_drop=0.0):
        super().__init__()
        assert qk_scale is None and attn_drop == 0.0 and proj_drop == 0.0
        self.dim, self.window_size, self.num_heads = dim, tuple(window_size), num_heads
        wd, wh, ww = self.window_size
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * wd - 1) * (2 * wh - 1) * (2 * ww - 1), num_heads))
        self.register_buffer("relative_position_index", relative_position_index(self.window_size))
        self.qkv = Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = Linear(dim, dim)
        nn.init.trunc_normal_(self.relative_position_bias_table, std=0.02)
        self._bias_cache = {}

    def _pack(self, device, dtype):
        self._bias_cache = {}

    def bias_dense(self, win):
        """(heads, N, N) fp32 bias of a (possibly clamped) window: table[index[:N, :N]] as the reference gathers it (:150-153)."""
        n = win[0] * win[1] * win[2]
        b = self._bias_cache.get(n)
        if b is None:
            idx = self.relative_position_index[:n, :n].reshape(-1)
            b = self.relative_position_bias_table.detach().float()[idx].reshape(n, n, self.num_heads)
            b = b.permute(2, 0, 1).contiguous().to(self.dev)
            self._bias_cache[n] = b
        return b


class SwinTransformerBlock3D(HipModule):
    """LN -> (shifted) window attention -> + shortcut -> LN -> Mlp -> + residual (reference: :170-275)."""

    def __init__(self, dim, num_heads, window_size=(2, 7, 7), shift_size=(0, 0, 0), mlp_ratio=4.0, qkv_bias=True,
                 qk_scale=None, drop=0.0, attn_drop=0.0, drop_path=0.0, use_checkpoint=False):
        super().__init__()
        assert drop == 0.0 and attn_drop == 0.0 and drop_path == 0.0, "inference only"
        assert all(0 <= s < w for s, w in zip(shift_size, window_size)), "shift_size must in 0-window_size"
        self.dim, self.num_heads = dim, num_heads
        self.window_size, self.shift_size, self.mlp_ratio = tuple(window_size), tuple(shift_size), mlp_ratio
        self.norm1 = LayerNorm(dim)
        self.attn = WindowAttention3D(dim, self.window_size, num_heads, qkv_bias=qkv_bias, qk_scale=qk_scale)
        self.norm2 = LayerNorm(dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))
        self.attn_dtype = torch.float16      # storage type of the attention kernel when the module runs in fp32

    def _attention(self, xt, B, D, H, W):
        # windows are clamped to the feature map (get_window_size :67-82); maps that are not multiples of the window are
        # padded inside the kernel the way forward_part1 pads norm1(x) with zeros (:218-223): a padding token's qkv row is
        # the qkv bias
        win, shift = get_window_size((D, H, W), self.window_size, self.shift_size)
        qkv = self.attn.qkv.run(self.norm1.run(xt))
        if qkv.dtype == torch.float32:
            qkv = qkv.to(self.attn_dtype)
        pad = None
        if any(s % w for s, w in zip((D, H, W), win)) and self.attn.qkv.bias is not None:
            pad = self.attn.qkv.bias.detach()
        ao = ops.window_attention3d(qkv, self.attn.bias_dense(win), B, D, H, W, self.dim, self.num_heads, win, shift, pad)
        return ao if ao.dtype == xt.dtype else ao.to(xt.dtype)

    def forward_part1(self, xt, B, D, H, W):
        """xt (B*D*H*W, C) tokens in (b, d, y, x) order -> attention branch before the shortcut add (reference: :212-246)."""
        return self.attn.proj.run(self._attention(xt, B, D, H, W))

    def forward(self, xt, B, D, H, W):
        """x = shortcut + part1(x); x = x + mlp(norm2(x)) (reference: :251-270); both adds are GEMM epilogues."""
        x1 = self.attn.proj.run(self._attention(xt, B, D, H, W), res=xt)
        m = self.mlp.fc1.run(self.norm2.run(x1), act=ACT_GELU)
        return self.mlp.fc2.run(m, res=x1)


class BasicLayer(HipModule):
    """One Swin stage: `depth` blocks alternating shift (0,0,0) / window // 2 (reference: :326-409)."""

    def __init__(self, dim, depth, num_heads, window_size=(1, 7, 7), mlp_ratio=4.0, qkv_bias=False, qk_scale=None,
                 drop=0.0, attn_drop=0.0, drop_path=0.0, downsample=None, use_checkpoint=False):
        super().__init__()
        assert downsample is None, "PatchMerging is not used by the reference's TDRQVAE stages"
        self.window_size = tuple(window_size)
        self.shift_size = tuple(i // 2 for i in self.window_size)
        self.depth, self.dim = depth, dim
        self.blocks = nn.ModuleList([
            SwinTransformerBlock3D(dim, num_heads, self.window_size, (0, 0, 0) if i % 2 == 0 else self.shift_size,
                                   mlp_ratio, qkv_bias, qk_scale, drop, attn_drop,
                                   drop_path[i] if isinstance(drop_path, list) else drop_path)
            for i in range(depth)])

    def forward_tokens(self, xt, B, D, H, W):
        """channels-last tokens (B*D*H*W, C) in, the same out: the form the rest of the build uses."""
        for blk in self.blocks:
            xt = blk(xt, B, D, H, W)
        return xt

    def forward(self, x):
        """x (B, C, D, H, W) -> (B, C, D, H, W): the reference's signature (:389-409); the two permutes are the only copies."""
        B, C, D, H, W = x.shape
        xt = x.permute(0, 2, 3, 4, 1).contiguous().reshape(B * D * H * W, C)
        if xt.dtype != self.dt:
            xt = xt.to(self.dt)
        y = self.forward_tokens(xt, B, D, H, W)
        return y.reshape(B, D, H, W, C).permute(0, 4, 1, 2, 3).contiguous()
