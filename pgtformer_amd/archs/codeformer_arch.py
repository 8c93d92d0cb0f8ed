"""Code-prediction transformer layer and AdaIN, HIP-backed (the two symbols PGTFormer takes from the
reference's archs/codeformer_arch.py: `TransformerSALayer` :102-137, `adaptive_instance_normalization`
:15-46). Token matrices are (B*L, E) row-major with rows in (b, t, y, x) order — the reference's
seq-first (L, B, E) with B=1 is the same memory."""
import torch.nn as nn

from .. import ops
from ..modules.rstt_layers import HipModule, LayerNorm, Linear, _f32, _is_x3, _pack_matrix
from ..ops import ACT_GELU


def adaptive_instance_normalization(content_feat, style_feat, eps=1e-5):
    """content/style: (N,H,W,C). (content - mean_c)/std_c * std_s + mean_s with per-(n,c) statistics over
    pixels and UNBIASED variance (reference: codeformer_arch.py:15-46); applied as one affine pass."""
    mc, vc = ops.channel_stats(content_feat)
    ms, vs = ops.channel_stats(style_feat)
    scale, shift = ops.adain_affine(mc, vc, ms, vs, eps)
    return ops.affine_act(content_feat, scale, shift)


class TransformerSALayer(HipModule):
    """Pre-LN self-attention (q = k = LN(x)+pos, v = LN(x)) + pre-LN GELU FFN."""

    def __init__(self, embed_dim, nhead=8, dim_mlp=2048, dropout=0.0, activation="gelu"):
        super().__init__()
        assert activation == "gelu" and dropout == 0.0
        self.embed_dim, self.nhead = embed_dim, nhead
        self.self_attn = nn.MultiheadAttention(embed_dim, nhead, dropout=dropout)  # parameter container
        self.linear1 = Linear(embed_dim, dim_mlp)
        self.linear2 = Linear(dim_mlp, embed_dim)
        self.norm1 = LayerNorm(embed_dim)
        self.norm2 = LayerNorm(embed_dim)

    def _pack(self, device, dtype):
        e = self.embed_dim
        w, b = self.self_attn.in_proj_weight.detach(), self.self_attn.in_proj_bias.detach()
        self.w_qk = _pack_matrix(w[:2 * e], device, dtype)   # applied to LN(x)+pos
        self.b_qk = _f32(b[:2 * e], device)
        self.w_v = _pack_matrix(w[2 * e:], device, dtype)    # applied to LN(x)
        self.b_v = _f32(b[2 * e:], device)
        self.w_o = _pack_matrix(self.self_attn.out_proj.weight, device, dtype)
        self.b_o = _f32(self.self_attn.out_proj.bias, device)

    def forward(self, tgt, B, L, query_pos=None):
        """tgt, query_pos: (B*L, E)."""
        e, hd = self.embed_dim, self.embed_dim // self.nhead
        x3 = _is_x3(self.dt)
        t2, t2p = self.norm1.run(tgt, pos=query_pos)
        qk = ops.linear(t2p, self.w_qk, self.b_qk, x3=x3)           # x3: (rows, 4E) = [hi q k | lo q k]
        v = ops.linear(t2, self.w_v, self.b_v, x3=x3)               # x3: (rows, 2E) = [hi v | lo v]
        if x3:
            ao = ops.mha(qk[:, :e], qk[:, e:2 * e], v[:, :e], B, L, self.nhead, hd, float(hd) ** -0.5, x3=(2 * e, 2 * e, e))
        else:
            ao = ops.mha(qk[:, :e], qk[:, e:], v, B, L, self.nhead, hd, float(hd) ** -0.5)
        tgt = ops.linear(ao, self.w_o, self.b_o, res=tgt, x3=x3)
        m = self.linear1.run(self.norm2.run(tgt), act=ACT_GELU)
        return self.linear2.run(m, res=tgt)
