"""`VectorQuantizer` of the reference's archs/vqgan_arch.py (:25-98), inference surface, HIP-backed: the nearest-code
look-up |z|^2 + |e|^2 - 2 z.e^T -> arg-min (:48-54), the code-book gather (:63) and the (1 + beta) * mean((z_q - z)^2)
loss value (:65).  Same state-dict key (`embedding.weight`); activations channels-last.  Everything else in that file
(VQAutoEncoder, Generator, discriminator) is never executed by PGTFormer inference ("not on the path")."""
import torch
import torch.nn as nn

from .. import ops
from ..modules.rstt_layers import HipModule


class VectorQuantizer(HipModule):
    def __init__(self, codebook_size, emb_dim, beta):
        super().__init__()
        self.codebook_size, self.emb_dim, self.beta = codebook_size, emb_dim, beta
        self.embedding = nn.Embedding(codebook_size, emb_dim)
        self.embedding.weight.data.uniform_(-1.0 / codebook_size, 1.0 / codebook_size)

    def _pack(self, device, dtype):
        w = self.embedding.weight.detach().float()
        self.book = w.to(device).contiguous()                                  # fp32 (K, D) for the gather
        self.book_t = w.to(device=device, dtype=dtype).contiguous()            # distance GEMM operand
        self.enorm = w.pow(2.0).sum(1).to(device).contiguous()

    @torch.no_grad()
    def forward_nhwc(self, z):
        """z (B,H,W,C) in the module dtype -> (z_q (B,H,W,C), loss fp32 device scalar, indices int32 (B*H*W,))."""
        b, h, w, c = z.shape
        z2 = z.reshape(b * h * w, c)
        if z2.dtype == torch.bfloat16 and c in (64, 128, 256, 512):
            idx = ops.rq_nearest(z2, self.book_t, ops.row_sumsq(z2), self.enorm)
        else:
            idx = ops.rq_argmin(ops.linear(z2, self.book_t, None, out_f32=True), ops.row_sumsq(z2), self.enorm)
        zq = ops.embed_rows(self.book, idx, z.dtype)
        loss = ops.commit_loss(z2, zq, scale=1.0 + self.beta)
        return ops.straight_through(z2, zq).reshape(b, h, w, c), loss, idx

    @torch.no_grad()
    def get_codebook_feat(self, indices, shape):
        """indices -> (B,H,W,C) code vectors (reference :86-98, channels-last here)."""
        zq = ops.embed_rows(self.book, indices.reshape(-1).to(torch.int32), self.dt)
        return zq.reshape(shape) if shape is not None else zq
