"""Stage-I temporal RQ-VAE (`TDCRQVAE3`), HIP-backed.

Mirror of the reference archs/tdcrqvae3_arch.py surface that the inference path touches: same class
names, constructor kwargs and state-dict keys (Encoder :460, Decoder :577, VQEmbedding :80,
RQBottleneck :206, TDCRQVAE3 :711).  Activations are channels-last (B*T, H, W, C); `forward` takes the
reference's (B*T, 3, H, W) fp32 tensor (or uint8 (B*T,H,W,3) frames) and returns the reference's
tuple.  The training-side quantiser - the EMA codebook update of VQEmbedding with its collectives (:128-199) - is
`VQEmbedding.forward` in training mode (fp32): statistics and update are HIP kernels (csrc/vq_ema.hip), the two
all-reduces of the reference travel as ONE buffer over RCCL.
"""
import functools
import inspect
import json
import os

import torch
import torch.nn as nn

from .. import ops
from ..modules.rstt_layers import (Conv2d, EncoderLayer, HipModule, Normalize, TDResnetBlock, _defect_t, _exact, _is_x3, _pack_matrix,
                                    _wants_wcomp, mark_exact_weights, mark_uncompensated, prepare_tree)
from ..ops import ACT_SILU, X3
from ..config import DEFAULT_PRECISION
from ..registry import ARCH_REGISTRY


class Upsample(HipModule):
    """nearest x2 + conv3x3 (reference: :34-52).  fp32 (parity) mode: one implicit-GEMM launch with the x2 resize folded
    into the gather.  bf16 modes: the four sub-pixel convolutions - output pixel (2y+py, 2x+px) only sees a 2x2
    neighbourhood of the low-resolution map, with the 3x3 taps that land on the same source pixel summed
    (rows: py=0 -> {y-1: k0, y: k1+k2}, py=1 -> {y: k0+k1, y+1: k2}; same for columns) - 4/9 of the FLOPs, identical
    mathematically (the sums are formed in fp32 before the cast to bf16)."""
    _ROWS = (((0,), (1, 2)), ((0, 1), (2,)))   # _ROWS[parity][tap a] = 3x3 taps merged into 2x2 tap a

    def __init__(self, in_channels, with_conv):
        super().__init__()
        assert with_conv
        self.conv = Conv2d(in_channels, in_channels, 3, padding=1)
        self.sub_w = None

    def _pack(self, device, dtype):
        self.sub_w = None
        if dtype == torch.float32:
            return
        w = self.conv.weight.detach().float()                       # (Cout, Cin, 3, 3)
        self.sub_w, self.sub_def, self._def4, self.sub_w2 = {}, {}, None, {}
        exact = _exact(self, dtype, w.shape[1]) and w.shape[0] % 8 == 0      # exact-weight stage: two-plane sub-pixel filters
        for py in (0, 1):
            for px in (0, 1):
                w2 = torch.stack([torch.stack([sum(w[:, :, ky, kx] for ky in self._ROWS[py][a] for kx in self._ROWS[px][b])
                                               for b in (0, 1)], -1) for a in (0, 1)], -2)      # (Cout, Cin, 2, 2)
                self.sub_w[(py, px)] = _pack_matrix(w2, device, dtype)
                self.sub_def[(py, px)] = _defect_t(w2, self.sub_w[(py, px)]) if _wants_wcomp(dtype, self) else None
                if exact:
                    self.sub_w2[(py, px)] = _pack_matrix(w2, device, dtype, w2=True)

    def forward(self, x):
        if self.sub_w is None:
            return self.conv.run(x, ups=True, gn=32)
        n, h, w, _ = x.shape
        cout = self.conv.out_channels
        out = torch.empty((n, 2 * h, 2 * w, cout), device=x.device, dtype=x.dtype)
        # a TDResnetBlock's GroupNorm follows: the four launches share one statistics workspace (4 sub-ranges)
        st = ops.GnStats(n, 4, h * w, cout, 32, x.device) if (ops.USE_EPILOGUE_GN and ops.gn_ok(n, h * w, cout)) else None
        mean = fb = None
        if self.sub_w2 and x.dtype == torch.float16 and x.data_ptr() % 16 == 0 and ops._ld_img(x) % 8 == 0:
            for i, ((py, px), w2) in enumerate(self.sub_w2.items()):      # exact weights: nothing to compensate
                ops.conv2d(x, w2, self.conv.pb, kh=2, kw=2, pad=(1 - py, py, 1 - px, px), out=out, out_parity=(py, px),
                           gn=None if st is None else (st, i), w2=cout)
            return out if st is None else st.bind(out, cout)
        if self.sub_def[(0, 0)] is not None and (h * w) % 512 == 0:
            if ops.USE_FRAME_BIAS and x.dtype in (torch.float16, torch.bfloat16):
                # the four sub-pixel convolutions read ONE operand: their defects side by side, one launch, (4, N, Cout) biases
                if getattr(self, "_def4", None) is None:
                    self._def4 = torch.cat([self.sub_def[k] for k in self.sub_w], 1).contiguous()
                    self._pb4 = None if self.conv.pb is None else self.conv.pb.repeat(4).contiguous()
                xb, nb = ops.banded(x)                          # (bands of the INPUT image = bands of every sub-pixel output)
                fb = ops.frame_bias(xb, self._def4, self._pb4, groups=4, scale_div=nb, sample_cells=ops.band_sample_cells(nb))
            else:
                mean = ops.sampled_channel_mean(x)
        for i, ((py, px), w2) in enumerate(self.sub_w.items()):
            b = fb[i] if fb is not None else (self.conv.pb if mean is None else ops.mean_field_bias(mean, self.sub_def[(py, px)], self.conv.pb))
            ops.conv2d(x, w2, b, kh=2, kw=2, pad=(1 - py, py, 1 - px, px), out=out, out_parity=(py, px),
                       gn=None if st is None else (st, i))
        return out if st is None else st.bind(out, cout)


class Downsample(HipModule):
    """pad (0,1,0,1) + conv3x3 stride 2 (reference: :55-76); the pad is the gather's bounds check."""

    def __init__(self, in_channels, with_conv):
        super().__init__()
        assert with_conv
        self.conv = Conv2d(in_channels, in_channels, 3, stride=2, padding=0, pad4=(0, 1, 0, 1))

    def forward(self, x):
        return self.conv.run(x, gn=32)      # a TDResnetBlock (GroupNorm first) follows every Downsample


class VQEmbedding(nn.Embedding, HipModule):
    """Codebook with EMA update (reference: :80-203).  Inference: nearest-code search + gather.  Training (`.train()`,
    fp32 modules, after `prepare`): `forward` also folds the batch into the EMA buffers and renormalises the codebook, in the
    reference's order - search with the current codebook, batch statistics, gather with the CURRENT codebook, then the
    update (:188-199).  The state lives on the device (`book`, `cs_ema_d`, `embed_ema_d`); `state_dict()` copies it back
    into the nn parameters / buffers first, so checkpoints see the trained values."""

    def __init__(self, n_embed, embed_dim, ema=True, decay=0.99, restart_unused_codes=True, eps=1e-5):
        nn.Embedding.__init__(self, n_embed + 1, embed_dim, padding_idx=n_embed)
        self.n_embed = n_embed
        self.ema, self.decay, self.eps, self.restart_unused_codes = ema, decay, eps, restart_unused_codes
        if ema:
            for p in self.parameters():
                p.requires_grad_(False)
        self.register_buffer("cluster_size_ema", torch.zeros(n_embed))
        self.register_buffer("embed_ema", self.weight[:-1, :].detach().clone())
        nn.Module.eval(self)      # inference-first build: the EMA path needs an explicit .train() (on this module or a parent)

    def _pack(self, device, dtype):
        w = self.weight.detach().float()
        self.book = w.to(device).contiguous()                      # fp32 (K+1, D) for the gather
        self.book_t = w[:-1].to(device=device, dtype=dtype).contiguous()   # (K, D) distance GEMM operand
        self.enorm = w[:-1].pow(2.0).sum(1).to(device).contiguous()  # |e_j|^2 (reference :111)
        self.cs_ema_d = self.cluster_size_ema.detach().float().to(device).contiguous()
        self.embed_ema_d = self.embed_ema.detach().float().to(device).contiguous()

    def distances_dot(self, x2d):
        """(rows, K) fp32 dot products x.e_j and |x|^2 (the two terms compute_distances combines, reference :100-117)."""
        return ops.linear(x2d, self.book_t, None, out_f32=True), ops.row_sumsq(x2d)

    def find_nearest_embedding(self, x2d):
        """x2d (rows, D) -> int32 codes: argmin_j |x|^2 + |e_j|^2 - 2 x.e_j (reference: :100-126).  bf16: one kernel with
        the arg-min inside the distance GEMM (no rows x K matrix in HBM); fp32: distance GEMM + row arg-min."""
        if x2d.dtype == torch.bfloat16 and x2d.shape[1] in (64, 128, 256, 512):
            return ops.rq_nearest(x2d, self.book_t, ops.row_sumsq(x2d), self.enorm)
        dot, xn = self.distances_dot(x2d)
        return ops.rq_argmin(dot, xn, self.enorm)

    # ---- training side (reference :128-186) ------------------------------------------------------------------------
    def _tile_with_noise(self, x, target_n, noise=None):
        """repeat the batch up to target_n rows and add U[0,1) * 0.01 / sqrt(D) (reference :128-136)."""
        b, d = x.shape
        n_rep = (target_n + b - 1) // b
        x = x.repeat(n_rep, 1)
        if noise is None:
            noise = torch.rand_like(x)
        return x + noise.to(x.device) * (0.01 / float(d) ** 0.5)

    @torch.no_grad()
    def batch_statistics(self, vectors, idxs, perm=None, noise=None):
        """First half of _update_buffers (:138-158, :163-171): this batch's per-code vector sums and counts as one flat
        buffer, all-reduced over the data-parallel group in ONE collective (the reference issues two), and the K candidate
        restart vectors (a random subset of the batch, broadcast from rank 0).  perm / noise: fixed draws for tests."""
        import torch.distributed as dist

        k, d = self.n_embed, self.weight.shape[1]
        x = vectors.reshape(-1, d)
        if x.dtype != torch.float32:
            raise TypeError("the EMA codebook update runs in fp32 (prepare(device, torch.float32))")
        x = x.contiguous()
        stats = ops.vq_cluster_stats(x, idxs.reshape(-1).to(torch.int32).contiguous(), k)
        multi = dist.is_available() and dist.is_initialized()
        if multi:
            dist.all_reduce(stats, op=dist.ReduceOp.SUM)
        restart = None
        if self.restart_unused_codes:
            if x.shape[0] < k:
                x = self._tile_with_noise(x, k, noise)
            if perm is None:
                perm = torch.randperm(x.shape[0], device=x.device)
            sel = perm[:k].to(device=x.device, dtype=torch.int32).contiguous()
            restart = ops.embed_rows(x, sel, torch.float32)        # row gather: the batch is the "codebook"
            if multi:
                dist.broadcast(restart, 0)
        return stats, restart

    @torch.no_grad()
    def apply_ema(self, stats, restart):
        """Second half of _update_buffers + _update_embedding (:160-186): EMA of counts and sums, restart of codes whose
        EMA count is below 1, codebook = embed_ema / normalised count; then the search operands follow the new codebook."""
        k = self.n_embed
        ops.vq_ema_update(self.cs_ema_d, self.embed_ema_d, stats, restart, self.book, self.decay, self.eps)
        new = self.book[:k]
        self.book_t = new if self.book_t.dtype == torch.float32 else new.to(self.book_t.dtype)
        self.enorm = ops.row_sumsq(new)

    def forward(self, inputs, perm=None, noise=None):
        """inputs (..., D) -> (embeds (..., D), codes int32 (...)) (reference :188-199)."""
        d = inputs.shape[-1]
        x2d = inputs.reshape(-1, d)
        idxs = self.find_nearest_embedding(x2d)
        pend = self.batch_statistics(x2d, idxs, perm, noise) if (self.training and self.ema) else None
        embeds = ops.embed_rows(self.book, idxs, inputs.dtype)
        if pend is not None:
            self.apply_ema(*pend)
        return embeds.reshape(inputs.shape), idxs.reshape(inputs.shape[:-1])

    def sync_host(self):
        """device state -> nn parameters / buffers (checkpointing)"""
        if hasattr(self, "book"):
            with torch.no_grad():
                self.weight.copy_(self.book.detach().cpu())
                self.cluster_size_ema.copy_(self.cs_ema_d.detach().cpu())
                self.embed_ema.copy_(self.embed_ema_d.detach().cpu())

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        self.sync_host()
        super()._save_to_state_dict(destination, prefix, keep_vars)


class RQBottleneck(HipModule):
    """Residual quantiser (reference: :206-368).  In training mode the codebooks fold every batch into their EMA buffers
    (VQEmbedding.forward :188-199) - statistics are taken on the residual the search saw, before it is updated."""

    def __init__(self, latent_shape, code_shape, n_embed, decay=0.99, shared_codebook=False,
                 restart_unused_codes=True, commitment_loss="cumsum"):
        super().__init__()
        assert len(code_shape) == len(latent_shape) == 3
        assert all(l % c == 0 for c, l in zip(code_shape[:2], latent_shape[:2]))
        self.latent_shape, self.code_shape = torch.Size(latent_shape), torch.Size(code_shape)
        self.shape_divisor = torch.Size([latent_shape[i] // code_shape[i] for i in range(3)])
        assert self.shape_divisor[0] == 1 and self.shape_divisor[1] == 1, "spatial folding unused by PGTFormer"
        depth = code_shape[-1]
        embed_dim = latent_shape[2]
        self.shared_codebook = shared_codebook
        self.n_embed = list(n_embed) if isinstance(n_embed, (list, tuple)) else [n_embed] * depth
        if shared_codebook:
            book = VQEmbedding(self.n_embed[0], embed_dim)
            self.codebooks = nn.ModuleList([book for _ in range(depth)])
        else:
            self.codebooks = nn.ModuleList([VQEmbedding(self.n_embed[i], embed_dim) for i in range(depth)])

    def quantize(self, x):
        """x (B,h,w,D) -> (aggregated quant (B,h,w,D), codes int32 (B,h,w,d)) (reference: :294-328)."""
        b, h, w, d = x.shape
        depth = self.code_shape[-1]
        rows = b * h * w
        x2 = x.reshape(rows, d)
        resid = x2 if depth == 1 else x2.clone()
        agg = torch.empty((rows, d), device=x.device, dtype=x.dtype)
        codes = []
        for i in range(depth):
            book = self.codebooks[i]
            c = book.find_nearest_embedding(resid)
            pend = book.batch_statistics(resid, c) if (book.training and book.ema) else None   # EMA update (:188-199)
            ops.embed_rows(book.book, c, x.dtype, out=agg, accumulate=i > 0, resid=resid if depth > 1 else None)
            if pend is not None:
                book.apply_ema(*pend)
            codes.append(c)
        return agg.reshape(b, h, w, d), torch.stack(codes, -1).reshape(b, h, w, depth)

    def forward(self, x):
        """x (B,h,w,D) -> (quants (B,h,w,D), commitment loss fp32 device scalar, codes int32 (B,h,w,d)) as
        RQBottleneck.forward in eval mode (reference :330-352): the loss is the mean over depths of
        mean((x - aggregated_quant_i)^2); the returned features are x + (quant - x) (the straight-through value)."""
        b, h, w, d = x.shape
        depth = self.code_shape[-1]
        rows = b * h * w
        x2 = x.reshape(rows, d)
        resid = x2 if depth == 1 else x2.clone()
        agg = torch.empty((rows, d), device=x.device, dtype=x.dtype)
        codes, loss = [], None
        for i in range(depth):
            book = self.codebooks[i]
            c = book.find_nearest_embedding(resid)
            pend = book.batch_statistics(resid, c) if (book.training and book.ema) else None   # EMA update (:188-199)
            ops.embed_rows(book.book, c, x.dtype, out=agg, accumulate=i > 0, resid=resid if depth > 1 else None)
            if pend is not None:
                book.apply_ema(*pend)
            loss = ops.commit_loss(x2, agg, out=loss, scale=1.0 / depth)      # mean over depths of mean((x - agg_i)^2)
            codes.append(c)
        quants = ops.straight_through(x2, agg)
        return quants.reshape(b, h, w, d), loss, torch.stack(codes, -1).reshape(b, h, w, depth)

    def get_soft_codes(self, x, temp=1.0, stochastic=False, generator=None):
        """soft codes softmax(-dist / temp) (B,h,w,d,K) fp32 and hard codes (B,h,w,d) (reference :429-457).  stochastic: the
        codes are drawn from the soft codes (the reference's torch.multinomial(soft_code, 1), :443-446) - one categorical
        draw per token by inverse CDF with uniforms from `generator` (a device torch.Generator, or None: the default one);
        same distribution, this build's random stream.  The residual carries the DRAWN code's embedding, as the reference's."""
        b, h, w, d = x.shape
        depth = self.code_shape[-1]
        rows = b * h * w
        resid = x.reshape(rows, d).clone()
        softs, codes = [], []
        for i in range(depth):
            book = self.codebooks[i]
            dot, xn = book.distances_dot(resid)
            soft, c = ops.rq_soft_codes(dot, xn, book.enorm, temp)
            if stochastic:
                u = torch.rand((rows,), device=soft.device, dtype=torch.float32, generator=generator)
                c = ops.sample_rows(soft, u)
            if i + 1 < depth:
                scratch = torch.empty_like(resid)
                ops.embed_rows(book.book, c, x.dtype, out=scratch, resid=resid)
            softs.append(soft)
            codes.append(c)
        k = softs[0].shape[1]
        return (torch.stack(softs, 1).reshape(b, h, w, depth, k), torch.stack(codes, -1).reshape(b, h, w, depth))

    def embed_code(self, code, dtype=torch.float32):
        """codes (B,h,w,d) int -> summed embeddings (B,h,w,D) (reference: :355-368)."""
        assert tuple(code.shape[1:]) == tuple(self.code_shape)
        b, h, w, depth = code.shape
        code = code.to(torch.int32)
        out = None
        for i in range(depth):
            ci = code[..., i].contiguous().reshape(-1)
            out = ops.embed_rows(self.codebooks[i].book, ci, dtype, out=out, accumulate=i > 0)
        return out.reshape(b, h, w, -1)


USE_CONV_IN_SPLIT = os.environ.get("PGT_CONV_IN_SPLIT", "1") != "0"     # A/B switch: split planes straight from the fp32 first conv


class Encoder(HipModule):
    def __init__(self, *, ch, out_ch, ch_mult=(1, 2, 4, 8), depths, num_res_blocks, num_heads, num_frames,
                 window_sizes, attn_resolutions, dropout=0.0, resamp_with_conv=True, in_channels, resolution,
                 z_channels, double_z=True, **ignore_kwargs):
        super().__init__()
        self.ch, self.num_resolutions, self.num_res_blocks = ch, len(ch_mult), num_res_blocks
        self.resolution, self.in_channels = resolution, in_channels
        self.conv_in = Conv2d(in_channels, ch, 3, padding=1, cin_pad="chunk")
        curr_res = resolution
        in_ch_mult = (1,) + tuple(ch_mult)
        self.down = nn.ModuleList()
        block_in = ch
        for i_level in range(self.num_resolutions):
            block, attn = nn.ModuleList(), nn.ModuleList()
            block_in = ch * in_ch_mult[i_level]
            block_out = ch * ch_mult[i_level]
            for _ in range(num_res_blocks):
                block.append(TDResnetBlock(in_channels=block_in, out_channels=block_out))
                block_in = block_out
                if curr_res in attn_resolutions:
                    attn.append(EncoderLayer(block_in, depths[i_level], num_heads=num_heads[i_level],
                                             num_frames=num_frames, window_size=window_sizes[i_level], mlp_ratio=1))
            down = nn.Module()
            down.block, down.attn = block, attn
            if i_level != self.num_resolutions - 1:
                down.downsample = Downsample(block_in, resamp_with_conv)
                curr_res //= 2
            self.down.append(down)
        self.mid = nn.Module()
        self.mid.block_1 = TDResnetBlock(in_channels=block_in, out_channels=block_in)
        self.mid.attn_1 = EncoderLayer(block_in, depths[-1], num_heads=num_heads[-1], num_frames=num_frames,
                                       window_size=window_sizes[-1], mlp_ratio=1)
        self.mid.block_2 = TDResnetBlock(in_channels=block_in, out_channels=block_in)
        self.norm_out = Normalize(block_in)
        self.conv_out = Conv2d(block_in, 2 * z_channels if double_z else z_channels, 3, padding=1)

    def first_attn_level(self):
        for i, lvl in enumerate(self.down):
            if len(lvl.attn) > 0:
                return i
        return self.num_resolutions

    def prepare_split(self, device):
        """bf16x3 mode: conv_in (3 input channels: no 64-channel K blocks for the LDS-DMA kernel) stays in exact fp32;
        every level, the middle blocks and conv_out run on split-half operands (the levels below the first temporal
        attention are computed once per FRAME by the overlap-aware driver)."""
        prepare_tree(self.conv_in, device, torch.float32)
        for m in (self.down, self.mid, self.norm_out, self.conv_out):
            prepare_tree(m, device, X3)
        self.dev, self.dt = device, X3

    @staticmethod
    def _convert(h, cur, want):
        if _is_x3(want) and not _is_x3(cur):
            assert cur == torch.float32, "split levels follow fp32 levels"
            return ops.to_x3(h)
        assert _is_x3(cur) == _is_x3(want) and (_is_x3(cur) or cur == want), (cur, want)
        return h

    def forward(self, x, return_multi_res_feats=False, feat_out=None, win=None, want_feats=None, feat_dtype=None):
        """x: (F, H, W, ops.input_channels(dtype)) channel-padded input (reference: :540-573).  feat_out: {level: (B*T,h,w,C) view} - the
        level's feature map is delivered in that view (a channel slice of the decoder-side concat buffer): written
        in place by the producing kernel when the dtypes and frame order allow it, else copied / gathered into it.

        win: None, or int32 device tensor (B*T,) of frame indices into x: the B windows of a batch are given as their
        UNIQUE frames (consecutive windows of the driver share 2 of 3 frames, reference inference.py:47-74).  Everything
        up to the first temporal attention is per-frame (conv_in, the res blocks and down-samplings of the levels without
        attention, and the first res block of the first level with attention: reference :546-555 with
        attn_resolutions starting at 128), so it runs once per frame and is gathered to window order (B*T frames) where
        the first EncoderLayer starts.  want_feats: levels whose feature maps the caller needs (None: all); the others
        are returned as None (a per-frame 512x512 map would otherwise be gathered for nothing).
        Feature maps of split levels are returned as their hi planes (half views: the value rounded to IEEE half), cast to
        feat_dtype / the dtype of the destination where that differs (the bf16 decoder of "bf16x3")."""
        feats = []
        cur = self.conv_in.dt
        # (the level-0 block starts with a GroupNorm: statistics from conv_in's epilogue unless a dtype conversion intervenes)
        if cur == torch.float32 and _is_x3(self.down[0].block[0].dt) and USE_CONV_IN_SPLIT:
            # exact-fp32 conv_in feeding a split-half level: the kernel stores the split planes itself (pgt_conv_desc::out_split)
            # - no fp32 tensor, no conversion pass over the largest activation of the model - and leaves the statistics too
            h = self.conv_in.run(x, gn=32, out_x3=True)
            cur = X3
        else:
            h = self.conv_in.run(x, gn=32 if self.down[0].block[0].dt == cur else None)
        per_frame = win is not None
        for i_level in range(self.num_resolutions):
            lvl = self.down[i_level]
            dst = None if feat_out is None else feat_out.get(i_level)
            has_attn = len(lvl.attn) > 0
            h = self._convert(h, cur, lvl.block[0].dt)
            cur = lvl.block[0].dt
            in_place = dst is not None and not _is_x3(cur) and dst.dtype == cur     # the producer can write dst itself
            top = i_level == self.num_resolutions - 1          # no Downsample after it: mid.block_1 (GroupNorm) follows
            for i_block in range(self.num_res_blocks):
                last = in_place and i_block == self.num_res_blocks - 1
                fin = i_block == self.num_res_blocks - 1
                gn_after = not fin or top                        # next op is a TDResnetBlock (another block / mid.block_1)
                h = lvl.block[i_block](h, out=dst if last and not has_attn and not per_frame else None,
                                       gn_next=gn_after and not has_attn and not per_frame)
                if has_attn:
                    if per_frame:
                        h = ops.gather_frames(h, win)
                        per_frame = False
                    h = lvl.attn[i_block](h, out=dst if last else None, gn_next=gn_after)
            wanted = return_multi_res_feats and (want_feats is None or i_level in want_feats)
            if not wanted:
                feats.append(None)
            else:
                # a split map is handed over as its hi plane: the value rounded to IEEE half (a strided view, no pass)
                f = h if not _is_x3(cur) else h[..., :h.shape[-1] // 2]
                want_dt = dst.dtype if dst is not None else (feat_dtype if (feat_dtype is not None and _is_x3(cur)) else f.dtype)
                if per_frame:     # per-frame level: cast on the unique frames, then gather to window order
                    f = ops.gather_frames(ops.cast(f, want_dt), win, out=dst)
                elif dst is not None and f.data_ptr() != dst.data_ptr():
                    f = ops.copy_into(f, dst)
                elif f.dtype != want_dt:
                    f = ops.cast(f, want_dt)
                feats.append(f)
            if i_level != self.num_resolutions - 1:
                h = lvl.downsample(h)
        if per_frame:
            h = ops.gather_frames(h, win)
        h = self._convert(h, cur, self.mid.block_1.dt)
        h = self.mid.block_2(self.mid.attn_1(self.mid.block_1(h), gn_next=True), gn_next=True)
        h = self.conv_out.run(self.norm_out.run(h, ACT_SILU))
        return (h, feats) if return_multi_res_feats else h


class Decoder(HipModule):
    def __init__(self, *, ch, out_ch, ch_mult=(1, 2, 4, 8), depths, num_res_blocks, num_heads, num_frames,
                 window_sizes, attn_resolutions, dropout=0.0, resamp_with_conv=True, in_channels, resolution,
                 z_channels, give_pre_end=False, **ignorekwargs):
        super().__init__()
        self.ch, self.num_frames = ch, num_frames
        self.num_resolutions, self.num_res_blocks = len(ch_mult), num_res_blocks
        self.resolution, self.give_pre_end = resolution, give_pre_end
        block_in = ch * ch_mult[self.num_resolutions - 1]
        curr_res = resolution // 2 ** (self.num_resolutions - 1)
        self.z_shape = (1, z_channels, curr_res, curr_res)
        self.conv_in = Conv2d(z_channels, block_in, 3, padding=1)
        self.mid = nn.Module()
        self.mid.block_1 = TDResnetBlock(in_channels=block_in, out_channels=block_in)
        self.mid.attn_1 = EncoderLayer(block_in, depths[-1], num_heads=num_heads[-1], num_frames=num_frames,
                                       window_size=window_sizes[-1], mlp_ratio=1)
        self.mid.block_2 = TDResnetBlock(in_channels=block_in, out_channels=block_in)
        self.up = nn.ModuleList()
        for i_level in reversed(range(self.num_resolutions)):
            block, attn = nn.ModuleList(), nn.ModuleList()
            block_out = ch * ch_mult[i_level]
            for _ in range(num_res_blocks + 1):
                block.append(TDResnetBlock(in_channels=block_in, out_channels=block_out))
                block_in = block_out
                if curr_res in attn_resolutions:
                    attn.append(EncoderLayer(block_in, depths[i_level], num_heads=num_heads[i_level],
                                             num_frames=num_frames, window_size=window_sizes[i_level], mlp_ratio=1))
            up = nn.Module()
            up.block, up.attn = block, attn
            if i_level != 0:
                up.upsample = Upsample(block_in, resamp_with_conv)
                curr_res *= 2
            self.up.insert(0, up)
        self.norm_out = Normalize(block_in)
        self.conv_out = Conv2d(block_in, out_ch, 3, padding=1)

    def forward(self, z, fuse=None, fuse_dst=None, mid=None):
        """z: (B*T, h, w, z_channels) (reference: :672-707; with `fuse`, the loop inlined in
        PGTFormer.forward, archs/pgtformer_arch.py:684-710). fuse(f_size:str, h) -> h.
        fuse_dst(f_size:str) -> (B*T,h,w,C) view or None: where the level's last block writes the feature map that
        goes into `fuse` (the `dec` slice of the fusion block's concat buffer: no copy later).
        mid = (k, size, kind): only frame k of every window is wanted; `size`/`kind` name the last temporal operation
        ("fuse": the fusion block at that size narrows to B frames itself; "attn": narrowed here after that level's
        EncoderLayers; "start": no temporal operation at all).  Everything after it runs on B frames."""
        self.last_z_shape = z.shape

        def narrow(hh):      # frame mid[0] of every window (index tensor cached: built once, before any graph capture)
            key = (hh.shape[0], mid[0], str(hh.device))
            cache = self.__dict__.setdefault("_fidx", {})
            if key not in cache:
                cache[key] = (torch.arange(hh.shape[0] // self.num_frames, dtype=torch.int32) * self.num_frames + mid[0]).to(hh.device)
            return ops.gather_frames(hh, cache[key])
        if mid is not None and mid[2] == "start":
            z = narrow(z)
        h = self.conv_in.run(z, gn=32)
        h = self.mid.block_2(self.mid.attn_1(self.mid.block_1(h), gn_next=True), gn_next=True)
        for i_level in reversed(range(self.num_resolutions)):
            lvl = self.up[i_level]
            dst = None if fuse_dst is None else fuse_dst(str(h.shape[2]))
            for i_block in range(self.num_res_blocks + 1):
                fin = i_block == self.num_res_blocks
                last = dst is not None and fin
                has_attn = len(lvl.attn) > 0
                # a GroupNorm follows unless this is the level's last block (then: fusion / up-sampling conv), except at the
                # full-resolution level whose last block feeds norm_out
                gn_after = (not fin) or (i_level == 0 and not self.give_pre_end)
                h = lvl.block[i_block](h, out=dst if last and not has_attn else None, gn_next=gn_after and not has_attn)
                if has_attn:
                    h = lvl.attn[i_block](h, out=dst if last else None, gn_next=gn_after)
            if mid is not None and mid[1:] == (str(h.shape[2]), "attn"):
                h = narrow(h)
            if fuse is not None:
                h = fuse(str(h.shape[2]), h)
            if i_level != 0:
                h = lvl.upsample(h)
        if self.give_pre_end:
            return h
        # the restored frames leave the last conv in fp32 whatever the decoder's storage type (3 channels: free), so the
        # 16-bit modes do not add an output rounding (2^-12 of [0, 1] in half) on top of their arithmetic
        return self.conv_out.run(h, affine_in=self.norm_out.coeffs(h, ACT_SILU), out_f32=True)


class HubMixin:
    """`from_pretrained` / `save_pretrained` with the file layout of huggingface_hub's PyTorchModelHubMixin, which the
    reference mixes into TDCRQVAE3 (archs/tdcrqvae3_arch.py:711; call site inference.py:118): `config.json` holds the
    constructor kwargs, `model.safetensors` the state dict, loaded strictly.  The constructor arguments of the
    outermost class are recorded at construction time (what the mixin serialises as config.json)."""

    def __init_subclass__(cls, **kw):
        super().__init_subclass__(**kw)
        orig = cls.__dict__.get("__init__")
        if orig is None:
            return

        @functools.wraps(orig)
        def init(self, *a, **k):
            if "_hub_config" not in self.__dict__:
                bound = inspect.signature(orig).bind(self, *a, **k)
                bound.apply_defaults()      # the reference's mixin stores defaulted constructor arguments too
                cfg = {}
                for name, val in list(bound.arguments.items())[1:]:
                    if inspect.signature(orig).parameters[name].kind is inspect.Parameter.VAR_KEYWORD:
                        cfg.update(val)
                    else:
                        cfg[name] = val
                object.__setattr__(self, "_hub_config", cfg)
            orig(self, *a, **k)

        cls.__init__ = init

    def save_pretrained(self, save_directory):
        from safetensors.torch import save_file
        os.makedirs(save_directory, exist_ok=True)
        with open(os.path.join(save_directory, "config.json"), "w") as f:
            json.dump(self._hub_config, f, indent=2, default=list)
        # shared_codebook registers one VQEmbedding under several names: safetensors wants each storage once
        sd, seen = {}, {}
        for k, v in self.state_dict().items():
            v = v.detach().cpu().contiguous()
            sd[k] = v.clone() if v.data_ptr() in seen else v
            seen[v.data_ptr()] = k
        save_file(sd, os.path.join(save_directory, "model.safetensors"))
        return save_directory

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, device=None, precision=DEFAULT_PRECISION, **model_kwargs):
        """`PGTFormer.from_pretrained("kepeng/pgtformer-base")` of the reference (inference.py:118).  A local directory
        holding config.json + model.safetensors is loaded directly; a hub id is resolved through huggingface_hub (needs
        network access or a populated cache).  With `device` the model is also prepared (weights repacked for the
        kernels) and ready to run; without it the caller calls .prepare(device, precision)."""
        path = str(pretrained_model_name_or_path)
        if not os.path.isdir(path):
            try:
                from huggingface_hub import snapshot_download
                path = snapshot_download(path, allow_patterns=["config.json", "model.safetensors"])
            except Exception as e:  # no network / unknown id
                raise FileNotFoundError(f"{pretrained_model_name_or_path!r} is not a local directory and could not be "
                                        f"fetched from the hub: {e}") from e
        from safetensors.torch import load_file
        with open(os.path.join(path, "config.json")) as f:
            cfg = json.load(f)
        cfg.update(model_kwargs)
        model = cls(**cfg)
        model.load_state_dict(load_file(os.path.join(path, "model.safetensors")), strict=True)
        nn.Module.eval(model)
        model.requires_grad_(False)
        return model.prepare(device, precision) if device is not None else model


@ARCH_REGISTRY.register()
class TDCRQVAE3(HubMixin, HipModule):
    """Stage-I temporal RQ-VAE (reference: :711-872). `prepare(device, precision)` must be called after
    weights are loaded (precision modes: see `prepare`)."""

    def __init__(self, *, embed_dim=64, n_embed=512, decay=0.99, loss_type="mse", latent_loss_weight=0.25,
                 bottleneck_type="rq", ddconfig=None, checkpointing=False, tf=3, **kwargs):
        super().__init__()
        assert loss_type in ("mse", "l1")
        assert bottleneck_type == "rq", "invalid 'bottleneck_type' (must be 'rq')"
        self.encoder = Encoder(**ddconfig)
        self.decoder = Decoder(**ddconfig)
        self.t = tf
        self.quantizer = RQBottleneck(latent_shape=kwargs["latent_shape"], code_shape=kwargs["code_shape"],
                                      n_embed=n_embed, decay=decay, shared_codebook=kwargs["shared_codebook"],
                                      restart_unused_codes=kwargs["restart_unused_codes"])
        self.code_shape = kwargs["code_shape"]
        self.quant_conv = Conv2d(ddconfig["z_channels"], embed_dim, 1)
        self.post_quant_conv = Conv2d(embed_dim, ddconfig["z_channels"], 1)
        self.loss_type, self.latent_loss_weight = loss_type, latent_loss_weight
        self.enc_dt = self.dec_dt = None

    # -- precision / weight repack ------------------------------------------------------------
    ENC_SIDE = ("encoder", "quant_conv", "quantizer", "conditionnet", "convpos", "feat_emb", "ft_layers", "idx_pred_layer")
    F32_IN_X3 = ("conditionnet", "convpos", "quantizer")    # bf16x3 mode: per-frame BiSeNet + codebook stay exact fp32

    def prepare(self, device="cuda", precision=DEFAULT_PRECISION):
        """Repack the weights for the kernels.  precision:
          "fp32"    exact-f32 MFMA everywhere (parity mode)
          "x3f16"   (default) the code-prediction branch on split-half operands (two IEEE-half planes, 3 f16 MFMAs per product, 22 significand
                    bits: the arg-max codes reproduce the fp32 reference) with its per-frame BiSeNet in fp32 storage;
                    decoder / SFT fusion in IEEE half (11 significand bits at the bf16 MFMA rate): restored frames within
                    1e-3 dB PSNR of the fp32 reference at a non-degenerate operating point (tests/golden/make_golden_r3.py).
                    RANGE: every fp32 -> half store saturates at +-65504 (no inf); a checkpoint whose decoder / code-branch
                    activations leave that range is clamped silently, so the first forward is checked - PGTFormer.check_range,
                    run by driver.WindowRunner on the first batch (raises, naming the layers; PGT_RANGE_CHECK=0 disables) -
                    and such a checkpoint is run with "bf16x3"
          "bf16x3"  as x3f16 with a bf16 decoder (8 significand bits: 5e-3 dB at that operating point; no half range limit)
          "mixed"   decoder bf16, the whole code-prediction branch in exact fp32
          "bf16"    bf16 everywhere (fastest; ~2 % of the codes differ from the fp32 reference with random weights)"""
        dts = {"fp32": (torch.float32, torch.float32), "bf16": (torch.bfloat16, torch.bfloat16),
               "mixed": (torch.float32, torch.bfloat16), "bf16x3": (X3, torch.bfloat16), "x3f16": (X3, torch.float16)}
        if precision not in dts:
            raise ValueError(f"precision must be one of {list(dts)}")
        self.enc_dt, self.dec_dt = dts[precision]
        self.precision = precision
        self.dev = torch.device(device)
        # exact-weight stages of the half decoder: marked before the repack; other modes: no layer is marked
        mark_exact_weights(self, False)
        mark_uncompensated(self, False)
        if self.dec_dt == torch.float16:
            for mod in self.exact_weight_modules(ops.EXACT_W_STAGES):
                mark_exact_weights(mod, True)
        if ops.WCOMP_STAGES is not None:      # compensation only in the named decoder stages
            for st in self.DECODER_STAGES:
                if st not in ops.WCOMP_STAGES:
                    for mod in self.exact_weight_modules((st,)):
                        mark_uncompensated(mod, True)
        for name, child in self.named_children():
            if name not in self.ENC_SIDE:
                prepare_tree(child, self.dev, self.dec_dt)
            elif not _is_x3(self.enc_dt):
                prepare_tree(child, self.dev, self.enc_dt)
            elif name == "encoder":
                child.prepare_split(self.dev)
            elif name == "conditionnet":
                child.prepare_x3f(self.dev)
            elif name in self.F32_IN_X3:
                prepare_tree(child, self.dev, torch.float32)
            else:
                prepare_tree(child, self.dev, X3)
        self.in_dt = self.encoder.conv_in.dt          # dtype the input frames are converted to
        self._prepare_extra()
        return self

    def _prepare_extra(self):
        pass

    # decoder stages by the resolution of their feature maps -> parameter-name prefixes (the stages of the weight-rounding ablation,
    # tests/precision_study3.py / profiles/r5_u_third_point_oracle_ablation.md)
    DECODER_STAGES = {"512": ("decoder.up.0.", "decoder.conv_out", "decoder.norm_out"),
                      "256": ("decoder.up.1.", "fuse_convs_dict.256."),
                      "128": ("decoder.up.2.", "fuse_convs_dict.128."),
                      "64": ("decoder.up.3.", "fuse_convs_dict.64."),
                      "32": ("decoder.up.4.", "decoder.mid.", "decoder.conv_in", "fuse_convs_dict.32.")}

    def exact_weight_modules(self, stages):
        """the sub-modules of the decoder stages `stages` (keys of DECODER_STAGES) that exist in this model"""
        out = []
        for st in stages:
            if st not in self.DECODER_STAGES:
                raise ValueError(f"PGT_EXACT_W: unknown decoder stage {st!r} (one of {list(self.DECODER_STAGES)})")
            for prefix in self.DECODER_STAGES[st]:
                try:
                    out.append(self.get_submodule(prefix.rstrip(".")))
                except AttributeError:
                    pass            # (stage-I models have no fusion blocks)
        return out

    def _check_ready(self):
        if self.enc_dt is None:
            raise RuntimeError("call model.prepare(device, precision) after loading weights")

    # -- stage-I API (reference: :760-813) ----------------------------------------------------
    def _ingest(self, x):
        """(B*T,3,H,W) fp32 in [0,1] or uint8 (B*T,H,W,3) -> raw / ImageNet-normalised (B*T,H,W,ops.input_channels(in_dt))."""
        self._check_ready()
        x = x.to(self.dev)
        if x.dim() == 5:                      # the reference's (b, t, c, h, w) clips (tdcrqvae3_arch.py:760-764)
            x = x.reshape(-1, *x.shape[2:])
        return ops.prep_input(x.contiguous(), self.in_dt)

    def encode(self, x):
        raw, _ = self._ingest(x)
        z_e = self.quant_conv.run(self.encoder(raw))   # (B*T,h,w,embed_dim) == reference's NHWC z_e
        return ops.from_x3(z_e) if _is_x3(self.enc_dt) else z_e

    def decode(self, z_q):
        z = self.post_quant_conv.run(ops.cast(z_q, self.dec_dt))
        return ops.nhwc_to_nchw_f32(self.decoder(z))

    @torch.no_grad()
    def forward(self, input, code_only=False):
        """(out | z_q, commitment loss (fp32 device scalar), codes) - reference :760-772 in eval mode."""
        z_e = self.encode(input)
        z_q, quant_loss, codes = self.quantizer(z_e)
        if code_only:
            return z_q, quant_loss, codes.long()
        return self.decode(z_q), quant_loss, codes.long()

    @torch.no_grad()
    def get_codes(self, input):
        return self.quantizer.quantize(self.encode(input))[1].long()

    @torch.no_grad()
    def get_codesbt(self, xs):
        """xs (b,t,c,h,w) (reference :797-802)."""
        b, t, c, h, w = xs.shape
        return self.get_codes(xs.reshape(b * t, c, h, w))

    @torch.no_grad()
    def get_soft_codes(self, xs, temp=1.0, stochastic=False, generator=None):
        soft, code = self.quantizer.get_soft_codes(self.encode(xs), temp=temp, stochastic=stochastic, generator=generator)
        return soft, code.long()

    @torch.no_grad()
    def decode_code(self, code):
        return self.decode(self.quantizer.embed_code(code.to(self.dev), self.dec_dt))

    def get_last_layer(self):
        return self.decoder.conv_out.weight
