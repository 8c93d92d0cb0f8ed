/* libpgt_hip.so — C-ABI of the MI355X-native PGTFormer forward path.
 *
 * Every entry point replaces a group of PyTorch/ATen calls on the reference's inference path
 * (kepengxu/PGTFormer); the reference has no native code, so the "FFI it would bind" is the set of
 * nn.Module forward()s cited per function below (file:line relative to the reference repo).
 *
 * Conventions
 *  - plain pointers and sizes only; all pointers are DEVICE pointers unless noted; the caller owns
 *    every buffer (inputs, outputs, workspaces); the library never allocates or frees on the path.
 *  - activations are channels-last: an image batch is (N, H, W, C) with pixel stride `ld*` elements
 *    (>= C, so channel slices of a wider buffer can be read/written in place); token matrices are
 *    (rows, C) with row stride `ld*`.
 *  - `dtype` selects the activation/weight storage type of the call: PGT_F32 (exact-f32 MFMA, parity
 *    mode) or PGT_BF16 (bf16 MFMA, fp32 accumulate).  Biases, norm gains/shifts, statistics and
 *    relative-position biases are always fp32.
 *  - every function enqueues on `stream` (a hipStream_t) and returns immediately: no internal
 *    synchronisation, HIP-graph capturable.  Return 0 on success, negative errno-style code on
 *    error; the message is available from pgt_last_error() (thread-local).
 */
#ifndef PGT_HIP_H
#define PGT_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef void* pgt_stream_t; /* hipStream_t */

/* PGT_F16X3: split storage on two IEEE-half planes, value = hi + lo (hi = half(v), lo = half(v - hi); conversions saturate at
 * +-65504): a tensor of C logical channels keeps two half planes per pixel / token row, hi at channel offset 0 and lo at an
 * explicit `*_lo` element offset (C for a dense tensor, the parent's width for a channel slice); row strides count 16-bit
 * elements (>= 2C).  Products are formed as hi*hi + lo*hi + hi*lo on the f16 MFMA with fp32 accumulation (22 significand
 * bits instead of 11; below 2^-3 the lo plane is subnormal and the resolution is absolute, 6e-8): the code-prediction branch
 * runs in this type so that the arg-max codes reproduce the fp32 reference (archs/pgtformer_arch.py:638-664) at
 * 16-bit-MFMA speed.  (Rounds 2-3a used two bf16 planes, 16 bits: its logit error of 6e-5 flipped one code in 16 000 at
 * near-ties; the half planes bring it to 1.5e-5) */
/* PGT_F16: IEEE half storage and operands (v_mfma_f32_32x32x16_f16, fp32 accumulate; fp32 -> half stores saturate at
 * +-65504): the decoder-side type of the default precision mode - 11 significand bits on the same MFMA rate as bf16's 8,
 * which is what holds |PSNR(build, GT) - PSNR(reference, GT)| under 1e-3 dB (reference decoder: fp32,
 * archs/pgtformer_arch.py:684-712).  Accepted by conv2d / linear, the norms, channel statistics, window attention,
 * embed_rows, copy2d, frame_to_u8, nhwc_to_nchw. */
enum { PGT_F32 = 0, PGT_BF16 = 1, PGT_F16X3 = 2, PGT_F16 = 3 };
enum { PGT_ACT_NONE = 0, PGT_ACT_RELU = 1, PGT_ACT_GELU = 2, PGT_ACT_SILU = 3, PGT_ACT_LEAKY02 = 4, PGT_ACT_SIGMOID = 5 };
enum { PGT_EPI_PLAIN = 0, PGT_EPI_SFT = 1 };

const char* pgt_version(void);
const char* pgt_last_error(void); /* host pointer, thread-local, valid until the next failing call */

/* ---- convolution / linear (implicit GEMM on MFMA) -------------------------------------------
 * Replaces nn.Conv2d(k=1|3|7, stride 1|2) and nn.Linear on the path: TDResnetBlock convs
 * (modules/rstt_layers.py:875-904), Downsample/Upsample (archs/tdcrqvae3_arch.py:45-76, with the
 * (0,1,0,1) pad and the nearest x2 resize folded in), q/kv/proj/fc1/fc2 (rstt_layers.py:126-234),
 * MHA in/out projections and FFN (archs/codeformer_arch.py:121-137), every conv of BiSeNet with its
 * eval-BatchNorm folded (archs/pgtformer_arch.py:40-379) and of Fuse_sft_block (:460-484).
 * w: (Cout, KH*KW*Cin) row-major, k index = (ky*KW + kx)*Cin + ci, dtype = d->dtype.
 * y[m, co] = epi(act(conv + bias[co])), m = (n*Ho + oy)*Wo + ox:
 *   PGT_EPI_PLAIN:  v = act(.) [+ residual[m*ldr + co]]; if post_relu v = max(v, 0)
 *   PGT_EPI_SFT:    v = dec + sft_w * (dec * act(.) + shift)          (pgtformer_arch.py:478-479)
 */
typedef struct pgt_conv_desc {
    int32_t dtype;
    int32_t N, H, W, Cin; /* stored input geometry                                         */
    int32_t ldx;          /* input pixel stride (elements)                                 */
    int32_t ups;          /* 1: convolve the nearest-x2 up-sampled image (2H x 2W)         */
    int32_t KH, KW, stride, pad_t, pad_l;
    int32_t Ho, Wo, Cout;
    int32_t ldy;
    int32_t act, post_relu;
    int32_t ldr;
    int32_t epi, ld_dec, ld_shift;
    float sft_w;
    int32_t out_f32;            /* 1: store y as fp32 even when dtype is bf16 (logits, distances) */
    int32_t force_bm, force_bn; /* 0 = heuristic; 64|128 pins the workgroup tile (tests, tuning)  */
    int32_t scalar_epilogue;    /* 1: force the element-wise epilogue (A/B tests); 0 = 16-byte path when legal */
    int32_t kernel;             /* 0 = auto; 1 = register-staged v1; 2 = LDS-DMA v2 (bf16, Cin % 64 == 0); 4 = phased v4; 5 = v4 + horizontal tap reuse; 6 = 64-input-channel 3x3 with the weights in registers (bf16 / half; split-half: Cout % 16 == 0); 7 = streaming linear for Cin == 256, Cout % 256 == 0 (bf16 / half, plain epilogue, 16-bit output): weights in registers, rows through LDS; 8 = the layers of 6 at W >= 128 with a ring of row images in LDS (every input row staged once), bias (+ residual) epilogue from registers, optional fused input affine (pgt_conv2d_affine_in) */
    int32_t splitk;             /* 0 = auto (needs a workspace); 1 = never; 2..16 = that many K slices           */
    int32_t stages;             /* unused (was the LDS pipeline depth of the removed kernel = 3)               */
    /* output placement: row index of output pixel m = orow_mul*m + orow_xmul*(m % Wo) + orow_off (orow_mul = 0: dense,
     * row m).  (4, -2, py*2*Wo + px) writes parity (py, px) of a 2x larger map: the four 2x2 sub-pixel convolutions
     * that replace nearest-x2 up-sampling + conv3x3 (archs/tdcrqvae3_arch.py:34-52).  Plain epilogue only (no
     * residual / SFT operands); kernels 1 and 4 (and 0 = auto).                                                  */
    int32_t orow_mul, orow_xmul, orow_off;
    /* dtype == PGT_F16X3: element offsets of the lo planes of x, y and the residual inside a pixel row (0 = Cin / Cout /
     * Cout, i.e. dense [hi | lo] tensors).  w then has 3*KH*KW*Cin columns: per filter tap and per 64-channel
     * block [w_hi | w_hi | w_lo] (64 each), matching the K order [x_hi | x_lo | x_hi] of that block.  y is split as well unless out_f32.  Kernels 4 and 6 (f16 MFMA,
     * Cin % 64 == 0); no SFT epilogue.                                                                              */
    int32_t x_lo, y_lo, r_lo;
    /* GroupNorm statistics of the output from the conv epilogue (pgt_conv2d_gn): gn_groups > 0 asks every workgroup tile
     * to write the per-group sum / sum of squares of its outputs (fp32, before the store rounding) into the caller's
     * statistics workspace - the input of pgt_groupnorm_from_partials, which replaces the separate statistics pass of the
     * following GroupNorm (TDResnetBlock: conv -> GroupNorm, modules/rstt_layers.py:875-904).  A tensor written by
     * several launches (the four sub-pixel convolutions of Upsample) uses gn_nsub sub-ranges, this launch being gn_sub.
     * Needs Cout % 8 == 0, Ho*Wo a multiple of the kernel's tile rows (<= 512), kernels 0, 1, 4, no split-K.            */
    int32_t gn_groups, gn_sub, gn_nsub;
    int32_t gn_img0, gn_nimg;   /* this call covers images gn_img0 .. gn_img0+N-1 of a gn_nimg-image tensor (0, 0 = all N) */
    int32_t res_f32;            /* PGT_F16X3 with out_f32: the residual is fp32 as well (ldr in floats) - split-half
                                 * ARITHMETIC on tensors that are stored in fp32 (BiSeNet's BasicBlocks)              */
    int32_t x3_fold;            /* PGT_F16X3, Cout == 64: the weight matrix has 128 rows and TWO K segments per tap and
                                 * 64-channel block (K = KH*KW*2*Cin, the input visited as [x_hi | x_lo]): rows 0..63 hold
                                 * [w_hi | w_hi], rows 64..127 [w_lo | 0]; y[n] = acc[n] + acc[n + 64] before activation /
                                 * residual.  A 64-channel layer then fills the 128-column tile with 2/3 of the K steps
                                 * (the standard form leaves half of the tile idle for three segments).                  */
    int32_t dec_lo, shift_lo;   /* PGT_F16X3 with PGT_EPI_SFT: element offsets of the lo planes of sft_dec / sft_shift
                                 * (0 = Cout): out = dec + sft_w * (dec * act(conv) + shift) on split operands       */
    int32_t bias_rows;          /* 0: `bias` holds Cout values.  > 0: `bias` is a (N*Ho*Wo / bias_rows, Cout) fp32 matrix, one
                                 * vector per bias_rows consecutive output pixels - a bias per frame (bias_rows = Ho*Wo; for
                                 * token rows: tokens per frame), the form pgt_mean_field_bias produces.  A multiple of 512
                                 * that divides N*Ho*Wo; single-plane dtypes; kernels 0, 1, 4, 5, 6, 7.                       */
    int32_t out_split;          /* dtype == PGT_F32 only: store y as split-half planes [hi | lo] (lo plane y_lo elements after
                                 * the hi plane, ldy in 16-bit elements) instead of fp32 - the exact-fp32 first conv of the encoder
                                 * (3 input channels) feeding the split-half levels without a conversion pass
                                 * (archs/tdcrqvae3_arch.py:540-546).  16-byte epilogue only: Cout % 8 == 0, y / ldy 16-byte aligned,
                                 * y_lo % 8 == 0, no residual, no split-K (anything else: -22).                                     */
    int32_t w2;                 /* PGT_F16 / PGT_BF16 only: EXACT-WEIGHT form.  w holds two 16-bit planes per filter row (the form
                                 * pgt_pack_conv_weight(..., x3_fold = 2) writes): 2 * ceil(Cout / 32) * 32 rows of KH*KW*Cin, per group of
                                 * 32 output channels [32 rows w_hi | 32 rows (w - w_hi) * 2048]; the kernels multiply the single-plane
                                 * operand by both (two MFMAs per product) and add acc_hi + acc_lo / 2048 in fp32 before the epilogue: the
                                 * layer computes with 22-bit weights, i.e. the weight-rounding error of a 16-bit layer (DESIGN.md section
                                 * 2.3; reference layers are fp32: modules/rstt_layers.py:875-904, archs/pgtformer_arch.py:421-484) is gone
                                 * and no mean-field bias is needed.  Kernel 8 (the 64-channel ring kernel, incl. pgt_conv2d_affine_in)
                                 * where it is legal, else kernel 4 (Cin % 64 == 0, 16-byte epilogue); no split-K.                      */
} pgt_conv_desc;

int pgt_conv2d(const pgt_conv_desc* d, const void* x, const void* w, const float* bias,
               const void* residual, const void* sft_dec, const void* sft_shift, void* y,
               pgt_stream_t stream);
/* Same, with a caller-owned scratch buffer that enables split-K on deep-K / small-M layers (the 32x32
 * feature maps): K slices accumulate fp32 partial tiles in `workspace`, a second kernel sums them in slice
 * order (deterministic) and applies the epilogue.  pgt_conv2d_workspace_bytes() returns the size the layer
 * would use (0: single pass, workspace may be NULL). */
size_t pgt_conv2d_workspace_bytes(const pgt_conv_desc* d);
int pgt_conv2d_ws(const pgt_conv_desc* d, const void* x, const void* w, const float* bias,
                  const void* residual, const void* sft_dec, const void* sft_shift, void* y,
                  void* workspace, size_t workspace_bytes, pgt_stream_t stream);

/* pgt_conv2d_ws + epilogue GroupNorm statistics (see pgt_conv_desc::gn_groups).  gn_workspace: fp32, 16-byte aligned,
 * pgt_conv_gn_workspace_bytes(N, nsub, Ho*Wo of one launch, groups) bytes, shared by the gn_nsub launches of a tensor. */
size_t pgt_conv_gn_workspace_bytes(int32_t N, int32_t nsub, int32_t HWsub, int32_t groups);
int pgt_conv2d_gn(const pgt_conv_desc* d, const void* x, const void* w, const float* bias,
                  const void* residual, const void* sft_dec, const void* sft_shift, void* y,
                  float* gn_workspace, void* workspace, size_t workspace_bytes, pgt_stream_t stream);

/* The conv of act(x * in_scale[n][c] + in_shift[n][c]): the GroupNorm apply + SiLU of the Normalize that PRECEDES the conv
 * (conv1 / conv2 of TDResnetBlock, modules/rstt_layers.py:875-904 with Normalize / nonlinearity :754-758; norm_out -> conv_out,
 * archs/tdcrqvae3_arch.py:672-707) fused into the operand load: the normalised tensor is never written.  in_scale / in_shift:
 * fp32 (N, Cin), the output of pgt_groupnorm_affine / pgt_groupnorm_from_partials; in_act: PGT_ACT_NONE or PGT_ACT_SILU.
 * Exists for the launches pgt_conv2d_affine_in_ok(d) accepts (returns 1): single-plane 16-bit dtypes, 3x3 stride 1 pad 1 same
 * size, Cin == 64, Cout <= 64, W >= 128 and H >= 4 powers of two, epilogue = bias (+ residual; bias_rows allowed) without
 * activation, 16-byte aligned rows unless Cout <= 32 - the full-resolution level, where a separate apply pass costs as much HBM
 * traffic as the conv itself.  Results equal pgt_affine_act followed by pgt_conv2d bit for bit in the operand (same arithmetic,
 * one rounding to the 16-bit type); the accumulation order is that of kernel 8.  Other launches: -22, run the two calls.        */
int pgt_conv2d_affine_in_ok(const pgt_conv_desc* d);
int pgt_conv2d_affine_in(const pgt_conv_desc* d, const void* x, const float* in_scale, const float* in_shift, int32_t in_act,
                         const void* w, const float* bias, const void* residual, void* y, pgt_stream_t stream);

/* ---- normalisation ---------------------------------------------------------------------------
 * GroupNorm(groups, eps) statistics -> per-(n,c) affine so that GN(x) = x*scale + shift
 * (Normalize(), rstt_layers.py:754-755; normalize(), pgtformer_arch.py:406-407).  Deterministic
 * two-stage reduction; `workspace` needs pgt_groupnorm_workspace_bytes() bytes. */
size_t pgt_groupnorm_workspace_bytes(int32_t N, int32_t HW, int32_t C, int32_t groups);
int pgt_groupnorm_affine(int32_t dtype, const void* x, int32_t ldx, int32_t N, int32_t HW, int32_t C,
                         int32_t groups, float eps, const float* gamma, const float* beta,
                         float* scale, float* shift, void* workspace, size_t workspace_bytes,
                         pgt_stream_t stream);
/* The same per-(n,c) affine from the statistics a producing conv left in `gn_workspace` (pgt_conv2d_gn): fixed-order
 * reduction in double, no pass over the tensor.  HWsub = output pixels per image of ONE producing launch, nsub launches. */
int pgt_groupnorm_from_partials(const float* gn_workspace, int32_t N, int32_t nsub, int32_t HWsub, int32_t C,
                                int32_t groups, float eps, const float* gamma, const float* beta, float* scale,
                                float* shift, pgt_stream_t stream);
/* y = act(x * scale[n,c] + shift[n,c])  (GN apply + SiLU/swish, AdaIN apply, eval-BN apply) */
int pgt_affine_act(int32_t dtype, const void* x, int32_t ldx, void* y, int32_t ldy, int32_t N,
                   int32_t HW, int32_t C, const float* scale, const float* shift, int32_t act,
                   pgt_stream_t stream);
/* LayerNorm over the last dim (nn.LayerNorm eps 1e-5; rstt_layers.py:298,335; codeformer_arch.py:128,134;
 * pgtformer_arch.py:532).  y = LN(x); if y2 != NULL also y2 = LN(x) + pos (q = k = LN(x) + query_pos). */
int pgt_layernorm(int32_t dtype, const void* x, int32_t ldx, int32_t rows, int32_t C,
                  const float* gamma, const float* beta, float eps, void* y, int32_t ldy,
                  const void* pos, int32_t ldpos, void* y2, int32_t ldy2, pgt_stream_t stream);
/* per-(n,c) mean and UNBIASED variance over HW pixels (calc_mean_std, codeformer_arch.py:15-29) */
int pgt_channel_stats(int32_t dtype, const void* x, int32_t ldx, int32_t N, int32_t HW, int32_t C,
                      float* mean, float* var_unbiased, pgt_stream_t stream);
/* ---- mean-field compensation of the weight rounding of a 16-bit layer (no reference counterpart: the reference's layers are
 * fp32, modules/rstt_layers.py:875-904, archs/pgtformer_arch.py:460-484; this keeps a half / bf16 layer's output unbiased).
 * y = W x + b with W rounded to W16 leaves (W - W16) x; the part of it that is constant over a frame, (W - W16) mean(x), is
 * put back as a per-frame bias (pgt_conv_desc::bias_rows):  out[r][o] = bias[o] + sum_k defect_t[k][o] * mean[r][k],
 * defect_t[k][o] = sum over filter taps of (W - W16)[o][k][tap] (fp32, K x Cout).
 * pgt_sampled_channel_mean: mean[n][c] of x (N, HW, C; pixel stride ldx) over a fixed sample of min(HW, 1024) pixels of each
 * frame: 16 consecutive pixels from each of up to 64 equal cells (K <= 3840 in pgt_mean_field_bias); pgt_sampled_pixel(HW, i) = the i-th sampled pixel index
 * (-1 past the end) so that a host can reproduce the sample. */
int pgt_sampled_channel_mean(int32_t dtype, const void* x, int32_t ldx, int32_t N, int32_t HW, int32_t C, float* mean,
                             pgt_stream_t stream);
int pgt_sampled_pixel(int32_t HW, int32_t i);
int pgt_mean_field_bias(const float* mean, const float* defect_t, const float* bias, int32_t R, int32_t K, int32_t Cout,
                        float* out, pgt_stream_t stream);
/* pgt_sampled_channel_mean + pgt_mean_field_bias in ONE launch (each of the two is at the launch-latency floor and sits in the
 * dependency chain in front of a compensated layer): out[n][o] = bias[o] + sum_k defect_t[k][o] * mean_n[k], mean_n over frame
 * n's pixel sample of x (N, HW, K; PGT_F16 / PGT_BF16) - or, with in_scale / in_shift (fp32 (N, K)), of the operand a layer
 * fed through pgt_conv2d_affine_in multiplies: in_act(x * in_scale + in_shift) rounded to the tensor's type.  K % 8 == 0.
 * out_groups = 1: out is (N, Cout).  out_groups = G > 1: defect_t / bias hold G layers side by side (Cout = G x Csub columns:
 * the four sub-pixel convolutions of an Upsample read ONE operand) and out is (G, N, Csub) - one contiguous per-frame bias matrix per
 * layer.  The N "frames" may be BANDS of images (N = images x bands, HW = pixels of a band: consecutive rows of the raster; the
 * conv then takes bias_rows = HW): the mean field is resolved down the image; scale_div = bands tells which coefficient row of
 * in_scale / in_shift (one per image) a band uses (1 otherwise); sample_cells (0 = 64) bounds the sample of a band to
 * sample_cells x 16 pixels (pgt_sampled_pixel_cells(HW, sample_cells, i) = its i-th pixel), so that 16 bands of a small map do not
 * add up to a pass over the whole tensor.
 * workspace: pgt_frame_bias_workspace_bytes(N, K, Cout) bytes of scratch (partial rows per 64-channel slice of K).
 * counters: N uint32, ZERO before the first call and left zero by every call (the workgroup that finishes a frame last adds
 * its partial rows in slice order - one fixed order whatever the arrival order: deterministic); calls that may run
 * CONCURRENTLY (other streams) need their own counters, consecutive calls on one stream share them. */
size_t pgt_frame_bias_workspace_bytes(int32_t N, int32_t K, int32_t Cout);
int pgt_frame_bias(int32_t dtype, const void* x, int32_t ldx, int32_t N, int32_t HW, int32_t K, const float* in_scale,
                   const float* in_shift, int32_t in_act, const float* defect_t, const float* bias, int32_t Cout,
                   int32_t out_groups, int32_t scale_div, int32_t sample_cells, float* out, void* workspace, size_t workspace_bytes,
                   uint32_t* counters, pgt_stream_t stream);
int pgt_sampled_pixel_cells(int32_t HW, int32_t sample_cells, int32_t i);
/* pgt_weight_defect: the (K x Cout) fp32 operand `defect_t` of pgt_mean_field_bias for a layer, from its fp32 reference weight
 * (Cout, Cin, KH, KW) (x out_scale[o] where given, e.g. the BatchNorm fold) and the packed 16-bit operand pgt_pack_conv_weight
 * wrote for it: defect_t[k][o] = sum over taps of (w * scale - packed)[o][k][tap], K = Cin_pad; sum_taps = 0 keeps one row per
 * (tap, k) (K = KH*KW*Cin_pad: a conv whose taps read different frames).  PGT_F16 / PGT_BF16.  With this a non-Python host
 * reaches the default precision mode's compensated layers from the header alone. */
int pgt_weight_defect(int32_t dtype, const float* w_oihw, int32_t Cout, int32_t Cin, int32_t KH, int32_t KW, int32_t Cin_pad,
                      const float* out_scale, const void* packed, int32_t sum_taps, float* defect_t, pgt_stream_t stream);
/* The sampled mean for a layer that reads NORMALISED rows (pgt_ln_linear below): mean[n][c] over frame n's sample of
 * half((x[p][c] - mu_p) * rstd_p), mu_p / rstd_p = the LayerNorm statistics of row p over its C channels (C = 256 or 512;
 * PGT_F16 / PGT_BF16).  workspace: pgt_sampled_rownorm_workspace_bytes bytes (partial sums of 128-row sample slices, summed in order). */
size_t pgt_sampled_rownorm_workspace_bytes(int32_t N, int32_t HW, int32_t C);
int pgt_sampled_rownorm_mean(int32_t dtype, const void* x, int32_t ldx, int32_t N, int32_t HW, int32_t C, float eps,
                             float* mean, void* workspace, pgt_stream_t stream);
/* AdaIN coefficients: scale = sqrt(var_s+eps)/sqrt(var_c+eps), shift = mean_s - mean_c*scale
 * (adaptive_instance_normalization, codeformer_arch.py:32-46); n = N*C entries */
int pgt_adain_affine(const float* mean_c, const float* var_c, const float* mean_s, const float* var_s,
                     float eps, float* scale, float* shift, int32_t n, pgt_stream_t stream);

/* ---- fused token-row chains of the 256-channel window-attention blocks (round 4) -------------------------------------
 * A VSTSREncoderTransformerBlock (modules/rstt_layers.py:284-338) is, per token row: LN1 -> [q | k | v] Linear -> window
 * attention -> proj Linear + shortcut -> LN2 -> fc1 -> GELU -> fc2 + residual.  Layer by layer a row crosses HBM 14 times; these
 * two entry points keep it on chip between its element-wise and GEMM steps (rows in registers as the MFMA B operand, weights
 * streaming through an LDS ring), so that only the block's input / output and the attention operands touch HBM.
 *
 * pgt_fold_layernorm: LN(x) W^T + b = xhat (W diag(gamma))^T + (b + W beta), xhat = (x - mean) / sqrt(var + eps).  w: fp32
 * (Cout, Cin) as the reference stores nn.Linear weights; w_out (may alias w) is then packed with pgt_pack_conv_weight;
 * bias_out (Cout) replaces the bias (bias may be NULL = 0).
 * pgt_ln_linear: y = half(xhat) W'^T + bias' on rows of Cin = 256 channels, PGT_F16; W' (Cout, 256) K-major half rows, Cout a
 * multiple of 128 up to 768; bias' (Cout) fp32, or with bias_rows > 0 a (rows / bias_rows, Cout) matrix - one vector per
 * bias_rows consecutive rows (a multiple of 256 that divides rows), the form pgt_mean_field_bias produces from
 * pgt_sampled_rownorm_mean.  Replaces norm1 + q / kv Linear (:298, :195-213).
 * pgt_attn_proj_mlp: x1 = attn Wproj^T + b_proj + shortcut (rounded to half);  y = x1 + fc2(GELU(fc1(LN2(x1)))) with
 * w3 = [Wproj; Wfc1 diag(gamma2); Wfc2] stacked (768, 256) K-major half rows, b_fc1 carrying W1 beta2 (pgt_fold_layernorm).
 * bias_rows = 0: the three biases are 256-vectors; > 0: each is a (rows / bias_rows, 256) matrix - one vector per bias_rows
 * consecutive rows (a multiple of 128 that divides rows: a frame), the form pgt_mean_field_bias produces.  Replaces proj +
 * shortcut add (:230-232, :329), norm2, Mlp (:126-132, mlp_ratio = 1: archs/tdcrqvae3_arch.py:499) and the residual add
 * (:335-337).  GELU is the exact-erf form (erf to 1.5e-7).  y may alias neither input.
 * pgt_attn_proj_mlp_sample: the operands of fc1 and fc2 never reach HBM in that launch, so the per-frame channel means their
 * weight-rounding compensation needs (pgt_mean_field_bias) are taken by running the chain up to the hidden row on the fixed
 * pixel sample of every frame (pgt_sampled_pixel; frames of HW rows): mean_ln[f][c] = mean of half(xhat) (fc1's operand),
 * mean_hid[f][c] = mean of half(GELU(fc1(xhat) + b_fc1)) (fc2's operand; fc1 with its plain bias), both (frames, 256) fp32.
 * b_proj: 256 values, or (frames, 256) with b_proj_per_frame.  workspace: pgt_attn_proj_mlp_sample_workspace_bytes bytes. */
int pgt_fold_layernorm(const float* w, const float* gamma, const float* beta, const float* bias, int32_t Cout, int32_t Cin,
                       float* w_out, float* bias_out, pgt_stream_t stream);
int pgt_ln_linear(int32_t dtype, const void* x, int32_t ldx, int32_t rows, int32_t Cin, float eps, const void* w,
                  const float* bias, int32_t bias_rows, int32_t Cout, void* y, int32_t ldy, pgt_stream_t stream);
int pgt_attn_proj_mlp(int32_t dtype, const void* attn, int32_t lda, const void* shortcut, int32_t lds, int32_t rows, int32_t C,
                      const void* w3, const float* b_proj, const float* b_fc1, const float* b_fc2, int32_t bias_rows,
                      float eps, void* y, int32_t ldy, pgt_stream_t stream);
size_t pgt_attn_proj_mlp_sample_workspace_bytes(int32_t frames, int32_t HW);
int pgt_attn_proj_mlp_sample(int32_t dtype, const void* attn, int32_t lda, const void* shortcut, int32_t lds, int32_t frames,
                             int32_t HW, int32_t C, const void* w3, const float* b_proj, int32_t b_proj_per_frame,
                             const float* b_fc1, float eps, void* workspace, float* mean_ln, float* mean_hid,
                             pgt_stream_t stream);

/* Split-half (PGT_F16X3) forms of the chains, for the encoder-side blocks: x / y split rows (lo planes x_lo / y_lo elements
 * after the hi planes), every product x_lo w_hi + x_hi w_lo + x_hi w_hi on the f16 MFMA, fp32 statistics; w in the
 * pgt_pack_conv_weight(PGT_F16X3) form of the FOLDED matrix.  pgt_ln_linear_x3: y = xhat W'^T + b' (norm1 + [q | k | v]).
 * pgt_ln_mlp_x3: y = x + fc2(GELU(fc1(LN(x)))) with w2 = [Wfc1 diag(gamma); Wfc2] stacked (512 packed rows), exact erf GELU; x is
 * re-read for the residual (y must not alias it).  The proj + shortcut step of a split block stays a pgt_conv2d launch. */
int pgt_ln_linear_x3(const void* x, int32_t ldx, int32_t x_lo, int32_t rows, int32_t Cin, float eps, const void* w,
                     const float* bias, int32_t Cout, void* y, int32_t ldy, int32_t y_lo, pgt_stream_t stream);
int pgt_ln_mlp_x3(const void* x, int32_t ldx, int32_t x_lo, int32_t rows, int32_t C, float eps, const void* w2,
                  const float* b_fc1, const float* b_fc2, void* y, int32_t ldy, int32_t y_lo, pgt_stream_t stream);

/* ---- attention -------------------------------------------------------------------------------
 * (T,Wh,Ww)-window multi-head self-attention with cyclic shift, relative-position bias and the
 * 9-region shift mask (WindowAttention3D.forward rstt_layers.py:195-234 + window_partition/reverse
 * :55-88 + torch.roll :307-327 + mask :552-568).  qkv: (B*T*H*W, 3C) rows in (b,t,y,x) order, columns
 * [q | k | v], head h at columns h*hd; q is NOT pre-scaled.  bias: dense (heads, N, N) fp32 with
 * N = T*wh*ww (gathered once at weight-pack time from relative_position_bias_table/_index).
 * out: (B*T*H*W, C) written at the un-shifted token positions. */
int pgt_window_attention(int32_t dtype, const void* qkv, int32_t ldqkv, void* out, int32_t ldo,
                         const float* bias, int32_t B, int32_t T, int32_t H, int32_t W, int32_t C,
                         int32_t heads, int32_t wh, int32_t ww, int32_t sh, int32_t sw,
                         pgt_stream_t stream);
/* Video-Swin form of the same attention (modules/swin.py: WindowAttention3D :85-167 + window_partition/reverse :38-64 +
 * the 3-axis torch.roll of SwinTransformerBlock3D.forward_part1 :212-246 + compute_mask :311-323): windows (wd,wh,ww) of
 * the (D,H,W) token grid, cyclic shift (sd,sh,sw), 27-region mask.  qkv: (B*D*H*W, 3C) rows in (b,d,y,x) order
 * (the fused `qkv` Linear's output, columns [q | k | v]); bias dense (heads, N, N) fp32, N = wd*wh*ww.
 * dtype PGT_BF16 or PGT_F16 with N a multiple of 48 (<= 192, e.g. 3x8x8) and (D,H,W) multiples of the window: the MFMA
 * kernel (fp16 MFMA: BASELINE.json configs[4]).  Everything else - PGT_F32 storage, other N (K, V and the N x N
 * probabilities of a window live in LDS: (2 N (hd + 4) + N (N + 1)) * 4 bytes <= 160 KiB, i.e. N <= ~170), feature maps that are
 * not multiples of the window - takes the general kernel, which pads the grid at the far end of D, H, W as
 * forward_part1 does (:218-223): `pad_row` (3C elements of `dtype`, or NULL = zeros) is the qkv row of a padding token,
 * i.e. the qkv Linear's bias; outputs of padding tokens are not written. */
int pgt_window_attention3d(int32_t dtype, const void* qkv, int32_t ldqkv, void* out, int32_t ldo,
                           const float* bias, const void* pad_row, int32_t B, int32_t D, int32_t H, int32_t W, int32_t C,
                           int32_t heads, int32_t wd, int32_t wh, int32_t ww, int32_t sd, int32_t sh,
                           int32_t sw, pgt_stream_t stream);
/* global multi-head attention, flash style (nn.MultiheadAttention inside TransformerSALayer,
 * codeformer_arch.py:105,129): per batch b, softmax(q k^T * scale) v over L tokens.
 * q,k,v: (B*L, heads*hd) row-major with row strides ldq/ldk/ldv. */
int pgt_mha(int32_t dtype, const void* q, int32_t ldq, const void* k, int32_t ldk, const void* v,
            int32_t ldv, void* out, int32_t ldo, int32_t B, int32_t L, int32_t heads, int32_t hd,
            float scale, pgt_stream_t stream);

/* ---- split-half (PGT_F16X3) forms of the normalisation / attention entry points ------------------------------
 * Same arithmetic as the functions above on tensors stored as [hi | lo] half planes; every tensor argument carries the
 * element offset of its lo plane (`*_lo`) next to its row stride.  Statistics, softmax and accumulation are fp32;
 * every MFMA product is hi*hi + lo*hi + hi*lo.  Used by the code-prediction branch (encoder levels with temporal
 * attention, quant_conv, feat_emb, the 9 TransformerSALayers, idx_pred_layer: archs/pgtformer_arch.py:626-649). */
int pgt_groupnorm_affine_x3(const void* x, int32_t ldx, int32_t x_lo, int32_t N, int32_t HW, int32_t C,
                            int32_t groups, float eps, const float* gamma, const float* beta, float* scale,
                            float* shift, void* workspace, size_t workspace_bytes, pgt_stream_t stream);
int pgt_affine_act_x3(const void* x, int32_t ldx, int32_t x_lo, void* y, int32_t ldy, int32_t y_lo, int32_t N,
                      int32_t HW, int32_t C, const float* scale, const float* shift, int32_t act,
                      pgt_stream_t stream);
int pgt_layernorm_x3(const void* x, int32_t ldx, int32_t x_lo, int32_t rows, int32_t C, const float* gamma,
                     const float* beta, float eps, void* y, int32_t ldy, int32_t y_lo, const void* pos,
                     int32_t ldpos, int32_t pos_lo, void* y2, int32_t ldy2, int32_t y2_lo, pgt_stream_t stream);
int pgt_window_attention_x3(const void* qkv, int32_t ldqkv, int32_t qkv_lo, void* out, int32_t ldo,
                            int32_t out_lo, const float* bias, int32_t B, int32_t T, int32_t H, int32_t W,
                            int32_t C, int32_t heads, int32_t wh, int32_t ww, int32_t sh, int32_t sw,
                            pgt_stream_t stream);
int pgt_mha_x3(const void* q, int32_t ldq, int32_t q_lo, const void* k, int32_t ldk, int32_t k_lo, const void* v,
               int32_t ldv, int32_t v_lo, void* out, int32_t ldo, int32_t out_lo, int32_t B, int32_t L,
               int32_t heads, int32_t hd, float scale, pgt_stream_t stream);
/* fp32 (rows, cols) <-> split-half planes (hi = half(v), lo = half(v - hi), saturating) */
int pgt_x3_split(const float* src, int32_t lds, void* dst, int32_t ldd, int32_t dst_lo, int64_t rows,
                 int32_t cols, pgt_stream_t stream);
/* split-half planes -> IEEE half rows (the encoder-side feature maps entering the PGT_F16 decoder's fusion blocks,
 * archs/pgtformer_arch.py:627-630: the value rounded to half, i.e. the hi plane up to ties) */
int pgt_x3_to_half(const void* src, int32_t lds, int32_t src_lo, void* dst, int32_t ldd, int64_t rows, int32_t cols,
                   pgt_stream_t stream);
int pgt_x3_merge(const void* src, int32_t lds, int32_t src_lo, float* dst, int32_t ldd, int64_t rows,
                 int32_t cols, pgt_stream_t stream);

/* One categorical draw per row of a (rows, K) probability matrix by inverse CDF: codes[r] = first j with
 * prob[r,0] + .. + prob[r,j] > u[r] * sum_j prob[r,j], u[r] uniform in [0, 1) supplied by the caller (the stochastic branch of
 * RQBottleneck.get_soft_codes: torch.multinomial(soft_code, 1), archs/tdcrqvae3_arch.py:443-446 - the same distribution,
 * the caller's random stream). */
int pgt_sample_rows(const float* prob, int32_t ld, int32_t rows, int32_t K, const float* u, int32_t* codes,
                    pgt_stream_t stream);

/* ---- weight repack (once, at load) --------------------------------------------------------------
 * The conv / linear kernels take their weights K-major; the reference stores nn.Conv2d weights as (Cout, Cin, KH, KW) and
 * nn.Linear weights as (Cout, Cin) (= KH = KW = 1).  pgt_pack_conv_weight writes the operand `pgt_conv2d` expects for
 * `dtype` from the reference tensor on the device: (Cout, KH*KW*Cin_pad) in fp32 / bf16 / half with the input channels
 * zero-padded to Cin_pad (3 -> 8, 57 -> 64, the [enc | dec | fut] concats -> multiples of 64); PGT_F16X3: the
 * [w_hi | w_hi | w_lo]-per-64-channel-block form, or with x3_fold (Cout == 64) the folded (128, KH*KW*2*Cin_pad) form.
 * PGT_F16 / PGT_BF16 with x3_fold == 2: the EXACT-WEIGHT form of pgt_conv_desc::w2 - (2 * ceil(Cout / 32) * 32, KH*KW*Cin_pad), per
 * group of 32 output channels 32 rows of w_hi = rn16(w) followed by 32 rows of rn16((w - w_hi) * 2048) (rows past Cout zero).
 * out_scale (Cout floats or NULL) multiplies every output channel in fp32 before the rounding: the eval-BatchNorm fold
 * of BiSeNet (archs/pgtformer_arch.py:40-68), whose factors and folded bias pgt_fold_batchnorm computes
 * (scale = gamma / sqrt(var + eps), bias = (conv_bias - mean) * scale + beta).  pgt_packed_weight_bytes sizes `packed`. */
size_t pgt_packed_weight_bytes(int32_t dtype, int32_t Cout, int32_t Cin_pad, int32_t KH, int32_t KW, int32_t x3_fold);
int pgt_pack_conv_weight(int32_t dtype, const float* w_oihw, int32_t Cout, int32_t Cin, int32_t KH, int32_t KW,
                         int32_t Cin_pad, const float* out_scale, int32_t x3_fold, void* packed, pgt_stream_t stream);
int pgt_fold_batchnorm(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                       float eps, const float* conv_bias, int32_t C, float* scale, float* bias, pgt_stream_t stream);

/* ---- quantiser -------------------------------------------------------------------------------
 * codes[r] = first argmax_j logits[r, j]  (logits.argmax(-1), pgtformer_arch.py:663) */
int pgt_argmax_rows(const float* logits, int32_t ld, int32_t rows, int32_t K, int32_t* codes,
                    pgt_stream_t stream);
/* nearest code: codes[r] = first argmin_j (xnorm[r] + enorm[j]) - 2*dot[r,j]
 * (VQEmbedding.compute_distances/find_nearest_embedding, tdcrqvae3_arch.py:100-126; also
 * VectorQuantizer.forward distance+argmin, archs/vqgan_arch.py:48-54) */
int pgt_rq_argmin(const float* dot, int32_t ld, const float* xnorm, const float* enorm, int32_t rows,
                  int32_t K, int32_t* codes, pgt_stream_t stream);
/* The same look-up with the arg-min inside the distance GEMM (no Ntok x K matrix in HBM): x (rows, D) and codebook
 * (K, D) in bf16, |x|^2 / |e|^2 fp32, D in {64,128,256,512}.  Same association and first-index tie rule; the dot
 * products are bit-identical to pgt_conv2d(out_f32) + pgt_rq_argmin on the same operands. */
int pgt_rq_nearest(int32_t dtype, const void* x, int32_t ldx, const void* codebook, const float* xnorm,
                   const float* enorm, int32_t rows, int32_t K, int32_t D, int32_t* codes, pgt_stream_t stream);
/* soft codes of the quantiser: soft[r, j] = softmax_j(-dist[r, j] / temp), codes[r] = argmin_j dist[r, j]
 * (RQBottleneck.get_soft_codes with stochastic=False, tdcrqvae3_arch.py:429-457) */
int pgt_rq_soft_codes(const float* dot, int32_t ld, const float* xnorm, const float* enorm, int32_t rows,
                      int32_t K, float temp, float* soft, int32_t* codes, pgt_stream_t stream);
/* commitment loss term: loss[0] (+)= scale * mean((x - q)^2) over rows x cols (one depth of
 * RQBottleneck.compute_commitment_loss :340-352; scale = 1/depth, accumulate = 1 from the second depth on); fp32 result
 * in device memory; deterministic two-stage reduction in a caller-owned workspace of pgt_commit_loss_workspace_bytes() */
size_t pgt_commit_loss_workspace_bytes(void);
int pgt_commit_loss(int32_t dtype, const void* x, int32_t ldx, const void* q, int32_t ldq, int64_t rows,
                    int32_t cols, float* loss, float scale, int32_t accumulate, void* workspace,
                    size_t workspace_bytes, pgt_stream_t stream);
/* y = x + (q - x): the value of the straight-through estimator in RQBottleneck.forward (:336), same fp32 order */
int pgt_straight_through(int32_t dtype, const void* x, int32_t ldx, const void* q, int32_t ldq, void* y,
                         int32_t ldy, int64_t rows, int32_t cols, pgt_stream_t stream);
/* Training-side quantiser (VQEmbedding EMA update, tdcrqvae3_arch.py:138-186), fp32.
 * pgt_vq_cluster_stats: stats[0 .. K*D) = per code the sum of the batch vectors assigned to it, stats[K*D .. K*D+K) = how
 * many (`one_hot @ vectors`, `one_hot.sum(1)` of _update_buffers :139-158) - one flat buffer, so that data-parallel ranks
 * combine both with ONE all-reduce (the reference issues two, :157-158).  Deterministic: vectors are added in row order.
 * pgt_vq_ema_update: cluster_size_ema / embed_ema <- decay * ema + (1 - decay) * stats (:160-161); codes whose EMA count
 * fell below 1 restart from restart[k] (K x D, NULL = restart_unused_codes off; :163-177); then the codebook rows
 * weight[k] = embed_ema[k] / (n (cs_k + eps) / (n + K eps)), n = sum(cs) (_update_embedding :179-186); the padding row
 * weight[K] is not touched. */
int pgt_vq_cluster_stats(const float* x, int32_t ldx, const int32_t* codes, int32_t rows, int32_t K, int32_t D,
                         float* stats, pgt_stream_t stream);
int pgt_vq_ema_update(float* cluster_size_ema, float* embed_ema, const float* stats, const float* restart,
                      float* weight, int32_t ldw, int32_t K, int32_t D, float decay, float one_minus_decay, float eps,
                      pgt_stream_t stream);
/* out[r,:] (+)= codebook[codes[r],:] ; optionally resid[r,:] -= codebook[codes[r],:]
 * (VQEmbedding.embed :201-203, RQBottleneck.embed_code :355-368, quantize loop :318-325).
 * codebook fp32 (K+1, D); out/resid dtype = `dtype`. accumulate: 0 = overwrite, 1 = add */
int pgt_embed_rows(int32_t dtype, const float* codebook, int32_t D, const int32_t* codes, int32_t rows,
                   void* out, int32_t ldo, int32_t accumulate, void* resid, int32_t ldres,
                   pgt_stream_t stream);
/* out[r] = sum_c x[r,c]^2 (fp32) */
int pgt_row_sumsq(int32_t dtype, const void* x, int32_t ldx, int32_t rows, int32_t C, float* out,
                  pgt_stream_t stream);

/* ---- BiSeNet glue / element-wise --------------------------------------------------------------
 * 3x3 stride-2 pad-1 max-pool, NHWC (nn.MaxPool2d(3,2,1), pgtformer_arch.py:84) */
int pgt_maxpool3x3s2(int32_t dtype, const void* x, int32_t N, int32_t H, int32_t W, int32_t C, void* y,
                     pgt_stream_t stream);
/* y[n,p,c] = x[n,p,c] * gate[n,c] + addvec[n,c] + addt[n,p,c]   (gate/addvec/addt optional; ARM and
 * FFM gating and the nearest-broadcast adds, pgtformer_arch.py:200-207, 235-247, 324-334) */
int pgt_gate_add(int32_t dtype, const void* x, int32_t ldx, int32_t N, int32_t HW, int32_t C,
                 const void* gate, const void* addvec, const void* addt, int32_t ldt, void* y,
                 int32_t ldy, pgt_stream_t stream);
/* bilinear resize with align_corners=True (F.interpolate, pgtformer_arch.py:375-376) */
int pgt_resize_bilinear_ac(int32_t dtype, const void* x, int32_t ldx, int32_t N, int32_t Hi, int32_t Wi,
                           int32_t C, void* y, int32_t ldy, int32_t Ho, int32_t Wo, pgt_stream_t stream);
/* strided 2-D copy with dtype conversion (channel concat, casts) */
int pgt_copy2d(int32_t src_dtype, const void* src, int32_t lds, int32_t dst_dtype, void* dst, int32_t ldd,
               int64_t rows, int32_t cols, pgt_stream_t stream);

/* frame gather: dst frame i <- src frame idx[i] (idx: n_dst int32 on the device).  A frame is `rows` rows of
 * `row_bytes` bytes (a multiple of 16) with independent row strides in bytes.  Replaces the index_select / stack that
 * turns per-frame tensors into per-window (B*T) tensors: consecutive windows of the reference driver share 2 of 3
 * frames (inference.py:47-74) and everything before the first temporal attention is per-frame
 * (archs/tdcrqvae3_arch.py:546-555), so it is computed once per frame and gathered here. */
int pgt_gather_frames(const void* src, int64_t src_row_stride, void* dst, int64_t dst_row_stride,
                      const int32_t* idx, int32_t n_dst, int64_t rows, int32_t row_bytes, pgt_stream_t stream);

/* dst[r, 0:row_bytes] = 0 for `rows` rows of stride ldd_bytes (16-byte granules): the zero pad channels of the
 * [enc | dec | fut | 0] concat buffers and of the 57 -> 64 channel parsing map (replaces torch.zeros / zero_()) */
int pgt_zero2d(void* dst, int64_t ldd_bytes, int64_t rows, int32_t row_bytes, pgt_stream_t stream);

/* ---- range telemetry of the default precision mode ---------------------------------------------------------------------
 * PGT_F16 / PGT_F16X3 stores saturate at +-65504 instead of producing inf: a checkpoint whose activations leave the half range
 * would be clamped silently.  pgt_count_saturated adds to *count (int32 on the device, zeroed by the caller) the number of
 * elements of a (rows x cols) IEEE-half matrix with row stride ldx (elements; the hi plane of a split tensor) that sit at the
 * limit or are not finite.  The Python driver runs it over every 16-bit tensor of the first forward (WindowRunner
 * check_range) and refuses to continue when a layer saturates - precision "bf16x3" has no such limit in the decoder. */
int pgt_count_saturated(const void* x, int64_t ldx, int64_t rows, int32_t cols, int32_t* count, pgt_stream_t stream);

/* ---- driver edges (inference.py:6-19) ----------------------------------------------------------
 * input window -> channels-last tensors of ONE 16-byte chunk per pixel (RGB + zeros: 4 channels in fp32, 8 in a 16-bit
 * dtype - the input-channel granularity of pgt_conv2d): raw = v/255 (encoder input) and
 * norm = (v/255 - mean)/std (BiSeNet input, transforms.Normalize pgtformer_arch.py:554-556,606).
 * src_kind 0: uint8 (N,H,W,3) HWC frames; 1: fp32 (N,3,H,W) in [0,1] */
int pgt_prep_input(int32_t dtype, const void* src, int32_t src_kind, int32_t N, int32_t H, int32_t W,
                   void* raw, void* norm, pgt_stream_t stream);
/* (N,H,W,C) activations -> fp32 (N,C,H,W) */
int pgt_nhwc_to_nchw_f32(int32_t dtype, const void* x, int32_t ldx, int32_t N, int32_t H, int32_t W,
                         int32_t C, float* y, pgt_stream_t stream);
/* one frame (H,W,C=3) -> uint8 HWC: floor(clamp(x,0,1)*255)  (truncation, inference.py:16-18) */
int pgt_frame_to_u8(int32_t dtype, const void* x, int32_t ldx, int32_t H, int32_t W, uint8_t* y,
                    pgt_stream_t stream);

/* ---- whole-graph entry: a recorded forward replayed by the library ("pgt_forward_window") -------------------
 * The reference's graph is Python (PGTFormer.forward, archs/pgtformer_arch.py:598-714; driver inference.py:12-19) and so is this
 * build's host.  pgtformer_amd/export.py records ONE forward of a prepared model - B sliding 3-frame windows, uint8 frames in,
 * restored uint8 middle frames out - as a tape of the calls of this header, pointers resolved to {persistent block (repacked
 * weights, tables, counters: stored in the file), workspace, input, output}; a non-Python host then needs three calls:
 *   pgt_program_load(path, &prog)       reads the file, uploads the persistent block (the only allocation)
 *   pgt_program_run(prog, in, out, ws, ws_bytes, stream)   replays the launches in order on `stream`: no allocation, no
 *                                       synchronisation, caller-owned buffers (pgt_program_io_bytes / _workspace_bytes give the
 *                                       sizes); results equal the Python host's forward bit for bit; hipGraph-capturable
 *   pgt_program_destroy(prog)
 * A program is specific to what was recorded (checkpoint, precision mode, window batch, frame size: pgt_program_info).  One
 * program serves one stream at a time (its persistent block holds the arrival counters of pgt_frame_bias).                   */
typedef struct pgt_program pgt_program;
int pgt_program_load(const char* path, pgt_program** out);
void pgt_program_destroy(pgt_program* prog);
size_t pgt_program_workspace_bytes(const pgt_program* prog);
int pgt_program_io_bytes(const pgt_program* prog, size_t* input_bytes, size_t* output_bytes);
const char* pgt_program_info(const pgt_program* prog);
int pgt_program_run(const pgt_program* prog, const void* input_u8, void* output_u8, void* workspace, size_t workspace_bytes,
                    pgt_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* PGT_HIP_H */
