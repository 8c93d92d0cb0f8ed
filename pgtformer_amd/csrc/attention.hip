// Attention kernels (K5 window attention, K8 global MHA of SURVEY.md §2.3).
//
// v1: storage-type generic (f32 parity mode and bf16) with LDS-staged K/V tiles and fp32 VALU
// contractions; roll / window partition / reverse are pure address arithmetic (no copies), the
// relative-position bias is a dense per-head (N,N) table and the shift mask is computed from the
// token's region id.  The MFMA variants replace the contractions, not the data flow.
#include "common.h"
#include "pgt_internal.h"

namespace {

// ---------------------------------------------------------------------------------------------
// Window attention: one workgroup per (window, head); thread (i = tid>>2, cg = tid&3) owns query
// row i and the key columns j = 4*jj + cg.
// ---------------------------------------------------------------------------------------------
template <typename T, int HD, int NMAX>
__global__ __launch_bounds__(NMAX * 4) void window_attn_kernel(const T* __restrict__ qkv, int ldqkv, T* __restrict__ out,
                                                               int ldo, const float* __restrict__ bias, int T_, int H, int W,
                                                               int C, int heads, int wh, int ww, int sh, int sw) {
    constexpr int KS = HD + 4;       // padded K/V row (floats): float4 reads, conflict-free across cg
    constexpr int NC = NMAX / 4;     // key columns per thread
    constexpr int PSTR = NMAX + 1;
    __shared__ __attribute__((aligned(16))) float ks[NMAX * KS];
    __shared__ __attribute__((aligned(16))) float vs[NMAX * KS];
    __shared__ float ps[NMAX * PSTR];
    __shared__ int tok[NMAX];
    __shared__ int reg[NMAX];

    const int N = T_ * wh * ww;
    const int nwx = W / ww, nwy = H / wh;
    int bid = blockIdx.x;
    const int head = bid % heads; bid /= heads;
    const int wx = bid % nwx; bid /= nwx;
    const int wy = bid % nwy;
    const int b = bid / nwy;
    const int tid = threadIdx.x;
    const float scale = rsqrtf((float)HD);
    const bool shifted = (sh > 0) || (sw > 0);

    if (tid < N) {
        const int s = tid % ww;
        const int r = (tid / ww) % wh;
        const int d = tid / (ww * wh);
        const int ys = wy * wh + r, xs = wx * ww + s;             // coordinates in the rolled frame
        const int y = (ys + sh) % H, x = (xs + sw) % W;           // source pixel (roll by -shift)
        tok[tid] = ((b * T_ + d) * H + y) * W + x;
        const int rh = ys < H - wh ? 0 : (ys < H - sh ? 1 : 2);   // img_mask regions (rstt_layers.py:552-563)
        const int rw = xs < W - ww ? 0 : (xs < W - sw ? 1 : 2);
        reg[tid] = rh * 3 + rw;
    }
    __syncthreads();
    // stage K and V head slices: N rows x HD
    for (int e = tid; e < N * HD; e += blockDim.x) {
        const int j = e / HD, d = e % HD;
        const T* row = qkv + (long)tok[j] * ldqkv + head * HD + d;
        ks[j * KS + d] = ldf(row + C);
        vs[j * KS + d] = ldf(row + 2 * C);
    }
    const int i = tid >> 2, cg = tid & 3;
    const bool rowok = i < N;
    float4 q[HD / 4];
    if (rowok) {
        const T* qr = qkv + (long)tok[i] * ldqkv + head * HD;
#pragma unroll
        for (int d4 = 0; d4 < HD / 4; ++d4) {
            q[d4].x = ldf(qr + 4 * d4 + 0) * scale;
            q[d4].y = ldf(qr + 4 * d4 + 1) * scale;
            q[d4].z = ldf(qr + 4 * d4 + 2) * scale;
            q[d4].w = ldf(qr + 4 * d4 + 3) * scale;
        }
    }
    __syncthreads();
    float s[NC];
    float mx = -3.0e38f;
    if (rowok) {
        const float* brow = bias + ((long)head * N + i) * N;
        const int ri = reg[i];
#pragma unroll
        for (int jj = 0; jj < NC; ++jj) {
            const int j = jj * 4 + cg;
            float a = -3.0e38f;
            if (j < N) {
                a = 0.f;
                const float4* kr = reinterpret_cast<const float4*>(ks + j * KS);
#pragma unroll
                for (int d4 = 0; d4 < HD / 4; ++d4) {
                    const float4 kv = kr[d4];
                    a += q[d4].x * kv.x + q[d4].y * kv.y + q[d4].z * kv.z + q[d4].w * kv.w;
                }
                a += brow[j];
                if (shifted && reg[j] != ri) a += -100.0f;
            }
            s[jj] = a;
            mx = fmaxf(mx, a);
        }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 1, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 2, 64));
    float sum = 0.f;
    if (rowok) {
#pragma unroll
        for (int jj = 0; jj < NC; ++jj) {
            const int j = jj * 4 + cg;
            const float e = j < N ? expf(s[jj] - mx) : 0.f;
            s[jj] = e;
            sum += e;
        }
    }
    sum += __shfl_xor(sum, 1, 64);
    sum += __shfl_xor(sum, 2, 64);
    if (rowok) {
        const float inv = 1.0f / sum;
#pragma unroll
        for (int jj = 0; jj < NC; ++jj) {
            const int j = jj * 4 + cg;
            if (j < N) ps[i * PSTR + j] = s[jj] * inv;
        }
    }
    __syncthreads();
    if (rowok) {
        constexpr int DQ = HD / 4;  // output channels per thread
        float o[DQ];
#pragma unroll
        for (int d = 0; d < DQ; ++d) o[d] = 0.f;
        for (int j = 0; j < N; ++j) {
            const float pj = ps[i * PSTR + j];
            const float4* vr = reinterpret_cast<const float4*>(vs + j * KS + cg * DQ);
#pragma unroll
            for (int d4 = 0; d4 < DQ / 4; ++d4) {
                const float4 vv = vr[d4];
                o[4 * d4 + 0] += pj * vv.x;
                o[4 * d4 + 1] += pj * vv.y;
                o[4 * d4 + 2] += pj * vv.z;
                o[4 * d4 + 3] += pj * vv.w;
            }
        }
        T* orow = out + (long)tok[i] * ldo + head * HD + cg * DQ;
#pragma unroll
        for (int d = 0; d < DQ; ++d) stf(orow + d, o[d]);
    }
}

// ---------------------------------------------------------------------------------------------
// Video-Swin window attention, general form (modules/swin.py:212-246): any window (wd, wh, ww) with N = wd*wh*ww <= NMAX
// tokens, feature maps that are NOT multiples of the window (the reference pads norm1(x) with zeros at the far end of D,
// H, W before the roll - :218-223 - so a padded token's qkv row is the qkv bias, `pad_row`, or zero; its output is
// cropped), 3-axis roll and the 27-region mask of compute_mask (:311-323) on the PADDED grid.  Same thread layout and fp32
// VALU contractions as window_attn_kernel above; K / V / P live in dynamic LDS sized by the run-time N.  This is the
// fallback of pgt_window_attention3d for what the MFMA kernel does not cover (N % 48 != 0, padded maps, fp32 storage).
// ---------------------------------------------------------------------------------------------
template <typename T, int HD, int NMAX>
__global__ __launch_bounds__(NMAX * 4) void window_attn3d_generic_kernel(const T* __restrict__ qkv, int ldqkv, T* __restrict__ out,
                                                                         int ldo, const float* __restrict__ bias,
                                                                         const T* __restrict__ pad_row, int D, int H, int W,
                                                                         int Dp, int Hp, int Wp, int C, int heads, int wd,
                                                                         int wh, int ww, int sd, int sh, int sw) {
    constexpr int KS = HD + 4;       // padded K/V row (floats): float4 reads, conflict-free across cg
    constexpr int NC = NMAX / 4;     // key columns per thread
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    const int N = wd * wh * ww;
    const int PSTR = N + 1;
    float* ks = smem_f;
    float* vs = ks + N * KS;
    float* ps = vs + N * KS;
    int* tok = reinterpret_cast<int*>(ps + N * PSTR);
    int* reg = tok + N;

    const int nwx = Wp / ww, nwy = Hp / wh, nwz = Dp / wd;
    int bid = blockIdx.x;
    const int head = bid % heads; bid /= heads;
    const int wx = bid % nwx; bid /= nwx;
    const int wy = bid % nwy; bid /= nwy;
    const int wz = bid % nwz;
    const int b = bid / nwz;
    const int tid = threadIdx.x;
    const float scale = rsqrtf((float)HD);
    const bool shifted = (sd > 0) || (sh > 0) || (sw > 0);

    if (tid < N) {
        const int s = tid % ww;
        const int r = (tid / ww) % wh;
        const int t = tid / (ww * wh);
        const int zs = wz * wd + t, ys = wy * wh + r, xs = wx * ww + s;     // coordinates in the rolled, padded grid
        const int z = (zs + sd) % Dp, y = (ys + sh) % Hp, x = (xs + sw) % Wp;   // source position (roll by -shift)
        tok[tid] = (z < D && y < H && x < W) ? ((b * D + z) * H + y) * W + x : -1;   // -1: a padding token
        const int rd = zs < Dp - wd ? 0 : (zs < Dp - sd ? 1 : 2);
        const int rh = ys < Hp - wh ? 0 : (ys < Hp - sh ? 1 : 2);
        const int rw = xs < Wp - ww ? 0 : (xs < Wp - sw ? 1 : 2);
        reg[tid] = (rd * 3 + rh) * 3 + rw;
    }
    __syncthreads();
    for (int e = tid; e < N * HD; e += blockDim.x) {
        const int j = e / HD, d = e % HD;
        const T* row = tok[j] >= 0 ? qkv + (long)tok[j] * ldqkv : pad_row;
        ks[j * KS + d] = row ? ldf(row + C + head * HD + d) : 0.f;
        vs[j * KS + d] = row ? ldf(row + 2 * C + head * HD + d) : 0.f;
    }
    const int i = tid >> 2, cg = tid & 3;
    const bool rowok = i < N && tok[i < N ? i : 0] >= 0;     // padding tokens are cropped: no output row
    float4 q[HD / 4];
    if (rowok) {
        const T* qr = qkv + (long)tok[i] * ldqkv + head * HD;
#pragma unroll
        for (int d4 = 0; d4 < HD / 4; ++d4) {
            q[d4].x = ldf(qr + 4 * d4 + 0) * scale;
            q[d4].y = ldf(qr + 4 * d4 + 1) * scale;
            q[d4].z = ldf(qr + 4 * d4 + 2) * scale;
            q[d4].w = ldf(qr + 4 * d4 + 3) * scale;
        }
    }
    __syncthreads();
    float s[NC];
    float mx = -3.0e38f;
    if (rowok) {
        const float* brow = bias + ((long)head * N + i) * N;
        const int ri = reg[i];
#pragma unroll
        for (int jj = 0; jj < NC; ++jj) {
            const int j = jj * 4 + cg;
            float a = -3.0e38f;
            if (j < N) {
                a = 0.f;
                const float4* kr = reinterpret_cast<const float4*>(ks + j * KS);
#pragma unroll
                for (int d4 = 0; d4 < HD / 4; ++d4) {
                    const float4 kv = kr[d4];
                    a += q[d4].x * kv.x + q[d4].y * kv.y + q[d4].z * kv.z + q[d4].w * kv.w;
                }
                a += brow[j];
                if (shifted && reg[j] != ri) a += -100.0f;
            }
            s[jj] = a;
            mx = fmaxf(mx, a);
        }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 1, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 2, 64));
    float sum = 0.f;
    if (rowok) {
#pragma unroll
        for (int jj = 0; jj < NC; ++jj) {
            const int j = jj * 4 + cg;
            const float e = j < N ? expf(s[jj] - mx) : 0.f;
            s[jj] = e;
            sum += e;
        }
    }
    sum += __shfl_xor(sum, 1, 64);
    sum += __shfl_xor(sum, 2, 64);
    if (rowok) {
        const float inv = 1.0f / sum;
#pragma unroll
        for (int jj = 0; jj < NC; ++jj) {
            const int j = jj * 4 + cg;
            if (j < N) ps[i * PSTR + j] = s[jj] * inv;
        }
    }
    __syncthreads();
    if (rowok) {
        constexpr int DQ = HD / 4;  // output channels per thread
        float o[DQ];
#pragma unroll
        for (int d = 0; d < DQ; ++d) o[d] = 0.f;
        for (int j = 0; j < N; ++j) {
            const float pj = ps[i * PSTR + j];
            const float4* vr = reinterpret_cast<const float4*>(vs + j * KS + cg * DQ);
#pragma unroll
            for (int d4 = 0; d4 < DQ / 4; ++d4) {
                const float4 vv = vr[d4];
                o[4 * d4 + 0] += pj * vv.x;
                o[4 * d4 + 1] += pj * vv.y;
                o[4 * d4 + 2] += pj * vv.z;
                o[4 * d4 + 3] += pj * vv.w;
            }
        }
        T* orow = out + (long)tok[i] * ldo + head * HD + cg * DQ;
#pragma unroll
        for (int d = 0; d < DQ; ++d) stf(orow + d, o[d]);
    }
}

// ---------------------------------------------------------------------------------------------
// Global MHA, flash style: one workgroup per (64 queries, head, batch); K/V streamed through LDS in
// 64-key tiles with online softmax.  thread (i = tid>>2, cg = tid&3): query row i, keys 4*jj+cg,
// output channels cg*HD/4 ...
// ---------------------------------------------------------------------------------------------
template <typename T, int HD>
__global__ __launch_bounds__(256) void mha_kernel(const T* __restrict__ q, int ldq, const T* __restrict__ k, int ldk,
                                                  const T* __restrict__ v, int ldv, T* __restrict__ out, int ldo, int L,
                                                  float scale) {
    constexpr int KS = HD + 4;
    constexpr int BQ = 64, BKV = 64, NC = BKV / 4, DQ = HD / 4;
    __shared__ __attribute__((aligned(16))) float ks[BKV * KS];
    __shared__ __attribute__((aligned(16))) float vs[BKV * KS];
    __shared__ float ps[BQ * (BKV + 1)];
    const int tid = threadIdx.x;
    const int head = blockIdx.y, b = blockIdx.z;
    const int q0 = blockIdx.x * BQ;
    const int i = tid >> 2, cg = tid & 3;
    const int qi = q0 + i;
    const bool rowok = qi < L;
    float4 qv[HD / 4];
    if (rowok) {
        const T* qr = q + ((long)b * L + qi) * ldq + head * HD;
#pragma unroll
        for (int d4 = 0; d4 < HD / 4; ++d4) {
            qv[d4].x = ldf(qr + 4 * d4 + 0) * scale;
            qv[d4].y = ldf(qr + 4 * d4 + 1) * scale;
            qv[d4].z = ldf(qr + 4 * d4 + 2) * scale;
            qv[d4].w = ldf(qr + 4 * d4 + 3) * scale;
        }
    }
    float m = -3.0e38f, l = 0.f;
    float o[DQ];
#pragma unroll
    for (int d = 0; d < DQ; ++d) o[d] = 0.f;

    for (int k0 = 0; k0 < L; k0 += BKV) {
        __syncthreads();  // previous tile fully consumed
        for (int e = tid; e < BKV * HD; e += 256) {
            const int j = e / HD, d = e % HD;
            const int kj = k0 + j;
            float kk = 0.f, vv = 0.f;
            if (kj < L) {
                kk = ldf(k + ((long)b * L + kj) * ldk + head * HD + d);
                vv = ldf(v + ((long)b * L + kj) * ldv + head * HD + d);
            }
            ks[j * KS + d] = kk;
            vs[j * KS + d] = vv;
        }
        __syncthreads();
        float s[NC];
        float mx = -3.0e38f;
#pragma unroll
        for (int jj = 0; jj < NC; ++jj) {
            const int j = jj * 4 + cg;
            float a = -3.0e38f;
            if (rowok && k0 + j < L) {
                a = 0.f;
                const float4* kr = reinterpret_cast<const float4*>(ks + j * KS);
#pragma unroll
                for (int d4 = 0; d4 < HD / 4; ++d4) {
                    const float4 kv = kr[d4];
                    a += qv[d4].x * kv.x + qv[d4].y * kv.y + qv[d4].z * kv.z + qv[d4].w * kv.w;
                }
            }
            s[jj] = a;
            mx = fmaxf(mx, a);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 1, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 2, 64));
        const float mnew = fmaxf(m, mx);
        const float alpha = expf(m - mnew);
        float sum = 0.f;
#pragma unroll
        for (int jj = 0; jj < NC; ++jj) {
            const int j = jj * 4 + cg;
            const float e = (rowok && k0 + j < L) ? expf(s[jj] - mnew) : 0.f;
            ps[i * (BKV + 1) + j] = e;
            sum += e;
        }
        sum += __shfl_xor(sum, 1, 64);
        sum += __shfl_xor(sum, 2, 64);
        l = l * alpha + sum;
        m = mnew;
        __syncthreads();
#pragma unroll
        for (int d = 0; d < DQ; ++d) o[d] *= alpha;
        for (int j = 0; j < BKV; ++j) {
            const float pj = ps[i * (BKV + 1) + j];
            const float4* vr = reinterpret_cast<const float4*>(vs + j * KS + cg * DQ);
#pragma unroll
            for (int d4 = 0; d4 < DQ / 4; ++d4) {
                const float4 vv = vr[d4];
                o[4 * d4 + 0] += pj * vv.x;
                o[4 * d4 + 1] += pj * vv.y;
                o[4 * d4 + 2] += pj * vv.z;
                o[4 * d4 + 3] += pj * vv.w;
            }
        }
    }
    if (rowok) {
        const float inv = 1.0f / l;
        T* orow = out + ((long)b * L + qi) * ldo + head * HD + cg * DQ;
#pragma unroll
        for (int d = 0; d < DQ; ++d) stf(orow + d, o[d] * inv);
    }
}

}  // namespace

extern "C" int pgt_window_attention(int32_t dtype, const void* qkv, int32_t ldqkv, void* out, int32_t ldo,
                                    const float* bias, int32_t B, int32_t T, int32_t H, int32_t W, int32_t C,
                                    int32_t heads, int32_t wh, int32_t ww, int32_t sh, int32_t sw, pgt_stream_t stream) {
    PGT_CHECK(qkv && out && bias, "window_attention: null argument");
    PGT_CHECK(H % wh == 0 && W % ww == 0, "window_attention: H=%d W=%d not multiples of window %dx%d", H, W, wh, ww);
    PGT_CHECK(sh >= 0 && sh < wh && sw >= 0 && sw < ww, "window_attention: shift must be in [0, window)");
    const int N = T * wh * ww;
    const int hd = C / heads;
    PGT_CHECK(C % heads == 0 && (hd == 32 || hd == 64), "window_attention: head_dim=%d unsupported (32, 64)", hd);
    if (dtype == PGT_BF16 || dtype == PGT_F16) {
        const int rc = pgt_window_attn_mfma(dtype == PGT_F16 ? 2 : 0, qkv, ldqkv, out, ldo, bias, B, T, H, W, C, heads, T, wh, ww, 0, sh, sw,
                                            (hipStream_t)stream);
        if (rc <= 0) return rc;   // 0 = launched, < 0 = error, 1 = shape not covered by the MFMA kernel
    }
    PGT_CHECK(N <= 64, "window_attention: %d tokens per window needs the bf16 MFMA kernel (N %% 48 == 0, <= 192); "
              "the generic kernel covers N <= 64", N);
    const int grid = B * (H / wh) * (W / ww) * heads;
    hipStream_t st = (hipStream_t)stream;
    const int nmax = N <= 48 ? 48 : 64;
#define WA_LAUNCH(TT, HD, NM)                                                                                    \
    hipLaunchKernelGGL((window_attn_kernel<TT, HD, NM>), dim3(grid), dim3(NM * 4), 0, st, (const TT*)qkv, ldqkv, \
                       (TT*)out, ldo, bias, T, H, W, C, heads, wh, ww, sh, sw)
    if (dtype == PGT_F32) {
        if (hd == 32) { if (nmax == 48) WA_LAUNCH(float, 32, 48); else WA_LAUNCH(float, 32, 64); }
        else          { if (nmax == 48) WA_LAUNCH(float, 64, 48); else WA_LAUNCH(float, 64, 64); }
    } else if (dtype == PGT_BF16) {
        if (hd == 32) { if (nmax == 48) WA_LAUNCH(bf16_t, 32, 48); else WA_LAUNCH(bf16_t, 32, 64); }
        else          { if (nmax == 48) WA_LAUNCH(bf16_t, 64, 48); else WA_LAUNCH(bf16_t, 64, 64); }
    } else if (dtype == PGT_F16) {
        if (hd == 32) { if (nmax == 48) WA_LAUNCH(half_t, 32, 48); else WA_LAUNCH(half_t, 32, 64); }
        else          { if (nmax == 48) WA_LAUNCH(half_t, 64, 48); else WA_LAUNCH(half_t, 64, 64); }
    } else {
        PGT_CHECK(false, "window_attention: bad dtype %d", dtype);
    }
#undef WA_LAUNCH
    PGT_LAUNCH_CHECK();
    return 0;
}

extern "C" int pgt_mha(int32_t dtype, const void* q, int32_t ldq, const void* k, int32_t ldk, const void* v,
                       int32_t ldv, void* out, int32_t ldo, int32_t B, int32_t L, int32_t heads, int32_t hd,
                       float scale, pgt_stream_t stream) {
    PGT_CHECK(q && k && v && out, "mha: null argument");
    PGT_CHECK(hd == 64 || hd == 32, "mha: head_dim=%d unsupported (32, 64)", hd);
    if (dtype == PGT_BF16 && hd == 64 && ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && ldo % 4 == 0)
        return pgt_mha_mfma_bf16(q, ldq, k, ldk, v, ldv, out, ldo, B, L, heads, scale, (hipStream_t)stream);
    const dim3 grid((L + 63) / 64, heads, B), blk(256);
    hipStream_t st = (hipStream_t)stream;
#define MHA_LAUNCH(TT, HD)                                                                                       \
    hipLaunchKernelGGL((mha_kernel<TT, HD>), grid, blk, 0, st, (const TT*)q, ldq, (const TT*)k, ldk, (const TT*)v, \
                       ldv, (TT*)out, ldo, L, scale)
    if (dtype == PGT_F32) { if (hd == 64) MHA_LAUNCH(float, 64); else MHA_LAUNCH(float, 32); }
    else if (dtype == PGT_BF16) { if (hd == 64) MHA_LAUNCH(bf16_t, 64); else MHA_LAUNCH(bf16_t, 32); }
    else if (dtype == PGT_F16) { if (hd == 64) MHA_LAUNCH(half_t, 64); else MHA_LAUNCH(half_t, 32); }
    else PGT_CHECK(false, "mha: bad dtype %d", dtype);
#undef MHA_LAUNCH
    PGT_LAUNCH_CHECK();
    return 0;
}

// ---- split-half (PGT_F16X3) forms: MFMA kernels only ------------------------------------------------------------
extern "C" int pgt_window_attention_x3(const void* qkv, int32_t ldqkv, int32_t qkv_lo, void* out, int32_t ldo,
                                       int32_t out_lo, const float* bias, int32_t B, int32_t T, int32_t H, int32_t W,
                                       int32_t C, int32_t heads, int32_t wh, int32_t ww, int32_t sh, int32_t sw,
                                       pgt_stream_t stream) {
    PGT_CHECK(qkv && out && bias, "window_attention_x3: null argument");
    PGT_CHECK(H % wh == 0 && W % ww == 0, "window_attention_x3: H=%d W=%d not multiples of window %dx%d", H, W, wh, ww);
    PGT_CHECK(sh >= 0 && sh < wh && sw >= 0 && sw < ww, "window_attention_x3: shift must be in [0, window)");
    PGT_CHECK(C % heads == 0 && qkv_lo >= 3 * C && ldqkv >= qkv_lo + 3 * C && out_lo >= C && ldo >= out_lo + C,
              "window_attention_x3: planes do not fit the rows (ldqkv=%d qkv_lo=%d ldo=%d out_lo=%d C=%d)", ldqkv, qkv_lo, ldo, out_lo, C);
    const int rc = pgt_window_attn_mfma(1, qkv, ldqkv, out, ldo, bias, B, T, H, W, C, heads, T, wh, ww, 0, sh, sw,
                                        (hipStream_t)stream, qkv_lo, out_lo);
    PGT_CHECK(rc != 1, "window_attention_x3: shape not covered by the MFMA kernel (N = T*wh*ww multiple of 48 up to 192, "
              "head_dim 32 or 64, 16-byte aligned rows)");
    return rc;
}

extern "C" int pgt_mha_x3(const void* q, int32_t ldq, int32_t q_lo, const void* k, int32_t ldk, int32_t k_lo, const void* v,
                          int32_t ldv, int32_t v_lo, void* out, int32_t ldo, int32_t out_lo, int32_t B, int32_t L,
                          int32_t heads, int32_t hd, float scale, pgt_stream_t stream) {
    PGT_CHECK(q && k && v && out, "mha_x3: null argument");
    PGT_CHECK(hd == 64, "mha_x3: head_dim=%d unsupported (64)", hd);
    return pgt_mha_mfma_bf16(q, ldq, k, ldk, v, ldv, out, ldo, B, L, heads, scale, (hipStream_t)stream, 1, q_lo, k_lo, v_lo, out_lo);
}

// Video-Swin form (modules/swin.py): windows along the depth axis too.  bf16 / fp16 on the MFMA kernel.
extern "C" int pgt_window_attention3d(int32_t dtype, const void* qkv, int32_t ldqkv, void* out, int32_t ldo,
                                      const float* bias, const void* pad_row, int32_t B, int32_t D, int32_t H, int32_t W,
                                      int32_t C, int32_t heads, int32_t wd, int32_t wh, int32_t ww, int32_t sd, int32_t sh,
                                      int32_t sw, pgt_stream_t stream) {
    PGT_CHECK(qkv && out && bias, "window_attention3d: null argument");
    PGT_CHECK(dtype == PGT_BF16 || dtype == PGT_F16 || dtype == PGT_F32, "window_attention3d: dtype must be PGT_F32, PGT_BF16 or PGT_F16 (got %d)", dtype);
    PGT_CHECK(wd > 0 && wh > 0 && ww > 0 && D > 0 && H > 0 && W > 0, "window_attention3d: bad window / feature map");
    PGT_CHECK(sd >= 0 && sd < wd && sh >= 0 && sh < wh && sw >= 0 && sw < ww, "window_attention3d: shift must be in [0, window)");
    PGT_CHECK(C % heads == 0, "window_attention3d: C=%d not divisible by heads=%d", C, heads);
    const bool whole = D % wd == 0 && H % wh == 0 && W % ww == 0;
    if (whole && dtype != PGT_F32) {
        const int rc = pgt_window_attn_mfma(dtype == PGT_F16 ? 2 : 0, qkv, ldqkv, out, ldo, bias, B, D, H, W, C, heads, wd, wh, ww,
                                            sd, sh, sw, (hipStream_t)stream);
        if (rc <= 0) return rc;   // 0 = launched, < 0 = error, 1 = shape not covered by the MFMA kernel
    }
    // general form: any N <= 256, feature map padded up to window multiples as the reference does
    const int N = wd * wh * ww, hd = C / heads;
    PGT_CHECK(N <= 256 && (hd == 16 || hd == 32 || hd == 64), "window_attention3d: %d tokens per window (<= 256) / head_dim %d (16, 32, 64) "
              "not covered", N, hd);
    const int Dp = (D + wd - 1) / wd * wd, Hp = (H + wh - 1) / wh * wh, Wp = (W + ww - 1) / ww * ww;
    const int grid = B * (Dp / wd) * (Hp / wh) * (Wp / ww) * heads;
    const int nmax = N <= 64 ? 64 : (N <= 128 ? 128 : 256);
    const int lds = (2 * N * (hd + 4) + N * (N + 1)) * 4 + 2 * N * 4;
    PGT_CHECK(lds <= 160 * 1024, "window_attention3d: %d tokens x head_dim %d needs %d bytes of LDS", N, hd, lds);
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipSuccess;
#define WG_LAUNCH(TT, HD, NM)                                                                                                  \
    do {                                                                                                                       \
        if (lds > 64 * 1024)                                                                                                   \
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&window_attn3d_generic_kernel<TT, HD, NM>),                  \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, lds);                                           \
        if (e == hipSuccess)                                                                                                   \
            hipLaunchKernelGGL((window_attn3d_generic_kernel<TT, HD, NM>), dim3(grid), dim3(NM * 4), lds, st, (const TT*)qkv,  \
                               ldqkv, (TT*)out, ldo, bias, (const TT*)pad_row, D, H, W, Dp, Hp, Wp, C, heads, wd, wh, ww, sd,  \
                               sh, sw);                                                                                        \
    } while (0)
#define WG_NM(TT, HD)                                                          \
    do {                                                                       \
        if (nmax == 64) WG_LAUNCH(TT, HD, 64);                                 \
        else if (nmax == 128) WG_LAUNCH(TT, HD, 128);                          \
        else WG_LAUNCH(TT, HD, 256);                                           \
    } while (0)
#define WG_HD(TT)                                                              \
    do {                                                                       \
        if (hd == 16) WG_NM(TT, 16);                                           \
        else if (hd == 32) WG_NM(TT, 32);                                      \
        else WG_NM(TT, 64);                                                    \
    } while (0)
    if (dtype == PGT_F32) WG_HD(float);
    else if (dtype == PGT_BF16) WG_HD(bf16_t);
    else WG_HD(half_t);
#undef WG_HD
#undef WG_NM
#undef WG_LAUNCH
    PGT_CHECK(e == hipSuccess, "window_attention3d: cannot reserve %d bytes of LDS: %s", lds, hipGetErrorString(e));
    PGT_LAUNCH_CHECK();
    return 0;
}
