// Normalisation kernels (K6, K7, K11 of SURVEY.md §2.3): HBM-bound, 16-byte vector accesses,
// every thread keeps a fixed channel chunk so per-channel coefficients live in registers.
#include "common.h"
#include "pgt_sample.h"
#include "pgt_internal.h"

namespace {

constexpr int kT = 256;

// ---------------------------------------------------------------------------------------------
// GroupNorm statistics, stage 1: per (image, pixel-chunk) partial sum / sum-of-squares per group.
// Thread -> (pixel slot, 16-byte channel chunk); the chunk is fixed per thread, pixels strided.
// ---------------------------------------------------------------------------------------------
// X3: x is a split tensor (T = x3p_t planes): the value of a channel is hi + lo, the lo plane sits xlo elements
// after the hi plane in every pixel row.
template <typename T, int NQ, bool X3 = false>
__global__ __launch_bounds__(kT) void gn_partial_kernel(const T* __restrict__ x, int ldx, int HW, int C, int groups,
                                                        int pix_per_blk, float* __restrict__ part, int xlo = 0) {
    constexpr int CH = Vec16<T>::N;
    extern __shared__ __attribute__((aligned(16))) float sm[];  // [slots][C] sums, then [slots][C] sumsq
    const int QC = C / CH;
    const int PS = NQ > 1 ? 1 : kT / QC;
    const int tid = threadIdx.x;
    const int n = blockIdx.y, blk = blockIdx.x;
    const int slot = NQ > 1 ? 0 : tid / QC;
    const int q0 = NQ > 1 ? tid : tid % QC;
    const bool active = NQ > 1 ? true : tid < PS * QC;
    float s[NQ][CH], ss[NQ][CH];
#pragma unroll
    for (int j = 0; j < NQ; ++j)
#pragma unroll
        for (int e = 0; e < CH; ++e) s[j][e] = ss[j][e] = 0.f;
    const int p_begin = blk * pix_per_blk;
    const int p_end = min(HW, p_begin + pix_per_blk);
    if (active) {
        // kU pixels per trip, all loads issued before the first add: one 16-byte load in flight per thread cannot
        // cover the HBM latency (2.6 TB/s measured); accumulation order per thread is unchanged (ascending pixels).
        constexpr int kU = 4;
        for (int p = p_begin + slot; p < p_end; p += kU * PS) {
            uint4 v[kU][NQ], vl[X3 ? kU : 1][NQ];
#pragma unroll
            for (int u = 0; u < kU; ++u) {
                const int pu = p + u * PS;
                const T* row = x + ((long)n * HW + (pu < p_end ? pu : p)) * ldx;
#pragma unroll
                for (int j = 0; j < NQ; ++j) {
                    const int q = q0 + j * kT;
                    v[u][j] = q < QC ? *reinterpret_cast<const uint4*>(row + q * CH) : make_uint4(0, 0, 0, 0);
                    if constexpr (X3) vl[u][j] = q < QC ? *reinterpret_cast<const uint4*>(row + xlo + q * CH) : make_uint4(0, 0, 0, 0);
                }
            }
#pragma unroll
            for (int u = 0; u < kU; ++u) {
                if (p + u * PS >= p_end) break;
#pragma unroll
                for (int j = 0; j < NQ; ++j) {
                    if (q0 + j * kT < QC) {
                        float f[CH];
                        if constexpr (X3) merge8(v[u][j], vl[u][j], f);
                        else Vec16<T>::unpack(v[u][j], f);
#pragma unroll
                        for (int e = 0; e < CH; ++e) { s[j][e] += f[e]; ss[j][e] += f[e] * f[e]; }
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
            const int q = q0 + j * kT;
            if (q < QC) {
#pragma unroll
                for (int e = 0; e < CH; ++e) {
                    sm[slot * C + q * CH + e] = s[j][e];
                    sm[(PS + slot) * C + q * CH + e] = ss[j][e];
                }
            }
        }
    }
    __syncthreads();
    if (tid < groups) {
        const int cpg = C / groups;
        float a = 0.f, b = 0.f;
        for (int sl = 0; sl < PS; ++sl)
            for (int c = tid * cpg; c < (tid + 1) * cpg; ++c) { a += sm[sl * C + c]; b += sm[(PS + sl) * C + c]; }
        float* o = part + (((long)n * gridDim.x + blk) * groups + tid) * 2;
        o[0] = a;
        o[1] = b;
    }
}

// stage 2: fixed-order reduction of the partials in double, then per-(n,c) affine coefficients
__global__ void gn_finalize_kernel(const float* __restrict__ part, int nblk, int HW, int C, int groups, float eps,
                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                   float* __restrict__ scale, float* __restrict__ shift) {
    __shared__ float s_mean[64], s_rstd[64];
    const int n = blockIdx.x;
    const int cpg = C / groups;
    // one wavefront per group: lanes stride the pixel-chunk partials, then a fixed-pattern butterfly in
    // double (same order every run -> deterministic)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nwv = blockDim.x >> 6;
    for (int g = wv; g < groups; g += nwv) {
        double a = 0.0, b = 0.0;
        for (int k = lane; k < nblk; k += 64) {
            const float* o = part + (((long)n * nblk + k) * groups + g) * 2;
            a += (double)o[0];
            b += (double)o[1];
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            a += __shfl_xor(a, o, 64);
            b += __shfl_xor(b, o, 64);
        }
        if (lane != 0) continue;
        const double cnt = (double)HW * cpg;
        const double mean = a / cnt;
        double var = b / cnt - mean * mean;
        if (var < 0.0) var = 0.0;
        s_mean[g] = (float)mean;
        s_rstd[g] = (float)(1.0 / sqrt(var + (double)eps));
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const int g = c / cpg;
        const float sc = s_rstd[g] * gamma[c];
        scale[(long)n * C + c] = sc;
        shift[(long)n * C + c] = beta[c] - s_mean[g] * sc;
    }
}

// The same from the statistics the producing conv's epilogue left behind (igemm_common.h: gn_tile_reduce).  ws = [8 header
// floats: tile rows of sub-launch s][sub 0: N x maxblk x groups x 2][sub 1 ...], maxblk = HWsub / 64 slots reserved per
// image of which HWsub / rows are used.  One workgroup per image, one wavefront per group, fixed order, double.
__global__ void gn_finalize_conv_kernel(const float* __restrict__ ws, int N, int nsub, int HWsub, int C, int groups, float eps,
                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                        float* __restrict__ scale, float* __restrict__ shift) {
    __shared__ float s_mean[64], s_rstd[64];
    const int n = blockIdx.x;
    const int cpg = C / groups, maxblk = HWsub / 64;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nwv = blockDim.x >> 6;
    for (int g = wv; g < groups; g += nwv) {
        double a = 0.0, b = 0.0;
        for (int sub = 0; sub < nsub; ++sub) {
            const int rows = (int)ws[sub];
            const int nblk = rows > 0 ? HWsub / rows : 0;
            const float* part = ws + 8 + (((long)sub * N + n) * maxblk) * groups * 2;
            for (int k = lane; k < nblk; k += 64) {
                a += (double)part[((long)k * groups + g) * 2];
                b += (double)part[((long)k * groups + g) * 2 + 1];
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            a += __shfl_xor(a, o, 64);
            b += __shfl_xor(b, o, 64);
        }
        if (lane != 0) continue;
        const double cnt = (double)nsub * HWsub * cpg;
        const double mean = a / cnt;
        double var = b / cnt - mean * mean;
        if (var < 0.0) var = 0.0;
        s_mean[g] = (float)mean;
        s_rstd[g] = (float)(1.0 / sqrt(var + (double)eps));
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const int g = c / cpg;
        const float sc = s_rstd[g] * gamma[c];
        scale[(long)n * C + c] = sc;
        shift[(long)n * C + c] = beta[c] - s_mean[g] * sc;
    }
}

// Activation of CH values with the (uniform) switch outside the element loop.  bf16 tensors take the hardware
// exp2 / rcp forms of SiLU and sigmoid (relative error ~1e-6, far below the bf16 rounding of the result: this kernel
// is otherwise VALU-bound on the IEEE division); fp32 tensors keep the exact expf / division of the parity path.
template <typename T, int CH> __device__ __forceinline__ void act_vec(float* f, int act) {
    if constexpr (sizeof(T) == 2) {
        if (act == ACT_SILU) {
#pragma unroll
            for (int e = 0; e < CH; ++e)
                f[e] = f[e] * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * f[e]));
            return;
        }
        if (act == ACT_SIGMOID) {
#pragma unroll
            for (int e = 0; e < CH; ++e)
                f[e] = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * f[e]));
            return;
        }
    }
    switch (act) {
        case ACT_NONE: break;
        case ACT_RELU:
#pragma unroll
            for (int e = 0; e < CH; ++e) f[e] = apply_act(f[e], ACT_RELU);
            break;
        case ACT_GELU:
#pragma unroll
            for (int e = 0; e < CH; ++e) f[e] = apply_act(f[e], ACT_GELU);
            break;
        case ACT_SILU:
#pragma unroll
            for (int e = 0; e < CH; ++e) f[e] = apply_act(f[e], ACT_SILU);
            break;
        case ACT_LEAKY02:
#pragma unroll
            for (int e = 0; e < CH; ++e) f[e] = apply_act(f[e], ACT_LEAKY02);
            break;
        case ACT_SIGMOID:
#pragma unroll
            for (int e = 0; e < CH; ++e) f[e] = apply_act(f[e], ACT_SIGMOID);
            break;
        default: break;
    }
}

// y = act(x*scale[n,c] + shift[n,c]): one (pixel, 16-byte channel chunk) per thread, the per-(n, c) coefficients re-read per
// thread (they live in L1 / L2).  Measured on MI355X against a form with register-resident coefficients and a pixel loop per
// thread (round 3, same-box A/B): 5.97 vs 7.4 ms per forward over the 49 GroupNorm / AdaIN applies - far more independent
// loads in flight.
template <typename T, bool X3 = false>
__global__ __launch_bounds__(256) void affine_act_kernel(const T* __restrict__ x, int ldx, T* __restrict__ y, int ldy, long HW,
                                                         int C, long total, const float* __restrict__ scale,
                                                         const float* __restrict__ shift, int act, int xlo, int ylo) {
    constexpr int CH = Vec16<T>::N;
    const int QC = C / CH;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const long pix = i / QC;
    const int q = (int)(i - pix * QC);
    const long n = pix / HW;
    const T* row = x + pix * ldx + q * CH;
    const uint4 v = *reinterpret_cast<const uint4*>(row);
    uint4 vl = make_uint4(0, 0, 0, 0);
    if constexpr (X3) vl = *reinterpret_cast<const uint4*>(row + xlo);
    float sc[CH], sh[CH], f[CH];
#pragma unroll
    for (int e = 0; e < CH; e += 4) {
        *reinterpret_cast<float4*>(sc + e) = *reinterpret_cast<const float4*>(scale + n * C + q * CH + e);
        *reinterpret_cast<float4*>(sh + e) = *reinterpret_cast<const float4*>(shift + n * C + q * CH + e);
    }
    if constexpr (X3) merge8(v, vl, f);
    else Vec16<T>::unpack(v, f);
#pragma unroll
    for (int e = 0; e < CH; ++e) f[e] = f[e] * sc[e] + sh[e];
    T* orow = y + pix * ldy + q * CH;
    if constexpr (X3) {   // exact expf / division: this type exists for precision
        act_vec<float, CH>(f, act);
        uint4 hi, lo;
        split8(f, hi, lo);
        *reinterpret_cast<uint4*>(orow) = hi;
        *reinterpret_cast<uint4*>(orow + ylo) = lo;
    } else {
        act_vec<T, CH>(f, act);
        *reinterpret_cast<uint4*>(orow) = Vec16<T>::pack(f);
    }
}

// ---------------------------------------------------------------------------------------------
// LayerNorm: one wavefront per row, C/64 contiguous elements per lane, two-pass in registers.
// ---------------------------------------------------------------------------------------------
// EPL contiguous elements of one row <-> floats; 8/16-byte vector accesses when VEC (caller checked alignment)
template <typename T, int EPL, bool VEC> struct RowIO {
    static __device__ __forceinline__ void ld(const T* p, float* v) {
#pragma unroll
        for (int e = 0; e < EPL; ++e) v[e] = ldf(p + e);
    }
    static __device__ __forceinline__ void st(T* p, const float* v) {
#pragma unroll
        for (int e = 0; e < EPL; ++e) stf(p + e, v[e]);
    }
};
template <int EPL> struct RowIO<bf16_t, EPL, true> {   // EPL in {4, 8, 16}
    static __device__ __forceinline__ void ld(const bf16_t* p, float* v) {
        if constexpr (EPL == 4) {
            const uint2 q = *reinterpret_cast<const uint2*>(p);
            v[0] = __uint_as_float(q.x << 16); v[1] = __uint_as_float(q.x & 0xffff0000u);
            v[2] = __uint_as_float(q.y << 16); v[3] = __uint_as_float(q.y & 0xffff0000u);
        } else {
#pragma unroll
            for (int i = 0; i < EPL / 8; ++i) Vec16<bf16_t>::unpack(reinterpret_cast<const uint4*>(p)[i], v + 8 * i);
        }
    }
    static __device__ __forceinline__ void st(bf16_t* p, const float* v) {
        if constexpr (EPL == 4) {
            *reinterpret_cast<uint2*>(p) = make_uint2(f2bf2(v[0], v[1]), f2bf2(v[2], v[3]));
        } else {
#pragma unroll
            for (int i = 0; i < EPL / 8; ++i) reinterpret_cast<uint4*>(p)[i] = Vec16<bf16_t>::pack(v + 8 * i);
        }
    }
};
template <int EPL> struct RowIO<half_t, EPL, true> {   // EPL in {4, 8, 16}
    static __device__ __forceinline__ void ld(const half_t* p, float* v) {
        if constexpr (EPL == 4) {
            const uint2 q = *reinterpret_cast<const uint2*>(p);
            const halfx2 a = __builtin_bit_cast(halfx2, q.x), b = __builtin_bit_cast(halfx2, q.y);
            v[0] = (float)a.x; v[1] = (float)a.y; v[2] = (float)b.x; v[3] = (float)b.y;
        } else {
#pragma unroll
            for (int i = 0; i < EPL / 8; ++i) Vec16<half_t>::unpack(reinterpret_cast<const uint4*>(p)[i], v + 8 * i);
        }
    }
    static __device__ __forceinline__ void st(half_t* p, const float* v) {
        if constexpr (EPL == 4) {
            *reinterpret_cast<uint2*>(p) = make_uint2(f2h2(v[0], v[1]), f2h2(v[2], v[3]));
        } else {
#pragma unroll
            for (int i = 0; i < EPL / 8; ++i) reinterpret_cast<uint4*>(p)[i] = Vec16<half_t>::pack(v + 8 * i);
        }
    }
};
template <int EPL> struct RowIO<float, EPL, true> {    // EPL in {4, 8, 16}
    static __device__ __forceinline__ void ld(const float* p, float* v) {
#pragma unroll
        for (int i = 0; i < EPL / 4; ++i) *reinterpret_cast<float4*>(v + 4 * i) = reinterpret_cast<const float4*>(p)[i];
    }
    static __device__ __forceinline__ void st(float* p, const float* v) {
#pragma unroll
        for (int i = 0; i < EPL / 4; ++i) reinterpret_cast<float4*>(p)[i] = *reinterpret_cast<const float4*>(v + 4 * i);
    }
};

// X3 (T = x3p_t): x, y, pos and y2 are split rows, the lo plane xlo / ylo / plo / y2lo elements after the hi plane.
// RPW rows per wavefront: all of their loads are issued before the first reduction (one 8- or 16-byte load per lane in
// flight ran the C = 256 half layers at 3.5 TB/s; the per-row arithmetic and reduction order are unchanged).
template <typename T, int EPL, bool VEC, bool X3 = false, int RPW = 1>
__global__ __launch_bounds__(kT) void layernorm_kernel(const T* __restrict__ x, int ldx, int rows, int C,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       float eps, T* __restrict__ y, int ldy, const T* __restrict__ pos,
                                                       int ldpos, T* __restrict__ y2, int ldy2, int xlo = 0, int ylo = 0,
                                                       int plo = 0, int y2lo = 0) {
    typedef RowIO<T, EPL, VEC> IO;
    const int lane = threadIdx.x & 63;
    const int row0 = (blockIdx.x * (kT / 64) + (threadIdx.x >> 6)) * RPW;
    if (row0 >= rows) return;
    const int c0 = lane * EPL;
    float v[RPW][EPL], pv[RPW][EPL], gm[EPL], bt[EPL];
#pragma unroll
    for (int u = 0; u < RPW; ++u) {
        const long row = row0 + u < rows ? row0 + u : row0;       // a ragged tail re-reads the first row (not stored)
        IO::ld(x + row * ldx + c0, v[u]);
        if constexpr (X3) {
            float vl[EPL];
            IO::ld(x + row * ldx + xlo + c0, vl);
#pragma unroll
            for (int e = 0; e < EPL; ++e) v[u][e] += vl[e];
        }
        if (y2) {
            IO::ld(pos + row * ldpos + c0, pv[u]);
            if constexpr (X3) {
                float pl[EPL];
                IO::ld(pos + row * ldpos + plo + c0, pl);
#pragma unroll
                for (int e = 0; e < EPL; ++e) pv[u][e] += pl[e];
            }
        }
    }
    RowIO<float, EPL, VEC>::ld(gamma + c0, gm);
    RowIO<float, EPL, VEC>::ld(beta + c0, bt);
    auto store = [&](T* dst, int lo_off, const float* f) {
        if constexpr (X3) {
            float hf[EPL], lf[EPL];
#pragma unroll
            for (int e = 0; e < EPL; ++e) { hf[e] = x3_hi_of(f[e]); lf[e] = f[e] - hf[e]; }
            IO::st(dst, hf);
            IO::st(dst + lo_off, lf);
        } else {
            IO::st(dst, f);
        }
    };
#pragma unroll
    for (int u = 0; u < RPW; ++u) {
        const long row = row0 + u;
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < EPL; ++e) s += v[u][e];
        const float mean = wave_sum(s) / (float)C;
        float ss = 0.f;
#pragma unroll
        for (int e = 0; e < EPL; ++e) { const float d = v[u][e] - mean; ss += d * d; }
        const float rstd = 1.0f / sqrtf(wave_sum(ss) / (float)C + eps);
        if (row >= rows) continue;                                // (after the wave-wide reductions: all lanes take part)
#pragma unroll
        for (int e = 0; e < EPL; ++e) v[u][e] = (v[u][e] - mean) * rstd * gm[e] + bt[e];
        store(y + row * ldy + c0, ylo, v[u]);
        if (y2) {
#pragma unroll
            for (int e = 0; e < EPL; ++e) pv[u][e] += v[u][e];
            store(y2 + row * ldy2 + c0, y2lo, pv[u]);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// per-(n,c) mean / unbiased variance over HW (AdaIN statistics; also global average pooling)
// grid (C/64, N), block 256 = 4 pixel slots x 64 channels
// ---------------------------------------------------------------------------------------------
constexpr int kStatSlots = 16;   // pixel slots per workgroup (1024 threads = 16 slots x 64 channels)
template <typename T>
__global__ __launch_bounds__(kStatSlots * 64) void channel_stats_kernel(const T* __restrict__ x, int ldx, int HW, int C,
                                                                        float* __restrict__ mean, float* __restrict__ var) {
    __shared__ float sm[kStatSlots][64];
    const int cl = threadIdx.x & 63;
    const int c = blockIdx.x * 64 + cl;
    const int slot = threadIdx.x >> 6;
    const int n = blockIdx.y;
    const bool ok = c < C;
    const T* base = x + (long)n * HW * ldx + c;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;   // 4 independent chains: loads stay in flight
    if (ok) {
        int p = slot;
        for (; p + 3 * kStatSlots < HW; p += 4 * kStatSlots) {
            s0 += ldf(base + (long)p * ldx);
            s1 += ldf(base + (long)(p + kStatSlots) * ldx);
            s2 += ldf(base + (long)(p + 2 * kStatSlots) * ldx);
            s3 += ldf(base + (long)(p + 3 * kStatSlots) * ldx);
        }
        for (; p < HW; p += kStatSlots) s0 += ldf(base + (long)p * ldx);
    }
    sm[slot][cl] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int k = 0; k < kStatSlots; ++k) tot += sm[k][cl];
    const float mu = tot / (float)HW;
    __syncthreads();
    float q0 = 0.f, q1 = 0.f;
    if (ok && var) {
        int p = slot;
        for (; p + kStatSlots < HW; p += 2 * kStatSlots) {
            const float d0 = ldf(base + (long)p * ldx) - mu, d1 = ldf(base + (long)(p + kStatSlots) * ldx) - mu;
            q0 += d0 * d0;
            q1 += d1 * d1;
        }
        for (; p < HW; p += kStatSlots) { const float d = ldf(base + (long)p * ldx) - mu; q0 += d * d; }
    }
    sm[slot][cl] = q0 + q1;
    __syncthreads();
    if (slot == 0 && ok) {
        mean[(long)n * C + c] = mu;
        if (var) {
            float qt = 0.f;
#pragma unroll
            for (int k = 0; k < kStatSlots; ++k) qt += sm[k][cl];
            var[(long)n * C + c] = qt / (float)(HW > 1 ? HW - 1 : 1);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Mean-field compensation of the weight rounding of 16-bit convs / linears.  A layer y = W x + b run
// with W rounded to half / bf16 leaves the error (W - W16) x; its part that is the same at every pixel of a frame,
// (W - W16) mean(x), is a per-frame bias and is put back: bias_n = b + D mean_n, D[o][k] = sum over the filter taps of
// (W - W16)[o][k][tap].  mean_n is taken over a fixed sample of <= 1024 pixels of the frame (an estimate good to 1-2 % of a
// term that is itself 2^-12 of the output): kSampleRun consecutive pixels from each of up to kSampleCells equal cells.
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void sampled_mean_kernel(const T* __restrict__ x, int ldx, int HW, int C, float* __restrict__ mean) {
    __shared__ float sm[32][64 + 1];
    const int cc = threadIdx.x & 7, pl = threadIdx.x >> 3;
    const int c0 = blockIdx.x * 64 + cc * 8;
    const int n = blockIdx.y;
    int run, cells, cell;
    mean_sample_geometry(HW, &run, &cells, &cell);
    const int S = cells * run;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    if (c0 < C) {
        const T* base = x + (long)n * HW * ldx + c0;
#pragma unroll 4
        for (int i = pl; i < S; i += 32) {
            float v[8];
            RowIO<T, 8, sizeof(T) == 2>::ld(base + (long)mean_sample_pixel(i, cell, run) * ldx, v);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += v[e];
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) sm[pl][cc * 8 + e] = acc[e];
    __syncthreads();
    if (threadIdx.x < 64) {
        const int c = blockIdx.x * 64 + threadIdx.x;
        float tot = 0.f;
#pragma unroll
        for (int k = 0; k < 32; ++k) tot += sm[k][threadIdx.x];
        if (c < C) mean[(long)n * C + c] = tot / (float)S;
    }
}

// The same mean for a layer that reads the NORMALISED rows of x (pgt_ln_linear: the LayerNorm's affine part lives in its
// weights): mean[n][c] over the frame's sample of half((x[p][c] - mu_p) * rstd_p), mu_p / rstd_p the LayerNorm statistics of
// row p over its C channels.  Stage 1: a workgroup (4 waves) per (frame, slice of 128 sample rows), a wavefront per row, eight
// rows' loads in flight at a time (one row per iteration is a chain of dependent HBM round trips); partial sums per slice.
// Stage 2: the slices summed in order (deterministic).
constexpr int kRnSlice = 128;
template <typename T, int EPL>
__global__ __launch_bounds__(256) void sampled_rownorm_partial_kernel(const T* __restrict__ x, int ldx, int HW, int C, float eps,
                                                                      float* __restrict__ part) {
    __shared__ float red[4][64 * EPL];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = blockIdx.y, slice = blockIdx.x;
    int run, cells, cell;
    mean_sample_geometry(HW, &run, &cells, &cell);
    const int S = cells * run;
    float acc[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) acc[e] = 0.f;
    const T* base = x + (long)n * HW * ldx + lane * EPL;
    constexpr int NB = 8;
    for (int i0 = slice * kRnSlice + wave; i0 < (slice + 1) * kRnSlice && i0 < S; i0 += 4 * NB) {
        float v[NB][EPL];
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int i = i0 + 4 * u;
            RowIO<T, EPL, true>::ld(base + (long)mean_sample_pixel(i < S ? i : i0, cell, run) * ldx, v[u]);
        }
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            float s = 0.f;
#pragma unroll
            for (int e = 0; e < EPL; ++e) s += v[u][e];
            const float mu = wave_sum(s) / (float)C;
            float ss = 0.f;
#pragma unroll
            for (int e = 0; e < EPL; ++e) { v[u][e] -= mu; ss += v[u][e] * v[u][e]; }
            const float rstd = 1.0f / sqrtf(wave_sum(ss) / (float)C + eps);
            const int i = i0 + 4 * u;
            if (i < S && i < (slice + 1) * kRnSlice) {
#pragma unroll
                for (int e = 0; e < EPL; ++e) {
                    T r;
                    stf(&r, v[u][e] * rstd);          // the rounding the consuming kernel applies to its operand
                    acc[e] += ldf(&r);
                }
            }
        }
    }
#pragma unroll
    for (int e = 0; e < EPL; ++e) red[wave][lane * EPL + e] = acc[e];
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256)
        part[((long)n * gridDim.x + slice) * C + c] = red[0][c] + red[1][c] + red[2][c] + red[3][c];
}

__global__ __launch_bounds__(256) void sampled_rownorm_finish_kernel(const float* __restrict__ part, int nslice, int C, int S,
                                                                     float* __restrict__ mean) {
    const int n = blockIdx.x;
    for (int c = threadIdx.x; c < C; c += 256) {
        float t = 0.f;
        for (int k = 0; k < nslice; ++k) t += part[((long)n * nslice + k) * C + c];
        mean[(long)n * C + c] = t / (float)S;
    }
}

// out[r][o] = bias[o] + sum_k defect_t[k][o] * mean[r][k]: a workgroup = OL outputs x kMfbFrames frames, K cut in 256 / OL
// slices that are summed in slice order - one fixed order whatever the grid
constexpr int kMfbFrames = 4;
template <int OL>
__global__ __launch_bounds__(256) void mean_field_bias_kernel(const float* __restrict__ mean, const float* __restrict__ defect_t,
                                                             const float* __restrict__ bias, int R, int K, int Cout, float* __restrict__ out) {
    constexpr int KS = 256 / OL;
    extern __shared__ float smk[];                     // [kMfbFrames][K] means, then [KS][kMfbFrames][OL] partial sums
    const int ol = threadIdx.x % OL, ks = threadIdx.x / OL;
    const int o = blockIdx.x * OL + ol, r0 = blockIdx.y * kMfbFrames;
    for (int i = threadIdx.x; i < kMfbFrames * K; i += 256) {
        const int r = r0 + i / K;
        smk[i] = r < R ? mean[(long)r * K + i % K] : 0.f;
    }
    __syncthreads();
    float acc[kMfbFrames];
#pragma unroll
    for (int f = 0; f < kMfbFrames; ++f) acc[f] = 0.f;
    const int kq = (K + KS - 1) / KS, k0 = ks * kq, k1 = k0 + kq < K ? k0 + kq : K;
    if (o < Cout) {
#pragma unroll 16
        for (int k = k0; k < k1; ++k) {
            const float d = defect_t[(long)k * Cout + o];
#pragma unroll
            for (int f = 0; f < kMfbFrames; ++f) acc[f] += d * smk[f * K + k];
        }
    }
    __syncthreads();
    float* red = smk;
#pragma unroll
    for (int f = 0; f < kMfbFrames; ++f) red[(ks * kMfbFrames + f) * OL + ol] = acc[f];
    __syncthreads();
    if (ks == 0 && o < Cout) {
        const float b = bias ? bias[o] : 0.f;
#pragma unroll
        for (int f = 0; f < kMfbFrames; ++f) {
            if (r0 + f >= R) break;
            float t = 0.f;
            for (int q = 0; q < KS; ++q) t += red[(q * kMfbFrames + f) * OL + ol];
            out[(long)(r0 + f) * Cout + o] = b + t;
        }
    }
}

// Sampled mean AND the matrix-vector product in ONE launch (round 5: the two launches above are each at the launch-latency
// floor, 11-18 us, and sit in the dependency chain in front of every compensated layer): out[n][o] = bias[o] +
// sum_k defect_t[k][o] * mean_n[k], mean_n[k] over frame n's pixel sample of the layer's operand - x itself, or
// T(in_act(x * in_scale[n][k] + in_shift[n][k])) when the layer reads its operand through the fused GroupNorm apply
// (pgt_conv2d_affine_in: the normalised tensor never exists).  Grid (K / 64 slices, frames), as sampled_mean_kernel: workgroup
// (j, n) takes the sampled sums of its 64 channels, multiplies them into a partial output row part[n][j][:], and the workgroup
// that arrives LAST for frame n (a self-resetting counter per frame) adds the partial rows in slice order: one fixed order
// whatever the arrival order - deterministic without a second launch.
template <typename T>
__global__ __launch_bounds__(256) void frame_bias_kernel(const T* __restrict__ x, int ldx, int HW, int K, const float* __restrict__ in_scale,
                                                         const float* __restrict__ in_shift, int in_act,
                                                         const float* __restrict__ defect_t, const float* __restrict__ bias,
                                                         int Cout, float* __restrict__ out, float* __restrict__ part,
                                                         unsigned* __restrict__ counters, int csub, int scale_div, int cells_max) {
    __shared__ float sm[32][64 + 1];
    __shared__ float mean[64];
    __shared__ int is_last;
    const int cc = threadIdx.x & 7, pl = threadIdx.x >> 3;
    const int j = blockIdx.x, KS = gridDim.x, n = blockIdx.y;
    const int c0 = j * 64 + cc * 8;
    int run, cells, cell;
    mean_sample_geometry(HW, &run, &cells, &cell, cells_max);
    const int S = cells * run;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    if (c0 < K) {
        float sc[8], sh[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { sc[e] = 1.f; sh[e] = 0.f; }
        if (in_scale) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {      // (scale_div frames - bands - per image: one coefficient row per image)
                sc[e] = in_scale[(long)(n / scale_div) * K + c0 + e];
                sh[e] = in_shift[(long)(n / scale_div) * K + c0 + e];
            }
        }
        const T* base = x + (long)n * HW * ldx + c0;
#pragma unroll 4
        for (int i = pl; i < S; i += 32) {
            float v[8];
            RowIO<T, 8, sizeof(T) == 2>::ld(base + (long)mean_sample_pixel(i, cell, run) * ldx, v);
            if (in_scale) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = v[e] * sc[e] + sh[e];
                act_vec<T, 8>(v, in_act);
#pragma unroll
                for (int e = 0; e < 8; ++e) {      // the rounding the consuming kernel applies to its operand
                    T r;
                    stf(&r, v[e]);
                    v[e] = ldf(&r);
                }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += v[e];
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) sm[pl][cc * 8 + e] = acc[e];
    __syncthreads();
    if (threadIdx.x < 64) {
        float tot = 0.f;
#pragma unroll
        for (int k = 0; k < 32; ++k) tot += sm[k][threadIdx.x];
        mean[threadIdx.x] = j * 64 + (int)threadIdx.x < K ? tot / (float)S : 0.f;
    }
    __syncthreads();
    const int kn = K - j * 64 < 64 ? K - j * 64 : 64;
    float* prow = part + ((long)n * KS + j) * Cout;
    // Cross-workgroup hand-over WITHOUT device-scope fences: on this chip a release / acquire fence at agent scope writes back
    // and invalidates the XCD's whole L2 (8 XCDs, one L2 each) - measured: 768 workgroups fencing per call cost 3.8 % of the
    // step and evict the concurrent forward's working set.  Instead the partial rows are written and read with agent-scope
    // RELAXED atomics (sc1 accesses: coherent at the memory side, per access), and the counter is bumped only after this
    // workgroup's stores have completed (vmcnt(0) + barrier).
    for (int o = threadIdx.x; o < Cout; o += 256) {
        const float* d = defect_t + (long)j * 64 * Cout + o;
        float a = 0.f;
#pragma unroll 8
        for (int k = 0; k < kn; ++k) a += d[(long)k * Cout] * mean[k];
        __hip_atomic_store(prow + o, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this thread's partial values have reached the coherent level ...
    __syncthreads();                                   // ... every thread's have
    if (threadIdx.x == 0) is_last = atomicInc(counters + n, (unsigned)(KS - 1)) == (unsigned)(KS - 1);
    __syncthreads();
    if (!is_last) return;
    for (int o = threadIdx.x; o < Cout; o += 256) {
        float t = bias ? bias[o] : 0.f;
        const float* pr = part + (long)n * KS * Cout + o;
        for (int q = 0; q < KS; ++q) t += __hip_atomic_load(pr + (long)q * Cout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // slice order
        // out_groups > 1: (groups, N, Cout / groups) - one contiguous (N, csub) bias matrix per group of output columns
        const int grp = o / csub;
        out[((long)grp * gridDim.y + n) * csub + (o - grp * csub)] = t;
    }
}

__global__ void adain_affine_kernel(const float* mc, const float* vc, const float* ms, const float* vs, float eps,
                                    float* scale, float* shift, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float sc = sqrtf(vs[i] + eps) / sqrtf(vc[i] + eps);
    scale[i] = sc;
    shift[i] = ms[i] - mc[i] * sc;
}

inline int gn_pix_per_block(int N, int HW, int QC) {
    const int ps = QC >= kT ? 1 : kT / QC;
    int target_blocks = 2048 / (N > 0 ? N : 1);
    if (target_blocks < 1) target_blocks = 1;
    int ppb = (HW + target_blocks - 1) / target_blocks;
    const int min_ppb = ps * 8;
    if (ppb < min_ppb) ppb = min_ppb;
    return ppb;
}

}  // namespace

extern "C" size_t pgt_groupnorm_workspace_bytes(int32_t N, int32_t HW, int32_t C, int32_t groups) {
    // upper bound independent of dtype: smallest chunk count is C/8
    const int ppb = gn_pix_per_block(N, HW, C / 8 > 0 ? C / 8 : 1);
    const int ppb4 = gn_pix_per_block(N, HW, C / 4 > 0 ? C / 4 : 1);
    const int nb = (HW + (ppb < ppb4 ? ppb : ppb4) - 1) / (ppb < ppb4 ? ppb : ppb4);
    return (size_t)N * nb * groups * 2 * sizeof(float);
}

template <typename T, bool X3 = false>
static int gn_affine_impl(const void* x, int ldx, int N, int HW, int C, int groups, float eps, const float* gamma,
                          const float* beta, float* scale, float* shift, void* ws, size_t ws_bytes, hipStream_t st,
                          int xlo = 0) {
    constexpr int CH = Vec16<T>::N;
    PGT_CHECK(C % CH == 0 && ldx % CH == 0, "groupnorm: C=%d, ldx=%d must be multiples of %d", C, ldx, CH);
    PGT_CHECK(C % groups == 0 && groups <= 64, "groupnorm: C=%d not divisible by groups=%d (<=64)", C, groups);
    const int QC = C / CH;
    PGT_CHECK(QC <= 2 * kT, "groupnorm: C=%d too large", C);
    const int ppb = gn_pix_per_block(N, HW, QC);
    const int nb = (HW + ppb - 1) / ppb;
    PGT_CHECK((size_t)N * nb * groups * 2 * sizeof(float) <= ws_bytes, "groupnorm: workspace too small");
    const int ps = QC > kT ? 1 : kT / QC;
    const size_t lds = (size_t)2 * ps * C * sizeof(float);
    float* part = (float*)ws;
    if (QC > kT)
        hipLaunchKernelGGL((gn_partial_kernel<T, 2, X3>), dim3(nb, N), dim3(kT), lds, st, (const T*)x, ldx, HW, C, groups, ppb, part, xlo);
    else
        hipLaunchKernelGGL((gn_partial_kernel<T, 1, X3>), dim3(nb, N), dim3(kT), lds, st, (const T*)x, ldx, HW, C, groups, ppb, part, xlo);
    PGT_LAUNCH_CHECK();
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(N), dim3(1024), 0, st, part, nb, HW, C, groups, eps, gamma, beta, scale, shift);
    PGT_LAUNCH_CHECK();
    return 0;
}

extern "C" int pgt_groupnorm_affine(int32_t dtype, const void* x, int32_t ldx, int32_t N, int32_t HW, int32_t C,
                                    int32_t groups, float eps, const float* gamma, const float* beta, float* scale,
                                    float* shift, void* workspace, size_t workspace_bytes, pgt_stream_t stream) {
    PGT_CHECK(x && gamma && beta && scale && shift && workspace, "groupnorm: null argument");
    PGT_CHECK(((uintptr_t)x & 15) == 0, "groupnorm: x must be 16-byte aligned");
    if (dtype == PGT_F32) return gn_affine_impl<float>(x, ldx, N, HW, C, groups, eps, gamma, beta, scale, shift, workspace, workspace_bytes, (hipStream_t)stream);
    if (dtype == PGT_BF16) return gn_affine_impl<bf16_t>(x, ldx, N, HW, C, groups, eps, gamma, beta, scale, shift, workspace, workspace_bytes, (hipStream_t)stream);
    if (dtype == PGT_F16) return gn_affine_impl<half_t>(x, ldx, N, HW, C, groups, eps, gamma, beta, scale, shift, workspace, workspace_bytes, (hipStream_t)stream);
    PGT_CHECK(false, "groupnorm: bad dtype %d", dtype);
}

extern "C" int pgt_groupnorm_from_partials(const float* gn_workspace, int32_t N, int32_t nsub, int32_t HWsub, int32_t C,
                                           int32_t groups, float eps, const float* gamma, const float* beta, float* scale,
                                           float* shift, pgt_stream_t stream) {
    PGT_CHECK(gn_workspace && gamma && beta && scale && shift, "groupnorm_from_partials: null argument");
    PGT_CHECK(C % groups == 0 && groups <= 64 && nsub >= 1 && nsub <= 8 && HWsub % 64 == 0, "groupnorm_from_partials: bad geometry");
    hipLaunchKernelGGL(gn_finalize_conv_kernel, dim3(N), dim3(1024), 0, (hipStream_t)stream, gn_workspace, N, nsub, HWsub, C,
                       groups, eps, gamma, beta, scale, shift);
    PGT_LAUNCH_CHECK();
    return 0;
}

template <typename T, bool X3 = false>
static int affine_act_impl(const void* x, int ldx, void* y, int ldy, int N, int HW, int C, const float* scale,
                           const float* shift, int act, hipStream_t st, int xlo = 0, int ylo = 0) {
    constexpr int CH = Vec16<T>::N;
    PGT_CHECK(C % CH == 0 && ldx % CH == 0 && ldy % CH == 0, "affine_act: C/ldx/ldy must be multiples of %d", CH);
    const long total = (long)N * HW * (C / CH);
    hipLaunchKernelGGL((affine_act_kernel<T, X3>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, (const T*)x, ldx, (T*)y, ldy,
                       (long)HW, C, total, scale, shift, act, xlo, ylo);
    PGT_LAUNCH_CHECK();
    return 0;
}

extern "C" int pgt_affine_act(int32_t dtype, const void* x, int32_t ldx, void* y, int32_t ldy, int32_t N, int32_t HW,
                              int32_t C, const float* scale, const float* shift, int32_t act, pgt_stream_t stream) {
    PGT_CHECK(x && y && scale && shift, "affine_act: null argument");
    PGT_CHECK((((uintptr_t)x | (uintptr_t)y) & 15) == 0, "affine_act: x and y must be 16-byte aligned");
    if (dtype == PGT_F32) return affine_act_impl<float>(x, ldx, y, ldy, N, HW, C, scale, shift, act, (hipStream_t)stream);
    if (dtype == PGT_BF16) return affine_act_impl<bf16_t>(x, ldx, y, ldy, N, HW, C, scale, shift, act, (hipStream_t)stream);
    if (dtype == PGT_F16) return affine_act_impl<half_t>(x, ldx, y, ldy, N, HW, C, scale, shift, act, (hipStream_t)stream);
    PGT_CHECK(false, "affine_act: bad dtype %d", dtype);
}

extern "C" int pgt_groupnorm_affine_x3(const void* x, int32_t ldx, int32_t x_lo, int32_t N, int32_t HW, int32_t C,
                                       int32_t groups, float eps, const float* gamma, const float* beta, float* scale,
                                       float* shift, void* workspace, size_t workspace_bytes, pgt_stream_t stream) {
    PGT_CHECK(x && gamma && beta && scale && shift && workspace, "groupnorm_x3: null argument");
    PGT_CHECK(((uintptr_t)x & 15) == 0 && x_lo % 8 == 0 && x_lo >= C && ldx >= x_lo + C, "groupnorm_x3: bad planes (ldx=%d x_lo=%d C=%d)", ldx, x_lo, C);
    return gn_affine_impl<x3p_t, true>(x, ldx, N, HW, C, groups, eps, gamma, beta, scale, shift, workspace, workspace_bytes, (hipStream_t)stream, x_lo);
}

extern "C" int pgt_affine_act_x3(const void* x, int32_t ldx, int32_t x_lo, void* y, int32_t ldy, int32_t y_lo, int32_t N,
                                 int32_t HW, int32_t C, const float* scale, const float* shift, int32_t act,
                                 pgt_stream_t stream) {
    PGT_CHECK(x && y && scale && shift, "affine_act_x3: null argument");
    PGT_CHECK((((uintptr_t)x | (uintptr_t)y) & 15) == 0 && x_lo % 8 == 0 && y_lo % 8 == 0 && ldx >= x_lo + C && ldy >= y_lo + C &&
              x_lo >= C && y_lo >= C, "affine_act_x3: bad planes");
    return affine_act_impl<x3p_t, true>(x, ldx, y, ldy, N, HW, C, scale, shift, act, (hipStream_t)stream, x_lo, y_lo);
}

template <typename T, bool X3 = false>
static int layernorm_impl(const void* x, int ldx, int rows, int C, const float* gamma, const float* beta, float eps,
                          void* y, int ldy, const void* pos, int ldpos, void* y2, int ldy2, hipStream_t st, int xlo = 0,
                          int ylo = 0, int plo = 0, int y2lo = 0) {
    const dim3 blk(kT);
    // 8/16-byte row accesses need 16-byte aligned rows on every tensor
    auto al = [](const void* ptr, int ld) { return ptr == nullptr || ((((uintptr_t)ptr) & 15) == 0 && ld % 8 == 0); };
    const bool vec = al(x, ldx) && al(y, ldy) && al(pos, ldpos) && al(y2, ldy2) && al(gamma, 8) && al(beta, 8) &&
                     xlo % 8 == 0 && ylo % 8 == 0 && plo % 8 == 0 && y2lo % 8 == 0;
    // rows per wavefront: enough bytes in flight per lane (2-byte types: 32 B, split rows read two planes)
#define LN_LAUNCH3(EPL, VEC, RPW)                                                                                         \
    hipLaunchKernelGGL((layernorm_kernel<T, EPL, VEC, X3, RPW>), dim3((rows + 4 * (RPW) - 1) / (4 * (RPW))), blk, 0, st,    \
                       (const T*)x, ldx, rows, C, gamma, beta, eps, (T*)y, ldy, (const T*)pos, ldpos, (T*)y2, ldy2, xlo, ylo, plo, y2lo)
#define LN_LAUNCH2(EPL, VEC) do { constexpr int rpw_ = (EPL) * (int)sizeof(T) * (X3 ? 2 : 1) >= 32 ? 1 : ((EPL) * (int)sizeof(T) * (X3 ? 2 : 1) >= 16 ? 2 : 4); \
                                  LN_LAUNCH3(EPL, VEC, rpw_); } while (0)
#define LN_LAUNCH(EPL) do { if (vec && EPL >= 4) LN_LAUNCH2(EPL, (EPL >= 4)); else LN_LAUNCH2(EPL, false); } while (0)
    switch (C) {
        case 64: LN_LAUNCH(1); break;
        case 128: LN_LAUNCH(2); break;
        case 256: LN_LAUNCH(4); break;
        case 512: LN_LAUNCH(8); break;
        case 1024: LN_LAUNCH(16); break;
        default: PGT_CHECK(false, "layernorm: C=%d unsupported (64,128,256,512,1024)", C);
    }
#undef LN_LAUNCH
#undef LN_LAUNCH2
#undef LN_LAUNCH3
    PGT_LAUNCH_CHECK();
    return 0;
}

extern "C" int pgt_layernorm_x3(const void* x, int32_t ldx, int32_t x_lo, int32_t rows, int32_t C, const float* gamma,
                                const float* beta, float eps, void* y, int32_t ldy, int32_t y_lo, const void* pos,
                                int32_t ldpos, int32_t pos_lo, void* y2, int32_t ldy2, int32_t y2_lo, pgt_stream_t stream) {
    PGT_CHECK(x && y && gamma && beta, "layernorm_x3: null argument");
    PGT_CHECK(!y2 || pos, "layernorm_x3: y2 requested without pos");
    PGT_CHECK(x_lo >= C && y_lo >= C && ldx >= x_lo + C && ldy >= y_lo + C, "layernorm_x3: bad planes");
    PGT_CHECK(!y2 || (pos_lo >= C && y2_lo >= C && ldpos >= pos_lo + C && ldy2 >= y2_lo + C), "layernorm_x3: bad pos / y2 planes");
    return layernorm_impl<x3p_t, true>(x, ldx, rows, C, gamma, beta, eps, y, ldy, pos, ldpos, y2, ldy2, (hipStream_t)stream,
                                        x_lo, y_lo, pos_lo, y2_lo);
}

extern "C" int pgt_layernorm(int32_t dtype, const void* x, int32_t ldx, int32_t rows, int32_t C, const float* gamma,
                             const float* beta, float eps, void* y, int32_t ldy, const void* pos, int32_t ldpos,
                             void* y2, int32_t ldy2, pgt_stream_t stream) {
    PGT_CHECK(x && y && gamma && beta, "layernorm: null argument");
    PGT_CHECK(!y2 || pos, "layernorm: y2 requested without pos");
    if (dtype == PGT_F32) return layernorm_impl<float>(x, ldx, rows, C, gamma, beta, eps, y, ldy, pos, ldpos, y2, ldy2, (hipStream_t)stream);
    if (dtype == PGT_BF16) return layernorm_impl<bf16_t>(x, ldx, rows, C, gamma, beta, eps, y, ldy, pos, ldpos, y2, ldy2, (hipStream_t)stream);
    if (dtype == PGT_F16) return layernorm_impl<half_t>(x, ldx, rows, C, gamma, beta, eps, y, ldy, pos, ldpos, y2, ldy2, (hipStream_t)stream);
    PGT_CHECK(false, "layernorm: bad dtype %d", dtype);
}

extern "C" int pgt_channel_stats(int32_t dtype, const void* x, int32_t ldx, int32_t N, int32_t HW, int32_t C,
                                 float* mean, float* var_unbiased, pgt_stream_t stream) {
    PGT_CHECK(x && mean, "channel_stats: null argument");
    const dim3 grid((C + 63) / 64, N), blk(kStatSlots * 64);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == PGT_F32)
        hipLaunchKernelGGL((channel_stats_kernel<float>), grid, blk, 0, st, (const float*)x, ldx, HW, C, mean, var_unbiased);
    else if (dtype == PGT_BF16)
        hipLaunchKernelGGL((channel_stats_kernel<bf16_t>), grid, blk, 0, st, (const bf16_t*)x, ldx, HW, C, mean, var_unbiased);
    else if (dtype == PGT_F16)
        hipLaunchKernelGGL((channel_stats_kernel<half_t>), grid, blk, 0, st, (const half_t*)x, ldx, HW, C, mean, var_unbiased);
    else
        PGT_CHECK(false, "channel_stats: bad dtype %d", dtype);
    PGT_LAUNCH_CHECK();
    return 0;
}

extern "C" int pgt_sampled_channel_mean(int32_t dtype, const void* x, int32_t ldx, int32_t N, int32_t HW, int32_t C, float* mean,
                                        pgt_stream_t stream) {
    PGT_CHECK(x && mean && N >= 1 && HW >= 1, "sampled_channel_mean: null argument");
    PGT_CHECK(C % 8 == 0 && ldx % 8 == 0 && ((uintptr_t)x & 15) == 0, "sampled_channel_mean: C=%d and ldx=%d must be multiples of 8, x 16-byte aligned", C, ldx);
    const dim3 grid((C + 63) / 64, N), blk(256);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == PGT_F32)
        hipLaunchKernelGGL((sampled_mean_kernel<float>), grid, blk, 0, st, (const float*)x, ldx, HW, C, mean);
    else if (dtype == PGT_BF16)
        hipLaunchKernelGGL((sampled_mean_kernel<bf16_t>), grid, blk, 0, st, (const bf16_t*)x, ldx, HW, C, mean);
    else if (dtype == PGT_F16)
        hipLaunchKernelGGL((sampled_mean_kernel<half_t>), grid, blk, 0, st, (const half_t*)x, ldx, HW, C, mean);
    else
        PGT_CHECK(false, "sampled_channel_mean: bad dtype %d", dtype);
    PGT_LAUNCH_CHECK();
    return 0;
}

extern "C" size_t pgt_sampled_rownorm_workspace_bytes(int32_t N, int32_t HW, int32_t C) {
    int run, cells, cell;
    if (N < 1 || HW < 1 || C < 1) return 0;
    mean_sample_geometry(HW, &run, &cells, &cell);
    return (size_t)N * ((cells * run + kRnSlice - 1) / kRnSlice) * C * sizeof(float);
}

extern "C" int pgt_sampled_rownorm_mean(int32_t dtype, const void* x, int32_t ldx, int32_t N, int32_t HW, int32_t C, float eps,
                                        float* mean, void* workspace, pgt_stream_t stream) {
    PGT_CHECK(x && mean && workspace && N >= 1 && HW >= 1, "sampled_rownorm_mean: null argument");
    PGT_CHECK((C == 256 || C == 512) && ldx % 8 == 0 && ((uintptr_t)x & 15) == 0, "sampled_rownorm_mean: C=%d (256 or 512), ldx=%d (multiple of 8)", C, ldx);
    int run, cells, cell;
    mean_sample_geometry(HW, &run, &cells, &cell);
    const int S = cells * run, nslice = (S + kRnSlice - 1) / kRnSlice;
    hipStream_t st = (hipStream_t)stream;
    float* part = (float*)workspace;
#define SRM(T_, E_) hipLaunchKernelGGL((sampled_rownorm_partial_kernel<T_, E_>), dim3(nslice, N), dim3(256), 0, st, (const T_*)x, ldx, HW, C, eps, part)
    if (dtype == PGT_BF16) { if (C == 256) SRM(bf16_t, 4); else SRM(bf16_t, 8); }
    else if (dtype == PGT_F16) { if (C == 256) SRM(half_t, 4); else SRM(half_t, 8); }
    else PGT_CHECK(false, "sampled_rownorm_mean: bad dtype %d", dtype);
#undef SRM
    PGT_LAUNCH_CHECK();
    hipLaunchKernelGGL(sampled_rownorm_finish_kernel, dim3(N), dim3(256), 0, st, part, nslice, C, S, mean);
    PGT_LAUNCH_CHECK();
    return 0;
}

extern "C" int pgt_sampled_pixel(int32_t HW, int32_t i) {
    int run, cells, cell;
    if (HW < 1 || i < 0) return -1;
    mean_sample_geometry(HW, &run, &cells, &cell);
    if (i >= cells * run) return -1;
    return mean_sample_pixel(i, cell, run);
}

extern "C" int pgt_sampled_pixel_cells(int32_t HW, int32_t sample_cells, int32_t i) {
    int run, cells, cell;
    if (HW < 1 || i < 0) return -1;
    mean_sample_geometry(HW, &run, &cells, &cell, sample_cells);
    if (i >= cells * run) return -1;
    return mean_sample_pixel(i, cell, run);
}

extern "C" int pgt_mean_field_bias(const float* mean, const float* defect_t, const float* bias, int32_t R, int32_t K, int32_t Cout,
                                   float* out, pgt_stream_t stream) {
    PGT_CHECK(mean && defect_t && out && R >= 1 && K >= 1 && Cout >= 1, "mean_field_bias: null argument");
    PGT_CHECK(K <= 3840, "mean_field_bias: K=%d > 3840", K);
    size_t lds = (size_t)kMfbFrames * K * sizeof(float);
    if (lds < 256 * kMfbFrames * sizeof(float)) lds = 256 * kMfbFrames * sizeof(float);
    const int fb = (R + kMfbFrames - 1) / kMfbFrames;
    hipStream_t st = (hipStream_t)stream;
    // narrow layers give their threads to the K axis instead (the 32-channel temporal mix has K = T * 2C up to 3072)
    if (Cout > 32) hipLaunchKernelGGL(mean_field_bias_kernel<64>, dim3((Cout + 63) / 64, fb), dim3(256), lds, st, mean, defect_t, bias, R, K, Cout, out);
    else if (Cout > 8) hipLaunchKernelGGL(mean_field_bias_kernel<32>, dim3((Cout + 31) / 32, fb), dim3(256), lds, st, mean, defect_t, bias, R, K, Cout, out);
    else hipLaunchKernelGGL(mean_field_bias_kernel<8>, dim3((Cout + 7) / 8, fb), dim3(256), lds, st, mean, defect_t, bias, R, K, Cout, out);
    PGT_LAUNCH_CHECK();
    return 0;
}

extern "C" size_t pgt_frame_bias_workspace_bytes(int32_t N, int32_t K, int32_t Cout) {
    if (N < 1 || K < 1 || Cout < 1) return 0;
    return (size_t)N * ((K + 63) / 64) * Cout * sizeof(float);
}

extern "C" int pgt_frame_bias(int32_t dtype, const void* x, int32_t ldx, int32_t N, int32_t HW, int32_t K, const float* in_scale,
                              const float* in_shift, int32_t in_act, const float* defect_t, const float* bias, int32_t Cout,
                              int32_t out_groups, int32_t scale_div, int32_t sample_cells, float* out, void* workspace,
                              size_t workspace_bytes, uint32_t* counters, pgt_stream_t stream) {
    PGT_CHECK(x && defect_t && out && workspace && counters && N >= 1 && HW >= 1 && Cout >= 1, "frame_bias: null argument");
    PGT_CHECK(out_groups >= 1 && Cout % out_groups == 0, "frame_bias: out_groups=%d must divide Cout=%d", out_groups, Cout);
    PGT_CHECK(scale_div >= 1 && N % scale_div == 0, "frame_bias: scale_div=%d must divide N=%d", scale_div, N);
    const int csub = Cout / out_groups;
    PGT_CHECK(K >= 8 && K % 8 == 0 && ldx % 8 == 0 && ((uintptr_t)x & 15) == 0, "frame_bias: K=%d (a multiple of 8) / ldx=%d / x alignment", K, ldx);
    PGT_CHECK((in_scale == nullptr) == (in_shift == nullptr), "frame_bias: in_scale and in_shift go together");
    PGT_CHECK(workspace_bytes >= pgt_frame_bias_workspace_bytes(N, K, Cout) && ((uintptr_t)workspace & 3) == 0, "frame_bias: workspace too small");
    const dim3 grid((K + 63) / 64, N), blk(256);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == PGT_BF16)
        hipLaunchKernelGGL((frame_bias_kernel<bf16_t>), grid, blk, 0, st, (const bf16_t*)x, ldx, HW, K, in_scale, in_shift, in_act, defect_t, bias, Cout, out, (float*)workspace, counters, csub, scale_div, sample_cells);
    else if (dtype == PGT_F16)
        hipLaunchKernelGGL((frame_bias_kernel<half_t>), grid, blk, 0, st, (const half_t*)x, ldx, HW, K, in_scale, in_shift, in_act, defect_t, bias, Cout, out, (float*)workspace, counters, csub, scale_div, sample_cells);
    else
        PGT_CHECK(false, "frame_bias: dtype %d (PGT_BF16 / PGT_F16: the compensated layers are the single-plane 16-bit ones)", dtype);
    PGT_LAUNCH_CHECK();
    return 0;
}

extern "C" int pgt_adain_affine(const float* mean_c, const float* var_c, const float* mean_s, const float* var_s,
                                float eps, float* scale, float* shift, int32_t n, pgt_stream_t stream) {
    PGT_CHECK(mean_c && var_c && mean_s && var_s && scale && shift, "adain_affine: null argument");
    hipLaunchKernelGGL(adain_affine_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, mean_c, var_c,
                       mean_s, var_s, eps, scale, shift, n);
    PGT_LAUNCH_CHECK();
    return 0;
}
