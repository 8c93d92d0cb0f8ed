// Quantiser look-ups (K9, K10), BiSeNet glue (K13) and the driver-edge conversions.  All HBM-bound.
#include "common.h"
#include "pgt_internal.h"
#include "igemm_common.h"

namespace {

// first-index arg-extremum of a row: one wavefront per row, lanes stride the row, ties -> lowest index
template <bool IS_MIN>
__device__ __forceinline__ void wave_argext(float& best, int& bi) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        const bool take = IS_MIN ? (ob < best || (ob == best && oi < bi)) : (ob > best || (ob == best && oi < bi));
        if (take) { best = ob; bi = oi; }
    }
}

__global__ __launch_bounds__(256) void argmax_rows_kernel(const float* __restrict__ x, int ld, int rows, int K,
                                                          int* __restrict__ codes) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + (long)row * ld;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int j = lane; j < K; j += 64) {
        const float v = xr[j];
        if (v > best || bi == 0x7fffffff) { best = v; bi = j; }   // ascending j: keeps the first max
    }
    wave_argext<false>(best, bi);
    if (lane == 0) codes[row] = bi;
}

__global__ __launch_bounds__(256) void rq_argmin_kernel(const float* __restrict__ dot, int ld, const float* __restrict__ xn,
                                                        const float* __restrict__ en, int rows, int K,
                                                        int* __restrict__ codes) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* dr = dot + (long)row * ld;
    const float x2 = xn[row];
    float best = INFINITY;
    int bi = 0x7fffffff;
    for (int j = lane; j < K; j += 64) {
        // distance as the reference forms it: (|x|^2 + |e|^2) + (-2) * <x,e>  (addmm, alpha=-2)
        const float v = (x2 + en[j]) - 2.0f * dr[j];
        if (v < best || bi == 0x7fffffff) { best = v; bi = j; }
    }
    wave_argext<true>(best, bi);
    if (lane == 0) codes[row] = bi;
}

template <typename T>
__global__ void embed_rows_kernel(const float* __restrict__ book, int D, const int* __restrict__ codes, int rows,
                                  T* __restrict__ out, int ldo, int accumulate, T* __restrict__ resid, int ldres) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)rows * D) return;
    const int r = (int)(i / D), c = (int)(i % D);
    const float e = book[(long)codes[r] * D + c];
    T* o = out + (long)r * ldo + c;
    stf(o, accumulate ? ldf(o) + e : e);
    if (resid) {
        T* rr = resid + (long)r * ldres + c;
        stf(rr, ldf(rr) - e);
    }
}

// the same with 8 consecutive channels per thread (D, ldo, ldres multiples of 8, 16-byte aligned rows): 32-byte codebook
// reads, 16-byte (bf16) / 32-byte (fp32) row accesses; element-wise arithmetic identical to the scalar form
template <typename T>
__global__ __launch_bounds__(256) void embed_rows_vec8_kernel(const float* __restrict__ book, int D, const int* __restrict__ codes,
                                                              int rows, T* __restrict__ out, int ldo, int accumulate,
                                                              T* __restrict__ resid, int ldres) {
    const int cpr = D >> 3;                                   // 8-channel chunks per row
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)rows * cpr) return;
    const int r = (int)(i / cpr), c = (int)(i - (long)r * cpr) * 8;
    float e[8], v[8];
    load8<float>(book + (long)codes[r] * D + c, e);
    T* o = out + (long)r * ldo + c;
    if (accumulate) {
        load8<T>(o, v);
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] += e[k];
        store8<T>(o, v);
    } else {
        store8<T>(o, e);
    }
    if (resid) {
        T* rr = resid + (long)r * ldres + c;
        load8<T>(rr, v);
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] -= e[k];
        store8<T>(rr, v);
    }
}

// |x_r|^2 per row, fp32.  One wave per row; every lane adds its elements in ascending column order (16-byte loads when the
// row allows), then the 64 partial sums are combined by a butterfly.
template <typename T>
__global__ __launch_bounds__(256) void row_sumsq_kernel(const T* __restrict__ x, int ldx, int rows, int C,
                                                        float* __restrict__ out, int vec) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float s = 0.f;
    const T* xr = x + (long)row * ldx;
    if (vec) {
        for (int c = lane * 8; c < C; c += 512) {
            float v[8];
            load8<T>(xr + c, v);
#pragma unroll
            for (int k = 0; k < 8; ++k) s += v[k] * v[k];
        }
    } else {
        for (int c = lane; c < C; c += 64) { const float v = ldf(xr + c); s += v * v; }
    }
    s = wave_sum(s);
    if (lane == 0) out[row] = s;
}

template <typename T>
__global__ void maxpool_kernel(const T* __restrict__ x, int N, int H, int W, int C, T* __restrict__ y) {
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)N * Ho * Wo * C) return;
    const int c = (int)(i % C);
    long t = i / C;
    const int ox = (int)(t % Wo); t /= Wo;
    const int oy = (int)(t % Ho);
    const int n = (int)(t / Ho);
    float m = -INFINITY;
    for (int dy = 0; dy < 3; ++dy) {
        const int iy = oy * 2 - 1 + dy;
        if (iy < 0 || iy >= H) continue;
        for (int dx = 0; dx < 3; ++dx) {
            const int ix = ox * 2 - 1 + dx;
            if (ix < 0 || ix >= W) continue;
            m = fmaxf(m, ldf(x + (((long)n * H + iy) * W + ix) * C + c));
        }
    }
    stf(y + i, m);
}

template <typename T>
__global__ void gate_add_kernel(const T* __restrict__ x, int ldx, int N, int HW, int C, const T* __restrict__ gate,
                                const T* __restrict__ addvec, const T* __restrict__ addt, int ldt, T* __restrict__ y,
                                int ldy) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)N * HW * C) return;
    const int c = (int)(i % C);
    const long pix = i / C;
    const int n = (int)(pix / HW);
    float v = ldf(x + pix * ldx + c);
    if (gate) v *= ldf(gate + (long)n * C + c);
    if (addvec) v += ldf(addvec + (long)n * C + c);
    if (addt) v += ldf(addt + pix * ldt + c);
    stf(y + pix * ldy + c, v);
}

template <typename T>
__global__ void resize_bilinear_kernel(const T* __restrict__ x, int ldx, int N, int Hi, int Wi, int C, T* __restrict__ y,
                                       int ldy, int Ho, int Wo) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)N * Ho * Wo * C) return;
    const int c = (int)(i % C);
    long t = i / C;
    const int ox = (int)(t % Wo); t /= Wo;
    const int oy = (int)(t % Ho);
    const int n = (int)(t / Ho);
    // align_corners=True source coordinate, as ATen's area_pixel_compute_source_index
    const float sy = Ho > 1 ? (float)(Hi - 1) / (float)(Ho - 1) : 0.f;
    const float sx = Wo > 1 ? (float)(Wi - 1) / (float)(Wo - 1) : 0.f;
    const float fy = sy * oy, fx = sx * ox;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < Hi - 1 ? 1 : 0), x1 = x0 + (x0 < Wi - 1 ? 1 : 0);
    const float ly = fy - y0, lx = fx - x0, hy = 1.f - ly, hx = 1.f - lx;
    const T* base = x + (long)n * Hi * Wi * ldx + c;
    const float v00 = ldf(base + ((long)y0 * Wi + x0) * ldx), v01 = ldf(base + ((long)y0 * Wi + x1) * ldx);
    const float v10 = ldf(base + ((long)y1 * Wi + x0) * ldx), v11 = ldf(base + ((long)y1 * Wi + x1) * ldx);
    stf(y + (((long)n * Ho + oy) * Wo + ox) * ldy + c, hy * (hx * v00 + lx * v01) + ly * (hx * v10 + lx * v11));
}

template <typename S, typename D>
__global__ void copy2d_kernel(const S* __restrict__ src, int lds, D* __restrict__ dst, int ldd, long rows, int cols) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * cols) return;
    const long r = i / cols;
    const int c = (int)(i % cols);
    stf(dst + r * ldd + c, ldf(src + r * lds + c));
}

// same-type rows whose widths, pitches and base addresses are multiples of 16 bytes: one 16-byte chunk per thread (the
// channel-slice copies into the fusion blocks' concat buffers: 0.7 GB per launch at 1.75 TB/s with the element form, round 4)
__global__ void copy2d_vec16_kernel(const uint4* __restrict__ src, long lds16, uint4* __restrict__ dst, long ldd16, long rows,
                                    unsigned cpr) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * cpr) return;
    long r;
    unsigned c;
    if (rows * cpr < (1l << 32)) {      // (uniform) 32-bit division
        const unsigned iu = (unsigned)i;
        r = iu / cpr;
        c = iu - (unsigned)r * cpr;
    } else {
        r = i / cpr;
        c = (unsigned)(i - r * cpr);
    }
    dst[r * ldd16 + c] = src[r * lds16 + c];
}

template <typename T>
__global__ void prep_input_kernel(const void* __restrict__ src, int kind, int N, int H, int W, T* __restrict__ raw,
                                  T* __restrict__ norm) {
    constexpr int CP = 16 / (int)sizeof(T);      // channels of an output pixel: RGB + zeros up to one 16-byte chunk
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;   // pixel index
    if (i >= (long)N * H * W) return;
    const float mean[3] = {0.485f, 0.456f, 0.406f};
    const float stdv[3] = {0.229f, 0.224f, 0.225f};
    float r[3];
    if (kind == 0) {
        const uint8_t* s = (const uint8_t*)src + i * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) r[c] = (float)s[c] / 255.0f;
    } else {
        const long hw = (long)H * W;
        const long n = i / hw, p = i % hw;
        const float* s = (const float*)src + n * 3 * hw + p;
#pragma unroll
        for (int c = 0; c < 3; ++c) r[c] = s[c * hw];
    }
    union { uint4 v; T e[CP]; } a, b;
    a.v = b.v = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        stf(a.e + c, r[c]);
        stf(b.e + c, (r[c] - mean[c]) / stdv[c]);
    }
    if (raw) *reinterpret_cast<uint4*>(raw + i * CP) = a.v;
    if (norm) *reinterpret_cast<uint4*>(norm + i * CP) = b.v;
}

template <typename T>
__global__ void nhwc_to_nchw_kernel(const T* __restrict__ x, int ldx, int N, int H, int W, int C, float* __restrict__ y) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;   // output index, NCHW
    if (i >= (long)N * C * H * W) return;
    const long hw = (long)H * W;
    const long p = i % hw;
    const int c = (int)((i / hw) % C);
    const long n = i / (hw * C);
    y[i] = ldf(x + (n * hw + p) * ldx + c);
}

template <typename T>
__global__ void frame_to_u8_kernel(const T* __restrict__ x, int ldx, long npix, uint8_t* __restrict__ y) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix * 3) return;
    const long p = i / 3;
    const int c = (int)(i % 3);
    float v = ldf(x + p * ldx + c);
    v = fminf(fmaxf(v, 0.f), 1.f) * 255.0f;
    y[i] = (uint8_t)v;   // truncation toward zero, like np.array(float, np.uint8)
}


// dst frame i <- src frame idx[i]; a frame is `rows` rows of `row_bytes` bytes (multiple of 16) with independent
// row strides (channel slices of wider buffers are valid on both sides).  One thread per 16-byte chunk.
__global__ __launch_bounds__(256) void gather_frames_kernel(const char* __restrict__ src, long src_row_stride,
                                                            char* __restrict__ dst, long dst_row_stride,
                                                            const int* __restrict__ idx, long rows, int chunks) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * chunks) return;
    const long r = i / chunks;
    const int c = (int)(i % chunks);
    const int f = blockIdx.y;
    const long sf = idx[f];
    const uint4 v = *reinterpret_cast<const uint4*>(src + (sf * rows + r) * src_row_stride + (long)c * 16);
    *reinterpret_cast<uint4*>(dst + ((long)f * rows + r) * dst_row_stride + (long)c * 16) = v;
}

// fp32 (rows, cols) -> split planes (hi at dst, lo at dst + dlo), 8 columns per thread
__global__ __launch_bounds__(256) void x3_split_kernel(const float* __restrict__ src, int lds, x3p_t* __restrict__ dst, int ldd,
                                                       int dlo, long rows, int cols8) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * cols8) return;
    const long r = i / cols8;
    const int c = (int)(i % cols8) * 8;
    float f[8];
    *reinterpret_cast<float4*>(f) = *reinterpret_cast<const float4*>(src + r * lds + c);
    *reinterpret_cast<float4*>(f + 4) = *reinterpret_cast<const float4*>(src + r * lds + c + 4);
    uint4 hi, lo;
    split8(f, hi, lo);
    *reinterpret_cast<uint4*>(dst + r * ldd + c) = hi;
    *reinterpret_cast<uint4*>(dst + r * ldd + dlo + c) = lo;
}
// split planes -> fp32, or IEEE half (the value rounded to half: the hi plane, up to ties)
template <typename D>
__global__ __launch_bounds__(256) void x3_merge_kernel(const x3p_t* __restrict__ src, int lds, int slo, D* __restrict__ dst,
                                                       int ldd, long rows, int cols8) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * cols8) return;
    const long r = i / cols8;
    const int c = (int)(i % cols8) * 8;
    float f[8];
    merge8(*reinterpret_cast<const uint4*>(src + r * lds + c), *reinterpret_cast<const uint4*>(src + r * lds + slo + c), f);
    store8<D>(dst + r * ldd + c, f);
}

// ---- stage-I quantiser extras (reference: archs/tdcrqvae3_arch.py:330-352, :429-457) --------------------------------
// partial sums of (x - q)^2 over a flat chunk of elements (fp32 accumulation, fixed order inside a block)
template <typename T>
__global__ __launch_bounds__(256) void sqdiff_partial_kernel(const T* __restrict__ x, int ldx, const T* __restrict__ q, int ldq,
                                                             long rows, int cols, float* __restrict__ part) {
    __shared__ float sm[4];
    const long total = rows * cols;
    const long per = (total + gridDim.x - 1) / gridDim.x;
    const long i0 = (long)blockIdx.x * per, i1 = min(total, i0 + per);
    float s = 0.f;
    for (long i = i0 + threadIdx.x; i < i1; i += 256) {
        const long r = i / cols;
        const int c = (int)(i % cols);
        const float d = ldf(x + r * ldx + c) - ldf(q + r * ldq + c);
        s += d * d;
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (sm[0] + sm[1]) + (sm[2] + sm[3]);
}
__global__ void sum_mean_kernel(const float* __restrict__ part, int n, double count, float* __restrict__ out, float scale,
                                int accumulate) {
    if (threadIdx.x || blockIdx.x) return;
    double a = 0.0;
    for (int i = 0; i < n; ++i) a += (double)part[i];   // fixed order: deterministic
    const float v = scale * (float)(a / count);
    out[0] = accumulate ? out[0] + v : v;
}
// straight-through value of RQBottleneck.forward: x + (q - x), evaluated in that order in fp32 (reference :336)
template <typename T>
__global__ void straight_through_kernel(const T* __restrict__ x, int ldx, const T* __restrict__ q, int ldq, T* __restrict__ y,
                                        int ldy, long rows, int cols) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * cols) return;
    const long r = i / cols;
    const int c = (int)(i % cols);
    const float xv = ldf(x + r * ldx + c);
    stf(y + r * ldy + c, xv + (ldf(q + r * ldq + c) - xv));
}
// soft codes: softmax_j(-dist[r, j] / temp) with dist as in rq_argmin_kernel, plus the hard arg-min (one wave per row)
__global__ __launch_bounds__(256) void rq_soft_codes_kernel(const float* __restrict__ dot, int ld, const float* __restrict__ xn,
                                                            const float* __restrict__ en, int rows, int K, float inv_temp,
                                                            float* __restrict__ soft, int* __restrict__ codes) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* dr = dot + (long)row * ld;
    const float x2 = xn[row];
    float best = INFINITY;
    int bi = 0x7fffffff;
    for (int j = lane; j < K; j += 64) {
        const float v = (x2 + en[j]) - 2.0f * dr[j];
        if (v < best || bi == 0x7fffffff) { best = v; bi = j; }
    }
    wave_argext<true>(best, bi);          // best = min distance -> max logit = -best * inv_temp
    float sum = 0.f;
    for (int j = lane; j < K; j += 64) sum += expf(-((x2 + en[j]) - 2.0f * dr[j]) * inv_temp - (-best * inv_temp));
    sum = wave_sum(sum);
    float* so = soft + (long)row * K;
    for (int j = lane; j < K; j += 64) so[j] = expf(-((x2 + en[j]) - 2.0f * dr[j]) * inv_temp - (-best * inv_temp)) / sum;
    if (lane == 0) codes[row] = bi;
}

// One categorical draw per row by inverse CDF (what torch.multinomial(prob, 1) samples, reference
// archs/tdcrqvae3_arch.py:443-446): codes[r] = the first j with prob[r, 0] + ... + prob[r, j] > u[r] * sum(prob[r, :]).
// One wavefront per row; lane l owns the contiguous slice [l*per, (l+1)*per): slice sums, an exclusive scan over the lanes
// (fixed order), then the owning lane walks its slice.  Deterministic for a given u.
__global__ __launch_bounds__(256) void sample_rows_kernel(const float* __restrict__ prob, int ld, int rows, int K,
                                                          const float* __restrict__ u, int* __restrict__ codes) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* pr = prob + (long)row * ld;
    const int per = (K + 63) / 64;
    const int j0 = lane * per, j1 = min(K, j0 + per);
    float mine = 0.f;
    for (int j = j0; j < j1; ++j) mine += pr[j];
    float incl = mine;                                     // inclusive scan over lanes (Hillis-Steele, fixed order)
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const float t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
    }
    const float total = __shfl(incl, 63, 64);
    const float target = u[row] * total;
    // the exclusive sum is the PREVIOUS lane's inclusive sum, bit for bit (incl - mine would differ from it in the last place and
    // could put the walk's running sum above the target before an entry with probability > 0 is reached)
    float excl = __shfl_up(incl, 1, 64);
    if (lane == 0) excl = 0.f;
    // the owning lane: the first one whose inclusive sum exceeds the target (the last non-empty lane if rounding leaves none)
    const unsigned long long hit = __ballot(incl > target && j0 < K);
    const int owner = hit ? __ffsll((long long)hit) - 1 : min(63, (K - 1) / per);
    if (lane == owner) {
        // torch.multinomial never returns a category of probability zero: only entries with pr[j] > 0 are accepted, and the
        // fall-back (rounding left no entry above the target) is the last such entry of the row
        float c = excl;
        int pick = -1;
        for (int j = j0; j < j1; ++j) {
            c += pr[j];
            if (c > target && pr[j] > 0.f) { pick = j; break; }
        }
        if (pick < 0) {
            for (int j = K - 1; j >= 0; --j)
                if (pr[j] > 0.f) { pick = j; break; }
            if (pick < 0) pick = j1 - 1;
        }
        codes[row] = pick;
    }
}

// zero a (rows, row_bytes) byte matrix with row stride ldd bytes (16-byte granules): pad channels of concat buffers
__global__ void zero2d_kernel(char* __restrict__ dst, long ldd, long rows, int chunks) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * chunks) return;
    *reinterpret_cast<uint4*>(dst + (i / chunks) * ldd + (i % chunks) * 16) = make_uint4(0, 0, 0, 0);
}

inline dim3 grid1d(long n, int blk = 256) { return dim3((unsigned)((n + blk - 1) / blk)); }

// ---- weight repack (once per model, at load): reference (Cout, Cin, KH, KW) fp32 -> the conv kernels' B operand -------------
// K-major rows (Cout, KH*KW*Cin_pad), k = (ky*KW + kx)*Cin_pad + ci, input channels zero-padded to Cin_pad, an optional
// per-output-channel factor (eval-BatchNorm fold) applied in fp32 before the rounding to D.
template <typename D>
__global__ __launch_bounds__(256) void pack_weight_kernel(const float* __restrict__ w, int Cout, int Cin, int KH, int KW, int Cin_pad,
                                                          const float* __restrict__ scale, D* __restrict__ out) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const int taps = KH * KW;
    if (i >= (long)Cout * taps * Cin_pad) return;
    const int ci = (int)(i % Cin_pad);
    const int tap = (int)((i / Cin_pad) % taps);
    const int co = (int)(i / ((long)Cin_pad * taps));
    float v = 0.f;
    if (ci < Cin) v = __fmul_rn(w[((long)co * Cin + ci) * taps + tap], scale ? scale[co] : 1.f);
    stf(out + i, v);
}
// exact-weight form of the single-plane 16-bit layers (pgt_conv_desc::w2): (2 * ceil32(Cout), KH*KW*Cin_pad), per group of 32 output
// channels [32 rows w_hi | 32 rows (w - w_hi) * 2048]: the lo plane is scaled into the weights' own range (no subnormals), the
// kernels add acc_hi + acc_lo / 2048.  One thread per (padded output channel, k).
template <typename D>
__global__ __launch_bounds__(256) void pack_weight_w2_kernel(const float* __restrict__ w, int Cout, int Cin, int KH, int KW, int Cin_pad,
                                                             const float* __restrict__ scale, D* __restrict__ out) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const int taps = KH * KW;
    const int c32 = (Cout + 31) / 32 * 32;
    const long K = (long)taps * Cin_pad;
    if (i >= (long)c32 * K) return;
    const int ci = (int)(i % Cin_pad);
    const int tap = (int)((i / Cin_pad) % taps);
    const int co = (int)(i / K);
    float v = 0.f;
    if (ci < Cin && co < Cout) v = __fmul_rn(w[((long)co * Cin + ci) * taps + tap], scale ? scale[co] : 1.f);
    D hi;
    stf(&hi, v);
    uint16_t hb = __builtin_bit_cast(uint16_t, hi);
    uint32_t hw = hb;
    x3_opaque(hw);                                   // (the lo plane is taken against the STORED hi bits: see x3_opaque)
    hb = (uint16_t)hw;
    hi = __builtin_bit_cast(D, hb);
    const float lo = (v - ldf(&hi)) * 2048.f;        // exact in fp32: |v - hi| <= ulp(hi) / 2
    const long k = i % K;
    D* o = out + ((long)(co >> 5) * 64 + (co & 31)) * K + k;
    o[0] = hi;
    stf(o + 32 * K, lo);
}
// split-half forms.  Standard: (Cout, KH*KW*3*Cin_pad), per tap and 64-channel block [w_hi | w_hi | w_lo] (the kernels visit
// the block's input planes as [x_hi | x_lo | x_hi]).  Folded (Cout == 64): (128, KH*KW*2*Cin_pad), rows 0..63 [w_hi | w_hi],
// rows 64..127 [w_lo | 0] per tap and block (pgt_conv_desc::x3_fold).
__global__ __launch_bounds__(256) void pack_weight_x3_kernel(const float* __restrict__ w, int Cout, int Cin, int KH, int KW, int Cin_pad,
                                                             const float* __restrict__ scale, x3p_t* __restrict__ out, int fold) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const int taps = KH * KW;
    if (i >= (long)Cout * taps * Cin_pad) return;
    const int ci = (int)(i % Cin_pad);
    const int tap = (int)((i / Cin_pad) % taps);
    const int co = (int)(i / ((long)Cin_pad * taps));
    float v = 0.f;
    if (ci < Cin) v = __fmul_rn(w[((long)co * Cin + ci) * taps + tap], scale ? scale[co] : 1.f);
    const float hf = x3_hi_of(v);
    const _Float16 hi = (_Float16)hf;
    const _Float16 lo = (_Float16)sat_half(v - hf);
    const int nblk = Cin_pad / 64, blk = ci / 64, c = ci % 64;
    if (!fold) {
        x3p_t* o = out + (long)co * taps * 3 * Cin_pad + ((long)tap * nblk + blk) * 192 + c;
        o[0].v = hi;
        o[64].v = hi;
        o[128].v = lo;
    } else {
        const long rowlen = (long)taps * 2 * Cin_pad, off = ((long)tap * nblk + blk) * 128 + c;
        x3p_t* top = out + (long)co * rowlen + off;
        x3p_t* bot = out + (long)(64 + co) * rowlen + off;
        top[0].v = hi;
        top[64].v = hi;
        bot[0].v = lo;
        bot[64].v = (_Float16)0.f;
    }
}
// eval-mode BatchNorm2d folded into the preceding conv: s = gamma / sqrt(var + eps), bias' = (bias - mean) * s + beta
// (separately rounded operations, as the reference's ATen ops round them)
__global__ void fold_bn_kernel(const float* gamma, const float* beta, const float* mean, const float* var, float eps,
                               const float* bias, int C, float* scale, float* bias_out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= C) return;
    const float s = __fdiv_rn(gamma[i], __fsqrt_rn(__fadd_rn(var[i], eps)));
    scale[i] = s;
    bias_out[i] = __fadd_rn(__fmul_rn(__fsub_rn(bias ? bias[i] : 0.f, mean[i]), s), beta[i]);
}

}  // namespace

#define DT_DISPATCH(dtype, NAME, CALL_F32, CALL_BF16)              \
    do {                                                           \
        if ((dtype) == PGT_F32) { CALL_F32; }                      \
        else if ((dtype) == PGT_BF16) { CALL_BF16; }               \
        else PGT_CHECK(false, NAME ": bad dtype %d", (int)(dtype)); \
        PGT_LAUNCH_CHECK();                                        \
        return 0;                                                  \
    } while (0)

// the same over the three storage types, CALL written once with T_ as the element type
#define DT_DISPATCH_T(dtype, NAME, ...)                                 \
    do {                                                                \
        if ((dtype) == PGT_F32) { using T_ = float; __VA_ARGS__; }       \
        else if ((dtype) == PGT_BF16) { using T_ = bf16_t; __VA_ARGS__; } \
        else if ((dtype) == PGT_F16) { using T_ = half_t; __VA_ARGS__; }  \
        else PGT_CHECK(false, NAME ": bad dtype %d", (int)(dtype));      \
        PGT_LAUNCH_CHECK();                                              \
        return 0;                                                        \
    } while (0)

extern "C" int pgt_argmax_rows(const float* logits, int32_t ld, int32_t rows, int32_t K, int32_t* codes,
                               pgt_stream_t stream) {
    PGT_CHECK(logits && codes && K > 0, "argmax_rows: bad argument");
    hipLaunchKernelGGL(argmax_rows_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, logits, ld, rows, K, codes);
    PGT_LAUNCH_CHECK();
    return 0;
}

extern "C" int pgt_rq_argmin(const float* dot, int32_t ld, const float* xnorm, const float* enorm, int32_t rows,
                             int32_t K, int32_t* codes, pgt_stream_t stream) {
    PGT_CHECK(dot && xnorm && enorm && codes && K > 0, "rq_argmin: bad argument");
    hipLaunchKernelGGL(rq_argmin_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, dot, ld, xnorm, enorm, rows, K, codes);
    PGT_LAUNCH_CHECK();
    return 0;
}

extern "C" int pgt_embed_rows(int32_t dtype, const float* codebook, int32_t D, const int32_t* codes, int32_t rows,
                              void* out, int32_t ldo, int32_t accumulate, void* resid, int32_t ldres,
                              pgt_stream_t stream) {
    PGT_CHECK(codebook && codes && out, "embed_rows: null argument");
    hipStream_t st = (hipStream_t)stream;
    const bool vec = D % 8 == 0 && ldo % 8 == 0 && (((uintptr_t)out | (uintptr_t)codebook) & 15) == 0 &&
                     (!resid || (ldres % 8 == 0 && ((uintptr_t)resid & 15) == 0));
    if (vec) {
        const dim3 g8 = grid1d((long)rows * (D / 8));
        DT_DISPATCH_T(dtype, "embed_rows",
                      hipLaunchKernelGGL((embed_rows_vec8_kernel<T_>), g8, dim3(256), 0, st, codebook, D, codes, rows, (T_*)out, ldo, accumulate, (T_*)resid, ldres));
    }
    const dim3 g = grid1d((long)rows * D);
    DT_DISPATCH_T(dtype, "embed_rows",
                  hipLaunchKernelGGL((embed_rows_kernel<T_>), g, dim3(256), 0, st, codebook, D, codes, rows, (T_*)out, ldo, accumulate, (T_*)resid, ldres));
}

extern "C" int pgt_row_sumsq(int32_t dtype, const void* x, int32_t ldx, int32_t rows, int32_t C, float* out,
                             pgt_stream_t stream) {
    PGT_CHECK(x && out, "row_sumsq: null argument");
    hipStream_t st = (hipStream_t)stream;
    const dim3 g((rows + 3) / 4);
    const int vec = C % 8 == 0 && ldx % 8 == 0 && ((uintptr_t)x & 15) == 0;
    DT_DISPATCH_T(dtype, "row_sumsq",
                  hipLaunchKernelGGL((row_sumsq_kernel<T_>), g, dim3(256), 0, st, (const T_*)x, ldx, rows, C, out, vec));
}

extern "C" int pgt_maxpool3x3s2(int32_t dtype, const void* x, int32_t N, int32_t H, int32_t W, int32_t C, void* y,
                                pgt_stream_t stream) {
    PGT_CHECK(x && y, "maxpool: null argument");
    hipStream_t st = (hipStream_t)stream;
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const dim3 g = grid1d((long)N * Ho * Wo * C);
    DT_DISPATCH_T(dtype, "maxpool",
                  hipLaunchKernelGGL((maxpool_kernel<T_>), g, dim3(256), 0, st, (const T_*)x, N, H, W, C, (T_*)y));
}

extern "C" int pgt_gate_add(int32_t dtype, const void* x, int32_t ldx, int32_t N, int32_t HW, int32_t C,
                            const void* gate, const void* addvec, const void* addt, int32_t ldt, void* y, int32_t ldy,
                            pgt_stream_t stream) {
    PGT_CHECK(x && y, "gate_add: null argument");
    hipStream_t st = (hipStream_t)stream;
    const dim3 g = grid1d((long)N * HW * C);
    DT_DISPATCH_T(dtype, "gate_add",
                  hipLaunchKernelGGL((gate_add_kernel<T_>), g, dim3(256), 0, st, (const T_*)x, ldx, N, HW, C, (const T_*)gate, (const T_*)addvec, (const T_*)addt, ldt, (T_*)y, ldy));
}

extern "C" int pgt_resize_bilinear_ac(int32_t dtype, const void* x, int32_t ldx, int32_t N, int32_t Hi, int32_t Wi,
                                      int32_t C, void* y, int32_t ldy, int32_t Ho, int32_t Wo, pgt_stream_t stream) {
    PGT_CHECK(x && y, "resize_bilinear: null argument");
    hipStream_t st = (hipStream_t)stream;
    const dim3 g = grid1d((long)N * Ho * Wo * C);
    DT_DISPATCH_T(dtype, "resize_bilinear",
                  hipLaunchKernelGGL((resize_bilinear_kernel<T_>), g, dim3(256), 0, st, (const T_*)x, ldx, N, Hi, Wi, C, (T_*)y, ldy, Ho, Wo));
}

extern "C" int pgt_copy2d(int32_t src_dtype, const void* src, int32_t lds, int32_t dst_dtype, void* dst, int32_t ldd,
                          int64_t rows, int32_t cols, pgt_stream_t stream) {
    PGT_CHECK(src && dst, "copy2d: null argument");
    hipStream_t st = (hipStream_t)stream;
    const dim3 b(256);
    if (src_dtype == dst_dtype && rows > 0 && cols > 0) {
        const int es = src_dtype == PGT_F32 ? 4 : 2, v = 16 / es;
        if (cols % v == 0 && lds % v == 0 && ldd % v == 0 && (((uintptr_t)src | (uintptr_t)dst) & 15) == 0) {
            hipLaunchKernelGGL(copy2d_vec16_kernel, grid1d((long)rows * (cols / v)), b, 0, st, (const uint4*)src, (long)(lds / v),
                               (uint4*)dst, (long)(ldd / v), (long)rows, (unsigned)(cols / v));
            PGT_LAUNCH_CHECK();
            return 0;
        }
    }
    const dim3 g = grid1d((long)rows * cols);
    if (src_dtype == PGT_F32 && dst_dtype == PGT_F32)
        hipLaunchKernelGGL((copy2d_kernel<float, float>), g, b, 0, st, (const float*)src, lds, (float*)dst, ldd, (long)rows, cols);
    else if (src_dtype == PGT_F32 && dst_dtype == PGT_BF16)
        hipLaunchKernelGGL((copy2d_kernel<float, bf16_t>), g, b, 0, st, (const float*)src, lds, (bf16_t*)dst, ldd, (long)rows, cols);
    else if (src_dtype == PGT_BF16 && dst_dtype == PGT_F32)
        hipLaunchKernelGGL((copy2d_kernel<bf16_t, float>), g, b, 0, st, (const bf16_t*)src, lds, (float*)dst, ldd, (long)rows, cols);
    else if (src_dtype == PGT_BF16 && dst_dtype == PGT_BF16)
        hipLaunchKernelGGL((copy2d_kernel<bf16_t, bf16_t>), g, b, 0, st, (const bf16_t*)src, lds, (bf16_t*)dst, ldd, (long)rows, cols);
    else if (src_dtype == PGT_F32 && dst_dtype == PGT_F16)
        hipLaunchKernelGGL((copy2d_kernel<float, half_t>), g, b, 0, st, (const float*)src, lds, (half_t*)dst, ldd, (long)rows, cols);
    else if (src_dtype == PGT_F16 && dst_dtype == PGT_F32)
        hipLaunchKernelGGL((copy2d_kernel<half_t, float>), g, b, 0, st, (const half_t*)src, lds, (float*)dst, ldd, (long)rows, cols);
    else if (src_dtype == PGT_F16 && dst_dtype == PGT_F16)
        hipLaunchKernelGGL((copy2d_kernel<half_t, half_t>), g, b, 0, st, (const half_t*)src, lds, (half_t*)dst, ldd, (long)rows, cols);
    else if (src_dtype == PGT_BF16 && dst_dtype == PGT_F16)
        hipLaunchKernelGGL((copy2d_kernel<bf16_t, half_t>), g, b, 0, st, (const bf16_t*)src, lds, (half_t*)dst, ldd, (long)rows, cols);
    else if (src_dtype == PGT_F16 && dst_dtype == PGT_BF16)
        hipLaunchKernelGGL((copy2d_kernel<half_t, bf16_t>), g, b, 0, st, (const half_t*)src, lds, (bf16_t*)dst, ldd, (long)rows, cols);
    else
        PGT_CHECK(false, "copy2d: bad dtypes %d -> %d", src_dtype, dst_dtype);
    PGT_LAUNCH_CHECK();
    return 0;
}

extern "C" int pgt_prep_input(int32_t dtype, const void* src, int32_t src_kind, int32_t N, int32_t H, int32_t W,
                              void* raw, void* norm, pgt_stream_t stream) {
    PGT_CHECK(src && (raw || norm), "prep_input: null argument");
    PGT_CHECK(src_kind == 0 || src_kind == 1, "prep_input: src_kind must be 0 (u8 NHWC) or 1 (f32 NCHW)");
    PGT_CHECK(((((uintptr_t)raw) | ((uintptr_t)norm)) & 15) == 0, "prep_input: raw / norm must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const dim3 g = grid1d((long)N * H * W);
    DT_DISPATCH_T(dtype, "prep_input",
                  hipLaunchKernelGGL((prep_input_kernel<T_>), g, dim3(256), 0, st, src, src_kind, N, H, W, (T_*)raw, (T_*)norm));
}

extern "C" int pgt_nhwc_to_nchw_f32(int32_t dtype, const void* x, int32_t ldx, int32_t N, int32_t H, int32_t W,
                                    int32_t C, float* y, pgt_stream_t stream) {
    PGT_CHECK(x && y, "nhwc_to_nchw: null argument");
    hipStream_t st = (hipStream_t)stream;
    const dim3 g = grid1d((long)N * C * H * W);
    DT_DISPATCH_T(dtype, "nhwc_to_nchw",
                  hipLaunchKernelGGL((nhwc_to_nchw_kernel<T_>), g, dim3(256), 0, st, (const T_*)x, ldx, N, H, W, C, y));
}

extern "C" int pgt_frame_to_u8(int32_t dtype, const void* x, int32_t ldx, int32_t H, int32_t W, uint8_t* y,
                               pgt_stream_t stream) {
    PGT_CHECK(x && y, "frame_to_u8: null argument");
    hipStream_t st = (hipStream_t)stream;
    const long npix = (long)H * W;
    const dim3 g = grid1d(npix * 3);
    DT_DISPATCH_T(dtype, "frame_to_u8",
                  hipLaunchKernelGGL((frame_to_u8_kernel<T_>), g, dim3(256), 0, st, (const T_*)x, ldx, npix, y));
}

extern "C" int pgt_gather_frames(const void* src, int64_t src_row_stride, void* dst, int64_t dst_row_stride,
                                 const int32_t* idx, int32_t n_dst, int64_t rows, int32_t row_bytes,
                                 pgt_stream_t stream) {
    PGT_CHECK(src && dst && idx && n_dst > 0 && rows > 0, "gather_frames: bad argument");
    PGT_CHECK(row_bytes > 0 && row_bytes % 16 == 0 && src_row_stride % 16 == 0 && dst_row_stride % 16 == 0 &&
              ((((uintptr_t)src) | ((uintptr_t)dst)) & 15) == 0,
              "gather_frames: rows must be 16-byte aligned multiples of 16 bytes (row_bytes=%d)", row_bytes);
    const int chunks = row_bytes / 16;
    const long per_frame = rows * chunks;
    PGT_CHECK((per_frame + 255) / 256 < (1L << 31) && n_dst < 65536, "gather_frames: grid too large");
    hipLaunchKernelGGL(gather_frames_kernel, dim3((unsigned)((per_frame + 255) / 256), n_dst), dim3(256), 0,
                       (hipStream_t)stream, (const char*)src, (long)src_row_stride, (char*)dst, (long)dst_row_stride, idx,
                       (long)rows, chunks);
    PGT_LAUNCH_CHECK();
    return 0;
}

extern "C" int pgt_x3_split(const float* src, int32_t lds, void* dst, int32_t ldd, int32_t dst_lo, int64_t rows,
                            int32_t cols, pgt_stream_t stream) {
    PGT_CHECK(src && dst && cols % 8 == 0 && lds % 4 == 0 && ldd % 8 == 0 && dst_lo % 8 == 0 && dst_lo >= cols &&
              ldd >= dst_lo + cols && ((((uintptr_t)src) | ((uintptr_t)dst)) & 15) == 0, "x3_split: bad argument / alignment");
    hipLaunchKernelGGL(x3_split_kernel, grid1d((long)rows * (cols / 8)), dim3(256), 0, (hipStream_t)stream, src, lds,
                       (x3p_t*)dst, ldd, dst_lo, (long)rows, cols / 8);
    PGT_LAUNCH_CHECK();
    return 0;
}

extern "C" int pgt_x3_merge(const void* src, int32_t lds, int32_t src_lo, float* dst, int32_t ldd, int64_t rows,
                            int32_t cols, pgt_stream_t stream) {
    PGT_CHECK(src && dst && cols % 8 == 0 && lds % 8 == 0 && ldd % 4 == 0 && src_lo % 8 == 0 && lds >= src_lo + cols &&
              ((((uintptr_t)src) | ((uintptr_t)dst)) & 15) == 0, "x3_merge: bad argument / alignment");
    hipLaunchKernelGGL(x3_merge_kernel<float>, grid1d((long)rows * (cols / 8)), dim3(256), 0, (hipStream_t)stream,
                       (const x3p_t*)src, lds, src_lo, dst, ldd, (long)rows, cols / 8);
    PGT_LAUNCH_CHECK();
    return 0;
}

extern "C" int pgt_x3_to_half(const void* src, int32_t lds, int32_t src_lo, void* dst, int32_t ldd, int64_t rows,
                              int32_t cols, pgt_stream_t stream) {
    PGT_CHECK(src && dst && cols % 8 == 0 && lds % 8 == 0 && ldd % 8 == 0 && src_lo % 8 == 0 && lds >= src_lo + cols &&
              ((((uintptr_t)src) | ((uintptr_t)dst)) & 15) == 0, "x3_to_half: bad argument / alignment");
    hipLaunchKernelGGL(x3_merge_kernel<half_t>, grid1d((long)rows * (cols / 8)), dim3(256), 0, (hipStream_t)stream,
                       (const x3p_t*)src, lds, src_lo, (half_t*)dst, ldd, (long)rows, cols / 8);
    PGT_LAUNCH_CHECK();
    return 0;
}

extern "C" size_t pgt_commit_loss_workspace_bytes(void) { return 1024 * sizeof(float); }

extern "C" int pgt_commit_loss(int32_t dtype, const void* x, int32_t ldx, const void* q, int32_t ldq, int64_t rows,
                               int32_t cols, float* loss, float scale, int32_t accumulate, void* workspace,
                               size_t workspace_bytes, pgt_stream_t stream) {
    PGT_CHECK(x && q && loss && workspace && workspace_bytes >= 1024 * sizeof(float), "commit_loss: bad argument / workspace");
    hipStream_t st = (hipStream_t)stream;
    const long total = (long)rows * cols;
    const int nb = (int)(total < 1024L * 4096 ? (total + 4095) / 4096 : 1024);
    float* part = (float*)workspace;
    if (dtype == PGT_F32)
        hipLaunchKernelGGL((sqdiff_partial_kernel<float>), dim3(nb), dim3(256), 0, st, (const float*)x, ldx, (const float*)q, ldq, (long)rows, cols, part);
    else if (dtype == PGT_BF16)
        hipLaunchKernelGGL((sqdiff_partial_kernel<bf16_t>), dim3(nb), dim3(256), 0, st, (const bf16_t*)x, ldx, (const bf16_t*)q, ldq, (long)rows, cols, part);
    else
        PGT_CHECK(false, "commit_loss: bad dtype %d", dtype);
    PGT_LAUNCH_CHECK();
    hipLaunchKernelGGL(sum_mean_kernel, dim3(1), dim3(64), 0, st, part, nb, (double)total, loss, scale, accumulate);
    PGT_LAUNCH_CHECK();
    return 0;
}

extern "C" int pgt_straight_through(int32_t dtype, const void* x, int32_t ldx, const void* q, int32_t ldq, void* y,
                                    int32_t ldy, int64_t rows, int32_t cols, pgt_stream_t stream) {
    PGT_CHECK(x && q && y, "straight_through: null argument");
    hipStream_t st = (hipStream_t)stream;
    const dim3 g = grid1d((long)rows * cols);
    DT_DISPATCH_T(dtype, "straight_through",
                  hipLaunchKernelGGL((straight_through_kernel<T_>), g, dim3(256), 0, st, (const T_*)x, ldx, (const T_*)q, ldq, (T_*)y, ldy, (long)rows, cols));
}

extern "C" int pgt_rq_soft_codes(const float* dot, int32_t ld, const float* xnorm, const float* enorm, int32_t rows,
                                 int32_t K, float temp, float* soft, int32_t* codes, pgt_stream_t stream) {
    PGT_CHECK(dot && xnorm && enorm && soft && codes && K > 0 && temp > 0.f, "rq_soft_codes: bad argument");
    hipLaunchKernelGGL(rq_soft_codes_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, dot, ld, xnorm, enorm,
                       rows, K, 1.0f / temp, soft, codes);
    PGT_LAUNCH_CHECK();
    return 0;
}

extern "C" int pgt_sample_rows(const float* prob, int32_t ld, int32_t rows, int32_t K, const float* u, int32_t* codes,
                               pgt_stream_t stream) {
    PGT_CHECK(prob && u && codes && K > 0 && ld >= K, "sample_rows: bad argument");
    hipLaunchKernelGGL(sample_rows_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, prob, ld, rows, K, u, codes);
    PGT_LAUNCH_CHECK();
    return 0;
}

extern "C" int pgt_zero2d(void* dst, int64_t ldd_bytes, int64_t rows, int32_t row_bytes, pgt_stream_t stream) {
    PGT_CHECK(dst && rows >= 0 && row_bytes > 0 && row_bytes % 16 == 0 && ldd_bytes % 16 == 0 && (((uintptr_t)dst) & 15) == 0,
              "zero2d: rows must be 16-byte aligned multiples of 16 bytes (row_bytes=%d)", row_bytes);
    if (rows == 0) return 0;
    hipLaunchKernelGGL(zero2d_kernel, grid1d((long)rows * (row_bytes / 16)), dim3(256), 0, (hipStream_t)stream, (char*)dst,
                       (long)ldd_bytes, (long)rows, row_bytes / 16);
    PGT_LAUNCH_CHECK();
    return 0;
}

// ---- rounding defect of a packed 16-bit weight (the operand of pgt_mean_field_bias) -------------------
// defect_t[k][o] = sum over the filter taps of (w[o][k][tap] * scale[o] - packed[o][tap * Cin_pad + k])  (sum_taps), or one row
// per (tap, k) (taps reading different frames: the composed temporal mix).  Sums in double, one thread per (k, o).
namespace {
template <typename T>
__global__ __launch_bounds__(256) void weight_defect_kernel(const float* __restrict__ w, int Cout, int Cin, int taps, int Cin_pad,
                                                            const float* __restrict__ scale, const T* __restrict__ packed,
                                                            int sum_taps, float* __restrict__ defect_t) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long K = sum_taps ? Cin_pad : (long)taps * Cin_pad;
    if (i >= K * Cout) return;
    const int o = (int)(i % Cout);
    const long kk = i / Cout;
    const float sc = scale ? scale[o] : 1.f;
    double acc = 0.0;
    const int t0 = sum_taps ? 0 : (int)(kk / Cin_pad), t1 = sum_taps ? taps : t0 + 1;
    const int k = (int)(kk % Cin_pad);
    for (int t = t0; t < t1; ++t) {
        const double ref = k < Cin ? (double)(w[((long)o * Cin + k) * taps + t] * sc) : 0.0;
        acc += ref - (double)ldf(packed + (long)o * taps * Cin_pad + (long)t * Cin_pad + k);
    }
    defect_t[kk * Cout + o] = (float)acc;
}
}  // namespace

extern "C" int pgt_weight_defect(int32_t dtype, const float* w_oihw, int32_t Cout, int32_t Cin, int32_t KH, int32_t KW, int32_t Cin_pad,
                                 const float* out_scale, const void* packed, int32_t sum_taps, float* defect_t, pgt_stream_t stream) {
    PGT_CHECK(w_oihw && packed && defect_t && Cout > 0 && Cin > 0 && KH > 0 && KW > 0 && Cin_pad >= Cin, "weight_defect: bad argument");
    PGT_CHECK(dtype == PGT_BF16 || dtype == PGT_F16, "weight_defect: dtype %d (single-plane 16-bit weights only)", dtype);
    const int taps = KH * KW;
    const long total = (long)(sum_taps ? Cin_pad : (long)taps * Cin_pad) * Cout;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == PGT_F16)
        hipLaunchKernelGGL(weight_defect_kernel<half_t>, grid1d(total), dim3(256), 0, st, w_oihw, Cout, Cin, taps, Cin_pad, out_scale,
                           (const half_t*)packed, sum_taps, defect_t);
    else
        hipLaunchKernelGGL(weight_defect_kernel<bf16_t>, grid1d(total), dim3(256), 0, st, w_oihw, Cout, Cin, taps, Cin_pad, out_scale,
                           (const bf16_t*)packed, sum_taps, defect_t);
    PGT_LAUNCH_CHECK();
    return 0;
}

// ---- range telemetry of the IEEE-half tensors (PGT_F16 / the hi plane of PGT_F16X3) ---------------------------------------
// fp32 -> half stores of the kernels SATURATE at +-65504 instead of producing inf (common.h sat_half): a tensor that hits the
// limit is silently clamped.  This pass counts the elements of a (rows x cols, row stride ld) half matrix that sit at the
// limit or are not finite (|bits| >= 0x7bff) into *count (int32, added atomically: the caller zeroes it).
namespace {
__global__ __launch_bounds__(256) void count_saturated_kernel(const uint16_t* __restrict__ x, long ld, long rows, int chunks,
                                                              int* __restrict__ count) {
    const long total = rows * chunks;
    int c = 0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long r = i / chunks;
        const int k = (int)(i - r * chunks);
        const uint4 q = *reinterpret_cast<const uint4*>(x + r * ld + k * 8);
        const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) c += ((w[j] & 0x7fffu) >= 0x7bffu) + (((w[j] >> 16) & 0x7fffu) >= 0x7bffu);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(count, c);
}
}  // namespace

extern "C" int pgt_count_saturated(const void* x, int64_t ldx, int64_t rows, int32_t cols, int32_t* count, pgt_stream_t stream) {
    PGT_CHECK(x && count && rows >= 0 && cols > 0 && cols % 8 == 0 && ldx % 8 == 0 && (((uintptr_t)x) & 15) == 0,
              "count_saturated: cols=%d and ldx must be multiples of 8, x 16-byte aligned", cols);
    if (rows == 0) return 0;
    long blocks = (rows * (cols / 8) + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(count_saturated_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x, (long)ldx,
                       (long)rows, cols / 8, count);
    PGT_LAUNCH_CHECK();
    return 0;
}

extern "C" size_t pgt_packed_weight_bytes(int32_t dtype, int32_t Cout, int32_t Cin_pad, int32_t KH, int32_t KW, int32_t x3_fold) {
    const size_t k = (size_t)KH * KW * Cin_pad;
    if (dtype == PGT_F32) return (size_t)Cout * k * 4;
    if (dtype == PGT_BF16 || dtype == PGT_F16) return (x3_fold == 2 ? (size_t)((Cout + 31) / 32 * 64) : (size_t)Cout) * k * 2;
    if (dtype == PGT_F16X3) return x3_fold ? (size_t)128 * 2 * k * 2 : (size_t)Cout * 3 * k * 2;
    return 0;
}

extern "C" int pgt_pack_conv_weight(int32_t dtype, const float* w_oihw, int32_t Cout, int32_t Cin, int32_t KH, int32_t KW,
                                    int32_t Cin_pad, const float* out_scale, int32_t x3_fold, void* packed, pgt_stream_t stream) {
    PGT_CHECK(w_oihw && packed && Cout > 0 && Cin > 0 && KH > 0 && KW > 0 && Cin_pad >= Cin, "pack_conv_weight: bad argument");
    hipStream_t st = (hipStream_t)stream;
    const long total = (long)Cout * KH * KW * Cin_pad;
    const dim3 g = grid1d(total);
    if (dtype == PGT_F16X3) {
        PGT_CHECK(Cin_pad % 64 == 0, "pack_conv_weight: split-half weights come in 64-channel K blocks (Cin_pad=%d)", Cin_pad);
        PGT_CHECK(!x3_fold || Cout == 64, "pack_conv_weight: the folded form is for 64 output channels (Cout=%d)", Cout);
        hipLaunchKernelGGL(pack_weight_x3_kernel, g, dim3(256), 0, st, w_oihw, Cout, Cin, KH, KW, Cin_pad, out_scale, (x3p_t*)packed, x3_fold);
        PGT_LAUNCH_CHECK();
        return 0;
    }
    if (x3_fold == 2) {      // exact-weight form (pgt_conv_desc::w2)
        PGT_CHECK(dtype == PGT_F16 || dtype == PGT_BF16, "pack_conv_weight: the exact-weight form (x3_fold = 2) goes with PGT_F16 / PGT_BF16");
        const dim3 g2 = grid1d((long)((Cout + 31) / 32 * 32) * KH * KW * Cin_pad);
        if (dtype == PGT_F16)
            hipLaunchKernelGGL((pack_weight_w2_kernel<half_t>), g2, dim3(256), 0, st, w_oihw, Cout, Cin, KH, KW, Cin_pad, out_scale, (half_t*)packed);
        else
            hipLaunchKernelGGL((pack_weight_w2_kernel<bf16_t>), g2, dim3(256), 0, st, w_oihw, Cout, Cin, KH, KW, Cin_pad, out_scale, (bf16_t*)packed);
        PGT_LAUNCH_CHECK();
        return 0;
    }
    PGT_CHECK(!x3_fold, "pack_conv_weight: x3_fold = 1 goes with dtype PGT_F16X3");
    DT_DISPATCH_T(dtype, "pack_conv_weight",
                  hipLaunchKernelGGL((pack_weight_kernel<T_>), g, dim3(256), 0, st, w_oihw, Cout, Cin, KH, KW, Cin_pad, out_scale, (T_*)packed));
}

extern "C" int pgt_fold_batchnorm(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                                  float eps, const float* conv_bias, int32_t C, float* scale, float* bias, pgt_stream_t stream) {
    PGT_CHECK(gamma && beta && running_mean && running_var && scale && bias && C > 0, "fold_batchnorm: bad argument");
    hipLaunchKernelGGL(fold_bn_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, gamma, beta, running_mean, running_var, eps,
                       conv_bias, C, scale, bias);
    PGT_LAUNCH_CHECK();
    return 0;
}
