// Implicit-GEMM convolution / linear layer on MFMA for gfx950 (K1, K2, K3, K4 of SURVEY.md §2.3).
//
//   y[m, co] = epi( sum_{ky,kx,ci} x[n, iy, ix, ci] * w[co, (ky*KW+kx)*Cin + ci] + bias[co] )
//
// Activations are channels-last (N,H,W,C): a pixel's channels are contiguous, so the GEMM A-operand
// (M = N*Ho*Wo output pixels, K = KH*KW*Cin) is gathered 16 bytes at a time straight from HBM with
// no im2col buffer; zero padding, the asymmetric (0,1,0,1) pad of the stride-2 Downsample
// (reference: archs/tdcrqvae3_arch.py:67-76) and the nearest x2 up-sampling of Upsample
// (reference: :45-52) are folded into the gather's address computation.  Linear layers and 1x1
// convs are the KH=KW=1 case (reference: nn.Linear / nn.Conv2d(k=1) call sites listed in §8a).
//
// Tile: BM x BN outputs per 256-thread workgroup (4 waves in 2x2), K consumed 128 bytes per row per
// step (64 bf16 / 32 f32).  Global->register prefetch of tile t+1 overlaps the MFMAs of tile t;
// LDS rows are padded 128->144 B so the ds_read_b128 fragment reads and ds_write_b128 staging
// writes are bank-conflict free.  bf16 uses v_mfma_f32_32x32x16_bf16, f32 uses the exact
// v_mfma_f32_32x32x2_f32 (parity mode); both accumulate in fp32.
#include <cstdlib>

#include "common.h"
#include "pgt_internal.h"
#include "igemm_common.h"

namespace {

constexpr int kThreads = 256;
constexpr long kV2MinBlocks = 1L << 40;  // v2 (LDS-DMA) is opt-in (kernel=2): measured no faster than v1 on MI355X (profiles/r1_igemm_pmc.md)
constexpr int kV2MinK = 256;
constexpr int kRowBytes = 128;          // K bytes per tile row
constexpr int kRowStride = kRowBytes + 16;  // padded LDS rows (144 B): conflict-free ds_read_b128 / ds_write_b128.
// (An unpadded XOR-swizzled image — swz128, used by igemm2 — fits 6 workgroups/CU but measured ~20 % slower in the
// whole model: profiles/r1_v3 vs r1_v4 kernel stats.)
__device__ __forceinline__ int lds_off(int row, int c) { return row * kRowStride + c * 16; }

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
    static __device__ __forceinline__ void run(const uint4& a, const uint4& b, f32x16& acc) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a),
                                                       __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
    }
};
template <> struct Mma<half_t> {
    static __device__ __forceinline__ void run(const uint4& a, const uint4& b, f32x16& acc) { acc = mma16<half_t>(a, b, acc); }
};
template <> struct Mma<float> {
    // lane half h holds 4 consecutive k of an 8-wide k group; MFMA j pairs (j, 4+j): the k order is
    // the same for A and B so the contraction is unchanged.
    static __device__ __forceinline__ void run(const uint4& a, const uint4& b, f32x16& acc) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
    }
};

template <typename T, int BM, int BN, bool SK>   // SK: split-K slice kernel (compile-time so the common path keeps its registers)
__global__ __launch_bounds__(kThreads) void igemm_kernel(ConvP p) {
    constexpr int ES = sizeof(T);
    constexpr int CH = 16 / ES;            // elements per 16-byte chunk
    constexpr int BK = kRowBytes / ES;     // K elements per tile step
    constexpr int AR = BM / 32;            // A rows staged per thread
    constexpr int BR = BN / 32;
    constexpr int MI = BM / 64;            // 32x32 MFMA tiles per wave along M
    constexpr int NI = BN / 64;
    constexpr int kTileBytes = (BM + BN) * kRowStride;
    constexpr int kStageBytes = (BM / 2) * (BN + 4) * 4;   // fp32 epilogue staging (two half-tile passes)
    __shared__ __attribute__((aligned(256))) char smem[kTileBytes > kStageBytes ? kTileBytes : kStageBytes];
    char* As = smem;
    char* Bs = smem + BM * kRowStride;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    // XCD-aware tile order: the 8 XCDs each take a contiguous run of tiles (n fastest) so the
    // workgroups that share an A tile / a halo run on the same L2 (bijective for any grid size).
    const int nblk = p.nbm * p.nbn;
    // split-K: workgroup = (K slice, output tile); slice s walks K tiles [s*kt_per_split, ...) and writes its
    // raw fp32 partial tile to workspace slab s (summed in fixed order by splitk_epilogue_kernel).
    const int split = SK ? (int)blockIdx.x / nblk : 0;
    const int bid = (int)blockIdx.x - split * nblk;
    const int xcd = bid & 7, q = nblk >> 3, r = nblk & 7;
    const int sw = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int m0 = (sw / p.nbn) * BM;
    const int n0 = (sw % p.nbn) * BN;
    if (SK) p.y += (long)split * p.M * p.Cout * 4;

    const int cc = tid & 7;    // 16-byte chunk column of this thread within the 128-byte tile row
    const int r0 = tid >> 3;   // first tile row of this thread (then +32, +64, ...)

    // per-row output-pixel coordinates for the A gather
    int iy0[AR], ix0[AR], pbase[AR];
    const int Hv = p.H << p.ups, Wv = p.W << p.ups;
#pragma unroll
    for (int i = 0; i < AR; ++i) {
        const int m = m0 + r0 + 32 * i;
        if (m < p.M) {
            const int ox = m % p.Wo;
            const int t = m / p.Wo;
            const int oy = t % p.Ho;
            const int n = t / p.Ho;
            iy0[i] = oy * p.stride - p.pad_t;
            ix0[i] = ox * p.stride - p.pad_l;
            pbase[i] = n * p.H * p.W;
        } else {
            iy0[i] = -(1 << 28);
            ix0[i] = 0;
            pbase[i] = 0;
        }
    }
    // K position of this thread's chunk: tap (ky,kx) and channel c, advanced incrementally
    const int nk_all = (p.K + BK - 1) / BK;
    const int kt0 = SK ? split * p.kt_per_split : 0;
    const int nk = SK ? min(nk_all, kt0 + p.kt_per_split) : nk_all;   // one past the last K tile
    int kg = kt0 * BK + cc * CH;
    int c = kg % p.Cin;
    int tap = kg / p.Cin;
    int ky = tap / p.KW, kx = tap % p.KW;

    // Per-row gather state for the CURRENT filter tap, recomputed only when this thread's tap changes (every
    // Cin/BK K tiles): byte offset of the source pixel (tensors are < 2 GiB), -1 = zero padding / out of range.
    int a_offs[AR];
    auto retap = [&]() {
        const bool kval = ky < p.KH;
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            const int iy = iy0[i] + ky, ix = ix0[i] + kx;
            const bool ok = kval && iy >= 0 && iy < Hv && ix >= 0 && ix < Wv;
            a_offs[i] = ok ? (pbase[i] + (iy >> p.ups) * p.W + (ix >> p.ups)) * p.ldx * ES : -1;
        }
    };
    retap();
    int b_offs[BR];   // byte offset of this thread's weight row (Cout*K*ES < 2 GiB), -1 = row >= Cout
#pragma unroll
    for (int i = 0; i < BR; ++i) {
        const int n = n0 + r0 + 32 * i;
        b_offs[i] = n < p.Cout ? n * p.K * ES : -1;
    }

    uint4 ra[AR], rb[BR];
    auto gload = [&]() {
        const char* xa = p.x + c * ES;
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            uint4 v = make_uint4(0, 0, 0, 0);
            if (a_offs[i] >= 0) v = *reinterpret_cast<const uint4*>(xa + a_offs[i]);
            ra[i] = v;
        }
        const char* wb = p.w + (long)kg * ES;
        const bool kin = kg < p.K;
#pragma unroll
        for (int i = 0; i < BR; ++i) {
            uint4 v = make_uint4(0, 0, 0, 0);
            if (kin && b_offs[i] >= 0) v = *reinterpret_cast<const uint4*>(wb + b_offs[i]);
            rb[i] = v;
        }
    };
    auto advance = [&]() {
        kg += BK;
        c += BK;
        if (c >= p.Cin) {
            do {
                c -= p.Cin;
                if (++kx == p.KW) { kx = 0; ++ky; }
            } while (c >= p.Cin);
            retap();
        }
    };
    auto sstore = [&]() {
#pragma unroll
        for (int i = 0; i < AR; ++i)
            *reinterpret_cast<uint4*>(As + lds_off(r0 + 32 * i, cc)) = ra[i];
#pragma unroll
        for (int i = 0; i < BR; ++i)
            *reinterpret_cast<uint4*>(Bs + lds_off(r0 + 32 * i, cc)) = rb[i];
    };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    gload();
    advance();
    sstore();
    __syncthreads();
    const char* a_rd = As + (wm * (BM / 2) + (lane & 31)) * kRowStride + (lane >> 5) * 16;
    const char* b_rd = Bs + (wn * (BN / 2) + (lane & 31)) * kRowStride + (lane >> 5) * 16;
    // the last K tile may hold fewer than four 32-byte steps (K = 36 or 196 of the 3-channel input layers, padded to 4 channels):
    // the steps past K multiply zeros; that tile is taken out of the main loop (whose body stays as it is) and runs only its steps
    const int ns_last = (p.K - (nk_all - 1) * BK + BK / 4 - 1) / (BK / 4);
    const bool tail = ns_last < 4 && nk == nk_all;
    const int nk_main = tail ? nk - 1 : nk;
    for (int kt = kt0; kt < nk_main; ++kt) {
        const bool more = kt + 1 < nk;
        if (more) {
            gload();
            advance();
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            uint4 af[MI], bfr[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i) af[i] = *reinterpret_cast<const uint4*>(a_rd + i * 32 * kRowStride + s * 32);
#pragma unroll
            for (int j = 0; j < NI; ++j) bfr[j] = *reinterpret_cast<const uint4*>(b_rd + j * 32 * kRowStride + s * 32);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) Mma<T>::run(af[i], bfr[j], acc[i][j]);
        }
        __syncthreads();
        if (more) {
            sstore();
            __syncthreads();
        }
    }

    if (tail) {
#pragma unroll 1
        for (int s = 0; s < ns_last; ++s) {
            uint4 af[MI], bfr[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i) af[i] = *reinterpret_cast<const uint4*>(a_rd + i * 32 * kRowStride + s * 32);
#pragma unroll
            for (int j = 0; j < NI; ++j) bfr[j] = *reinterpret_cast<const uint4*>(b_rd + j * 32 * kRowStride + s * 32);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) Mma<T>::run(af[i], bfr[j], acc[i][j]);
        }
        __syncthreads();
    }

    // epilogue: bias, activation, residual, (post-ReLU | SFT modulate), store
    if (p.vec_epi) {
        // Vectorised path: act(acc + bias) is staged through LDS (fp32, two half-tile passes) so that every
        // thread then handles 8 consecutive output channels of one pixel: 16-byte residual/dec/shift loads
        // and 16-byte stores instead of 2-byte scattered ones.
        constexpr int SROW = BN + 4;
        float* stage = reinterpret_cast<float*>(smem);
        static_assert((BM / 2) * SROW * 4 <= (int)sizeof(smem), "stage buffer must fit the tile LDS");
        const T* res = reinterpret_cast<const T*>(p.res);
        const T* dec = reinterpret_cast<const T*>(p.dec);
        const T* shf = reinterpret_cast<const T*>(p.shift);
        const bool gn = p.gn_part != nullptr;   // epilogue GroupNorm statistics (uniform)
        float gs[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) gs[e] = 0.f;
        for (int pass = 0; pass < 2; ++pass) {
            if (wm == pass) {
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    const int cl = wn * (BN / 2) + j * 32 + (lane & 31);
                    const int n = n0 + cl;
                    const float bv = (p.bias && n < p.Cout) ? bias_of(p, m0)[n] : 0.f;
#pragma unroll
                    for (int i = 0; i < MI; ++i)
#pragma unroll
                        for (int e = 0; e < 16; ++e) {
                            const int rl = i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                            stage[rl * SROW + cl] = acc[i][j][e] + bv;
                        }
                }
            }
            __syncthreads();
            for (int cidx = tid; cidx < (BM / 2) * (BN / 8); cidx += kThreads) {
                const int rl = cidx / (BN / 8), c8 = (cidx % (BN / 8)) * 8;
                const int m = m0 + pass * (BM / 2) + rl, n = n0 + c8;
                if (m >= p.M || n >= p.Cout) continue;
                float v[8];
                *reinterpret_cast<float4*>(v) = *reinterpret_cast<const float4*>(stage + rl * SROW + c8);
                *reinterpret_cast<float4*>(v + 4) = *reinterpret_cast<const float4*>(stage + rl * SROW + c8 + 4);
                apply_act8(v, p.act);
                if (p.epi == 1) {
                    float d[8], s[8];
                    load8<T>(dec + (long)m * p.ld_dec + n, d);
                    load8<T>(shf + (long)m * p.ld_shift + n, s);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = d[e] + p.sft_w * (d[e] * v[e] + s[e]);
                } else {
                    if (res) {
                        float r[8];
                        load8<T>(res + (long)m * p.ldr + n, r);
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] += r[e];
                    }
                    if (p.post_relu) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
                    }
                }
                if (gn) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) { gs[e] += v[e]; gs[8 + e] += v[e] * v[e]; }
                }
                if (p.out_split) {      // (fp32 launches only) split-half planes straight from the fp32 values
                    uint4 hi, lo;
                    split8(v, hi, lo);
                    half_t* yh = reinterpret_cast<half_t*>(p.y) + out_row(p, m) * p.ldy + n;
                    *reinterpret_cast<uint4*>(yh) = hi;
                    *reinterpret_cast<uint4*>(yh + p.ylo) = lo;
                } else if (p.out_f32) store8<float>(reinterpret_cast<float*>(p.y) + out_row(p, m) * p.ldy + n, v);
                else store8<T>(reinterpret_cast<T*>(p.y) + out_row(p, m) * p.ldy + n, v);
            }
            __syncthreads();
        }
        if (gn) gn_tile_reduce<BN / 8, 4>(p, gs, stage, tid, m0, n0, BM);
        return;
    }
    T* y = reinterpret_cast<T*>(p.y);
    const T* res = reinterpret_cast<const T*>(p.res);
    const T* dec = reinterpret_cast<const T*>(p.dec);
    const T* shf = reinterpret_cast<const T*>(p.shift);
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int n = n0 + wn * (BN / 2) + j * 32 + (lane & 31);
        if (n >= p.Cout) continue;
        const float bv = p.bias ? bias_of(p, m0)[n] : 0.f;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wm * (BM / 2) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                if (m >= p.M) continue;
                float v = apply_act(acc[i][j][e] + bv, p.act);
                if (p.epi == 1) {
                    // SFT: out = dec + w * (dec * scale + shift)  (reference: pgtformer_arch.py:478-479)
                    const float d = ldf(dec + (long)m * p.ld_dec + n);
                    const float s = ldf(shf + (long)m * p.ld_shift + n);
                    v = d + p.sft_w * (d * v + s);
                } else {
                    if (res) v += ldf(res + (long)m * p.ldr + n);
                    if (p.post_relu) v = v > 0.f ? v : 0.f;
                }
                if (p.out_f32) reinterpret_cast<float*>(p.y)[out_row(p, m) * p.ldy + n] = v;
                else stf(y + out_row(p, m) * p.ldy + n, v);
            }
        }
    }
}

template <typename T, int BM, int BN> int launch(const ConvP& p0, hipStream_t st) {
    ConvP p = p0;
    p.nbm = (p.M + BM - 1) / BM;
    p.nbn = (p.Cout + BN - 1) / BN;
    if (p.gn_part)
        PGT_CHECK(p.vec_epi && p.splitk <= 1 && p.gn_hw % BM == 0 && BN % p.gn_cpg == 0,
                  "pgt_conv2d: GroupNorm statistics need the 16-byte epilogue, no split-K and HW %% %d == 0 (HW=%d)", BM, p.gn_hw);
    if (p.splitk > 1) {
        if constexpr (BM == 64 && BN == 64)
            hipLaunchKernelGGL((igemm_kernel<T, BM, BN, true>), dim3(p.nbm * p.nbn * p.splitk), dim3(kThreads), 0, st, p);
        else
            PGT_CHECK(false, "split-K runs on 64x64 tiles only");
    } else {
        hipLaunchKernelGGL((igemm_kernel<T, BM, BN, false>), dim3(p.nbm * p.nbn), dim3(kThreads), 0, st, p);
    }
    PGT_LAUNCH_CHECK();
    return 0;
}

// Sum the split-K slabs in slice order (deterministic) and apply the conv epilogue, 8 channels per thread.
template <typename T>
__global__ __launch_bounds__(256) void splitk_epilogue_kernel(ConvP p, const float* __restrict__ ws, int slices) {
    const long nchunk = (long)p.M * (p.Cout / 8);
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nchunk) return;
    const int m = (int)(i / (p.Cout / 8)), n = (int)(i % (p.Cout / 8)) * 8;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 0.f;
    for (int s = 0; s < slices; ++s) {
        float t[8];
        load8<float>(ws + ((long)s * p.M + m) * p.Cout + n, t);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += t[e];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = apply_act(v[e] + (p.bias ? bias_of(p, m)[n + e] : 0.f), p.act);
    if (p.epi == 1) {
        float d[8], sh[8];
        load8<T>(reinterpret_cast<const T*>(p.dec) + (long)m * p.ld_dec + n, d);
        load8<T>(reinterpret_cast<const T*>(p.shift) + (long)m * p.ld_shift + n, sh);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = d[e] + p.sft_w * (d[e] * v[e] + sh[e]);
    } else {
        if (p.res) {
            float rr[8];
            load8<T>(reinterpret_cast<const T*>(p.res) + (long)m * p.ldr + n, rr);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += rr[e];
        }
        if (p.post_relu) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
        }
    }
    if (p.out_f32) store8<float>(reinterpret_cast<float*>(p.y) + out_row(p, m) * p.ldy + n, v);
    else store8<T>(reinterpret_cast<T*>(p.y) + out_row(p, m) * p.ldy + n, v);
}

// Split-K policy: deep-K layers whose 64x64 tiling leaves most of the chip idle (32x32-resolution convs).
inline int choose_splitk(long M, int Cout, int K, int bk) {
    const long tiles = ((M + 63) / 64) * ((Cout + 63) / 64);
    const int nk = (K + bk - 1) / bk;
    if (tiles >= 512 || nk < 32) return 1;
    int s = (int)((1536 + tiles - 1) / tiles);
    if (s > 8) s = 8;
    while (s > 1 && nk / s < 8) --s;
    return s;
}

template <typename T> int dispatch(const ConvP& p, hipStream_t st, int force_bm, int force_bn) {
    // Tile choice measured with tools/bench_igemm.py on MI355X (profiles/r1_igemm_shapes_v1.txt): this
    // 2-barrier register-staged pipeline wants MANY resident workgroups per CU, so BM=64 wins almost
    // everywhere; BN=128 only pays once K is deep and there are still >= 2 workgroups per CU; Cout<=64
    // layers at 512x512 / 256x256 amortise their weight tile better with BM=128.
    int bm = 64, bn = 64;
    if (p.Cout <= 64) {
        if (p.K >= 128 && p.M >= 65536) bm = 128;
    } else if (p.K >= 1024 && (long)((p.M + 63) / 64) * ((p.Cout + 127) / 128) >= 512) {
        bn = 128;
    }
    if (force_bm) bm = force_bm;
    if (force_bn) bn = force_bn;
    if (bm == 128 && bn == 128) return launch<T, 128, 128>(p, st);
    if (bm == 128 && bn == 64) return launch<T, 128, 64>(p, st);
    if (bm == 64 && bn == 128) return launch<T, 64, 128>(p, st);
    return launch<T, 64, 64>(p, st);
}

}  // namespace

// number of K slices pgt_conv2d_ws would use for this layer (1 = single pass)
static int planned_splitk(const pgt_conv_desc* d) {
    if (d->splitk == 1 || d->kernel == 2 || d->scalar_epilogue || d->Cout % 8 != 0 || d->dtype == PGT_F16X3 || d->gn_groups > 0 || d->out_split || d->w2) return 1;
    const long M = (long)d->N * d->Ho * d->Wo;
    const int K = d->KH * d->KW * d->Cin;
    const int bk = d->dtype == PGT_F32 ? 32 : 64;
    if (d->splitk > 1) return d->splitk <= 16 && (K + bk - 1) / bk >= d->splitk ? d->splitk : 1;
    return choose_splitk(M, d->Cout, K, bk);
}

// statistics workspace of the call in flight on this thread (set by pgt_conv2d_gn around pgt_conv2d_ws)
static thread_local float* g_gn_ws = nullptr;
// per-(image, channel) affine + activation applied to the operand on load (set by pgt_conv2d_affine_in around pgt_conv2d_ws)
static thread_local const float* g_in_scale = nullptr;
static thread_local const float* g_in_shift = nullptr;
static thread_local int g_in_act = 0;

// A/B switch of the ring kernel for the 64-channel 3x3 layers at W >= 128 (igemm8.hip): PGT_C64_RING=0 keeps igemm6 on them
static bool use_ring_kernel() {
    static const bool on = [] {
        const char* e = getenv("PGT_C64_RING");
        return !(e && e[0] == '0');
    }();
    return on;
}

// igemm8.hip covers this launch (single-plane 16-bit 3x3, Cin == 64, Cout <= 64, stride 1, same size, W >= 128, bias (+ residual)
// epilogue); `aligned`: every epilogue operand allows 16-byte accesses (else only the <= 32-output-channel form, which stores
// element-wise: conv_out)
static bool ring_legal(const pgt_conv_desc* d, bool aligned, bool has_res) {
    auto pow2 = [](int v) { return v > 0 && (v & (v - 1)) == 0; };
    return (d->dtype == PGT_BF16 || d->dtype == PGT_F16) && d->orow_mul == 0 && d->Cin == 64 && d->Cout >= 1 && d->Cout <= 64 &&
           d->KH == 3 && d->KW == 3 && d->stride == 1 && d->ups == 0 && d->pad_t == 1 && d->pad_l == 1 && d->Ho == d->H &&
           d->Wo == d->W && pow2(d->W) && pow2(d->H) && d->W >= 128 && d->H >= 4 && d->ldx % 8 == 0 && d->epi == 0 && d->act == 0 &&
           !d->post_relu && d->gn_groups == 0 && d->splitk <= 1 && d->force_bm == 0 && d->force_bn == 0 && (aligned || d->Cout <= 32) &&
           (!has_res || (long)d->N * d->H * d->W * d->ldr * 2 < (1L << 32)) &&
           // a strip of >= 4 rows x 128 columns takes ONE bias vector: bias bands (bias_rows pixels of the raster) must be whole strips
           (d->bias_rows == 0 || d->bias_rows % (4 * d->W) == 0) &&
           // exact weights: IEEE half only (the ring kernel's two-plane form)
           (!d->w2 || (d->dtype == PGT_F16 && (aligned || d->Cout <= 16)));
}

extern "C" size_t pgt_conv2d_workspace_bytes(const pgt_conv_desc* d) {
    if (!d) return 0;
    const int s = planned_splitk(d);
    return s > 1 ? (size_t)s * d->N * d->Ho * d->Wo * d->Cout * sizeof(float) : 0;
}

extern "C" int pgt_conv2d_ws(const pgt_conv_desc* d, const void* x, const void* w, const float* bias,
                             const void* residual, const void* sft_dec, const void* sft_shift, void* y,
                             void* workspace, size_t workspace_bytes, pgt_stream_t stream) {
    PGT_CHECK(d && x && w && y, "pgt_conv2d: null argument");
    PGT_CHECK(d->dtype == PGT_F32 || d->dtype == PGT_BF16 || d->dtype == PGT_F16X3 || d->dtype == PGT_F16, "pgt_conv2d: bad dtype %d", d->dtype);
    const bool x3 = d->dtype == PGT_F16X3;
    const bool f16 = d->dtype == PGT_F16;
    const int es = d->dtype == PGT_F32 ? 4 : 2;
    const int ch = 16 / es;
    PGT_CHECK(d->Cin > 0 && d->Cin % ch == 0, "pgt_conv2d: Cin=%d must be a multiple of %d", d->Cin, ch);
    PGT_CHECK(d->ldx % ch == 0 && d->ldx >= d->Cin, "pgt_conv2d: ldx=%d must be a multiple of %d and >= Cin", d->ldx, ch);
    PGT_CHECK(((uintptr_t)x & 15) == 0 && ((uintptr_t)w & 15) == 0, "pgt_conv2d: x and w must be 16-byte aligned");
    PGT_CHECK(d->KH >= 1 && d->KW >= 1 && d->stride >= 1 && d->Cout >= 1 && d->N >= 1, "pgt_conv2d: bad geometry");
    PGT_CHECK(d->ups == 0 || d->ups == 1, "pgt_conv2d: ups must be 0 or 1");
    PGT_CHECK(d->ldy >= d->Cout && (!d->out_split || d->ldy >= (d->y_lo ? d->y_lo : d->Cout) + d->Cout), "pgt_conv2d: ldy < Cout");
    // the gather offsets of every kernel are 32-bit byte offsets
    PGT_CHECK((long)d->N * d->H * d->W * d->ldx * es < (1L << 31), "pgt_conv2d: the input tensor must be smaller than 2 GiB "
              "(N=%d H=%d W=%d ldx=%d): split the batch", d->N, d->H, d->W, d->ldx);
    PGT_CHECK(d->epi == 0 || (sft_dec && sft_shift), "pgt_conv2d: SFT epilogue needs dec and shift");
    ConvP p;
    p.x = (const char*)x; p.w = (const char*)w; p.bias = bias; p.res = (const char*)residual;
    p.dec = (const char*)sft_dec; p.shift = (const char*)sft_shift; p.y = (char*)y;
    p.N = d->N; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.ldx = d->ldx; p.ups = d->ups;
    p.KH = d->KH; p.KW = d->KW; p.stride = d->stride; p.pad_t = d->pad_t; p.pad_l = d->pad_l;
    p.Ho = d->Ho; p.Wo = d->Wo; p.Cout = d->Cout; p.ldy = d->ldy; p.act = d->act;
    p.post_relu = d->post_relu; p.ldr = d->ldr; p.epi = d->epi; p.ld_dec = d->ld_dec;
    p.ld_shift = d->ld_shift; p.sft_w = d->sft_w; p.out_f32 = d->out_f32;
    p.M = d->N * d->Ho * d->Wo;
    p.K = d->KH * d->KW * d->Cin * (x3 ? (d->x3_fold ? 2 : 3) : 1);
    p.gn_part = nullptr; p.gn_hdr = nullptr; p.gn_cpg = p.gn_G = p.gn_maxblk = p.gn_hw = 0;
    p.in_scale = g_in_scale; p.in_shift = g_in_shift; p.in_act = g_in_act;
    if (d->gn_groups > 0) {
        PGT_CHECK(g_gn_ws != nullptr, "pgt_conv2d: gn_groups set without a statistics workspace (use pgt_conv2d_gn)");
        PGT_CHECK(d->Cout % d->gn_groups == 0 && d->Cout % 8 == 0 && d->gn_nsub >= 1 && d->gn_nsub <= 8 && d->gn_sub >= 0 &&
                  d->gn_sub < d->gn_nsub, "pgt_conv2d: bad GroupNorm statistics request (Cout=%d groups=%d sub=%d/%d)", d->Cout,
                  d->gn_groups, d->gn_sub, d->gn_nsub);
        PGT_CHECK(d->kernel == 0 || d->kernel == 1 || d->kernel == 4, "pgt_conv2d: the statistics epilogue exists in kernels 1 and 4 (kernel=%d)", d->kernel);
        const int hw = d->Ho * d->Wo;
        PGT_CHECK(hw % 64 == 0, "pgt_conv2d: GroupNorm statistics need Ho*Wo %% 64 == 0 (got %d)", hw);
        p.gn_cpg = d->Cout / d->gn_groups;
        p.gn_G = d->gn_groups;
        p.gn_maxblk = hw / 64;
        p.gn_hw = hw;
        const int nimg = d->gn_nimg > 0 ? d->gn_nimg : d->N;
        PGT_CHECK(d->gn_img0 >= 0 && d->gn_img0 + d->N <= nimg, "pgt_conv2d: gn_img0=%d + N=%d exceeds gn_nimg=%d", d->gn_img0, d->N, nimg);
        p.gn_hdr = g_gn_ws + d->gn_sub;
        p.gn_part = g_gn_ws + 8 + ((long)d->gn_sub * nimg + d->gn_img0) * p.gn_maxblk * p.gn_G * 2;
    }
    p.x3 = x3 ? (d->x3_fold ? 2 : 1) : 0;
    p.f16 = f16 ? 1 : 0;
    p.dlo = d->dec_lo ? d->dec_lo : d->Cout;
    p.slo = d->shift_lo ? d->shift_lo : d->Cout;
    p.nw = (x3 && d->x3_fold) ? 128 : (d->w2 ? (d->Cout + 31) / 32 * 64 : d->Cout);
    p.w2 = d->w2 ? 1 : 0;
    PGT_CHECK(!d->w2 || ((f16 || d->dtype == PGT_BF16) && d->splitk <= 1 && !d->ups && !d->out_split && !d->x3_fold && (d->kernel == 0 || d->kernel == 4 || d->kernel == 8)),
              "pgt_conv2d: the exact-weight form (w2) takes PGT_F16 / PGT_BF16 operands, kernel 0, 4 or 8, no split-K, no fused up-sampling");
    p.bias_rows = d->bias_rows;
    p.out_split = d->out_split ? 1 : 0;
    PGT_CHECK(d->bias_rows == 0 || (bias && d->bias_rows > 0 && d->bias_rows % 512 == 0 && p.M % d->bias_rows == 0 && !x3 &&
                                    d->kernel != 2 && d->kernel != 3),
              "pgt_conv2d: bias_rows=%d (a bias vector per frame) needs a bias, a multiple of 512 rows that divides M=%d, a single-plane dtype and kernel 0, 1, 4, 5 or 6", d->bias_rows, p.M);
    PGT_CHECK(!d->x3_fold || (x3 && d->Cout == 64 && d->gn_groups == 0), "pgt_conv2d: x3_fold is the 64-output-channel form of dtype PGT_F16X3 (no statistics epilogue)");
    p.res_f32 = (x3 && d->res_f32) ? 1 : 0;
    PGT_CHECK(!d->res_f32 || (x3 && d->out_f32), "pgt_conv2d: res_f32 goes with dtype PGT_F16X3 and out_f32");
    p.xlo = d->x_lo ? d->x_lo : d->Cin;
    p.ylo = d->y_lo ? d->y_lo : d->Cout;
    p.rlo = d->r_lo ? d->r_lo : d->Cout;
    p.orow_mul = d->orow_mul; p.orow_xmul = d->orow_xmul; p.orow_off = d->orow_off;
    const bool placed = d->orow_mul != 0;
    PGT_CHECK(!placed || (!residual && d->epi == 0), "pgt_conv2d: output placement (orow_*) takes the plain epilogue without residual");
    PGT_CHECK(!placed || d->kernel == 0 || d->kernel == 1 || d->kernel == 4, "pgt_conv2d: output placement needs kernel 0, 1 or 4");
    p.nbm = p.nbn = 0;
    p.splitk = 1;
    p.kt_per_split = 0;
    // 16-byte epilogue accesses need 8-channel granularity and 16-byte aligned rows on every operand
    auto al = [](const void* ptr, int ld) { return ptr == nullptr || ((((uintptr_t)ptr) & 15) == 0 && ld % 8 == 0); };
    p.vec_epi = (d->Cout % 8 == 0) && al(y, d->ldy) && al(residual, d->ldr) &&
                (d->epi == 0 || (al(sft_dec, d->ld_dec) && al(sft_shift, d->ld_shift))) && !d->scalar_epilogue;
    // split-half planes out of the fp32 kernel exist in the 16-byte epilogue only (the element-wise epilogue would write fp32
    // into the half buffer): checked AFTER vec_epi is known, and the lo plane must start on a 16-byte boundary
    PGT_CHECK(!d->out_split || (d->dtype == PGT_F32 && p.vec_epi && !residual && d->epi == 0 && !d->out_f32 && d->orow_mul == 0 &&
                                d->splitk <= 1 && p.ylo % 8 == 0),
              "pgt_conv2d: out_split needs dtype PGT_F32, Cout %% 8 == 0, the plain 16-byte epilogue without residual, no split-K, "
              "16-byte aligned rows and y_lo %% 8 == 0");
    hipStream_t st = (hipStream_t)stream;

    // ---- the 64-channel 3x3 layers of the 512x512 level: ring of row images, optional fused input affine (igemm8.hip)
    {
        const bool v8 = ring_legal(d, p.vec_epi != 0, residual != nullptr);
        PGT_CHECK(d->kernel != 8 || v8, "pgt_conv2d: kernel=8 needs a single-plane 16-bit 3x3 stride-1 same-size conv, Cin == 64, Cout <= 64, "
                  "W >= 128 and H >= 4 powers of two, bias (+ residual) epilogue without activation, 16-byte aligned rows unless Cout <= 32");
        PGT_CHECK(!p.in_scale || (v8 && (d->kernel == 0 || d->kernel == 8) && p.in_shift && (p.in_act == ACT_NONE || p.in_act == ACT_SILU)),
                  "pgt_conv2d_affine_in: this launch has no fused-operand form (pgt_conv2d_affine_in_ok), or in_act is neither none nor SiLU");
        if (v8 && (d->kernel == 8 || p.in_scale || d->w2 || (d->kernel == 0 && use_ring_kernel()))) return pgt_igemm8_launch(&p, st);
    }
    if (d->w2) {      // exact-weight form outside the ring kernel: the phased LDS-DMA kernel, both planes against one staged operand tile
        PGT_CHECK(f16 && d->Cin % 64 == 0 && p.vec_epi && ((uintptr_t)x & 15) == 0 && d->ldx % 8 == 0 && d->KH * d->KW <= 30 &&
                  (long)p.nw * p.K * 2 < (1L << 31),
                  "pgt_conv2d: the exact-weight form needs IEEE half, Cin %% 64 == 0 (Cin=%d), a 16-byte-legal epilogue (Cout %% 8 == 0, aligned rows) "
                  "and <= 30 taps - or a launch the ring kernel covers", d->Cin);
        const int bn = d->force_bn ? d->force_bn : (p.nw <= 128 ? 128 : 256);
        PGT_CHECK(!p.gn_part || (p.gn_hw % (bn == 256 ? 256 : 512) == 0 && (bn / 2) % p.gn_cpg == 0),
                  "pgt_conv2d: exact-weight form with epilogue statistics needs Ho*Wo %% %d == 0 and whole channel groups per %d-channel tile", bn == 256 ? 256 : 512, bn / 2);
        const int rc = pgt_igemm4_launch(&p, bn, st);
        PGT_CHECK(rc != 1, "pgt_conv2d: the exact-weight form has no %d-row weight tile (128, 256)", bn);
        return rc;
    }

    if (x3) {   // split-half operands: the phase-interleaved LDS-DMA kernel with the three-segment K order
        PGT_CHECK(d->Cin % 64 == 0 && d->ups == 0 && d->KH * d->KW <= 30,
                  "pgt_conv2d: bf16x3 needs Cin %% 64 == 0 (Cin=%d) and no fused up-sampling", d->Cin);
        PGT_CHECK(d->epi == 0 || (!d->out_f32 && !d->x3_fold && d->ld_dec >= p.dlo + d->Cout && d->ld_shift >= p.slo + d->Cout &&
                                  p.dlo % 8 == 0 && p.slo % 8 == 0),
                  "pgt_conv2d: bf16x3 SFT epilogue takes split dec / shift planes (dec_lo=%d ld_dec=%d shift_lo=%d ld_shift=%d)", p.dlo, d->ld_dec, p.slo, d->ld_shift);
        PGT_CHECK(d->ldx >= p.xlo + d->Cin && p.xlo % 8 == 0 && p.xlo >= d->Cin, "pgt_conv2d: bf16x3 x_lo=%d / ldx=%d do not hold [hi | lo] planes of %d channels", p.xlo, d->ldx, d->Cin);
        PGT_CHECK(d->out_f32 || (d->ldy >= p.ylo + d->Cout && p.ylo % 8 == 0 && p.ylo >= d->Cout), "pgt_conv2d: bf16x3 y_lo=%d / ldy=%d do not hold [hi | lo] planes of %d channels", p.ylo, d->ldy, d->Cout);
        PGT_CHECK(!residual || p.res_f32 || (d->ldr >= p.rlo + d->Cout && p.rlo % 8 == 0), "pgt_conv2d: bf16x3 residual planes");
        PGT_CHECK(!residual || !p.res_f32 || d->ldr % 4 == 0, "pgt_conv2d: fp32 residual rows must be 16-byte aligned");
        PGT_CHECK(p.vec_epi && (long)d->Cout * p.K * 2 < (1L << 31), "pgt_conv2d: bf16x3 needs Cout %% 8 == 0, 16-byte aligned rows and weights < 2 GiB");
        // the 64-input-channel 3x3 layers of the full-resolution levels: weights in registers, no operand streaming (kernel 6)
        auto p2 = [](int v) { return v > 0 && (v & (v - 1)) == 0; };
        const bool v6x = d->Cin == 64 && d->Cout <= 64 && d->Cout % 16 == 0 && d->KH == 3 && d->KW == 3 && d->stride == 1 &&
                         d->pad_t == 1 && d->pad_l == 1 && d->Ho == d->H && d->Wo == d->W && p2(d->W) && p2(d->H) && d->W >= 32 &&
                         d->H * d->W >= 64 && !d->x3_fold && d->epi == 0 && d->gn_groups == 0 && d->orow_mul == 0 && d->ldx % 8 == 0;
        PGT_CHECK(d->kernel != 6 || v6x, "pgt_conv2d: kernel=6 with bf16x3 needs a 3x3 stride-1 same-size conv, Cin == 64, Cout %% 16 == 0 (<= 64), power-of-two maps, split output");
        if (v6x && (d->kernel == 0 || d->kernel == 6) && d->force_bn == 0) return pgt_igemm6x3_launch(&p, st);
        const int rc = pgt_igemm4_launch(&p, d->x3_fold ? 128 : (d->force_bn ? d->force_bn : (d->Cout <= 128 ? 128 : 256)), st);
        PGT_CHECK(rc != 1, "pgt_conv2d: bf16x3 has no %d-column tile (128, 256)", d->force_bn);
        return rc;
    }

    // ---- K = 256 linears on very many rows: the streaming kernel (igemm7.hip), the static choice from 65 536 rows up
    {
        const bool v7 = d->dtype != PGT_F32 && d->KH == 1 && d->KW == 1 && d->stride == 1 && d->ups == 0 && d->Cin == 256 &&
                        d->Cout % 256 == 0 && d->epi == 0 && !d->out_f32 && d->gn_groups == 0 && d->orow_mul == 0 && p.vec_epi &&
                        d->pad_t == 0 && d->pad_l == 0 && d->Ho == d->H && d->Wo == d->W && d->ldx % 8 == 0 && d->ldy % 8 == 0 &&
                        (!residual || d->ldr % 8 == 0) && ((uintptr_t)x & 15) == 0;
        PGT_CHECK(d->kernel != 7 || v7, "pgt_conv2d: kernel=7 is the K = 256 linear: bf16 / half, 1x1, Cin == 256, Cout %% 256 == 0, plain epilogue, 16-bit output");
        // (wider layers - the 256 -> 768 q|k|v projection - would re-read their rows once per 256 columns: measured 38 % slower
        //  than the phased kernel, so only Cout == 256 takes it by default)
        if (v7 && (d->kernel == 7 || (d->kernel == 0 && d->Cout == 256 && p.M >= 65536 && d->force_bm == 0 && d->force_bn == 0 &&
                                      d->splitk <= 1 && !d->scalar_epilogue)))
            return pgt_igemm7_launch(&p, st);
    }

    // ---- split-K: slices write fp32 partial tiles to the workspace, a second kernel sums + applies the epilogue
    const int slices = (workspace && p.vec_epi) ? planned_splitk(d) : 1;
    if (slices > 1) {
        PGT_CHECK(((uintptr_t)workspace & 15) == 0 && workspace_bytes >= (size_t)slices * p.M * p.Cout * sizeof(float),
                  "pgt_conv2d: split-K workspace too small or misaligned");
        ConvP ps = p;
        ps.bias = nullptr; ps.res = nullptr; ps.dec = nullptr; ps.shift = nullptr;
        ps.act = 0; ps.post_relu = 0; ps.epi = 0; ps.out_f32 = 1; ps.vec_epi = 1;
        ps.y = (char*)workspace; ps.ldy = p.Cout;
        ps.orow_mul = ps.orow_xmul = ps.orow_off = 0;   // the fp32 slabs are dense; only the reduce kernel places rows
        ps.splitk = slices;
        const int bk = d->dtype == PGT_F32 ? 32 : 64;
        const int nk = (p.K + bk - 1) / bk;
        ps.kt_per_split = (nk + slices - 1) / slices;
        int rc = d->dtype == PGT_F32 ? launch<float, 64, 64>(ps, st) : f16 ? launch<half_t, 64, 64>(ps, st) : launch<bf16_t, 64, 64>(ps, st);
        if (rc) return rc;
        const long nchunk = (long)p.M * (p.Cout / 8);
        const dim3 grid((unsigned)((nchunk + 255) / 256));
        if (d->dtype == PGT_F32)
            hipLaunchKernelGGL((splitk_epilogue_kernel<float>), grid, dim3(256), 0, st, p, (const float*)workspace, slices);
        else if (f16)
            hipLaunchKernelGGL((splitk_epilogue_kernel<half_t>), grid, dim3(256), 0, st, p, (const float*)workspace, slices);
        else
            hipLaunchKernelGGL((splitk_epilogue_kernel<bf16_t>), grid, dim3(256), 0, st, p, (const float*)workspace, slices);
        PGT_LAUNCH_CHECK();
        return 0;
    }

    if (d->dtype == PGT_F32) return dispatch<float>(p, st, d->force_bm, d->force_bn);
    // v2 (LDS-DMA, 128-row tiles) is opt-in (kernel = 2); kernel: 0 auto, 1 v1, 2 v2
    const bool v2_legal = d->Cin % 64 == 0 && p.vec_epi && ((uintptr_t)x & 15) == 0 && d->ldx % 8 == 0;
    PGT_CHECK(!f16 || (d->kernel != 2 && d->kernel != 3), "pgt_conv2d: kernels 2 and 3 are bf16-only tuning candidates");
    PGT_CHECK(d->kernel != 2 || v2_legal, "pgt_conv2d: kernel=2 needs bf16, Cin %% 64 == 0 and a 16-byte-legal epilogue");
    // v3 (large tiles, 8-16 waves): additionally stride 1, no up-sampling, <= 32 taps, 32-bit byte offsets
    const bool v3_legal = v2_legal && !placed && d->stride == 1 && d->ups == 0 && d->KH * d->KW <= 32 &&
                          (long)d->N * d->H * d->W * d->ldx * 2 < (1L << 31);
    // auto: 256x256 tiles (16 waves, 2 LDS stages) win on the Cout >= 256 convs once the grid covers the chip at
    // least twice (profiles/r1_igemm_shapes_v7.txt: 256->256 3x3 @ 12x128x128 363 -> 300 us)
    if (d->kernel == 0 && v3_legal && d->force_bm == 0 && d->force_bn == 0 && d->Cout % 256 == 0 && p.K >= 1024 &&
        (long)((p.M + 255) / 256) * (d->Cout / 256) >= 512)
        return pgt_igemm4_launch(&p, 256, st);
    // v4 (phase-interleaved 8-wave schedule; 256x256 or 512x128 tiles): any stride, nearest-2x up-sampling allowed
    const bool v4_legal = v2_legal && d->KH * d->KW <= 30 && (long)d->N * d->H * d->W * d->ldx * 2 < (1L << 31) &&
                          (long)d->Cout * p.K * 2 < (1L << 31);
    PGT_CHECK(d->kernel != 4 || v4_legal, "pgt_conv2d: kernel=4 needs bf16, Cin %% 64 == 0, <= 30 taps, tensors < 2 GiB");
    // v5 (v4's 256x256 schedule + horizontal tap reuse): 3-wide filters, stride 1, same-size power-of-two maps
    auto pow2 = [](int v) { return v > 0 && (v & (v - 1)) == 0; };
    const bool v5_legal = v4_legal && !placed && d->stride == 1 && d->ups == 0 && d->KW == 3 && d->KH <= 8 && d->Ho == d->H &&
                          d->Wo == d->W && pow2(d->W) && pow2(d->H) && d->W >= 32;
    PGT_CHECK(d->kernel != 5 || v5_legal, "pgt_conv2d: kernel=5 needs a 3-wide stride-1 same-size conv on power-of-two maps (W >= 32)");
    if (d->kernel == 5) return pgt_igemm5_launch(&p, st);
    // v6 (register-resident weights, persistent, halo images): the 64-channel 3x3 layers
    const bool v6_legal = !placed && d->Cin == 64 && d->Cout <= 64 && d->KH == 3 && d->KW == 3 && d->stride == 1 && d->ups == 0 &&
                          d->Ho == d->H && d->Wo == d->W && pow2(d->W) && pow2(d->H) && d->W >= 32 && d->ldx % 8 == 0;
    PGT_CHECK(d->kernel != 6 || v6_legal, "pgt_conv2d: kernel=6 needs a 3x3 stride-1 same-size conv, Cin == 64, Cout <= 64, power-of-two maps");
    if (d->kernel == 6) return pgt_igemm6_launch(&p, st);
    // ---- automatic selection (kernel = 0, no pinned tile): a STATIC function of the launch shape, so that two processes
    //      run the same kernels and write the same bits.  The rules restate what timing-based tuning picked on MI355X for
    //      the model's shapes (profiles/r2_v9_autotune_table_b16.json): the register-weight kernel for the 64-channel
    //      3x3 layers; the phase-interleaved LDS-DMA kernel wherever it is legal and its 256x256 / 512x128 tiles cover at
    //      least half of the chip's 256 CUs (Cout <= 128: the 512x128 tile); 128x64 tiles for the remaining <= 64-channel
    //      outputs of the full-resolution maps; the register-staged 64-wide tiles otherwise.
    if (d->kernel == 0 && d->force_bm == 0 && d->force_bn == 0) {
        if (v6_legal && !p.gn_part) return pgt_igemm6_launch(&p, st);
        const int bn4 = d->Cout <= 128 ? 128 : 256, bm4 = bn4 == 256 ? 256 : 512;
        const long tiles4 = (long)((p.M + bm4 - 1) / bm4) * ((d->Cout + bn4 - 1) / bn4);
        const bool gn4 = !p.gn_part || (p.gn_hw % bm4 == 0 && bn4 % p.gn_cpg == 0 && d->ups == 0);
        if (v4_legal && d->Cout > 64 && tiles4 >= 128 && gn4) {
            const int rc = pgt_igemm4_launch(&p, bn4, st);
            if (rc != 1) return rc;
        }
    }
    if (d->kernel == 4) {
        const int rc = pgt_igemm4_launch(&p, d->force_bn ? d->force_bn : (d->Cout <= 128 ? 128 : 256), st);
        PGT_CHECK(rc != 1, "pgt_conv2d: kernel=4 has no %d-column tile (128, 256)", d->force_bn);
        return rc;
    }
    PGT_CHECK(d->kernel != 3, "pgt_conv2d: kernel=3 (the large-tile LDS-DMA variant of rounds 1-2) was removed; use 0, 1, 2, 4, 5, 6, 7");
    if (v2_legal && !placed && d->kernel != 1 && !p.gn_part && !f16) {
        const int bn = (d->force_bn == 64 || (d->force_bn == 0 && d->Cout <= 64)) ? 64 : 128;
        const long blocks = (long)((p.M + 127) / 128) * ((p.Cout + bn - 1) / bn);
        if (d->kernel == 2 || (d->kernel == 0 && d->force_bm == 0 && d->force_bn == 0 && blocks >= kV2MinBlocks && p.K >= kV2MinK))
            return pgt_igemm2_launch(&p, bn, d->force_bm >= 2 && d->force_bm <= 4 ? d->force_bm : 3, st);
    }
    if (f16) return dispatch<half_t>(p, st, d->force_bm, d->force_bn);
    return dispatch<bf16_t>(p, st, d->force_bm, d->force_bn);
}

extern "C" int pgt_conv2d(const pgt_conv_desc* d, const void* x, const void* w, const float* bias,
                          const void* residual, const void* sft_dec, const void* sft_shift, void* y,
                          pgt_stream_t stream) {
    return pgt_conv2d_ws(d, x, w, bias, residual, sft_dec, sft_shift, y, nullptr, 0, stream);
}

extern "C" int pgt_conv2d_affine_in_ok(const pgt_conv_desc* d) {
    return d && ring_legal(d, true, false) ? 1 : 0;
}

extern "C" int pgt_conv2d_affine_in(const pgt_conv_desc* d, const void* x, const float* in_scale, const float* in_shift,
                                    int32_t in_act, const void* w, const float* bias, const void* residual, void* y,
                                    pgt_stream_t stream) {
    PGT_CHECK(d && in_scale && in_shift, "pgt_conv2d_affine_in: null argument");
    g_in_scale = in_scale; g_in_shift = in_shift; g_in_act = in_act;
    const int rc = pgt_conv2d_ws(d, x, w, bias, residual, nullptr, nullptr, y, nullptr, 0, stream);
    g_in_scale = g_in_shift = nullptr; g_in_act = 0;
    return rc;
}

extern "C" size_t pgt_conv_gn_workspace_bytes(int32_t N, int32_t nsub, int32_t HWsub, int32_t groups) {
    return (8 + (size_t)nsub * N * (HWsub / 64 > 0 ? HWsub / 64 : 1) * groups * 2) * sizeof(float);
}

extern "C" int pgt_conv2d_gn(const pgt_conv_desc* d, const void* x, const void* w, const float* bias,
                             const void* residual, const void* sft_dec, const void* sft_shift, void* y,
                             float* gn_workspace, void* workspace, size_t workspace_bytes, pgt_stream_t stream) {
    PGT_CHECK(d && d->gn_groups > 0 && gn_workspace && (((uintptr_t)gn_workspace) & 15) == 0, "pgt_conv2d_gn: needs gn_groups > 0 and a 16-byte aligned statistics workspace");
    g_gn_ws = gn_workspace;
    const int rc = pgt_conv2d_ws(d, x, w, bias, residual, sft_dec, sft_shift, y, workspace, workspace_bytes, stream);
    g_gn_ws = nullptr;
    return rc;
}
