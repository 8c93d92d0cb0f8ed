// Implicit-GEMM conv / linear, second generation (bf16): LDS-DMA staged, double-buffered, 64-wide wave tiles.
//
// What changes against igemm.hip (v1, kept for f32, ragged channel counts and tiny layers):
//  * global -> LDS goes through `global_load_lds_dwordx4` (no VGPR round trip, no ds_write pass).  The
//    DMA writes lane-linear (1 KiB per wave instruction), so the LDS image is UNPADDED and the
//    bank-conflict fix is an XOR swizzle applied on the SOURCE side: the lane that lands in 16-byte slot
//    p fetches the chunk whose swizzled position is p; fragment reads apply the same involution.
//    A 256-byte "super row" holds two tile rows (16 slots); slot' = slot ^ (superrow & 15).
//  * zero padding / ragged M,N: invalid lanes point their source at a 64-byte zero page (a device
//    global) — the DMA has no per-lane predicate.
//  * two LDS stages, ONE barrier per K tile: wait own DMA (tile t) -> barrier -> issue DMA (tile t+1)
//    into the stage compute(t-1) just released -> 4 k-steps of MFMA on tile t.
//  * workgroup tile 128 x BN (BN = 64 | 128), 4 waves 2x2, wave tile 64 x BN/2: one A + one B fragment
//    read feeds 2 x (BN/64) MFMAs, halving LDS read traffic per flop against v1's 64-row tiles.
//  * requires Cin % 64 == 0 so a 64-deep K tile never straddles a filter tap: (ky,kx) is uniform per tile.
#include "common.h"
#include "pgt_internal.h"
#include "igemm_common.h"

__device__ uint4 g_pgt_zero_page[4];   // 64 B of zeros: source of padded / out-of-range 16-byte chunks

namespace {

constexpr int kThreads = 256;
constexpr int BM = 128;

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// byte offset of (row, 16-byte chunk c) inside a swizzled tile of 128-byte rows
__device__ __forceinline__ int swz(int row, int c) {
    const int sr = row >> 1;
    return sr * 256 + (((((row & 1) << 3) | c) ^ (sr & 15)) << 4);
}

template <int BN, int NST, bool FAST>   // FAST: stride 1, no up-sampling, KH*KW <= 32 -> scalar tap offset + row bitmask
__global__ __launch_bounds__(kThreads) void igemm2_kernel(ConvP p) {
    constexpr int NI = BN / 64;             // 32-wide MFMA tiles per wave along N
    constexpr int TILE_A = BM * 128, TILE_B = BN * 128, STAGE = TILE_A + TILE_B;
    constexpr int QA = BM / 32;             // A DMA calls per wave per tile (8 rows per call, 4 waves)
    constexpr int QB = BN / 32;
    extern __shared__ __attribute__((aligned(1024))) char smem[];   // NST * STAGE bytes

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int nblk = p.nbm * p.nbn;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, q8 = nblk >> 3, r8 = nblk & 7;
    const int sw = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    const int m0 = (sw / p.nbn) * BM;
    const int n0 = (sw % p.nbn) * BN;
    const char* zero = reinterpret_cast<const char*>(g_pgt_zero_page);

    // ---- DMA roles: call q covers super rows 4q..4q+3; this lane lands in slot (lane & 15) of super row
    //      4q + (lane >> 4) and therefore fetches (row, chunk) = inverse swizzle of that position.
    int a_iy0[QA], a_ix0[QA], a_c8[QA];
    long a_pix[QA];
    const int Hv = p.H << p.ups, Wv = p.W << p.ups;
#pragma unroll
    for (int i = 0; i < QA; ++i) {
        const int sr = (wave + 4 * i) * 4 + (lane >> 4);
        const int slot = (lane & 15) ^ (sr & 15);
        const int row = 2 * sr + (slot >> 3);
        a_c8[i] = (slot & 7) * 8;
        const int m = m0 + row;
        if (m < p.M) {
            const int ox = m % p.Wo;
            const int t = m / p.Wo;
            const int oy = t % p.Ho;
            a_iy0[i] = oy * p.stride - p.pad_t;
            a_ix0[i] = ox * p.stride - p.pad_l;
            a_pix[i] = (long)(t / p.Ho) * p.H * p.W;
        } else {
            a_iy0[i] = -(1 << 28);
            a_ix0[i] = 0;
            a_pix[i] = 0;
        }
    }
    // FAST gather state: byte offset of the tap-(0,0) source pixel (+ this lane's chunk) and a bit per filter
    // tap telling whether that tap is inside the image for this row; per K tile the address is then
    // x + a_base[i] + (uniform tap/channel offset), one add and one select per DMA.
    int a_base[QA];
    unsigned a_mask[QA];
    if (FAST) {
#pragma unroll
        for (int i = 0; i < QA; ++i) {
            a_base[i] = (int)(((a_pix[i] + (long)a_iy0[i] * p.W + a_ix0[i]) * p.ldx + a_c8[i]) * 2);
            unsigned mk = 0;
            for (int t = 0; t < p.KH * p.KW; ++t) {
                const int iy = a_iy0[i] + t / p.KW, ix = a_ix0[i] + t % p.KW;
                if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) mk |= 1u << t;
            }
            a_mask[i] = mk;
        }
    }
    const char* b_src[QB];
#pragma unroll
    for (int i = 0; i < QB; ++i) {
        const int sr = (wave + 4 * i) * 4 + (lane >> 4);
        const int slot = (lane & 15) ^ (sr & 15);
        const int n = n0 + 2 * sr + (slot >> 3);
        b_src[i] = n < p.Cout ? p.w + ((long)n * p.K + (slot & 7) * 8) * 2 : nullptr;
    }
    int ky = 0, kx = 0, c0 = 0;   // filter tap and first input channel of the current K tile (uniform)

    auto issue = [&](int kt, int stage) {
        const unsigned sa = lds_addr(smem) + stage * STAGE;
        const unsigned sb = sa + TILE_A;
        if (FAST) {
            const int tap = ky * p.KW + kx;
            const int s_off = ((ky * p.W + kx) * p.ldx + c0) * 2;   // wave-uniform
#pragma unroll
            for (int i = 0; i < QA; ++i) {
                const char* src = ((a_mask[i] >> tap) & 1u) ? p.x + (long)(a_base[i] + s_off) : zero;
                glds16(src, sa + (wave + 4 * i) * 1024);
            }
        } else {
#pragma unroll
            for (int i = 0; i < QA; ++i) {
                const int iy = a_iy0[i] + ky, ix = a_ix0[i] + kx;
                const char* src = zero;
                if (iy >= 0 && iy < Hv && ix >= 0 && ix < Wv)
                    src = p.x + ((a_pix[i] + (long)(iy >> p.ups) * p.W + (ix >> p.ups)) * p.ldx + c0 + a_c8[i]) * 2;
                glds16(src, sa + (wave + 4 * i) * 1024);
            }
        }
#pragma unroll
        for (int i = 0; i < QB; ++i) {
            const char* src = b_src[i] ? b_src[i] + (long)kt * 128 : zero;
            glds16(src, sb + (wave + 4 * i) * 1024);
        }
        c0 += 64;
        if (c0 == p.Cin) {
            c0 = 0;
            if (++kx == p.KW) { kx = 0; ++ky; }
        }
    };

    f32x16 acc[2][NI];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // fragment read offsets (swizzled): rows of this lane, chunk = 2*s + h
    const int h = lane >> 5;
    int a_off[2][4], b_off[NI][4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
        for (int i = 0; i < 2; ++i) a_off[i][s] = swz(wm * 64 + i * 32 + (lane & 31), 2 * s + h);
#pragma unroll
        for (int j = 0; j < NI; ++j) b_off[j][s] = TILE_A + swz(wn * (BN / 2) + j * 32 + (lane & 31), 2 * s + h);
    }

    const int nk = p.K / 64;
    // prologue: NST-1 tiles in flight
#pragma unroll
    for (int t = 0; t < NST - 1; ++t)
        if (t < nk) issue(t, t);
    int cur = 0, nxt = NST - 1;   // stage holding tile kt / stage to refill with tile kt+NST-1
    for (int kt = 0; kt < nk; ++kt) {
        // this wave's share of tile kt has landed; the (up to NST-2) younger tiles may stay in flight.
        // vmcnt counts this wave's outstanding DMA instructions: QA+QB per tile.
        if (NST == 2 || kt + 1 >= nk) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (NST == 3 || kt + 2 >= nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(QA + QB) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (QA + QB)) : "memory");
        __builtin_amdgcn_s_barrier();   // everyone's share landed; the stage read by compute(kt-1) is free
        asm volatile("" ::: "memory");  // compiler fence: no LDS read of this tile may be hoisted above the barrier
        if (kt + NST - 1 < nk) issue(kt + NST - 1, nxt);
        const char* st = smem + cur * STAGE;
        cur = cur + 1 == NST ? 0 : cur + 1;
        nxt = nxt + 1 == NST ? 0 : nxt + 1;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            uint4 af[2], bfr[NI];
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const uint4*>(st + a_off[i][s]);
#pragma unroll
            for (int j = 0; j < NI; ++j) bfr[j] = *reinterpret_cast<const uint4*>(st + b_off[j][s]);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[i]),
                                                                       __builtin_bit_cast(bf16x8, bfr[j]), acc[i][j], 0, 0, 0);
        }
    }
    __syncthreads();

    // ---- epilogue (16-byte path only; the launcher guarantees legality): stage act(acc + bias) in LDS as
    //      fp32, two 64-row passes, then 8 channels of one pixel per thread.
    constexpr int SROW = BN + 4;
    float* stage = reinterpret_cast<float*>(smem);
    static_assert(64 * SROW * 4 <= NST * STAGE, "epilogue stage must fit");
    const bf16_t* res = reinterpret_cast<const bf16_t*>(p.res);
    const bf16_t* dec = reinterpret_cast<const bf16_t*>(p.dec);
    const bf16_t* shf = reinterpret_cast<const bf16_t*>(p.shift);
    for (int pass = 0; pass < 2; ++pass) {
        if (wm == pass) {
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const int cl = wn * (BN / 2) + j * 32 + (lane & 31);
                const int n = n0 + cl;
                const float bv = (p.bias && n < p.Cout) ? p.bias[n] : 0.f;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int rl = i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                        stage[rl * SROW + cl] = acc[i][j][e] + bv;
                    }
            }
        }
        __syncthreads();
        for (int cidx = tid; cidx < 64 * (BN / 8); cidx += kThreads) {
            const int rl = cidx / (BN / 8), c8 = (cidx % (BN / 8)) * 8;
            const int m = m0 + pass * 64 + rl, n = n0 + c8;
            if (m >= p.M || n >= p.Cout) continue;
            float v[8];
            *reinterpret_cast<float4*>(v) = *reinterpret_cast<const float4*>(stage + rl * SROW + c8);
            *reinterpret_cast<float4*>(v + 4) = *reinterpret_cast<const float4*>(stage + rl * SROW + c8 + 4);
            apply_act8(v, p.act);
            if (p.epi == 1) {
                float d[8], s[8];
                load8<bf16_t>(dec + (long)m * p.ld_dec + n, d);
                load8<bf16_t>(shf + (long)m * p.ld_shift + n, s);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = d[e] + p.sft_w * (d[e] * v[e] + s[e]);
            } else {
                if (res) {
                    float r[8];
                    load8<bf16_t>(res + (long)m * p.ldr + n, r);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += r[e];
                }
                if (p.post_relu) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
                }
            }
            if (p.out_f32) store8<float>(reinterpret_cast<float*>(p.y) + (long)m * p.ldy + n, v);
            else store8<bf16_t>(reinterpret_cast<bf16_t*>(p.y) + (long)m * p.ldy + n, v);
        }
        __syncthreads();
    }
}

}  // namespace

template <int BN, int NST, bool FAST> static int launch3(const ConvP& p, hipStream_t st) {
    constexpr int bytes = NST * (BM + BN) * 128;
    static bool attr_set = false;   // > 64 KiB of dynamic LDS needs the opt-in attribute (once per kernel)
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm2_kernel<BN, NST, FAST>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e != hipSuccess) { pgt_set_error("igemm2: cannot reserve %d B of LDS: %s", bytes, hipGetErrorString(e)); return -12; }
        attr_set = true;
    }
    hipLaunchKernelGGL((igemm2_kernel<BN, NST, FAST>), dim3(p.nbm * p.nbn), dim3(kThreads), bytes, st, p);
    PGT_LAUNCH_CHECK();
    return 0;
}
template <int BN, int NST> static int launch2(const ConvP& p, hipStream_t st) {
    // FAST needs 32-bit byte offsets (tensor < 2 GiB) and a <= 32-tap filter
    const bool fast = p.stride == 1 && p.ups == 0 && p.KH * p.KW <= 32 &&
                      (long)p.N * p.H * p.W * p.ldx * 2 < (1L << 31);
    return fast ? launch3<BN, NST, true>(p, st) : launch3<BN, NST, false>(p, st);
}

// bf16 only.  bn: 64 | 128; stages: 2 | 3 | 4 (LDS = stages * (128 + bn) * 128 bytes).
int pgt_igemm2_launch(const void* pv, int bn, int stages, hipStream_t st) {
    ConvP p = *reinterpret_cast<const ConvP*>(pv);
    p.nbm = (p.M + BM - 1) / BM;
    p.nbn = (p.Cout + bn - 1) / bn;
    if (bn == 128) {
        if (stages == 2) return launch2<128, 2>(p, st);
        if (stages == 4) return launch2<128, 4>(p, st);
        return launch2<128, 3>(p, st);
    }
    if (stages == 2) return launch2<64, 2>(p, st);
    if (stages == 4) return launch2<64, 4>(p, st);
    return launch2<64, 3>(p, st);
}
