// Implicit-GEMM conv / linear, large-tile LDS-DMA kernel (bf16): BM x BN workgroup tiles of 256x128 / 256x256 /
// 128x256 with one 64x64 wave tile per wavefront (8 or 16 waves per workgroup).
//
// Why: PMC on the 128x128 kernels (profiles/r1_igemm_pmc.md) shows the effective DMA latency of a 32 KiB K tile
// (~2-3 k cycles: 512 cycles of L1 issue + L2 latency + queueing) exceeds the tile's compute phase (512 MFMA cycles
// per SIMD), so the waves park at the barrier whatever the staging mechanism.  A 256-row tile doubles or quadruples
// the MFMA cycles per byte staged (1024-2048 cycles per K tile per SIMD, 2-4 waves per SIMD inside ONE workgroup) so
// a 1-2 tile look-ahead covers the latency, and halves the L2->LDS traffic per flop.
//
// Same machinery as igemm2.hip: global_load_lds_dwordx4 into an unpadded XOR-swizzled LDS image (swizzle applied on
// the source address), zero page for padding / ragged edges, counted vmcnt + raw s_barrier, scalar-tap + row-bitmask
// gather addressing (stride-1, non-upsampled convs and linears only; Cin % 64 == 0), fp32 LDS-staged 16-byte
// epilogue.  Waves are laid out WM x WN; DMA call q (1 KiB = 8 tile rows) is issued by wave q % NWAVES.
#include "common.h"
#include "pgt_internal.h"
#include "igemm_common.h"

__device__ uint4 g_pgt_zero_page3[4];   // 64 B of zeros (per translation unit: no relocatable device code)

namespace {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int BM, int BN, int NST>
__global__ __launch_bounds__((BM / 64) * (BN / 64) * 64) void igemm3_kernel(ConvP p) {
    constexpr int WM = BM / 64, WN = BN / 64, NWAVES = WM * WN, NTHREADS = NWAVES * 64;
    constexpr int TILE_A = BM * 128, TILE_B = BN * 128, STAGE = TILE_A + TILE_B;
    constexpr int QA = (BM / 8) / NWAVES;   // A DMA calls per wave per K tile (8 rows per call)
    constexpr int QB = (BN / 8) / NWAVES;
    static_assert(QA >= 1 && QB >= 1 && (BM / 8) % NWAVES == 0 && (BN / 8) % NWAVES == 0, "tile/wave mismatch");
    extern __shared__ __attribute__((aligned(1024))) char smem[];   // NST * STAGE bytes

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int nblk = p.nbm * p.nbn;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, q8 = nblk >> 3, r8 = nblk & 7;
    const int sw = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    const int m0 = (sw / p.nbn) * BM;
    const int n0 = (sw % p.nbn) * BN;
    const char* zero = reinterpret_cast<const char*>(g_pgt_zero_page3);

    // DMA roles (see igemm2.hip): call q covers super rows 4q..4q+3; lane lands in slot (lane & 15) of super row
    // 4q + (lane >> 4) and fetches the inverse-swizzled (row, chunk).
    int a_base[QA];
    unsigned a_mask[QA];
#pragma unroll
    for (int i = 0; i < QA; ++i) {
        const int sr = (wave + NWAVES * i) * 4 + (lane >> 4);
        const int slot = (lane & 15) ^ (sr & 15);
        const int row = 2 * sr + (slot >> 3);
        const int c8 = (slot & 7) * 8;
        const int m = m0 + row;
        unsigned mk = 0;
        int base = 0;
        if (m < p.M) {
            const int ox = m % p.Wo;
            const int t = m / p.Wo;
            const int oy = t % p.Ho;
            const int iy0 = oy - p.pad_t, ix0 = ox - p.pad_l;
            base = (int)((((long)(t / p.Ho) * p.H + iy0) * p.W + ix0) * p.ldx + c8) * 2;
            for (int tt = 0; tt < p.KH * p.KW; ++tt) {
                const int iy = iy0 + tt / p.KW, ix = ix0 + tt % p.KW;
                if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) mk |= 1u << tt;
            }
        }
        a_base[i] = base;
        a_mask[i] = mk;
    }
    const char* b_src[QB];
#pragma unroll
    for (int i = 0; i < QB; ++i) {
        const int sr = (wave + NWAVES * i) * 4 + (lane >> 4);
        const int slot = (lane & 15) ^ (sr & 15);
        const int n = n0 + 2 * sr + (slot >> 3);
        b_src[i] = n < p.Cout ? p.w + ((long)n * p.K + (slot & 7) * 8) * 2 : nullptr;
    }
    int ky = 0, kx = 0, c0 = 0;   // filter tap and first input channel of the next K tile to issue (uniform)

    auto issue = [&](int kt, int stage) {
        const unsigned sa = lds_addr(smem) + stage * STAGE;
        const unsigned sb = sa + TILE_A;
        const int tap = ky * p.KW + kx;
        const int s_off = ((ky * p.W + kx) * p.ldx + c0) * 2;   // wave-uniform
#pragma unroll
        for (int i = 0; i < QA; ++i) {
            const char* src = ((a_mask[i] >> tap) & 1u) ? p.x + (long)(a_base[i] + s_off) : zero;
            glds16(src, sa + (wave + NWAVES * i) * 1024);
        }
#pragma unroll
        for (int i = 0; i < QB; ++i) {
            const char* src = b_src[i] ? b_src[i] + (long)kt * 128 : zero;
            glds16(src, sb + (wave + NWAVES * i) * 1024);
        }
        c0 += 64;
        if (c0 == p.Cin) {
            c0 = 0;
            if (++kx == p.KW) { kx = 0; ++ky; }
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int h = lane >> 5;
    int a_off[2][4], b_off[2][4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            a_off[i][s] = swz128(wm * 64 + i * 32 + (lane & 31), 2 * s + h);
            b_off[i][s] = TILE_A + swz128(wn * 64 + i * 32 + (lane & 31), 2 * s + h);
        }
    }

    const int nk = p.K / 64;
#pragma unroll
    for (int t = 0; t < NST - 1; ++t)
        if (t < nk) issue(t, t);
    int cur = 0, nxt = NST - 1;
    for (int kt = 0; kt < nk; ++kt) {
        if (NST == 2 || kt + 1 >= nk) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (NST == 3 || kt + 2 >= nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(QA + QB) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (QA + QB)) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (kt + NST - 1 < nk) issue(kt + NST - 1, nxt);
        const char* st = smem + cur * STAGE;
        cur = cur + 1 == NST ? 0 : cur + 1;
        nxt = nxt + 1 == NST ? 0 : nxt + 1;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            uint4 af[2], bfr[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                af[i] = *reinterpret_cast<const uint4*>(st + a_off[i][s]);
                bfr[i] = *reinterpret_cast<const uint4*>(st + b_off[i][s]);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[i]),
                                                                       __builtin_bit_cast(bf16x8, bfr[j]), acc[i][j], 0, 0, 0);
        }
    }
    __syncthreads();

    // ---- epilogue: WM passes of 64 rows; act(acc + bias) staged in LDS as fp32, then 8 channels of one pixel per
    //      thread with 16-byte residual / dec / shift loads and stores.
    constexpr int SROW = BN + 4;
    float* stage = reinterpret_cast<float*>(smem);
    static_assert(64 * SROW * 4 <= NST * STAGE, "epilogue stage must fit");
    const bf16_t* res = reinterpret_cast<const bf16_t*>(p.res);
    const bf16_t* dec = reinterpret_cast<const bf16_t*>(p.dec);
    const bf16_t* shf = reinterpret_cast<const bf16_t*>(p.shift);
    for (int pass = 0; pass < WM; ++pass) {
        if (wm == pass) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int cl = wn * 64 + j * 32 + (lane & 31);
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
        for (int cidx = tid; cidx < 64 * (BN / 8); cidx += NTHREADS) {
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

template <int BM, int BN, int NST> int launch(const ConvP& p0, hipStream_t st) {
    ConvP p = p0;
    p.nbm = (p.M + BM - 1) / BM;
    p.nbn = (p.Cout + BN - 1) / BN;
    constexpr int bytes = NST * (BM + BN) * 128;
    static_assert(bytes <= 160 * 1024, "LDS budget");
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm3_kernel<BM, BN, NST>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e != hipSuccess) { pgt_set_error("igemm3: cannot reserve %d B of LDS: %s", bytes, hipGetErrorString(e)); return -12; }
        attr_set = true;
    }
    hipLaunchKernelGGL((igemm3_kernel<BM, BN, NST>), dim3(p.nbm * p.nbn), dim3((BM / 64) * (BN / 64) * 64), bytes, st, p);
    PGT_LAUNCH_CHECK();
    return 0;
}

}  // namespace

// bf16, stride 1, no up-sampling, Cin % 64 == 0, KH*KW <= 32, tensor < 2 GiB, 16-byte-legal epilogue (caller checks).
// (bm, bn, stages): (256,128,2|3), (256,256,2), (128,256,2|3).  Returns 1 if the combination is not built.
int pgt_igemm3_launch(const void* pv, int bm, int bn, int stages, hipStream_t st) {
    const ConvP& p = *reinterpret_cast<const ConvP*>(pv);
    if (bm == 256 && bn == 128) return stages == 2 ? launch<256, 128, 2>(p, st) : launch<256, 128, 3>(p, st);
    if (bm == 256 && bn == 256) return launch<256, 256, 2>(p, st);
    if (bm == 128 && bn == 256) return stages == 2 ? launch<128, 256, 2>(p, st) : launch<128, 256, 3>(p, st);
    return 1;
}
