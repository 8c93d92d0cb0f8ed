// Implicit-GEMM conv / linear, 256x256 tile, 8 waves, phase-interleaved schedule (bf16, gfx950).
//
// Why another kernel: igemm.hip / igemm2.hip / igemm3.hip all run "wait for the K tile -> barrier -> read fragments ->
// MFMA" once per K tile, so the matrix pipes idle while fragments are read and while the barrier collects the waves
// (MFMA busy 25-35 %, profiles/r1_igemm_pmc.md).  This kernel follows the CDNA4 guide's 8-phase structure instead:
//
//   * 8 waves as 2 (M) x 4 (N); each wave owns 128 x 64 outputs = 4 x 2 accumulators of 32x32 (128 acc registers).
//   * one K tile (64 deep) is consumed in FOUR phases, one 64 x 32 quadrant of the wave tile each (8 MFMAs):
//     (A0,B0) (A0,B1) (A1,B1) (A1,B0) - fragment reads per phase 12 / 4 / 8 / 0.
//   * the two wave rows run one barrier apart (wave row 1 executes one extra s_barrier up front, wave row 0 one at the
//     end): while one row multiplies, the other one reads its fragments and issues DMA - each SIMD holds one wave of
//     each row, so its matrix pipe always has a wave in the MFMA segment.
//   * global -> LDS by LDS-DMA (glds16, inline asm so that hipcc does not drain it) in UNITS of 16 KiB = the rows that
//     one phase reads: A0 = rows {0..63, 128..191}, A1 = the other rows, B0 = columns {64c..64c+31}, B1 the rest.
//     Unit s = 4 kt + {A0,B0,B1,A1} is issued in phase g = s - 6 and first read in phase 4 kt + {0,0,1,2}:
//       WAR  the previous occupant of the same LDS bytes (K tile kt-2) was last read >= 2 phases before the issue, so
//            both wave rows have retired those reads and passed a barrier;
//       RAW  after issuing, every wave waits vmcnt(8) (= at most the 4 newest units in flight) BEFORE the phase's first
//            barrier, which retires every unit read in the NEXT phase - by either wave row.
//     So four units (64 KiB) are always in flight and nothing is ever waited for within 4 phases of its issue.
//
// Swizzle (swz128) and epilogue as igemm3.hip; stride 1, no up-sampling, Cin % 64 == 0.
#include "common.h"
#include "pgt_internal.h"
#include "igemm_common.h"

// Probe hooks (tools/igemm4_probe.hip compiles this file with -DPGT_PROBE=<bits>; the library build has none):
//   1 no DMA in the main loop   2 no fragment reads   4 no s_setprio   8 no wave-row stagger   16 no epilogue
//   32 time stamps (s_memtime) of wave 0 into g_pgt_probe_ts[block][4]: start, loop start, loop end, end
#ifndef PGT_PROBE
#define PGT_PROBE 0
#endif
#if PGT_PROBE & 32
__device__ unsigned long long g_pgt_probe_ts[4096][4];
#define PGT_STAMP(i) do { if (tid == 0) g_pgt_probe_ts[blockIdx.x & 4095][i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define PGT_STAMP(i) do {} while (0)
#endif

namespace {

constexpr int TILE4 = 256 * 128;        // bytes of one operand tile (256 rows of 64 bf16)
constexpr int STAGE4 = 2 * TILE4;       // A tile + B tile

#define PGT_FENCE() do { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define PGT_BARRIER() do { PGT_FENCE(); __builtin_amdgcn_s_barrier(); PGT_FENCE(); } while (0)

__global__ __launch_bounds__(512) void igemm4_kernel(ConvP p) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];   // 2 * STAGE4 bytes

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int nblk = p.nbm * p.nbn;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, q8 = nblk >> 3, r8 = nblk & 7;
    const int sw = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    const int m0 = (sw / p.nbn) * 256;
    const int n0 = (sw % p.nbn) * 256;
    const unsigned lds0 = lds_addr(smem);
    PGT_STAMP(0);

    // ---- DMA roles.  A piece (h, g): tile rows g*128 + h*64 + wave*8 .. +7.  B piece (h, g): j = wave + 8g,
    //      tile columns (j>>2)*64 + h*32 + (j&3)*8 .. +7.  Lane l lands in slot (l & 15) of super row r0/2 + (l >> 4)
    //      and therefore fetches the inverse-swizzled (row, chunk).
    //      Sources are addressed through buffer descriptors: voffset = the lane's pixel / weight-row byte offset (or
    //      kOob for padding taps, rows >= M, columns >= Cout: an out-of-range buffer load returns zeros), soffset =
    //      the wave-uniform tap / channel / K-tile offset.  The A descriptor's base is moved back by the top-left
    //      padding so that voffset is the UNPADDED pixel address (never negative).
    constexpr unsigned kOob = 0x80000000u;
    const long x_back = ((long)p.pad_t * p.W + p.pad_l) * p.ldx * 2;
    const v4i rsrc_x = make_rsrc(p.x - x_back, (unsigned)((long)p.N * p.H * p.W * p.ldx * 2 + x_back));
    const v4i rsrc_w = make_rsrc(p.w, (unsigned)((long)p.Cout * p.K * 2));
    unsigned a_pix[2][2], a_mask[2][2], a_sel[2][2], b_off[2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            {
                const int r0 = g * 128 + h * 64 + wave * 8;
                const int sr = (r0 >> 1) + (lane >> 4);
                const int slot = (lane & 15) ^ (sr & 15);
                const int m = m0 + 2 * sr + (slot >> 3);
                const int c8 = (slot & 7) * 8;
                unsigned mk = 0, pix = 0;
                if (m < p.M) {
                    const int ox = m % p.Wo;
                    const int t = m / p.Wo;
                    const int oy = t % p.Ho;
                    pix = (unsigned)(((((long)(t / p.Ho) * p.H + oy) * p.W + ox) * p.ldx + c8) * 2);
                    int tt = 0;
                    for (int fy = 0; fy < p.KH; ++fy)
                        for (int fx = 0; fx < p.KW; ++fx, ++tt) {
                            const int iy = oy - p.pad_t + fy, ix = ox - p.pad_l + fx;
                            if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) mk |= 1u << tt;
                        }
                }
                a_pix[h][g] = pix;
                a_mask[h][g] = mk;
                a_sel[h][g] = (mk & 1u) ? pix : kOob;
            }
            {
                const int j = wave + 8 * g;
                const int cc = (j >> 2) * 64 + h * 32 + (j & 3) * 8;
                const int sr = (cc >> 1) + (lane >> 4);
                const int slot = (lane & 15) ^ (sr & 15);
                const int n = n0 + 2 * sr + (slot >> 3);
                b_off[h][g] = n < p.Cout ? (unsigned)((n * p.K + (slot & 7) * 8) * 2) : kOob;
            }
        }

    // issue state: filter tap / first channel of the K tile whose A units are issued next (wave-uniform)
    int ky = 0, kx = 0, c0 = 0, s_off = 0;
    auto issue_a = [&](int h, int buf) {
#pragma unroll
        for (int g = 0; g < 2; ++g)
            bufdma16(a_sel[h][g], rsrc_x, s_off, lds0 + buf * STAGE4 + (g * 128 + h * 64 + wave * 8) * 128);
    };
    auto advance = [&]() {
        c0 += 64;
        s_off += 128;
        if (c0 == p.Cin) {
            c0 = 0;
            if (++kx == p.KW) { kx = 0; ++ky; }
            const int tap = ky * p.KW + kx;
            s_off = (ky * p.W + kx) * p.ldx * 2;
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int g = 0; g < 2; ++g) a_sel[h][g] = ((a_mask[h][g] >> tap) & 1u) ? a_pix[h][g] : kOob;
        }
    };
    auto issue_b = [&](int h, int buf, int kt) {
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const int j = wave + 8 * g;
            bufdma16(b_off[h][g], rsrc_w, kt * 128, lds0 + buf * STAGE4 + TILE4 + ((j >> 2) * 64 + h * 32 + (j & 3) * 8) * 128);
        }
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // fragment read offsets: row (lane & 31) of the wave's block, chunk 2 ks + (lane >> 5); +32 rows = +4096 bytes
    // and +64 rows = +8192 bytes (16 / 32 super rows: the swizzle key is unchanged).
    const int hh = lane >> 5;
    int a_rd[4], b_rd[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        a_rd[ks] = swz128(wr * 128 + (lane & 31), 2 * ks + hh);
        b_rd[ks] = TILE4 + swz128(wc * 64 + (lane & 31), 2 * ks + hh);
    }

    const int nk = p.K / 64;
    // ---- prologue: units 0..5 = K tile 0 (A0 B0 B1 A1) and K tile 1 (A0 B0)
    issue_a(0, 0);
    issue_b(0, 0, 0);
    issue_b(1, 0, 0);
    issue_a(1, 0);
    advance();
    if (nk > 1) {
        issue_a(0, 1);
        issue_b(0, 1, 1);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    }
    PGT_BARRIER();
    if (!(PGT_PROBE & 8) && wr == 1) PGT_BARRIER();   // wave row 1 runs one barrier behind wave row 0

    uint4 fa[2][4], fb0[4], fb1[4];
#if PGT_PROBE & 2
    for (int ks = 0; ks < 4; ++ks) fa[0][ks] = fa[1][ks] = fb0[ks] = fb1[ks] = make_uint4(lane, ks, 0, 0);
#endif
    PGT_STAMP(1);

#define PGT_PHASE(Q, BUF, KT)                                                                                          \
    {                                                                                                                  \
        const char* st_ = smem + (BUF) * STAGE4;                                                                       \
        if (PGT_PROBE & 2) {                                                                                           \
        } else if ((Q) == 0) {                                                                                         \
            _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) fb0[ks] = *reinterpret_cast<const uint4*>(st_ + b_rd[ks]); \
            _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                              \
                _Pragma("unroll") for (int ks = 0; ks < 4; ++ks)                                                       \
                    fa[i][ks] = *reinterpret_cast<const uint4*>(st_ + a_rd[ks] + i * 4096);                            \
        } else if ((Q) == 1) {                                                                                         \
            _Pragma("unroll") for (int ks = 0; ks < 4; ++ks)                                                           \
                fb1[ks] = *reinterpret_cast<const uint4*>(st_ + b_rd[ks] + 4096);                                      \
        } else if ((Q) == 2) {                                                                                         \
            _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                              \
                _Pragma("unroll") for (int ks = 0; ks < 4; ++ks)                                                       \
                    fa[i][ks] = *reinterpret_cast<const uint4*>(st_ + a_rd[ks] + 8192 + i * 4096);                     \
        }                                                                                                              \
        asm volatile("" ::: "memory");                                                                                 \
        if (!(PGT_PROBE & 1) && (KT) + ((Q) < 2 ? 1 : 2) < nk) {                                                       \
            if ((Q) == 0) issue_b(1, (BUF) ^ 1, (KT) + 1);                                                             \
            else if ((Q) == 1) { issue_a(1, (BUF) ^ 1); advance(); }                                                   \
            else if ((Q) == 2) issue_a(0, (BUF));                                                                      \
            else issue_b(0, (BUF), (KT) + 2);                                                                          \
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                                                           \
        } else {                                                                                                       \
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                           \
        }                                                                                                              \
        PGT_BARRIER();                                                                                                 \
        if (!(PGT_PROBE & 4)) __builtin_amdgcn_s_setprio(1);                                                           \
        _Pragma("unroll") for (int ks = 0; ks < 4; ++ks)                                                               \
            _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                              \
                acc[((Q) >> 1) * 2 + i][((Q) == 1 || (Q) == 2) ? 1 : 0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(     \
                    __builtin_bit_cast(bf16x8, fa[i][ks]),                                                             \
                    __builtin_bit_cast(bf16x8, ((Q) == 1 || (Q) == 2) ? fb1[ks] : fb0[ks]),                            \
                    acc[((Q) >> 1) * 2 + i][((Q) == 1 || (Q) == 2) ? 1 : 0], 0, 0, 0);                                 \
        if (!(PGT_PROBE & 4)) __builtin_amdgcn_s_setprio(0);                                                           \
        PGT_BARRIER();                                                                                                 \
    }

    for (int kt = 0; kt < nk; kt += 2) {
        PGT_PHASE(0, 0, kt)
        PGT_PHASE(1, 0, kt)
        PGT_PHASE(2, 0, kt)
        PGT_PHASE(3, 0, kt)
        if (kt + 1 < nk) {
            PGT_PHASE(0, 1, kt + 1)
            PGT_PHASE(1, 1, kt + 1)
            PGT_PHASE(2, 1, kt + 1)
            PGT_PHASE(3, 1, kt + 1)
        }
    }
#undef PGT_PHASE
    if (!(PGT_PROBE & 8) && wr == 0) PGT_BARRIER();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    PGT_STAMP(2);
#if PGT_PROBE & 16
    if (acc[0][0][0] + acc[1][1][3] + acc[2][0][5] + acc[3][1][7] == 12345.f) reinterpret_cast<float*>(p.y)[tid] = 1.f;
    PGT_STAMP(3);
    return;
#endif

    // ---- epilogue: 4 passes of 64 rows (wave row wr, half ih); act(acc + bias) staged in LDS as fp32, then 8 channels
    //      of one pixel per thread with 16-byte residual / dec / shift loads and stores.
    constexpr int SROW = 256 + 4;
    float* stage = reinterpret_cast<float*>(smem);
    static_assert(64 * SROW * 4 <= 2 * STAGE4, "epilogue stage must fit");
    const bf16_t* res = reinterpret_cast<const bf16_t*>(p.res);
    const bf16_t* dec = reinterpret_cast<const bf16_t*>(p.dec);
    const bf16_t* shf = reinterpret_cast<const bf16_t*>(p.shift);
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
        if (wr == (pass >> 1)) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int cl = wc * 64 + j * 32 + (lane & 31);
                const int n = n0 + cl;
                const float bv = (p.bias && n < p.Cout) ? p.bias[n] : 0.f;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int rl = i * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
                        stage[rl * SROW + cl] = acc[(pass & 1) * 2 + i][j][e] + bv;
                    }
            }
        }
        __syncthreads();
        for (int cidx = tid; cidx < 64 * 32; cidx += 512) {
            const int rl = cidx >> 5, c8 = (cidx & 31) * 8;
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
    PGT_STAMP(3);
}

}  // namespace

// bf16, stride 1, no up-sampling, Cin % 64 == 0, KH*KW <= 32, tensor < 2 GiB, 16-byte-legal epilogue (caller checks).
int pgt_igemm4_launch(const void* pv, hipStream_t st) {
    ConvP p = *reinterpret_cast<const ConvP*>(pv);
    p.nbm = (p.M + 255) / 256;
    p.nbn = (p.Cout + 255) / 256;
    constexpr int bytes = 2 * STAGE4;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm4_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e != hipSuccess) { pgt_set_error("igemm4: cannot reserve %d B of LDS: %s", bytes, hipGetErrorString(e)); return -12; }
        attr_set = true;
    }
    hipLaunchKernelGGL(igemm4_kernel, dim3(p.nbm * p.nbn), dim3(512), bytes, st, p);
    PGT_LAUNCH_CHECK();
    return 0;
}
