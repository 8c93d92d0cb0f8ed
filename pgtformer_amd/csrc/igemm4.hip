// Implicit-GEMM conv / linear with the phase-interleaved 8-wave schedule (bf16, gfx950).
// Workgroup tiles 256x256 (waves 2 x 4) and 512x128 (waves 4 x 2); every wave owns 128 x 64 outputs.
//
// Why another kernel: igemm.hip / igemm2.hip / igemm3.hip all run "wait for the K tile -> barrier -> read fragments ->
// MFMA" once per K tile, so the matrix pipes idle while fragments are read and while the barrier collects the waves
// (MFMA busy 25-35 %, profiles/r1_igemm_pmc.md).  This kernel follows the CDNA4 guide's 8-phase structure instead:
//
//   * 8 waves as WR (M) x WC (N); each wave owns 128 x 64 outputs = 4 x 2 accumulators of 32x32 (128 acc registers).
//   * one K tile (64 deep) is consumed in FOUR phases, one 64 x 32 quadrant of the wave tile each (8 MFMAs):
//     (A0,B0) (A0,B1) (A1,B1) (A1,B0) - fragment reads per phase 12 / 4 / 8 / 0.
//   * waves 4-7 run one barrier behind waves 0-3 (they execute one extra s_barrier up front, waves 0-3 one at the
//     end): while one half multiplies, the other one reads its fragments and issues DMA - each SIMD holds one wave of
//     each half, so its matrix pipe always has a wave in the MFMA segment.
//   * global -> LDS by LDS-DMA through buffer descriptors (bufdma16: inline asm, so hipcc does not drain it) in UNITS
//     = the rows one phase reads: A0 = rows {128g .. 128g+63}, A1 = the other rows, B0 = columns {64c .. 64c+31}, B1
//     the rest.  K tile kt's units are issued in phases 4kt-6 (A0), 4kt-5 (B0), 4kt-4 (B1), 4kt-3 (A1) - for the
//     512-row tile half of each A unit moves to the neighbouring B phase (3-2-2-3 instead of 1-4-4-1 DMAs per
//     phase) - and first read in phases 4kt + {0, 0, 1, 2}:
//       WAR  the bytes a unit overwrites (K tile kt-2) were last read >= 2 phases before the issue, so both wave
//            halves have retired those reads and passed a barrier;
//       RAW  after its issues every phase waits vmcnt(NV), NV = the wave's DMA count of any 4 consecutive phases,
//            BEFORE its first barrier: that retires everything issued >= 4 phases earlier, i.e. every unit read in
//            the NEXT phase - by either half.
//     So a full K tile (64-80 KiB) is always in flight and nothing is waited for within 4 phases of its issue.
//   * padding taps, rows >= M and columns >= Cout use an out-of-range buffer offset (the DMA then writes zeros); the
//     per-lane source offset of the current filter tap is recomputed when the tap changes (every Cin/64 K tiles),
//     which also covers strided and nearest-2x up-sampled inputs; the K-tile channel offset is the scalar soffset.
//
// Swizzle (swz128) and epilogue as igemm3.hip.  Preconditions: bf16, Cin % 64 == 0, KH*KW <= 30, tensors < 2 GiB.
#include <atomic>

#include "common.h"
#include "pgt_internal.h"
#include "igemm_common.h"
#include "igemm_epi.h"

// Probe hooks (tools/igemm4_probe.hip compiles this file with -DPGT_PROBE=<bits>; the library build has none):
//   1 no DMA in the main loop   2 no fragment reads   4 no s_setprio   8 no wave stagger   16 no epilogue
//   32 time stamps (s_memtime) of wave 0 into g_pgt_probe_ts[block][8]: start, loop start, loop end, end,
//      addresses ready (before the first DMA), prologue DMA issued
#ifndef PGT_PROBE
#define PGT_PROBE 0
#endif
#if PGT_PROBE & 32
__device__ unsigned long long g_pgt_probe_ts[4096][8];
#define PGT_STAMP(i) do { if (tid == 0) g_pgt_probe_ts[blockIdx.x & 4095][i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define PGT_STAMP(i) do {} while (0)
#endif

namespace {

#define PGT_FENCE() do { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define PGT_BARRIER() do { PGT_FENCE(); __builtin_amdgcn_s_barrier(); PGT_FENCE(); } while (0)
#define PGT_VMWAIT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")

// dynamic LDS: two K-tile stages, or the fp32 epilogue stage of WR*64 rows if that is larger
constexpr int lds_bytes4(int wr, int wc) {
    const int stages = 2 * (wr * 128 + wc * 64) * 128, epi = wr * 64 * (wc * 64 + 4) * 4;   // = epi_stage_bytes<wr, wc>()
    return stages > epi ? stages : epi;
}

// W2: exact-weight form (pgt_conv_desc::w2): the weight matrix has 2 * ceil32(Cout) rows, per 32 output channels [32 rows w_hi |
// 32 rows w_lo * 2048], so a wave's 64 tile columns are the hi and lo products of the SAME 32 output channels: after the main
// loop acc[.][0] += acc[.][1] / 2048 and the epilogue runs on a tile of half the columns.  The operand tile is staged once and
// multiplied by both planes (2x the MFMA and weight traffic, 1x the activation traffic).
template <int WR, int WC, bool UPS, bool X3 = false, bool GN = false, typename T = bf16_t, bool W2 = false>
__global__ __launch_bounds__(512) void igemm4_kernel(ConvP p) {
    static_assert(!(W2 && X3), "exact weights are a single-plane form");
    static_assert(WR * WC == 8, "8 waves");
    constexpr int BM = WR * 128, BN = WC * 64;
    constexpr int TILE_A = BM * 128, TILE_B = BN * 128, STAGE = TILE_A + TILE_B;   // bytes; K tile = 64 bf16 = 128 B
    constexpr int PA = WR, PB = WC / 2;          // DMA pieces (8 rows x 128 B) per wave in one A / B unit
    constexpr int NV = 2 * (PA + PB);            // pieces issued in any 4 consecutive phases = allowed in flight
    constexpr int SA = PA > 2 ? PA / 2 : 0;      // A pieces moved to the neighbouring B phase (evens out 4-2-... issue)
    constexpr int kLds = lds_bytes4(WR, WC);
    static_assert(PB >= 1 && 2 * STAGE <= kLds && kLds <= 160 * 1024, "tile");
    constexpr unsigned kOob = 0x80000000u;
    extern __shared__ __attribute__((aligned(1024))) char smem[];   // kLds bytes

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WC, wc = wave % WC;
    const int late = wave >> 2;   // waves 4-7 share SIMDs with 0-3 and run one barrier behind them
    const int nblk = p.nbm * p.nbn;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, q8 = nblk >> 3, r8 = nblk & 7;
    const int sw = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    const int m0 = (sw / p.nbn) * BM;
    const int n0 = (sw % p.nbn) * BN;
    const unsigned lds0 = lds_addr(smem);
    PGT_STAMP(0);
#if PGT_PROBE & 32
    if (tid == 0) {   // where and when (constant 100 MHz clock) this workgroup started
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        g_pgt_probe_ts[blockIdx.x & 4095][6] = ((unsigned long long)xcc << 32) | hw;
        g_pgt_probe_ts[blockIdx.x & 4095][7] = __builtin_readcyclecounter();
    }
#endif

    // ---- DMA roles.  A piece (h, g): tile rows g*128 + h*64 + wave*8 .. +7 (g < WR).  B piece (h, g): j = wave + 8g
    //      (g < PB), tile columns (j>>2)*64 + h*32 + (j&3)*8 .. +7.  Lane l lands in slot (l & 15) of super row
    //      r0/2 + (l >> 4) and therefore fetches the inverse-swizzled (row, chunk).
    //      a_pix = byte offset of the input pixel under filter tap (0,0) (may be "negative" for padding: only used
    //      when the tap is valid), a_mask = valid-tap bits (+ the output pixel's y/x parity in bits 30/31 for UPS).
    const v4i rsrc_x = make_rsrc(p.x, (unsigned)((long)p.N * p.H * p.W * p.ldx * 2));
    const v4i rsrc_w = make_rsrc(p.w, (unsigned)((long)p.nw * p.K * 2));
    const int Hv = UPS ? 2 * p.H : p.H, Wv = UPS ? 2 * p.W : p.W;   // virtual (up-sampled) input size
    const bool pointwise = p.KH == 1 && p.KW == 1 && p.stride == 1 && p.pad_t == 0 && p.pad_l == 0;
    int a_pix[2][PA];
    unsigned a_mask[2][PA], a_sel[2][PA], b_off[2][PB];
    int ky = 0, kx = 0, c0 = 0;   // filter tap / first channel of the K tile whose A units are issued next (uniform)
    // X3: every 64-channel block of a tap is visited three times, K order [x_hi | x_lo | x_hi] per block (the weights hold
    // [w_hi | w_hi | w_lo] per block): `seg` is the visit, a_soff the byte offset of the K tile inside the pixel row.
    int seg = 0, a_soff = 0;

    auto setup_b = [&](int h) {
#pragma unroll
        for (int g = 0; g < PB; ++g) {
            const int j = wave + 8 * g;
            const int cc = (j >> 2) * 64 + h * 32 + (j & 3) * 8;
            const int sr = (cc >> 1) + (lane >> 4);
            const int slot = (lane & 15) ^ (sr & 15);
            const int n = n0 + 2 * sr + (slot >> 3);
            b_off[h][g] = n < p.nw ? (unsigned)((n * p.K + (slot & 7) * 8) * 2) : kOob;
        }
    };
    auto setup_a = [&](int h) {   // 32-bit arithmetic throughout: every tensor is < 2 GiB
#pragma unroll
        for (int g = 0; g < PA; ++g) {
            const int r0 = g * 128 + h * 64 + wave * 8;
            const int sr = (r0 >> 1) + (lane >> 4);
            const int slot = (lane & 15) ^ (sr & 15);
            const int m = m0 + 2 * sr + (slot >> 3);
            const int c8 = (slot & 7) * 8;
            unsigned mk = 0;
            int pix = 0;
            if (m < p.M) {
                if (!UPS && pointwise) {   // 1x1, stride 1, no padding: the output row IS the input pixel
                    pix = (m * p.ldx + c8) * 2;
                    mk = 1u;
                } else {
                    int ox, oy, img;
                    if (p.wo_shift >= 0) {   // power-of-two feature maps: no integer division
                        ox = m & (p.Wo - 1);
                        const int t = m >> p.wo_shift;
                        oy = t & (p.Ho - 1);
                        img = t >> p.ho_shift;
                    } else {
                        ox = m % p.Wo;
                        const int t = m / p.Wo;
                        oy = t % p.Ho;
                        img = t / p.Ho;
                    }
                    const int iy0 = oy * p.stride - p.pad_t, ix0 = ox * p.stride - p.pad_l;   // virtual coordinates
                    if (UPS) {
                        pix = (((img * p.H + (oy >> 1)) * p.W + (ox >> 1)) * p.ldx + c8) * 2;
                        mk = ((unsigned)(oy & 1) << 30) | ((unsigned)(ox & 1) << 31);
                    } else {
                        pix = (((img * p.H + iy0) * p.W + ix0) * p.ldx + c8) * 2;
                    }
                    unsigned xb = 0;   // valid columns of one filter row, replicated for every valid filter row
                    for (int fx = 0; fx < p.KW; ++fx) xb |= ((unsigned)(ix0 + fx) < (unsigned)Wv ? 1u : 0u) << fx;
                    for (int fy = 0; fy < p.KH; ++fy)
                        if ((unsigned)(iy0 + fy) < (unsigned)Hv) mk |= xb << (fy * p.KW);
                }
            }
            a_pix[h][g] = pix;
            a_mask[h][g] = mk;
        }
    };
    auto select_tap = [&](int h) {   // per-lane source offset of tap (ky, kx), or kOob
        const int tap = ky * p.KW + kx;
#pragma unroll
        for (int g = 0; g < PA; ++g) {
            int off;
            if (UPS) {   // source pixel = floor(virtual / 2): the step depends on the output pixel's parity
                const int dy = ((int)((a_mask[h][g] >> 30) & 1u) - p.pad_t + ky) >> 1;
                const int dx = ((int)(a_mask[h][g] >> 31) - p.pad_l + kx) >> 1;
                off = (dy * p.W + dx) * p.ldx * 2;
            } else {
                off = (ky * p.W + kx) * p.ldx * 2;
            }
            a_sel[h][g] = ((a_mask[h][g] >> tap) & 1u) ? (unsigned)(a_pix[h][g] + off) : kOob;
        }
    };
    // DMA of pieces [g0, g1) of A unit h / of B unit h into stage `buf`
    auto issue_a = [&](int h, int buf, int g0, int g1) {
#pragma unroll
        for (int g = 0; g < PA; ++g)
            if (g >= g0 && g < g1)
                bufdma16(a_sel[h][g], rsrc_x, a_soff, lds0 + buf * STAGE + (g * 128 + h * 64 + wave * 8) * 128);
    };
    auto advance = [&]() {
        if (X3 && ++seg < (p.x3 == 2 ? 2 : 3)) {
            // next plane of the SAME 64-channel block: K order per block is [x_hi | x_lo | x_hi] (weights [w_hi | w_hi | w_lo]),
            // so the second read of the hi plane follows the first by two K tiles and hits the cache.  Folded form (x3 == 2,
            // 64 output channels): [x_hi | x_lo] against rows [w_hi | w_hi] and, in the other tile half, [w_lo | 0]
            a_soff = (c0 + (seg == 1 ? p.xlo : 0)) * 2;
            return;
        }
        seg = 0;
        c0 += 64;
        if (c0 == p.Cin) {
            c0 = 0;
            if (++kx == p.KW) { kx = 0; ++ky; }
            select_tap(0);
            select_tap(1);
        }
        a_soff = c0 * 2;
    };
    auto issue_b = [&](int h, int buf, int kt) {
#pragma unroll
        for (int g = 0; g < PB; ++g) {
            const int j = wave + 8 * g;
            bufdma16(b_off[h][g], rsrc_w, kt * 128, lds0 + buf * STAGE + TILE_A + ((j >> 2) * 64 + h * 32 + (j & 3) * 8) * 128);
        }
    };

    // ---- prologue: K tile 0 (B0 B1 A0 A1) and the first two units of K tile 1 (A0 B0); each DMA leaves as soon as
    //      its addresses exist, so the first (cold) transfers overlap the remaining address arithmetic.
    const int nk = p.K / 64;
    setup_b(0);
    issue_b(0, 0, 0);
    setup_b(1);
    issue_b(1, 0, 0);
    setup_a(0);
    select_tap(0);
    issue_a(0, 0, 0, PA);
    setup_a(1);
    select_tap(1);
    issue_a(1, 0, 0, PA);
    PGT_STAMP(4);
    advance();
    if (nk > 1) {
        issue_a(0, 1, 0, PA);
        issue_b(0, 1, 1);
    }

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
        b_rd[ks] = TILE_A + swz128(wc * 64 + (lane & 31), 2 * ks + hh);
    }

    // A0 and B0 of K tile 0 must have landed: everything issued after A0 may stay in flight
    if (nk > 1) PGT_VMWAIT(2 * PA + PB);
    else PGT_VMWAIT(PA);
    PGT_STAMP(5);
    PGT_BARRIER();
    if (!(PGT_PROBE & 8) && late) PGT_BARRIER();

    uint4 fa[2][4], fb0[4], fb1[4];
#if PGT_PROBE & 2
    for (int ks = 0; ks < 4; ++ks) fa[0][ks] = fa[1][ks] = fb0[ks] = fb1[ks] = make_uint4(lane, ks, 0, 0);
#endif
    PGT_STAMP(1);

#define PGT_PHASE(Q, BUF, KT)                                                                                          \
    {                                                                                                                  \
        const char* st_ = smem + (BUF) * STAGE;                                                                        \
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
            if ((Q) == 0) { issue_b(1, (BUF) ^ 1, (KT) + 1); issue_a(1, (BUF) ^ 1, 0, SA); }                           \
            else if ((Q) == 1) { issue_a(1, (BUF) ^ 1, SA, PA); advance(); }                                           \
            else if ((Q) == 2) issue_a(0, (BUF), 0, PA - SA);                                                          \
            else { issue_a(0, (BUF), PA - SA, PA); issue_b(0, (BUF), (KT) + 2); }                                      \
            PGT_VMWAIT(NV);                                                                                            \
        } else {                                                                                                       \
            PGT_VMWAIT(0);                                                                                             \
        }                                                                                                              \
        PGT_BARRIER();                                                                                                 \
        if (!(PGT_PROBE & 4)) __builtin_amdgcn_s_setprio(1);                                                           \
        _Pragma("unroll") for (int ks = 0; ks < 4; ++ks)                                                               \
            _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                              \
                acc[((Q) >> 1) * 2 + i][((Q) == 1 || (Q) == 2) ? 1 : 0] = mma16<T>(                                    \
                    fa[i][ks], ((Q) == 1 || (Q) == 2) ? fb1[ks] : fb0[ks],                                             \
                    acc[((Q) >> 1) * 2 + i][((Q) == 1 || (Q) == 2) ? 1 : 0]);                                          \
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
    if (!(PGT_PROBE & 8) && !late) PGT_BARRIER();
    PGT_VMWAIT(0);
    __syncthreads();
    PGT_STAMP(2);
#if PGT_PROBE & 16
    if (acc[0][0][0] + acc[1][1][3] + acc[2][0][5] + acc[3][1][7] == 12345.f) reinterpret_cast<float*>(p.y)[tid] = 1.f;
    PGT_STAMP(3);
    return;
#endif

    if constexpr (W2) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][0][e] = __builtin_fmaf(acc[i][1][e], kW2Inv, acc[i][0][e]);
        epilogue_128x64<WR, WC, false, GN, T, 1>(p, acc, smem, m0, n0 >> 1, tid, lane, wr, wc);
    } else {
        epilogue_128x64<WR, WC, X3, GN, T>(p, acc, smem, m0, n0, tid, lane, wr, wc);
    }
    PGT_STAMP(3);
}

template <int WR, int WC, bool UPS, bool X3 = false, bool GN = false, typename T = bf16_t, bool W2 = false> int launch4(const ConvP& p0, hipStream_t st) {
    ConvP p = p0;
    constexpr int BM = WR * 128, BN = WC * 64, bytes = lds_bytes4(WR, WC);
    const bool pow2 = (p.Wo & (p.Wo - 1)) == 0 && (p.Ho & (p.Ho - 1)) == 0;
    p.wo_shift = pow2 ? __builtin_ctz(p.Wo) : -1;
    p.ho_shift = pow2 ? __builtin_ctz(p.Ho) : -1;
    p.nbm = (p.M + BM - 1) / BM;
    p.nbn = (p.nw + BN - 1) / BN;
    if (GN) PGT_CHECK(p.gn_hw % BM == 0 && (W2 ? BN / 2 : BN) % p.gn_cpg == 0, "igemm4: GroupNorm statistics need HW %% %d == 0 (HW=%d) and whole groups per tile", BM, p.gn_hw);
    // the attribute is per device and per function: set it once per (device, instantiation), thread-safe
    static std::atomic<unsigned long long> attr_set{0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!((attr_set.load(std::memory_order_acquire) >> (dev & 63)) & 1ull)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm4_kernel<WR, WC, UPS, X3, GN, T, W2>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e != hipSuccess) { pgt_set_error("igemm4: cannot reserve %d B of LDS: %s", bytes, hipGetErrorString(e)); return -12; }
        attr_set.fetch_or(1ull << (dev & 63), std::memory_order_release);
    }
    hipLaunchKernelGGL((igemm4_kernel<WR, WC, UPS, X3, GN, T, W2>), dim3(p.nbm * p.nbn), dim3(512), bytes, st, p);
    PGT_LAUNCH_CHECK();
    return 0;
}

}  // namespace

// bf16, Cin % 64 == 0, KH*KW <= 30, tensors < 2 GiB, 16-byte-legal epilogue (caller checks).  bn = 256: 256x256
// tiles; bn = 128: 512x128 tiles.  Returns 1 if the tile is not built.
int pgt_igemm4_launch(const void* pv, int bn, hipStream_t st) {
    const ConvP& p = *reinterpret_cast<const ConvP*>(pv);
    if (p.w2) {   // exact-weight form: IEEE half, plain (not up-sampled) inputs; bn counts WEIGHT rows (2 per output channel)
        if (!p.f16 || p.x3 || p.ups) return 1;
        if (p.gn_part) {
            if (bn == 256) return launch4<2, 4, false, false, true, half_t, true>(p, st);
            if (bn == 128) return launch4<4, 2, false, false, true, half_t, true>(p, st);
            return 1;
        }
        if (bn == 256) return launch4<2, 4, false, false, false, half_t, true>(p, st);
        if (bn == 128) return launch4<4, 2, false, false, false, half_t, true>(p, st);
        return 1;
    }
    if (p.x3) {   // split operands on two half planes (no up-sampled inputs on that path)
        if (p.ups) return 1;
        if (p.gn_part) {
            if (bn == 256) return launch4<2, 4, false, true, true, x3p_t>(p, st);
            if (bn == 128) return launch4<4, 2, false, true, true, x3p_t>(p, st);
            return 1;
        }
        if (bn == 256) return launch4<2, 4, false, true, false, x3p_t>(p, st);
        if (bn == 128) return launch4<4, 2, false, true, false, x3p_t>(p, st);
        return 1;
    }
    if (p.f16) {   // IEEE half operands (PGT_F16): the same schedule on v_mfma_f32_32x32x16_f16
        if (p.gn_part) {
            if (p.ups) return 1;
            if (bn == 256) return launch4<2, 4, false, false, true, half_t>(p, st);
            if (bn == 128) return launch4<4, 2, false, false, true, half_t>(p, st);
            return 1;
        }
        if (bn == 256) return p.ups ? launch4<2, 4, true, false, false, half_t>(p, st) : launch4<2, 4, false, false, false, half_t>(p, st);
        if (bn == 128) return p.ups ? launch4<4, 2, true, false, false, half_t>(p, st) : launch4<4, 2, false, false, false, half_t>(p, st);
        return 1;
    }
    if (p.gn_part) {   // epilogue statistics: plain (not up-sampled) inputs
        if (p.ups) return 1;
        if (bn == 256) return launch4<2, 4, false, false, true>(p, st);
        if (bn == 128) return launch4<4, 2, false, false, true>(p, st);
        return 1;
    }
    if (bn == 256) return p.ups ? launch4<2, 4, true>(p, st) : launch4<2, 4, false>(p, st);
    if (bn == 128) return p.ups ? launch4<4, 2, true>(p, st) : launch4<4, 2, false>(p, st);
    return 1;
}
