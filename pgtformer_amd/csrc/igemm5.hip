// 3x3-style convolutions with horizontal tap reuse: the igemm4.hip schedule (256x256 tile, 8 waves, phase-interleaved,
// staggered wave halves) fed with ONE LDS image of the input rows per (filter row ky, 64-channel block) that serves the
// three horizontal taps kx = 0, 1, 2 - the A fragments of tap kx are the same LDS rows shifted by kx pixels.
//
// Why: with every tap fetched separately igemm4 moves 64 KiB of L2 -> LDS per 64-deep K tile; sharing the A image between
// the kx taps cuts the A side to a third (32 -> ~11.6 KiB per K tile, 44 KiB in total) and the DMA instruction count
// per wave from 8 to 5.7 per K tile.  Measured (tools/igemm4_probe.hip -DPGT_PROBE_V5, 256->256 3x3 @ 12x128x128):
// 598 ticks per phase against 619 for igemm4 and 527 with neither DMA nor fragment reads; DMA instructions cost ~14,
// their memory traffic ~33, fragment reads ~24 ticks per phase.  Both kernels are within 15 % of the MFMA-only loop;
// what remains is the clock (1.6-1.9 GHz under MFMA load) and the per-tile setup + epilogue (~18 % of a tile at
// K = 2304).  The autotuner picks this kernel where it wins (small margins).
//
//   * K order: for ky, for channel block cb (64 ch), for kx - a GROUP is the 3 K tiles (12 phases) of one (ky, cb).
//   * A image of a group ("extended tile"): the tile's 256 output pixels are consecutive pixels of the feature map, i.e.
//     BM/S segments of S = min(W, 256) pixels of one image row; each segment is stored with one extra pixel on both
//     sides (S + 2 rows of 128 B, out-of-image pixels = zeros via out-of-range DMA offsets).  Output pixel x of a
//     segment reads row x + kx.  The image is XOR-swizzled with key (row >> 1) & 7 on the 16-byte chunk index, which
//     is bank-conflict free for ds_read_b128 fragment reads at ANY row offset (8 rows of equal parity in a 16-lane
//     group always carry 8 different keys), so the shifted reads cost nothing.
//   * double-buffered A images: group g+1 is fetched during group g, one 1-KiB piece per wave in phases 0, 2, 4, 5, 6
//     (the phases with fragment reads carry one DMA); the B operand lives in a RING of five 16-KiB units (unit 2kt =
//     B0 of K tile kt, 2kt+1 = B1), unit v = 2 pieces per wave is issued in the odd phase 2v - 7 and first read in
//     phase 2v (B0) / 2v - 1 (B1): 6-7 phases ahead (the DMA round trip under load is ~2400 cycles).
//   * every phase issues the same DMAs in every group (past the end of K they read out of range and land in a buffer
//     nobody reads), so the wait before the phase's first barrier is the compile-time constant
//     vmcnt(DMAs issued in the 5 newest phases): it retires everything issued >= 5 phases earlier, which covers every
//     read of the next phase; all units are issued >= 6 phases ahead of their first read and >= 2 phases after the
//     last read of the bytes they replace (same RAW / WAR argument as igemm4.hip).
//
// Preconditions (checked by the caller): bf16, stride 1, no up-sampling, KW == 3, pad_l == 1, Ho == H, Wo == W, W and H
// powers of two, W >= 32, Cin % 64 == 0, (H * W) % 256 == 0, tensors < 2 GiB.
#include "common.h"
#include "pgt_internal.h"
#include "igemm_common.h"
#include "igemm_epi.h"

// Probe hooks (tools/igemm4_probe.hip with -DPGT_PROBE_V5 -DPGT_PROBE=<bits>; none in the library build):
//   1 every main-loop DMA reads out of range (issue + LDS zero-fill, no memory traffic)   2 no main-loop DMA
//   4 no vmcnt wait in the phases (racy: timing only)   8 no fragment reads   32 s_memtime stamps
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

constexpr int kNPA = 5;                       // A pieces (8 rows x 128 B) per wave and group: 40 KiB >= 272 rows
constexpr int kABuf = kNPA * 8 * 1024;        // bytes of one A image
constexpr int kUnitB = 128 * 128;             // bytes of one B unit (128 columns x 64 k)
constexpr int kRing = 5;                      // B units resident
constexpr int kOffB = 2 * kABuf;
constexpr int kLds5 = 2 * kABuf + kRing * kUnitB;   // 160 KiB (>= the 130 KiB epilogue stage)
static_assert(kLds5 >= epi_stage_bytes<2, 4>() && kLds5 <= 160 * 1024, "LDS budget");

// DMAs one wave issues in phase p of a group (period 12): one A piece in phases 0, 2, 4, 5, 6 (the phases that carry
// fragment reads get one DMA), one B unit = 2 pieces in every odd phase (few or no fragment reads); and the count of
// the 5 newest phases
constexpr int issued_a(int p) { return (p == 0 || p == 2 || p == 4 || p == 5 || p == 6) ? 1 : 0; }
constexpr int a_piece(int p) { return p == 0 ? 0 : p == 2 ? 1 : p == 4 ? 2 : p == 5 ? 3 : 4; }
constexpr int issued(int p) { return issued_a(p) + (p % 2 == 1 ? 2 : 0); }
constexpr int newest5(int p) {
    int n = 0;
    for (int d = 0; d < 5; ++d) n += issued((p + 12 - d) % 12);
    return n;
}

// byte offset of (row e, 16-byte chunk c) in the extended A image
__device__ __forceinline__ int swzx(int e, int c) { return e * 128 + ((c ^ ((e >> 1) & 7)) << 4); }

template <typename T>   // 16-bit operand type: bf16_t or half_t (PGT_F16)
__global__ __launch_bounds__(512) void igemm5_kernel(ConvP p) {
    constexpr unsigned kOob = 0x80000000u;
    extern __shared__ __attribute__((aligned(1024))) char smem[];   // kLds5 bytes

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int late = wave >> 2;
    const int nblk = p.nbm * p.nbn;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, q8 = nblk >> 3, r8 = nblk & 7;
    const int sw = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    const int m0 = (sw / p.nbn) * 256;
    const int n0 = (sw % p.nbn) * 256;
    const unsigned lds0 = lds_addr(smem);
    PGT_STAMP(0);

    const v4i rsrc_x = make_rsrc(p.x, (unsigned)((long)p.N * p.H * p.W * p.ldx * 2));
    const v4i rsrc_w = make_rsrc(p.w, (unsigned)((long)p.Cout * p.K * 2));
    const int S = p.W < 256 ? p.W : 256;          // pixels per segment (power of two >= 32)
    const int s_shift = p.W < 256 ? p.wo_shift : 8;
    const int S2 = S + 2, segs = 256 >> s_shift, E = 256 + 2 * segs;
    const int ncb = p.Cin >> 6;
    const int row_bytes = p.W * p.ldx * 2;

    // ---- B roles: unit h holds columns (c, h, 0..31), c = 0..3, as local rows c*32 + (0..31) (swz128 on local rows);
    //      piece j = wave + 8g covers local rows 8j .. 8j+7
    unsigned b_off[2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const int sr = (wave + 8 * g) * 4 + (lane >> 4);
            const int slot = (lane & 15) ^ (sr & 15);
            const int lr = 2 * sr + (slot >> 3);
            const int n = n0 + (lr >> 5) * 64 + h * 32 + (lr & 31);
            b_off[h][g] = n < p.Cout ? (unsigned)((n * p.K + (slot & 7) * 8) * 2) : kOob;
        }
    // B iterator: soffset of the weight K tile (ky, kx, cb) = ((ky*KW + kx)*Cin + cb*64) * 2 bytes, in K-tile order
    struct BIter {
        int kx, cb, base, soff;   // base = soffset of (ky, kx = 0, cb = 0)
        __device__ __forceinline__ void next(int ncb_, int cin2, int kw) {
            if (++kx == 3) {
                kx = 0;
                if (++cb == ncb_) { cb = 0; base += kw * cin2; }
            }
            soff = base + kx * cin2 + cb * 128;
        }
    };
    const int cin2 = p.Cin * 2;
    auto issue_b = [&](int h, int slot, int soff) {
#pragma unroll
        for (int g = 0; g < 2; ++g)
            bufdma16(b_off[h][g], rsrc_w, soff, lds0 + kOffB + slot * kUnitB + (wave + 8 * g) * 1024);
    };
    BIter itb{0, 0, 0, 0};   // K tile whose B units are issued next
    int ws = 0;              // ring slot written next
    issue_b(0, 0, 0);        // unit 0 = B0(kt 0): needed in phase 0, so it leaves first
    ws = 1;

    // ---- A roles: piece i of this wave = image rows 8 (wave + 8 i) + (lane >> 3), physical chunk lane & 7
    int a_pix[kNPA];
    unsigned a_msk[kNPA], a_sel[kNPA];
#pragma unroll
    for (int i = 0; i < kNPA; ++i) {
        const int e = 8 * (wave + 8 * i) + (lane >> 3);
        const int c = (lane & 7) ^ ((e >> 1) & 7);
        int seg = 0;
        for (int k = 1; k < segs; ++k) seg += e >= k * S2 ? 1 : 0;
        const int xx = e - seg * S2;
        const int mseg = m0 + (seg << s_shift);
        unsigned mk = 0;
        int pix = 0;
        if (e < E && mseg < p.M) {
            const int ox0 = mseg & (p.W - 1);
            const int t = mseg >> p.wo_shift;
            const int oy = t & (p.H - 1), img = t >> p.ho_shift;
            const int ix = ox0 - p.pad_l + xx;
            if ((unsigned)ix < (unsigned)p.W) {
                pix = (((img * p.H + oy - p.pad_t) * p.W + ix) * p.ldx + c * 8) * 2;
                for (int ky = 0; ky < p.KH; ++ky) mk |= ((unsigned)(oy - p.pad_t + ky) < (unsigned)p.H ? 1u : 0u) << ky;
            }
        }
        a_pix[i] = pix;
        a_msk[i] = mk;
    }
    int ky_a = 0, cb_a = 0;   // group whose A image is issued next
    auto select_row = [&]() {
#pragma unroll
        for (int i = 0; i < kNPA; ++i)
            a_sel[i] = ((a_msk[i] >> ky_a) & 1u) ? (unsigned)(a_pix[i] + ky_a * row_bytes) : kOob;
    };
    auto issue_a = [&](int i, int buf) {
        bufdma16(a_sel[i], rsrc_x, cb_a * 128, lds0 + buf * kABuf + (wave + 8 * i) * 1024);
    };
    auto next_group = [&]() {
        if (++cb_a == ncb) {
            cb_a = 0;
            ++ky_a;
            select_row();
        }
    };
    select_row();
#pragma unroll
    for (int i = 0; i < kNPA; ++i) issue_a(i, 0);   // A image of group 0
    next_group();
    issue_b(1, 1, 0);         // units 1..3 = B1(kt 0), B0(kt 1), B1(kt 1) may still be in flight at phase 0
    itb.next(ncb, cin2, p.KW);
    issue_b(0, 2, itb.soff);
    issue_b(1, 3, itb.soff);
    itb.next(ncb, cin2, p.KW);
    ws = 4;

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // fragment reads: A row of output pixel r = wr*128 + ih*64 + i*32 + (lane & 31) under tap kx is image row
    // e0[ih][i] + kx; B as igemm4 (swz128).
    const int hh = lane >> 5;
    int e0[2][2], b_rd[4];
#pragma unroll
    for (int ih = 0; ih < 2; ++ih)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = wr * 128 + ih * 64 + i * 32 + (lane & 31);
            e0[ih][i] = (r >> s_shift) * S2 + (r & (S - 1));
        }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) b_rd[ks] = kOffB + swz128(wc * 32 + (lane & 31), 2 * ks + hh);

    const int G = p.KH * ncb;   // groups; K tiles = 3 G
    PGT_VMWAIT(6);              // unit 0 and the A image have landed
    PGT_BARRIER();
    if (late) PGT_BARRIER();

    uint4 fa[2][4], fb0[4], fb1[4];
    int bq[4];
#if PGT_PROBE & 8
    for (int ks = 0; ks < 4; ++ks) fa[0][ks] = fa[1][ks] = fb0[ks] = fb1[ks] = make_uint4(lane, ks, 0, 0);
#endif
#if PGT_PROBE & 1
    for (int i = 0; i < kNPA; ++i) a_msk[i] = 0;
    select_row();
    for (int h = 0; h < 2; ++h) for (int g2 = 0; g2 < 2; ++g2) b_off[h][g2] = kOob;
#endif
    PGT_STAMP(1);
    PGT_STAMP(4);
    PGT_STAMP(5);

#define PGT_READ_A(IH, KX)                                                                                             \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                                    \
        const int e_ = e0[IH][i] + (KX);                                                                               \
        const int ab_ = a_img + e_ * 128 + ((hh ^ ((e_ >> 1) & 7)) << 4);                                              \
        _Pragma("unroll") for (int ks = 0; ks < 4; ++ks)                                                               \
            fa[i][ks] = *reinterpret_cast<const uint4*>(smem + (ab_ ^ (ks << 5)));                                     \
    }

#define PGT_PHASE5(P)                                                                                                  \
    {                                                                                                                  \
        constexpr int J_ = (P) / 4, Q_ = (P) % 4;                                                                      \
        if (PGT_PROBE & 8) {                                                                                           \
        } else if (Q_ == 0) {                                                                                          \
            _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) bq[ks] = b_rd[ks] + rs * kUnitB;                          \
            _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) fb0[ks] = *reinterpret_cast<const uint4*>(smem + bq[ks]); \
            rs = rs + 1 == kRing ? 0 : rs + 1;                                                                         \
            PGT_READ_A(0, J_)                                                                                          \
        } else if (Q_ == 1) {                                                                                          \
            _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) bq[ks] = b_rd[ks] + rs * kUnitB;                          \
            _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) fb1[ks] = *reinterpret_cast<const uint4*>(smem + bq[ks]); \
            rs = rs + 1 == kRing ? 0 : rs + 1;                                                                         \
        } else if (Q_ == 2) {                                                                                          \
            PGT_READ_A(1, J_)                                                                                          \
        }                                                                                                              \
        asm volatile("" ::: "memory");                                                                                 \
        if (issued_a(P) && !(PGT_PROBE & 2)) issue_a(a_piece(P), (g + 1) & 1);                                         \
        if ((Q_ == 1 || Q_ == 3) && !(PGT_PROBE & 2)) {   /* units 2kt+4 = B0(kt+2), 2kt+5 = B1(kt+2) */               \
            issue_b(Q_ >> 1, ws, itb.soff);                                                                            \
            ws = ws + 1 == kRing ? 0 : ws + 1;                                                                         \
            if (Q_ == 3) itb.next(ncb, cin2, p.KW);                                                                    \
        }                                                                                                              \
        if (!(PGT_PROBE & 6)) PGT_VMWAIT(newest5(P));                                                                                      \
        PGT_BARRIER();                                                                                                 \
        __builtin_amdgcn_s_setprio(1);                                                                                 \
        _Pragma("unroll") for (int ks = 0; ks < 4; ++ks)                                                               \
            _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                              \
                acc[(Q_ >> 1) * 2 + i][(Q_ == 1 || Q_ == 2) ? 1 : 0] = mma16<T>(                                       \
                    fa[i][ks], (Q_ == 1 || Q_ == 2) ? fb1[ks] : fb0[ks],                                               \
                    acc[(Q_ >> 1) * 2 + i][(Q_ == 1 || Q_ == 2) ? 1 : 0]);                                             \
        __builtin_amdgcn_s_setprio(0);                                                                                 \
        PGT_BARRIER();                                                                                                 \
    }

    int rs = 0;   // ring slot read next
    for (int g = 0; g < G; ++g) {
        const int a_img = (g & 1) * kABuf;
        PGT_PHASE5(0)
        PGT_PHASE5(1)
        PGT_PHASE5(2)
        PGT_PHASE5(3)
        PGT_PHASE5(4)
        PGT_PHASE5(5)
        PGT_PHASE5(6)
        next_group();   // all pieces of group g+1 are out: step the issue state to group g+2
        PGT_PHASE5(7)
        PGT_PHASE5(8)
        PGT_PHASE5(9)
        PGT_PHASE5(10)
        PGT_PHASE5(11)
    }
#undef PGT_PHASE5
#undef PGT_READ_A
    if (!late) PGT_BARRIER();
    PGT_VMWAIT(0);
    __syncthreads();
    PGT_STAMP(2);
    epilogue_128x64<2, 4, false, false, T>(p, acc, smem, m0, n0, tid, lane, wr, wc);
    PGT_STAMP(3);
}

}  // namespace

// See the preconditions at the top of the file; the caller checks them.
int pgt_igemm5_launch(const void* pv, hipStream_t st) {
    ConvP p = *reinterpret_cast<const ConvP*>(pv);
    p.wo_shift = __builtin_ctz(p.Wo);
    p.ho_shift = __builtin_ctz(p.Ho);
    p.nbm = (p.M + 255) / 256;
    p.nbn = (p.Cout + 255) / 256;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm5_kernel<bf16_t>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, kLds5);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm5_kernel<half_t>), hipFuncAttributeMaxDynamicSharedMemorySize, kLds5);
        if (e != hipSuccess) { pgt_set_error("igemm5: cannot reserve %d B of LDS: %s", kLds5, hipGetErrorString(e)); return -12; }
        attr_set = true;
    }
    if (p.f16) hipLaunchKernelGGL(igemm5_kernel<half_t>, dim3(p.nbm * p.nbn), dim3(512), kLds5, st, p);
    else hipLaunchKernelGGL(igemm5_kernel<bf16_t>, dim3(p.nbm * p.nbn), dim3(512), kLds5, st, p);
    PGT_LAUNCH_CHECK();
    return 0;
}
