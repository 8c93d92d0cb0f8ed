// 3x3 convolution, 64 input channels, <= 64 output channels (bf16 / half), W >= 128: the full-resolution (512x512) layers of
// the decoder - with the GroupNorm apply + SiLU of the PRECEDING Normalize fused into the operand load
// (reference: modules/rstt_layers.py:754-758 `Normalize` / `nonlinearity` feeding conv1 / conv2 of TDResnetBlock :875-904 and
// `conv_out`, archs/tdcrqvae3_arch.py:672-707).
//
// igemm6.hip (the kernel this replaces on these layers) loads three rows-with-halo images per 128-pixel tile: every input
// row crosses L2 -> LDS three times, a tile is one serial chain (DMA -> wait -> 72 MFMAs -> LDS stage -> store) and the
// matrix pipes idle 75 % of the time (0.25 of the MFMA peak at 0.36 of HBM: under both roofs).  A transform of the operand
// in LDS would have to touch every element three times there.
//
// Design: a persistent workgroup walks DOWN a strip of 128 columns.  A ring of FOUR row images (130 pixels x 64 channels,
// pixel rows padded to 144 bytes: ds_read_b128 of 32 consecutive pixels is bank-conflict free, 9 r mod 16 being a bijection
// on the lane groups' residues) lives in LDS; an output row needs three of them, the fourth is being filled by LDS-DMA
// while the current row multiplies: every input row crosses L2 -> LDS ONCE (+ 2 halo rows per strip), and the GroupNorm
// apply + SiLU runs ONCE per element, in place on the freshly landed row, by the wave that requested it (no extra barrier).
//   * weights: the 9 x 64 filter of a wave's 32 output channels stays in registers (144 VGPRs) as the MFMA **A** operand,
//     rows permuted so that a lane's 16 accumulators are two runs of 8 CONSECUTIVE channels of one pixel: the epilogue
//     (bias from LDS, activation, residual, rounding) stores 16 bytes straight from the registers - no LDS stage, no barrier;
//   * pixels: B operand, ds_read_b128 with IMMEDIATE offsets only (filter row = ring slot, tap column, k-step and pixel block
//     are constants of the 4x unrolled row loop): one address register, no address arithmetic in the loop;
//   * one barrier per output row; the residual row is requested before the MFMAs, the stores of row y drain under row y + 1;
//   * 77 KiB of LDS and <= 256 VGPRs: two workgroups per CU, one multiplies while the other stores / transforms.
//
// Exact-weight form (W2, pgt_conv_desc::w2; IEEE half): the weight-rounding error of these layers is what the PSNR contract's margin
// hung on (profiles/r5_u_third_point_oracle_ablation.md: -1.7e-3 dB from the 3x3 layers of the 512x512 stage alone), so the kernel
// also exists with BOTH planes of the weights in registers: a wave owns 16 output channels instead of 32 and its 32 MFMA rows
// are [w_hi (16 channels) | w_lo * 2048 (the same 16 channels)] - still 144 weight registers -, row permutation such that a
// lane's accumulators e and e + 8 are the hi and lo products of ONE (pixel, channel): y = acc[e] + acc[e + 8] / 2048.  The four
// waves cover 64 channels x all 128 pixels of the row in two passes of 64 pixels (32 accumulator registers live, as before):
// twice the MFMAs and fragment reads per output row, no extra L2 -> LDS traffic, the fused GroupNorm apply unchanged.
//
// Preconditions (caller): 16-bit single-plane operands, KH = KW = 3, stride 1, pad 1, no up-sampling, Cin == 64, Cout <= 64,
// Ho == H, Wo == W, W a power of two >= 128, H a power of two >= 4, epilogue = bias (+ residual) only (no activation, no
// SFT: what the residual blocks and conv_out need), in_act none or SiLU, input < 2 GiB, residual < 4 GiB.
#include <type_traits>

#include "common.h"
#include "pgt_internal.h"
#include "igemm_common.h"

namespace {

constexpr int kRowB = 144;                       // bytes of one pixel row in LDS (128 + 16 padding)
constexpr int kPxRow = 130;                      // pixels of a row image: 128 + one halo pixel on both sides
constexpr int kPieces8 = 19;                     // 1-KiB DMA pieces per row image (19456 >= 130 * 144 = 18720)
constexpr int kSlot8 = kPieces8 * 1024;
constexpr int kTab8 = 4 * kSlot8;                // tables behind the ring: bias[64] | in_scale[64] | in_shift[64] (fp32)
constexpr int kLds8 = kTab8 + 3 * 256;           // 78592 bytes
static_assert(kPxRow * kRowB <= kSlot8, "row image must fit its slot");
static_assert(3 * kSlot8 + (32 + 2) * kRowB + 3 * 32 + 16 < 65536, "ds_read immediate offsets are 16 bits");

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// 16 bytes per lane through a buffer descriptor into registers, as ONE opaque statement (the compiler neither waits for it
// nor counts it): completion is the caller's `wait_all` below
__device__ __forceinline__ u32x4 bufload16(unsigned voff, v4i rsrc, int soff, int /*imm*/) {
    u32x4 r;
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(r) : "v"(voff), "s"(rsrc), "s"(soff) : "memory");
    return r;
}
__device__ __forceinline__ u32x4 bufload16_32(unsigned voff, v4i rsrc, int soff) {   // the same, 32 bytes further
    u32x4 r;
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:32" : "=v"(r) : "v"(voff), "s"(rsrc), "s"(soff) : "memory");
    return r;
}
// every vector-memory operation of this wave has completed (LDS-DMA pieces landed, residual registers valid)
__device__ __forceinline__ void wait_all(u32x4& a, u32x4& b, u32x4& c, u32x4& d) {
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "memory");
}
__device__ __forceinline__ void wait_all(u32x4& a, u32x4& b) { asm volatile("s_waitcnt vmcnt(0)" : "+v"(a), "+v"(b) : : "memory"); }
__device__ __forceinline__ void wait_all(u32x4& a) { asm volatile("s_waitcnt vmcnt(0)" : "+v"(a) : : "memory"); }
__device__ __forceinline__ void wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

template <typename T> __device__ __forceinline__ void unpack_u(const u32x4& q, float* f) {
    Vec16<T>::unpack(make_uint4(q.x, q.y, q.z, q.w), f);
}

// NCB = 32-channel blocks of the output (2: waves 2 x 2, 64 pixels x 32 channels each; 1: four waves of 32 pixels);
// FUSE: the operand is act(x * in_scale[n][c] + in_shift[n][c]) (the GroupNorm apply of the preceding Normalize + SiLU)
//   W2: exact-weight form - NCB = 2: wave w owns channels 16 w .. 16 w + 15 and the whole row (two passes of two 32-pixel blocks);
//       NCB = 1 (Cout <= 16): four waves of 32 pixels, channels 0 .. 15
template <typename T, int NCB, bool FUSE, bool W2 = false>
__global__ __launch_bounds__(256, 2) void conv3x3_c64_ring_kernel(ConvP p, int R, int nys, int nxs, int nitems) {
    constexpr unsigned kOob = 0x80000000u;
    constexpr int NPB = NCB == 2 ? 2 : 1;         // 32-pixel blocks per wave (and pass)
    constexpr int NPASS = (W2 && NCB == 2) ? 2 : 1;   // passes of 64 pixels over the row
    constexpr int NH2 = W2 ? 1 : 2;               // runs of 8 consecutive output channels per lane
    extern __shared__ __attribute__((aligned(1024))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = W2 ? (NCB == 2 ? 0 : wave) : (NCB == 2 ? wave >> 1 : wave);
    const int wc = W2 ? (NCB == 2 ? wave : 0) : (NCB == 2 ? wave & 1 : 0);
    const int hh = lane >> 5, l31 = lane & 31;
    const unsigned lds0 = lds_addr(smem);
    float* tab = reinterpret_cast<float*>(smem + kTab8);
    const v4i rsrc_x = make_rsrc(p.x, (unsigned)((long)p.N * p.H * p.W * p.ldx * 2));
    const v4i rsrc_r = make_rsrc(p.res ? p.res : p.x, p.res ? (unsigned)((long)p.M * p.ldr * 2) : 0u);

    // ---- weights: A fragment of (tap, ks) for A row i = lane & 31 -> output channel wc * 32 + sigma(i); row 8q + 4h + j of
    // the 32x32 result sits in accumulator e = 4q + j of the lanes with lane >> 5 == h, so sigma(8q + 4h + j) =
    // 16 (q >> 1) + 8 h + 4 (q & 1) + j gives those lanes channels 8h .. 8h + 7 in e = 0..7 and 16 + 8h .. in e = 8..15
    // W2: rows q = 0, 1 carry w_hi and q = 2, 3 w_lo * 2048 of channel wc * 16 + 8 h + 4 (q & 1) + j: e < 8 and e + 8 pair up
    uint4 wreg[9][4];
    {
        const int q = l31 >> 3, h = (l31 >> 2) & 1, j = l31 & 3;
        const int n = W2 ? wc * 16 + 8 * h + 4 * (q & 1) + j : wc * 32 + 16 * (q >> 1) + 8 * h + 4 * (q & 1) + j;
        const bool live = n < p.Cout;      // (rows past Cout multiply zeros)
        // row of the weight matrix: channel n, or in the exact-weight form plane (q >> 1) of channel n
        const long wrow = W2 ? (long)(n >> 5) * 64 + (q >> 1) * 32 + (n & 31) : (long)n;
        const uint4* wp = reinterpret_cast<const uint4*>(p.w + ((live ? wrow : 0) * p.K + hh * 8) * 2);
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                uint4 v = wp[t * 8 + ks * 2];   // (t*64 + ks*16) elements = (t*8 + ks*2) x 16 bytes
                v.x = live ? v.x : 0u; v.y = live ? v.y : 0u; v.z = live ? v.z : 0u; v.w = live ? v.w : 0u;
                wreg[t][ks] = v;
            }
    }
    const int cb = (W2 ? wc * 16 : wc * 32) + 8 * hh;   // this lane's channels: cb .. cb + 7 (and cb + 16 .. cb + 23 without W2)
    const int pxb = (NCB == 2 ? wr * 64 : wr * 32) + l31;      // its pixel (block i: + 32 i; pass ps: + 64 ps) inside the 128-pixel row
    const char* lbase = smem + pxb * kRowB + hh * 16;          // B fragment of (slot, i, kx, ks): + slot*kSlot8 + (32 i + kx)*kRowB + 32 ks
    // chunk (s % 9; 8 = the padding) of the 16 bytes this lane moves in its k-th DMA piece: 4 bits each, the same for every row
    unsigned cpack = 0;
#pragma unroll
    for (int k = 0; k < 5; ++k) cpack |= (unsigned)((64 * (wave + 4 * k) + lane) % 9) << (4 * k);
    const bool vec = NCB == 2 || p.vec_epi != 0;       // (the launcher sends 16-byte-illegal epilogues to the one-block form only)
    const bool has_res = p.res != nullptr;
    const int ysz = p.out_f32 ? 4 : 2;

    for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
        const int xs = item % nxs, t1 = item / nxs;
        const int ys = t1 % nys, img = t1 / nys;
        const int y0 = ys * R, x0 = xs * 128;
        // ---- per item: the DMA lanes' source offsets inside an input row (piece q = wave + 4 k: LDS bytes 1024 q + 16 lane
        // = pixel slot s / 9, chunk s % 9 with s = 64 q + lane; chunk 8 is the padding) and what they hold
        unsigned voff[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const int s = 64 * (wave + 4 * k) + lane;
            const int e = s / 9, c = s - 9 * e;
            const int ix = x0 - 1 + e;
            const bool ok = c < 8 && e < kPxRow && (unsigned)ix < (unsigned)p.W;
            voff[k] = ok ? (unsigned)((ix * p.ldx + c * 8) * 2) : kOob;
        }
        // tables (the previous item's last barrier has passed: nobody reads them any more)
        {
            const int m0 = (img * p.H + y0) * p.W;
            if (tid < 64) tab[tid] = (p.bias && tid < p.Cout) ? bias_of(p, m0)[tid] : 0.f;
            if constexpr (FUSE) {
                if (tid >= 64 && tid < 128) tab[tid] = p.in_scale[(long)img * 64 + tid - 64];
                if (tid >= 128 && tid < 192) tab[tid] = p.in_shift[(long)img * 64 + tid - 128];
            }
        }
        auto issue_row = [&](int rel, int slot) __attribute__((always_inline)) {      // input row y0 - 1 + rel -> ring slot
            const int iy = y0 - 1 + rel;
            const bool rowok = (unsigned)iy < (unsigned)p.H;
            const int soff = rowok ? ((img * p.H + iy) * p.W) * p.ldx * 2 : 0;
#pragma unroll
            for (int k = 0; k < 5; ++k)
                if (k < 4 || wave < kPieces8 - 16)
                    bufdma16(rowok ? voff[k] : kOob, rsrc_x, soff, lds0 + slot * kSlot8 + (wave + 4 * k) * 1024);
        };
        auto transform_row = [&](int rel, int slot) __attribute__((always_inline)) {  // in place on the pieces THIS wave requested (they have landed)
            if constexpr (FUSE) {
                const int iy = y0 - 1 + rel;
                if ((unsigned)iy >= (unsigned)p.H) return;       // a padding row stays zero
                unsigned cp = cpack;
                asm volatile("" : "+v"(cp));      // (opaque: keeps hipcc from hoisting 15 table addresses out of the row loop)
                uint4 d[5];
#pragma unroll
                for (int k = 0; k < 5; ++k)
                    if (k < 4 || wave < kPieces8 - 16) d[k] = *reinterpret_cast<const uint4*>(smem + slot * kSlot8 + (wave + 4 * k) * 1024 + lane * 16);
                    else d[k] = make_uint4(0, 0, 0, 0);
#pragma unroll
                for (int k = 0; k < 5; ++k)
                    if (k < 4 || wave < kPieces8 - 16) {
                        const int c = (cp >> (4 * k)) & 7;      // (the padding chunk reads coefficients of chunk 0: discarded)
                        float f[8], sc[8], sh[8];
                        Vec16<T>::unpack(d[k], f);
                        *reinterpret_cast<float4*>(sc) = *reinterpret_cast<const float4*>(tab + 64 + c * 8);
                        *reinterpret_cast<float4*>(sc + 4) = *reinterpret_cast<const float4*>(tab + 64 + c * 8 + 4);
                        *reinterpret_cast<float4*>(sh) = *reinterpret_cast<const float4*>(tab + 128 + c * 8);
                        *reinterpret_cast<float4*>(sh + 4) = *reinterpret_cast<const float4*>(tab + 128 + c * 8 + 4);
#pragma unroll
                        for (int e = 0; e < 8; ++e) f[e] = f[e] * sc[e] + sh[e];
                        if (p.in_act == ACT_SILU) {          // the form of affine_act_kernel (norms.hip): same bits
#pragma unroll
                            for (int e = 0; e < 8; ++e)
                                f[e] = f[e] * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * f[e]));
                        }
                        uint4 o = Vec16<T>::pack(f);
                        const bool live = voff[k] != kOob;       // padding pixels / the padding chunk keep what the DMA wrote (zeros)
                        o.x = live ? o.x : d[k].x;
                        o.y = live ? o.y : d[k].y;
                        o.z = live ? o.z : d[k].z;
                        o.w = live ? o.w : d[k].w;
                        *reinterpret_cast<uint4*>(smem + slot * kSlot8 + (wave + 4 * k) * 1024 + lane * 16) = o;
                    }
            }
        };

        issue_row(0, 0);
        issue_row(1, 1);
        issue_row(2, 2);
        wait_all();
        if constexpr (FUSE) __syncthreads();           // the coefficient tables are written
        transform_row(0, 0);
        transform_row(1, 1);
        transform_row(2, 2);
        __syncthreads();

        // one output row; PH = (row index inside the strip) & 3 = ring slot of its first filter row
        auto step = [&](auto ph_tag, int t) __attribute__((always_inline)) {
            constexpr int PH = decltype(ph_tag)::value;
            const int y = y0 + t;
            const int mrow = (img * p.H + y) * p.W + x0;            // first pixel of this output row
            if (t + 3 <= R + 1) issue_row(t + 3, (PH + 3) & 3);     // the row the NEXT step needs; its slot was last read in step t - 1
            char* yrow = p.y + (long)mrow * p.ldy * ysz;
#pragma unroll
            for (int ps = 0; ps < NPASS; ++ps) {
            const char* lb = lbase + ps * 64 * kRowB;         // (second pass: + 64 pixels; a register add, the rest stays immediate)
            // the residual chunks of this pass are requested before its MFMAs (the second pass's after the first pass's stores)
            u32x4 rr[NPB][NH2];
#pragma unroll
            for (int i = 0; i < NPB; ++i)
#pragma unroll
                for (int h2 = 0; h2 < NH2; ++h2) rr[i][h2] = u32x4{0u, 0u, 0u, 0u};
            if (has_res && vec) {
                const int soff = (int)((unsigned)mrow * (unsigned)p.ldr * 2u);      // (< 4 GiB by the launcher's check: 32-bit unsigned)
#pragma unroll
                for (int i = 0; i < NPB; ++i) {
                    const unsigned vo = (unsigned)(((pxb + 64 * ps + 32 * i) * p.ldr + cb) * 2);
                    rr[i][0] = bufload16(cb < p.Cout ? vo : kOob, rsrc_r, soff, 0);
                    if constexpr (NH2 == 2) rr[i][1] = bufload16_32(cb + 16 < p.Cout ? vo : kOob, rsrc_r, soff);
                }
            }
            f32x16 acc[NPB];
            {
                float bv[16];
                *reinterpret_cast<float4*>(bv) = *reinterpret_cast<const float4*>(tab + cb);
                *reinterpret_cast<float4*>(bv + 4) = *reinterpret_cast<const float4*>(tab + cb + 4);
                if constexpr (W2) {
#pragma unroll
                    for (int e = 8; e < 16; ++e) bv[e] = 0.f;          // the lo products start from zero
                } else {
                    *reinterpret_cast<float4*>(bv + 8) = *reinterpret_cast<const float4*>(tab + cb + 16);
                    *reinterpret_cast<float4*>(bv + 12) = *reinterpret_cast<const float4*>(tab + cb + 20);
                }
#pragma unroll
                for (int i = 0; i < NPB; ++i)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[i][e] = bv[e];
            }
            // 36 groups g = (tap, k-step) of NPB products; the B fragments of group g + 2 are requested before group g
            // multiplies (ds_read latency ~ two groups of MFMA time); sched_barrier keeps hipcc from re-serialising them
            {
                constexpr int NG = 36, D = 2;
                uint4 fb[D + 1][NPB];
                auto frag = [&](int g, int i) __attribute__((always_inline)) {
                    const int tap = g >> 2, ks = g & 3, ky = tap / 3, kx = tap - 3 * ky;
                    return *reinterpret_cast<const uint4*>(lb + ((PH + ky) & 3) * kSlot8 + (32 * i + kx) * kRowB + 32 * ks);
                };
#pragma unroll
                for (int g = 0; g < D; ++g)
#pragma unroll
                    for (int i = 0; i < NPB; ++i) fb[g][i] = frag(g, i);
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    if (g + D < NG) {
#pragma unroll
                        for (int i = 0; i < NPB; ++i) fb[(g + D) % (D + 1)][i] = frag(g + D, i);
                    }
#pragma unroll
                    for (int i = 0; i < NPB; ++i) acc[i] = mma16<T>(wreg[g >> 2][g & 3], fb[g % (D + 1)][i], acc[i]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // the row requested at the top of the step has landed by now, and so have this pass's residual chunks
            if constexpr (NPB == 2 && NH2 == 2) wait_all(rr[0][0], rr[0][1], rr[1][0], rr[1][1]);
            else if constexpr (NPB == 2) wait_all(rr[0][0], rr[1][0]);
            else if constexpr (NH2 == 2) wait_all(rr[0][0], rr[0][1]);
            else wait_all(rr[0][0]);
            // ---- epilogue straight from the accumulators: 8 consecutive channels of one pixel per 16-byte store
#pragma unroll
            for (int i = 0; i < NPB; ++i)
#pragma unroll
                for (int h2 = 0; h2 < NH2; ++h2) {
                    const int ch = cb + 16 * h2;
                    if (ch >= p.Cout) continue;
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = W2 ? __builtin_fmaf(acc[i][8 + e], kW2Inv, acc[i][e]) : acc[i][8 * h2 + e];
                    const int px = pxb + 64 * ps + 32 * i;
                    const int yo = px * p.ldy + ch;       // (element offset inside the output row: small)
                    if (vec) {
                        if (has_res) {
                            float r[8];
                            unpack_u<T>(rr[i][h2], r);
#pragma unroll
                            for (int e = 0; e < 8; ++e) v[e] += r[e];
                        }
                        if (p.out_f32) store8<float>(reinterpret_cast<float*>(yrow) + yo, v);
                        else store8<T>(reinterpret_cast<T*>(yrow) + yo, v);
                    } else {           // Cout % 8 != 0 or unaligned rows (the 64 -> 3 output conv): one value at a time
                        const T* res = reinterpret_cast<const T*>(p.res);
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            if (ch + e >= p.Cout) continue;
                            float u = v[e];
                            if (has_res) u += ldf(res + ((long)mrow + px) * p.ldr + ch + e);
                            if (p.out_f32) reinterpret_cast<float*>(yrow)[yo + e] = u;
                            else stf(reinterpret_cast<T*>(yrow) + yo + e, u);
                        }
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);      // (keeps the transform's registers out of the epilogue's live range)
            if (t + 3 <= R + 1) transform_row(t + 3, (PH + 3) & 3);
            __syncthreads();
        };
#pragma unroll 1
        for (int t = 0; t < R; t += 4) {
            step(std::integral_constant<int, 0>{}, t);
            step(std::integral_constant<int, 1>{}, t + 1);
            step(std::integral_constant<int, 2>{}, t + 2);
            step(std::integral_constant<int, 3>{}, t + 3);
        }
    }
}

template <typename T, int NCB, bool FUSE, bool W2 = false> int launch8(const ConvP& p, int R, int nys, int nxs, int nitems, int grid, hipStream_t st) {
    static bool configured = false;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_c64_ring_kernel<T, NCB, FUSE, W2>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, kLds8);
        if (e != hipSuccess) {
            pgt_set_error("igemm8: cannot reserve %d bytes of LDS: %s", kLds8, hipGetErrorString(e));
            return -5;
        }
        configured = true;
    }
    hipLaunchKernelGGL((conv3x3_c64_ring_kernel<T, NCB, FUSE, W2>), dim3(grid), dim3(256), kLds8, st, p, R, nys, nxs, nitems);
    PGT_LAUNCH_CHECK();
    return 0;
}

template <typename T> int launch8_t(const ConvP& p, int R, int nys, int nxs, int nitems, int grid, hipStream_t st) {
    const bool fuse = p.in_scale != nullptr;
    if constexpr (std::is_same<T, half_t>::value) {
        if (p.w2) {      // exact-weight form: 16 output channels per wave
            if (p.Cout <= 16) return fuse ? launch8<T, 1, true, true>(p, R, nys, nxs, nitems, grid, st) : launch8<T, 1, false, true>(p, R, nys, nxs, nitems, grid, st);
            return fuse ? launch8<T, 2, true, true>(p, R, nys, nxs, nitems, grid, st) : launch8<T, 2, false, true>(p, R, nys, nxs, nitems, grid, st);
        }
    }
    if (p.w2) { pgt_set_error("igemm8: the exact-weight form is IEEE half only"); return -22; }
    if (p.Cout <= 32) return fuse ? launch8<T, 1, true>(p, R, nys, nxs, nitems, grid, st) : launch8<T, 1, false>(p, R, nys, nxs, nitems, grid, st);
    return fuse ? launch8<T, 2, true>(p, R, nys, nxs, nitems, grid, st) : launch8<T, 2, false>(p, R, nys, nxs, nitems, grid, st);
}

}  // namespace

// See the preconditions at the top of the file; the caller checks them.
int pgt_igemm8_launch(const void* pv, hipStream_t st) {
    ConvP p = *reinterpret_cast<const ConvP*>(pv);
    static int n_cu = 0;
    if (n_cu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
            pgt_set_error("igemm8: cannot query the device");
            return -5;
        }
        n_cu = prop.multiProcessorCount;
    }
    const int grid_max = 2 * n_cu;
    const int nxs = p.W / 128;
    // rows per strip: as long as possible (2 halo rows are re-read per strip) while every workgroup still gets >= 4 strips
    int R = p.H < 64 ? p.H : 64;
    while (R > 8 && (long)p.N * (p.H / R) * nxs < 4L * grid_max) R >>= 1;
    if (R > p.H) R = p.H;
    // a strip takes ONE bias vector (bias_of at its first pixel): with a bias per band of the frame (pgt_conv_desc::bias_rows) a
    // strip must not straddle bands (the caller - ring_legal - made sure that 4-row strips fit)
    while (p.bias_rows > 0 && R > 4 && p.bias_rows % (R * p.W) != 0) R >>= 1;
    const int nys = p.H / R;
    const int nitems = p.N * nys * nxs;
    const int grid = nitems < grid_max ? nitems : grid_max;
    return p.f16 ? launch8_t<half_t>(p, R, nys, nxs, nitems, grid, st) : launch8_t<bf16_t>(p, R, nys, nxs, nitems, grid, st);
}
