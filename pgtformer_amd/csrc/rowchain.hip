// Token-row chains of the 256-channel window-attention blocks, fused (round 4).
//
// A VSTSREncoderTransformerBlock (reference: modules/rstt_layers.py:284-338, Mlp :126-132, the q / kv / proj Linears of
// WindowAttention3D :195-234) is, per token row of C = 256 channels,
//      LN1 -> [q | k | v] Linear ............. window attention ............. proj Linear + shortcut -> LN2 -> fc1 -> GELU -> fc2 + x1
// Run layer by layer, a row crosses HBM 14 times (10 KB per row in half, 20 KB in split-half) for 0.84 MFLOP: every one of
// those launches is HBM-bound.  The two kernels here keep a row ON CHIP between its element-wise and
// GEMM steps, so that only the attention operands and the block's input / output touch HBM (5 KB per row):
//
//   chain 0  "ln_linear":  y = LN(x) W^T + b                       (x: rows x 256, W: Cout x 256, Cout a multiple of 128)
//   chain 1  "proj_mlp":   x1 = a Wp^T + bp + s;  y = x1 + fc2(GELU(fc1(LN(x1))))      (a = attention output, s = shortcut)
//
// Design - the ROWS live in registers, the WEIGHTS stream through LDS:
//   * swapped GEMM on v_mfma_f32_16x16x32_f16: C^T[out column, row] = W[out column, k] . X^T[k, row].  A wave owns 16 rows
//     (RT = 1) or 32 (RT = 2: two row tiles under the same weight fragments); its B operand - X^T, 8 k-steps of 32 - is the
//     row itself, 16 bytes per lane and k-step: lane (g, n) holds x[row n][32 ks + 8 g .. + 8].
//   * the weights of 32 output columns (one CHUNK: 32 x 512 B = 16 KiB, K-major rows as pgt_pack_conv_weight writes them)
//     enter a ring of NS LDS slots by LDS-DMA (buffer_load ... lds), XOR-swizzled on the source side so that the A-fragment
//     ds_read_b128 is conflict-free; every wave reads every chunk once.  A chunk is two 16-column MFMA tiles whose rows are
//     interleaved so that lane (g, n) ends up with the EIGHT CONSECUTIVE output columns 32 q + 8 g .. + 8 of row n:
//       - that is one 16-byte store of the output row, and
//       - packed to half it is exactly the B fragment of k-step q of the NEXT GEMM: the chain never leaves the registers
//         (x1, LN2(x1) and the GELU'd hidden row are 32 VGPRs each).
//   * LayerNorm: a row is spread over 4 lanes (64 values each): two-pass statistics in registers, two cross-lane adds.
//   * one barrier per chunk (DMA of chunk c + NS - 1 is issued right after the barrier that retires chunk c - 1); counted
//     s_waitcnt vmcnt so that NS - 2 chunks stay in flight across the barrier.
//   * persistent workgroups (8 waves = 128 RT rows per tile), one per CU.
// Numerics are those of the unfused launches: x1, the normalised rows and the hidden row are rounded to half exactly where the
// layer-by-layer path stores them; accumulation and statistics are fp32.  Two differences, both at load time: the LayerNorm's
// affine part is folded into the following Linear (pgt_fold_layernorm: W diag(gamma), b + W beta), so the kernels normalise
// without per-channel operands; GELU is the exact-erf form with erf from Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, three
// orders below the half rounding that follows it).
//   chain 3  the sampled pass of chain 1 (pgt_attn_proj_mlp_sample): the same code on the <= 1024 sampled rows of every frame,
//            up to the hidden row, leaving column sums - the per-frame means the weight-rounding compensation of fc1 / fc2
//            needs for operands that never reach HBM.
//   split-half forms (rowchain_x3_kernel, further down): LN -> Linear and LN -> Mlp -> residual on two half planes.
// What bounds these kernels (tools/rowchain_probe.py): the chip's power limit - switching work off
// shortens them whatever the work is, re-scheduling the same work (pipelined epilogues, prefetched rows, 4 / 8 / 16 waves, two
// workgroups per CU) changes their cycle count and clock in opposite directions and leaves the time where it was.
#include <atomic>
#include <cstdlib>
#include <cstring>

#include "common.h"
#include "pgt_internal.h"
#include "igemm_common.h"
#include "pgt_sample.h"

// Probe hooks (tools/rowchain_probe.py compiles this file with -DRC_PROBE=<bits>; the library build has none):
//   1 no MFMAs   2 no A-fragment reads   4 no weight DMA   8 GELU -> identity   16 no normalisation   32 no output stores
//   64 rows not loaded   128 no chunk barriers / waits (timing only: results are wrong)
//   256 workgroup 0 leaves its shader-clock ticks (s_memtime) and 100 MHz ticks in the first 16 bytes of y: the clock it ran at
#ifndef RC_PROBE
#define RC_PROBE 0
#endif
// build switches (A/B through tools/rowchain_probe.py --define)
#ifndef RC_PIPE
#define RC_PIPE 1          // chain 1: the element-wise step of chunk q runs inside chunk q + 1
#endif
#ifndef RC_PREFETCH
#define RC_PREFETCH 0      // chain 1: next tile's rows requested during the last GEMM (measured: no gain, +48 VGPRs)
#endif
#ifndef RC_PREFETCH0
#define RC_PREFETCH0 0     // chain 0: next tile's rows requested three chunks before the tile ends (measured: no gain)
#endif
#ifndef RC_KG1
#define RC_KG1 2           // chain 1: k-steps per fragment group
#endif
#ifndef RC_KG0
#define RC_KG0 4           // chain 0: k-steps per fragment group
#endif

namespace {

typedef _Float16 rc_half8 __attribute__((ext_vector_type(8)));

#define RC_FENCE() do { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define RC_BARRIER() do { RC_FENCE(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); RC_FENCE(); } while (0)
#define RC_VMWAIT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")

struct RowChainP {
    const char* x;        // chain 0: input rows; chain 1: attention output rows
    const char* res;      // chain 1: shortcut rows (the block's input)
    char* y;
    const char* w;        // chain 0: (ncol, 256); chain 1: [Wproj; Wfc1; Wfc2] = (768, 256); K-major half rows
    const float* b0;      // chain 0: bias (ncol); chain 1: proj bias (256); one vector per b0_rows rows when b0_rows > 0
    const float* b1;      // chain 1: fc1 bias
    const float* b2;      // chain 1: fc2 bias
    int ldx, ldr, ldy;    // row strides in elements
    int M, ncol, b0_rows;
    float eps;
    int HW;               // chain 3 (sampled pass): rows per frame; M = frames * HW
    float* stats;         // chain 3: partial column sums [frame][128-row sample tile][wave][2][256]
};

constexpr int kRcNS = 4;                        // ring slots
constexpr int kRcChunk = 32 * 512;              // bytes of one weight chunk (32 output columns x 256 k, half)
constexpr int kRcMaxCol = 768;
constexpr int kRcLds = kRcNS * kRcChunk + 2 * kRcMaxCol * 4;   // ring + two bias buffers

__device__ __forceinline__ f32x4 rc_mma(const uint4& a, const uint4& b, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(rc_half8, a), __builtin_bit_cast(rc_half8, b), c, 0, 0, 0);
}

// exact-erf GELU with erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7 on erf)
__device__ __forceinline__ float rc_gelu(float x) {
    const float z = x * 0.70710678118654752440f, az = fabsf(z);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, az, 1.0f));
    float p = fmaf(t, 1.061405429f, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    p *= t;
    const float e = __builtin_amdgcn_exp2f(az * az * -1.44269504088896340736f);
    const float er = copysignf(fmaf(-p, e, 1.0f), z);
    const float hx = 0.5f * x;
    return fmaf(hx, er, hx);
}

// LayerNorm statistics of the 16 rows a wave holds as B fragments (lane (g, n): 8 values of row n per k-step), applied WITHOUT
// the affine part: b <- half((b - mean) * rstd).  gamma and beta live in the weights of the GEMM that follows
// (pgt_fold_layernorm: W' = W diag(gamma), b' = b + W beta - LN(x) W^T + b = xhat W'^T + b'), so the normalisation needs no
// per-channel operands.  Two-pass statistics in fp32, as layernorm_kernel (norms.hip).
__device__ __forceinline__ void rc_normalize(uint4 (&b)[8], float eps) {
    float v[64];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) Vec16<half_t>::unpack(b[ks], v + 8 * ks);
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 64; ++e) s += v[e];
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    const float mean = s * (1.0f / 256.0f);
    float ss = 0.f;
#pragma unroll
    for (int e = 0; e < 64; ++e) { v[e] -= mean; ss = fmaf(v[e], v[e], ss); }
    ss += __shfl_xor(ss, 16, 64);
    ss += __shfl_xor(ss, 32, 64);
    const float rstd = 1.0f / sqrtf(ss * (1.0f / 256.0f) + eps);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[8 * ks + e] *= rstd;
        b[ks] = Vec16<half_t>::pack(v + 8 * ks);
    }
}

// NW waves per workgroup (8 or 16), RT row tiles of 16 rows per wave
template <int MODE, int RT, int NW>
__global__ __launch_bounds__(64 * NW) void rowchain_kernel(RowChainP p, int ntiles) {
    static_assert(MODE == 0 || RT == 1, "the three-GEMM chain keeps one row tile per wave");
    static_assert(MODE == 0 || MODE == 1 || MODE == 3, "chain 0: LN -> Linear; 1: proj -> Mlp; 3: chain 1 on the sampled rows, statistics only");
    constexpr bool CHAIN = MODE != 0;
    static_assert(NW == 4 || NW == 8 || NW == 16, "16 DMA pieces per chunk are dealt to 4, 8 or 16 waves");
    constexpr int kRcWaves = NW;
    constexpr int kRcPPC = 16 / NW;                     // DMA pieces (1 KiB) per wave and chunk
    constexpr int TR = kRcWaves * 16 * RT;              // rows per workgroup tile
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    float* const bias_l = reinterpret_cast<float*>(smem + kRcNS * kRcChunk);    // two buffers of kRcMaxCol floats

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, n = lane & 15;
#if RC_PROBE & 256
    const unsigned long long clk_t0 = __builtin_amdgcn_s_memtime(), clk_r0 = __builtin_readcyclecounter();
#endif
    const int NQ = MODE == 0 ? p.ncol / 32 : MODE == 1 ? 24 : 16;        // chunks per tile (sampled pass: proj and fc1 only)
    const int my_tiles = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int total = my_tiles * NQ;                    // chunks this workgroup consumes
    const unsigned lds0 = lds_addr(smem);
    const v4i rsrc_w = make_rsrc(p.w, (unsigned)(NQ * kRcChunk));

    // ---- DMA role: piece j = PPC wave + i holds chunk rows 2 j, 2 j + 1; lane l -> row a = 2 j + (l >> 5), slot l & 31, which
    //      receives source chunk (slot ^ f(a)), f(a) = ((a >> 3) << 2) | (a & 3)
    unsigned dma_off[kRcPPC];
#pragma unroll
    for (int i = 0; i < kRcPPC; ++i) {
        const int a = 2 * (kRcPPC * wave + i) + (lane >> 5);
        const int f = ((a >> 3) << 2) | (a & 3);
        dma_off[i] = (unsigned)(a * 512 + (((lane & 31) ^ f) << 4));
    }
    int iss_q = 0, iss_slot = 0, iss_idx = 0;           // next chunk to issue: index inside the tile, ring slot, running index
    auto issue = [&]() {
#pragma unroll
        for (int i = 0; i < kRcPPC; ++i)
            bufdma16(dma_off[i], rsrc_w, iss_q * kRcChunk, lds0 + iss_slot * kRcChunk + (kRcPPC * wave + i) * 1024);
        iss_q = iss_q + 1 == NQ ? 0 : iss_q + 1;
        iss_slot = (iss_slot + 1) & (kRcNS - 1);
        ++iss_idx;
    };
    for (int i = 0; i < kRcNS - 1 && i < total && !(RC_PROBE & 4); ++i) issue();

    // ---- static operands into LDS (visible after the first chunk barrier).  Chain 1: [b_proj | b_fc1 | b_fc2], all three one
    //      vector per frame when b0_rows > 0; the sampled pass: b_proj per frame when b0_rows > 0, b_fc1 always a vector
    const int nb0 = MODE == 0 ? p.ncol : MODE == 1 ? 768 : 256;
    if (MODE == 3)
        for (int i = tid; i < 256; i += 64 * kRcWaves) bias_l[256 + i] = bias_l[kRcMaxCol + 256 + i] = p.b1[i];
    if (p.b0_rows == 0)
        for (int i = tid; i < nb0; i += 64 * kRcWaves)
            bias_l[i] = !CHAIN ? (p.b0 ? p.b0[i] : 0.f) : i < 256 ? p.b0[i] : i < 512 ? p.b1[i - 256] : p.b2[i - 512];
    int cur_frame = -1, cur_buf = 0;

    // A fragment of (16-column tile t, k-step ks) of a slot: lane (g, m = n) reads chunk row 8 (m >> 2) + 4 t + (m & 3), 16-byte
    // chunk 4 ks + g, stored in slot (4 ks + g) ^ m: byte (abase ^ (ks << 6)) + 2048 t
    const int abase = (8 * (n >> 2) + (n & 3)) * 512 + ((g ^ n) << 4);
    int idx = 0;                                        // running index of the chunk being consumed
    // One chunk: retire its DMA, release the slot of the previous one, keep the ring full, multiply.  `between(part)` is called
    // once per k-group, after that group's fragment reads have been issued and before its MFMAs: the place where the
    // element-wise work of the PREVIOUS chunk goes, so that it overlaps this chunk's LDS latency and matrix pipe time.
    constexpr int KG = MODE == 0 ? RC_KG0 : RC_KG1;     // k-steps per group (2 KG fragments = 8 KG VGPRs in flight)
    constexpr int NPART = 8 / KG;
    // `extra`: this wave requested the next tile's rows (kXL ordinary loads) AFTER the DMA of the chunk waited for here: the
    // counter retires in order, so they may stay in flight (waiting them out here is what made a prefetch useless)
    constexpr int kXL = MODE == 0 ? 8 * RT : 16;
    auto chunk_mma = [&](f32x4 (&acc)[RT][2], const uint4 (&b)[RT][8], auto&& between, bool extra = false) {
        const int after = total - 1 - idx;              // chunks issued after this one (capped by the ring)
        if (!(RC_PROBE & 128)) {
            if (after >= kRcNS - 2) {
                if (extra) RC_VMWAIT((kRcNS - 2) * kRcPPC + kXL);
                else RC_VMWAIT((kRcNS - 2) * kRcPPC);
            } else if (after == 1) RC_VMWAIT(kRcPPC);
            else RC_VMWAIT(0);
            RC_BARRIER();
        }
        if (!(RC_PROBE & 4) && iss_idx < total) issue();
        const char* sp = smem + (idx & (kRcNS - 1)) * kRcChunk;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            acc[rt][0] = f32x4{0.f, 0.f, 0.f, 0.f};
            acc[rt][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int k4 = 0; k4 < 8; k4 += KG) {
            uint4 a0[KG], a1[KG];
#pragma unroll
            for (int j = 0; j < KG; ++j) {
                if (RC_PROBE & 2) { a0[j] = a1[j] = make_uint4(lane, k4 + j, 0x3c003c00u, idx); continue; }
                a0[j] = *reinterpret_cast<const uint4*>(sp + (abase ^ ((k4 + j) << 6)));
                a1[j] = *reinterpret_cast<const uint4*>(sp + 2048 + (abase ^ ((k4 + j) << 6)));
            }
            between(k4 / KG);
#pragma unroll
            for (int j = 0; j < KG; ++j)
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    if (RC_PROBE & 1) {
                        acc[rt][0][j & 3] += __uint_as_float(a0[j].x ^ b[rt][k4 + j].y);
                        acc[rt][1][j & 3] += __uint_as_float(a1[j].z ^ b[rt][k4 + j].w);
                        continue;
                    }
                    acc[rt][0] = rc_mma(a0[j], b[rt][k4 + j], acc[rt][0]);
                    acc[rt][1] = rc_mma(a1[j], b[rt][k4 + j], acc[rt][1]);
                }
            __builtin_amdgcn_sched_barrier(0);
        }
        ++idx;
    };
    auto nothing = [](int) {};
    // the 8 consecutive columns 32 q + 8 g .. + 8 of this lane's row: accumulator (t, r) = column 4 t + r of them
    auto cols8 = [&](const f32x4 (&acc)[2], float* v) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) v[4 * t + r] = acc[t][r];
    };
    auto add_bias = [&](float* v, const float* bp) {
        float bv[8];
        *reinterpret_cast<float4*>(bv) = *reinterpret_cast<const float4*>(bp);
        *reinterpret_cast<float4*>(bv + 4) = *reinterpret_cast<const float4*>(bp + 4);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += bv[e];
    };
    auto load_rows = [&](uint4 (&dst)[8], const char* base, int ld, long row, int tile) {
        const long rr = row < p.M ? row : p.M - 1;
        const uint4* xp = reinterpret_cast<const uint4*>(base + (rr * ld + 8 * g) * 2);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) dst[ks] = (RC_PROBE & 64) ? make_uint4(0x3c003800u + lane, 0x38003c00u, tile, ks) : xp[4 * ks];
    };

    uint4 nb[MODE != 0 ? 8 : 1], nsc[MODE != 0 ? 8 : 1];      // chain 1: the next tile's rows, requested during this tile's last GEMM
    uint4 nb0r[MODE == 0 ? RT : 1][8];                         // chain 0: the same, requested kRcNS - 1 chunks before the tile ends
    bool have_next = false;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long row0 = (long)tile * TR;
        // ---- per-frame biases: refreshed into the other LDS buffer when the tile enters a new frame
        int s_run = 1, s_cells = 1, s_cell = 1, tpf = 1;           // sampled pass: sample geometry, 128-row sample tiles per frame
        if (MODE == 3) {
            mean_sample_geometry(p.HW, &s_run, &s_cells, &s_cell);
            tpf = (s_cells * s_run + TR - 1) / TR;
        }
        if (p.b0_rows > 0) {
            const int frame = MODE == 3 ? tile / tpf : (int)(row0 / p.b0_rows);
            if (frame != cur_frame) {
                cur_frame = frame;
                cur_buf ^= 1;
                for (int i = tid; i < nb0; i += 64 * kRcWaves) {
                    const float* src = !CHAIN ? p.b0 + (long)frame * nb0 : i < 256 ? p.b0 + (long)frame * 256 : i < 512 ? p.b1 + (long)frame * 256 - 256 : p.b2 + (long)frame * 256 - 512;
                    bias_l[cur_buf * kRcMaxCol + i] = src[i];
                }
            }
        }
        const float* bl = bias_l + cur_buf * kRcMaxCol;
        // ---- this wave's rows as B fragments
        long rows[RT];
        uint4 b[RT][8];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) rows[rt] = row0 + (wave * RT + rt) * 16 + n;
        bool valid = true;                              // sampled pass: this lane's row is one of the frame's sample
        if (MODE == 3) {
            const int f = tile / tpf, i = (tile - f * tpf) * TR + wave * 16 + n, S = s_cells * s_run;
            valid = i < S;
            rows[0] = (long)f * p.HW + mean_sample_pixel(valid ? i : S - 1, s_cell, s_run);
        }
        // sampled pass: column sums of this wave's (valid) rows, into stats[tile][wave][which][256]
        auto column_sums = [&](const uint4 (&t8)[8], int which) {
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                float v[8];
                Vec16<half_t>::unpack(t8[ks], v);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float x = valid ? v[e] : 0.f;
#pragma unroll
                    for (int o = 1; o < 16; o <<= 1) x += __shfl_xor(x, o, 64);
                    v[e] = x;
                }
                if (n == 0) {
                    float* dst = p.stats + (((long)tile * kRcWaves + wave) * 2 + which) * 256 + 32 * ks + 8 * g;
                    *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
                    *reinterpret_cast<float4*>(dst + 4) = make_float4(v[4], v[5], v[6], v[7]);
                }
            }
        };
        if constexpr (MODE == 0) {
            if (have_next) {
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                    for (int ks = 0; ks < 8; ++ks) b[rt][ks] = nb0r[rt][ks];
            } else {
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) load_rows(b[rt], p.x, p.ldx, rows[rt], tile);
            }
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) if (!(RC_PROBE & 16)) rc_normalize(b[rt], p.eps);
            have_next = false;
            for (int q = 0; q < NQ; ++q) {
                if (RC_PREFETCH0 && q == NQ - (kRcNS - 1) && tile + (int)gridDim.x < ntiles) {
                    // the next tile's rows: requested kRcNS - 1 chunks before the end of this one
                    have_next = true;
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt)
                        load_rows(nb0r[rt], p.x, p.ldx, (long)(tile + gridDim.x) * TR + (wave * RT + rt) * 16 + n, tile);
                }
                f32x4 acc[RT][2];
                chunk_mma(acc, b, nothing, have_next);
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    float v[8];
                    cols8(acc[rt], v);
                    add_bias(v, bl + 32 * q + 8 * g);
                    if (rows[rt] < p.M && (!(RC_PROBE & 32) || v[0] == 123.456f))
                        *reinterpret_cast<uint4*>(p.y + (rows[rt] * p.ldy + 32 * q + 8 * g) * 2) = Vec16<half_t>::pack(v);
                }
            }
        } else {
            // shortcut rows, in the layout of the GEMM output (8 consecutive columns per chunk)
            uint4 sc[8];
            if (have_next) {
#pragma unroll
                for (int q = 0; q < 8; ++q) { b[0][q] = nb[q]; sc[q] = nsc[q]; }
            } else {
                load_rows(b[0], p.x, p.ldx, rows[0], tile);
                load_rows(sc, p.res, p.ldr, rows[0], tile);
            }
            uint4 x1[1][8], hid[1][8];
            f32x4 accs[2][1][2];                        // this chunk's accumulators and the previous chunk's (epilogue pending)
            // The element-wise step of chunk q runs inside chunk q + 1 (RC_PIPE), one quarter per k-group: v holds the chunk's 8
            // columns across the groups.
            float v[8];
            // ---- proj + bias + shortcut -> x1 (rounded to half: the tensor the layer-by-layer path stores)
            auto epi0 = [&](int q, const f32x4 (&acc)[2], int part) {
                if (part == 0) { cols8(acc, v); add_bias(v, bl + 32 * q + 8 * g); }
                if (part == NPART - 1) {
                    float s[8];
                    Vec16<half_t>::unpack(sc[q], s);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += s[e];
                    x1[0][q] = Vec16<half_t>::pack(v);
                    x3_opaque(x1[0][q]);                // finished HERE: keeps the compiler from sinking the epilogues below the loop
                }
            };
            // ---- fc1 + bias + GELU -> hidden row (half)
            auto epi1 = [&](int q, const f32x4 (&acc)[2], int part) {
                if (part == 0) { cols8(acc, v); add_bias(v, bl + 256 + 32 * q + 8 * g); }
                constexpr int per = 8 / NPART;
#pragma unroll
                for (int e = part * per; e < (part + 1) * per; ++e) v[e] = (RC_PROBE & 8) ? v[e] : rc_gelu(v[e]);
                if (part == NPART - 1) {
                    hid[0][q] = Vec16<half_t>::pack(v);
                    x3_opaque(hid[0][q]);
                }
            };
            // ---- fc2 + bias + x1 -> out
            auto epi2 = [&](int q, const f32x4 (&acc)[2], int part) {
                if (part == 0) { cols8(acc, v); add_bias(v, bl + 512 + 32 * q + 8 * g); }
                if (part == NPART - 1) {
                    float s[8];
                    Vec16<half_t>::unpack(x1[0][q], s);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += s[e];
                    if (rows[0] < p.M && (!(RC_PROBE & 32) || v[0] == 123.456f))
                        *reinterpret_cast<uint4*>(p.y + (rows[0] * p.ldy + 32 * q + 8 * g) * 2) = Vec16<half_t>::pack(v);
                }
            };
            auto gemm = [&](const uint4 (&bop)[1][8], auto&& epi, bool pre = false) {
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const bool extra = pre && q < kRcNS - 1;
                    if (RC_PIPE) {
                        chunk_mma(accs[q & 1], bop, [&](int part) { if (q > 0) epi(q - 1, accs[(q - 1) & 1][0], part); }, extra);
                    } else {
                        chunk_mma(accs[0], bop, nothing, extra);
#pragma unroll
                        for (int part = 0; part < NPART; ++part) epi(q, accs[0][0], part);
                    }
                }
                if (RC_PIPE) {
#pragma unroll
                    for (int part = 0; part < NPART; ++part) epi(7, accs[1][0], part);
                }
            };
            gemm(b, epi0);
            // ---- LN2
#pragma unroll
            for (int q = 0; q < 8; ++q) b[0][q] = x1[0][q];
            if (!(RC_PROBE & 16)) rc_normalize(b[0], p.eps);
            if (MODE == 3) column_sums(b[0], 0);
            gemm(b, epi1);
            if (MODE == 3) {
                column_sums(hid[0], 1);
                continue;
            }
            // ---- the next tile's rows are requested here: they have the whole last GEMM to arrive
            have_next = RC_PREFETCH && MODE == 1 && tile + (int)gridDim.x < ntiles;
            if (have_next) {
                const long nrow = (long)(tile + gridDim.x) * TR + wave * 16 + n;
                load_rows(nb, p.x, p.ldx, nrow, tile);
                load_rows(nsc, p.res, p.ldr, nrow, tile);
            }
            gemm(hid, epi2, have_next);
        }
    }
    RC_VMWAIT(0);
#if RC_PROBE & 256
    if (blockIdx.x == 0 && tid == 0) {
        reinterpret_cast<unsigned long long*>(p.y)[0] = __builtin_amdgcn_s_memtime() - clk_t0;
        reinterpret_cast<unsigned long long*>(p.y)[1] = __builtin_readcyclecounter() - clk_r0;
    }
#endif
}

// ------------------------------------------------------------------------------------------------------------------------------
// Split-half (PGT_F16X3) forms for the encoder-side blocks: rows and weights on two IEEE-half planes, every product taken as
// x_lo w_hi + x_hi w_lo + x_hi w_hi (small terms first, as every split kernel of the library), fp32 accumulation / statistics.
//   MODE 0  y = xhat W'^T + b'                                   (LayerNorm -> Linear, Cout a multiple of 128 up to 768)
//   MODE 2  y = x + fc2(GELU(fc1(xhat)))                         (LayerNorm -> Mlp -> residual; x is re-read for the residual)
// A chunk is 32 output columns x [w_hi (512 B) | w_lo (512 B)] = 32 KiB, taken from the [w_hi | w_hi | w_lo]-per-64-channel
// rows of pgt_pack_conv_weight(PGT_F16X3); the ring holds four of them.  The three-GEMM chain of the half form would need
// three split rows per wave (192 VGPRs): proj + shortcut stays on the phased kernel for split blocks.
constexpr int kRcChunkX = 2 * kRcChunk;
constexpr int kRcLdsX = kRcNS * kRcChunkX + kRcMaxCol * 4;

template <int MODE, int RT>
__global__ __launch_bounds__(512) void rowchain_x3_kernel(RowChainP p, int ntiles, int xlo, int ylo) {
    static_assert(MODE == 0 || (MODE == 2 && RT == 1), "modes");
    constexpr int NWV = 8, PPC = 4;                     // 32 DMA pieces per chunk: waves 0-3 the hi plane, 4-7 the lo plane
    constexpr int TR = NWV * 16 * RT;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    float* const bias_l = reinterpret_cast<float*>(smem + kRcNS * kRcChunkX);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, n = lane & 15;
    const int NQ = MODE == 0 ? p.ncol / 32 : 16;
    const int my_tiles = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int total = my_tiles * NQ;
    const unsigned lds0 = lds_addr(smem);
    const v4i rsrc_w = make_rsrc(p.w, (unsigned)(NQ * 32 * 1536));
    // piece j = 4 wave + i: plane j >> 4, rows 2 (j & 15), + 1 of the chunk; lane -> row a, slot l & 31 <- source chunk ck
    unsigned dma_off[PPC];
#pragma unroll
    for (int i = 0; i < PPC; ++i) {
        const int j = PPC * wave + i, plane = j >> 4;
        const int a = 2 * (j & 15) + (lane >> 5);
        const int f = ((a >> 3) << 2) | (a & 3);
        const int ck = (lane & 31) ^ f;                                  // 16-byte chunk of the 256 k: 64-channel block ck >> 3
        dma_off[i] = (unsigned)(a * 1536 + (ck >> 3) * 384 + plane * 256 + (ck & 7) * 16);
    }
    int iss_q = 0, iss_slot = 0, iss_idx = 0;
    auto issue = [&]() {
#pragma unroll
        for (int i = 0; i < PPC; ++i) {
            const int j = PPC * wave + i;
            bufdma16(dma_off[i], rsrc_w, iss_q * (32 * 1536), lds0 + iss_slot * kRcChunkX + (j >> 4) * kRcChunk + (j & 15) * 1024);
        }
        iss_q = iss_q + 1 == NQ ? 0 : iss_q + 1;
        iss_slot = (iss_slot + 1) & (kRcNS - 1);
        ++iss_idx;
    };
    for (int i = 0; i < kRcNS - 1 && i < total; ++i) issue();
    for (int i = tid; i < NQ * 32; i += 512)
        bias_l[i] = MODE == 0 ? (p.b0 ? p.b0[i] : 0.f) : (i < 256 ? p.b0[i] : p.b1[i - 256]);

    const int abase = (8 * (n >> 2) + (n & 3)) * 512 + ((g ^ n) << 4);
    int idx = 0;
    auto chunk_mma = [&](f32x4 (&acc)[RT][2], const uint4 (&bh)[RT][8], const uint4 (&bl)[RT][8]) {
        const int after = total - 1 - idx;
        if (after >= kRcNS - 2) RC_VMWAIT((kRcNS - 2) * PPC);
        else if (after == 1) RC_VMWAIT(PPC);
        else RC_VMWAIT(0);
        RC_BARRIER();
        if (iss_idx < total) issue();
        const char* sp = smem + (idx & (kRcNS - 1)) * kRcChunkX;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            acc[rt][0] = f32x4{0.f, 0.f, 0.f, 0.f};
            acc[rt][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int k2 = 0; k2 < 8; k2 += 2) {             // two k-steps = 8 fragments (hi / lo planes of two column tiles) in flight
            uint4 ah[2][2], al[2][2];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    ah[j][t] = *reinterpret_cast<const uint4*>(sp + 2048 * t + (abase ^ ((k2 + j) << 6)));
                    al[j][t] = *reinterpret_cast<const uint4*>(sp + kRcChunk + 2048 * t + (abase ^ ((k2 + j) << 6)));
                }
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        acc[rt][t] = rc_mma(ah[j][t], bl[rt][k2 + j], acc[rt][t]);     // small terms first
                        acc[rt][t] = rc_mma(al[j][t], bh[rt][k2 + j], acc[rt][t]);
                        acc[rt][t] = rc_mma(ah[j][t], bh[rt][k2 + j], acc[rt][t]);
                    }
            __builtin_amdgcn_sched_barrier(0);
        }
        ++idx;
    };
    auto cols8 = [&](const f32x4 (&acc)[2], const float* bp, float* v) {
        float bv[8];
        *reinterpret_cast<float4*>(bv) = *reinterpret_cast<const float4*>(bp);
        *reinterpret_cast<float4*>(bv + 4) = *reinterpret_cast<const float4*>(bp + 4);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) v[4 * t + r] = acc[t][r] + bv[4 * t + r];
    };
    // rows (hi + lo) -> normalised rows as split B fragments; statistics as layernorm_kernel<.., X3> (norms.hip)
    auto load_norm = [&](uint4 (&bh)[8], uint4 (&bl)[8], long row) {
        const long rr = row < p.M ? row : p.M - 1;
        const uint4* xh = reinterpret_cast<const uint4*>(p.x + (rr * p.ldx + 8 * g) * 2);
        const uint4* xl = reinterpret_cast<const uint4*>(p.x + (rr * p.ldx + xlo + 8 * g) * 2);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) { bh[ks] = xh[4 * ks]; bl[ks] = xl[4 * ks]; }
        float v[64];
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) merge8(bh[ks], bl[ks], v + 8 * ks);
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < 64; ++e) s += v[e];
        s += __shfl_xor(s, 16, 64);
        s += __shfl_xor(s, 32, 64);
        const float mean = s * (1.0f / 256.0f);
        float ss = 0.f;
#pragma unroll
        for (int e = 0; e < 64; ++e) { v[e] -= mean; ss = fmaf(v[e], v[e], ss); }
        ss += __shfl_xor(ss, 16, 64);
        ss += __shfl_xor(ss, 32, 64);
        const float rstd = 1.0f / sqrtf(ss * (1.0f / 256.0f) + p.eps);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[8 * ks + e] *= rstd;
            split8(v + 8 * ks, bh[ks], bl[ks]);
        }
    };

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long row0 = (long)tile * TR;
        long rows[RT];
        uint4 bh[RT][8], bl[RT][8];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            rows[rt] = row0 + (wave * RT + rt) * 16 + n;
            load_norm(bh[rt], bl[rt], rows[rt]);
        }
        if constexpr (MODE == 0) {
            for (int q = 0; q < NQ; ++q) {
                f32x4 acc[RT][2];
                chunk_mma(acc, bh, bl);
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    float v[8];
                    cols8(acc[rt], bias_l + 32 * q + 8 * g, v);
                    uint4 oh, ol;
                    split8(v, oh, ol);
                    if (rows[rt] < p.M) {
                        char* yp = p.y + (rows[rt] * p.ldy + 32 * q + 8 * g) * 2;
                        *reinterpret_cast<uint4*>(yp) = oh;
                        *reinterpret_cast<uint4*>(yp + (long)ylo * 2) = ol;
                    }
                }
            }
        } else {
            uint4 hh[1][8], hl[1][8];
            // ---- fc1 + bias + exact GELU -> hidden row (split)
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                f32x4 acc[1][2];
                chunk_mma(acc, bh, bl);
                float v[8];
                cols8(acc[0], bias_l + 32 * q + 8 * g, v);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = 0.5f * v[e] * (1.f + erff(v[e] * 0.70710678118654752440f));
                split8(v, hh[0][q], hl[0][q]);
                x3_opaque(hh[0][q]);
                x3_opaque(hl[0][q]);
            }
            // ---- the residual rows (the kernel's own input), requested before the last GEMM
            uint4 rh[8], rl[8];
            {
                const long rr = rows[0] < p.M ? rows[0] : p.M - 1;
                const uint4* xh = reinterpret_cast<const uint4*>(p.x + (rr * p.ldx + 8 * g) * 2);
                const uint4* xl = reinterpret_cast<const uint4*>(p.x + (rr * p.ldx + xlo + 8 * g) * 2);
#pragma unroll
                for (int q = 0; q < 8; ++q) { rh[q] = xh[4 * q]; rl[q] = xl[4 * q]; }
            }
            // ---- fc2 + bias + x -> out
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                f32x4 acc[1][2];
                chunk_mma(acc, hh, hl);
                float v[8], r[8];
                cols8(acc[0], bias_l + 256 + 32 * q + 8 * g, v);
                merge8(rh[q], rl[q], r);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += r[e];
                uint4 oh, ol;
                split8(v, oh, ol);
                if (rows[0] < p.M) {
                    char* yp = p.y + (rows[0] * p.ldy + 32 * q + 8 * g) * 2;
                    *reinterpret_cast<uint4*>(yp) = oh;
                    *reinterpret_cast<uint4*>(yp + (long)ylo * 2) = ol;
                }
            }
        }
    }
    RC_VMWAIT(0);
}

int rc_cus();

template <int MODE, int RT, int NW> int rc_launch(const RowChainP& p, hipStream_t st, int ntiles_override = 0) {
    constexpr int TR = NW * 16 * RT;
    const int ntiles = ntiles_override ? ntiles_override : (p.M + TR - 1) / TR;
    static std::atomic<unsigned long long> attr_set{0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!((attr_set.load(std::memory_order_acquire) >> (dev & 63)) & 1ull)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&rowchain_kernel<MODE, RT, NW>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, kRcLds);
        if (e != hipSuccess) { pgt_set_error("rowchain: cannot reserve %d B of LDS: %s", kRcLds, hipGetErrorString(e)); return -12; }
        attr_set.fetch_or(1ull << (dev & 63), std::memory_order_release);
    }
    const int cus = rc_cus();
    if (cus <= 0) { pgt_set_error("rowchain: cannot query the device"); return -5; }
    const int wgs = cus * (NW == 4 ? 2 : 1);           // 4-wave workgroups: two per CU (LDS 2 x 70 KiB, 8 waves), out of step with each other
    const int grid = ntiles < wgs ? ntiles : wgs;
    hipLaunchKernelGGL((rowchain_kernel<MODE, RT, NW>), dim3(grid), dim3(64 * NW), kRcLds, st, p, ntiles);
    PGT_LAUNCH_CHECK();
    return 0;
}

int rc_common_checks(const char* who, const RowChainP& p, int tr) {
    PGT_CHECK(p.x && p.y && p.w && p.M >= 1, "%s: null argument", who);
    PGT_CHECK(p.ldx % 8 == 0 && p.ldy % 8 == 0 && ((uintptr_t)p.x & 15) == 0 && ((uintptr_t)p.y & 15) == 0 && ((uintptr_t)p.w & 15) == 0,
              "%s: rows must be 16-byte aligned (ldx=%d ldy=%d)", who, p.ldx, p.ldy);
    PGT_CHECK(p.b0_rows == 0 || (p.b0 && p.b0_rows % tr == 0 && p.M % p.b0_rows == 0),
              "%s: bias_rows=%d must be a multiple of the %d-row tile and divide rows=%d", who, p.b0_rows, tr, p.M);
    return 0;
}

int rc_cus() {
    static std::atomic<int> n_cu{0};
    if (n_cu.load(std::memory_order_relaxed) == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
        n_cu.store(prop.multiProcessorCount, std::memory_order_relaxed);
    }
    return n_cu.load(std::memory_order_relaxed);
}

// w_out[o][k] = w[o][k] * gamma[k];  bias_out[o] = bias[o] + sum_k w[o][k] * beta[k]   (one wavefront per output row, fixed order)
__global__ __launch_bounds__(256) void fold_layernorm_kernel(const float* __restrict__ w, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, const float* __restrict__ bias,
                                                             int Cout, int Cin, float* __restrict__ w_out, float* __restrict__ bias_out) {
    const int o = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (o >= Cout) return;
    float s = 0.f;
    for (int k = lane; k < Cin; k += 64) {
        const float v = w[(long)o * Cin + k];
        w_out[(long)o * Cin + k] = v * gamma[k];
        s = fmaf(v, beta[k], s);
    }
    s = wave_sum(s);
    if (lane == 0) bias_out[o] = (bias ? bias[o] : 0.f) + s;
}

}  // namespace

// LayerNorm's affine part folded into the Linear that follows it: LN(x) W^T + b = xhat (W diag(gamma))^T + (b + W beta), xhat the
// normalised row.  w: fp32 (Cout, Cin) as the reference stores it; w_out (may alias w) is then packed with pgt_pack_conv_weight.
extern "C" int pgt_fold_layernorm(const float* w, const float* gamma, const float* beta, const float* bias, int32_t Cout,
                                  int32_t Cin, float* w_out, float* bias_out, pgt_stream_t stream) {
    PGT_CHECK(w && gamma && beta && w_out && bias_out && Cout >= 1 && Cin >= 1, "fold_layernorm: null argument");
    hipLaunchKernelGGL(fold_layernorm_kernel, dim3((Cout + 3) / 4), dim3(256), 0, (hipStream_t)stream, w, gamma, beta, bias, Cout, Cin, w_out, bias_out);
    PGT_LAUNCH_CHECK();
    return 0;
}

// y = xhat W'^T + bias' on rows of 256 channels (PGT_F16), xhat = (x - mean) / sqrt(var + eps) rounded to half: norm1 + the fused
// [q | k | v] projection of a window-attention block (modules/rstt_layers.py:298, 195-213) - or any LayerNorm -> Linear pair with
// Cin = 256 - with the LayerNorm's gamma / beta folded into w / bias by pgt_fold_layernorm.
extern "C" int pgt_ln_linear(int32_t dtype, const void* x, int32_t ldx, int32_t rows, int32_t Cin, float eps, const void* w,
                             const float* bias, int32_t bias_rows, int32_t Cout, void* y, int32_t ldy, pgt_stream_t stream) {
    PGT_CHECK(dtype == PGT_F16, "ln_linear: dtype %d (PGT_F16 only)", dtype);
    PGT_CHECK(Cin == 256 && Cout >= 128 && Cout % 128 == 0 && Cout <= kRcMaxCol, "ln_linear: Cin=%d (256), Cout=%d (multiple of 128, <= %d)", Cin, Cout, kRcMaxCol);
    RowChainP p{};
    p.x = (const char*)x; p.y = (char*)y; p.w = (const char*)w;
    p.b0 = bias; p.b0_rows = bias_rows; p.eps = eps;
    p.ldx = ldx; p.ldy = ldy; p.M = rows; p.ncol = Cout;
    PGT_CHECK(ldx >= Cin && ldy >= Cout, "ln_linear: ldx=%d ldy=%d", ldx, ldy);
    const int cus = rc_cus();
    PGT_CHECK(cus > 0, "ln_linear: cannot query the device");
    // two row tiles per wave (256-row workgroup tiles: half the weight traffic and half the LDS reads per row) once they still
    // fill the chip.  PGT_RC_LN = r1w8 | r2w8 | r1w16 pins the variant (tuning).
    const char* e = getenv("PGT_RC_LN");
    const int forced = !e ? 0 : !strcmp(e, "r1w8") ? 1 : !strcmp(e, "r2w8") ? 2 : !strcmp(e, "r1w16") ? 3 : !strcmp(e, "r2w4") ? 4 : !strcmp(e, "r1w4") ? 5 : 0;
    int var = forced ? forced : (rows >= 256 * cus ? 2 : 1);
    if ((var == 2 || var == 3) && bias_rows % 256 != 0) var = 1;
    if (int rc = rc_common_checks("ln_linear", p, var == 5 ? 64 : (var == 1 || var == 4) ? 128 : 256)) return rc;
    hipStream_t st = (hipStream_t)stream;
    switch (var) {
        case 1: return rc_launch<0, 1, 8>(p, st);
        case 2: return rc_launch<0, 2, 8>(p, st);
        case 3: return rc_launch<0, 1, 16>(p, st);
        case 4: return rc_launch<0, 2, 4>(p, st);
        default: return rc_launch<0, 1, 4>(p, st);
    }
}

namespace {
template <int MODE, int RT> int rc_launch_x3(const RowChainP& p, int xlo, int ylo, hipStream_t st) {
    constexpr int TR = 128 * RT;
    const int ntiles = (p.M + TR - 1) / TR;
    static std::atomic<unsigned long long> attr_set{0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!((attr_set.load(std::memory_order_acquire) >> (dev & 63)) & 1ull)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&rowchain_x3_kernel<MODE, RT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, kRcLdsX);
        if (e != hipSuccess) { pgt_set_error("rowchain: cannot reserve %d B of LDS: %s", kRcLdsX, hipGetErrorString(e)); return -12; }
        attr_set.fetch_or(1ull << (dev & 63), std::memory_order_release);
    }
    const int cus = rc_cus();
    if (cus <= 0) { pgt_set_error("rowchain: cannot query the device"); return -5; }
    hipLaunchKernelGGL((rowchain_x3_kernel<MODE, RT>), dim3(ntiles < cus ? ntiles : cus), dim3(512), kRcLdsX, st, p, ntiles, xlo, ylo);
    PGT_LAUNCH_CHECK();
    return 0;
}
}  // namespace

// Split-half forms (PGT_F16X3; the encoder-side blocks).  x / y: split rows, lo planes x_lo / y_lo elements after the hi planes;
// w: pgt_pack_conv_weight(PGT_F16X3) of the folded matrix ((Cout, 3 * 256) halves: [w_hi | w_hi | w_lo] per 64-channel block).
extern "C" int pgt_ln_linear_x3(const void* x, int32_t ldx, int32_t x_lo, int32_t rows, int32_t Cin, float eps, const void* w,
                                const float* bias, int32_t Cout, void* y, int32_t ldy, int32_t y_lo, pgt_stream_t stream) {
    PGT_CHECK(Cin == 256 && Cout >= 128 && Cout % 128 == 0 && Cout <= kRcMaxCol, "ln_linear_x3: Cin=%d (256), Cout=%d (multiple of 128, <= %d)", Cin, Cout, kRcMaxCol);
    PGT_CHECK(x_lo % 8 == 0 && y_lo % 8 == 0 && x_lo >= Cin && y_lo >= Cout && ldx >= x_lo + Cin && ldy >= y_lo + Cout, "ln_linear_x3: lo planes (x_lo=%d y_lo=%d)", x_lo, y_lo);
    RowChainP p{};
    p.x = (const char*)x; p.y = (char*)y; p.w = (const char*)w;
    p.b0 = bias; p.eps = eps; p.ldx = ldx; p.ldy = ldy; p.M = rows; p.ncol = Cout;
    if (int rc = rc_common_checks("ln_linear_x3", p, 128)) return rc;
    const int cus = rc_cus();
    const char* e = getenv("PGT_RC_LNX3");              // r1 | r2 pins the variant (tuning)
    const bool r2 = e ? !strcmp(e, "r2") : rows >= 256 * cus;
    return r2 ? rc_launch_x3<0, 2>(p, x_lo, y_lo, (hipStream_t)stream) : rc_launch_x3<0, 1>(p, x_lo, y_lo, (hipStream_t)stream);
}

// y = x + fc2(GELU(fc1(LN(x)))) on split rows: norm2 + Mlp + residual of a window-attention block (modules/rstt_layers.py:335-337,
// 126-132) in one launch.  w2 = [Wfc1 diag(gamma); Wfc2] stacked (512 rows) in the PGT_F16X3 packed form, b_fc1 carrying W1 beta.
extern "C" int pgt_ln_mlp_x3(const void* x, int32_t ldx, int32_t x_lo, int32_t rows, int32_t C, float eps, const void* w2,
                             const float* b_fc1, const float* b_fc2, void* y, int32_t ldy, int32_t y_lo, pgt_stream_t stream) {
    PGT_CHECK(C == 256 && b_fc1 && b_fc2, "ln_mlp_x3: C=%d (256), biases required", C);
    PGT_CHECK(x_lo % 8 == 0 && y_lo % 8 == 0 && x_lo >= C && y_lo >= C && ldx >= x_lo + C && ldy >= y_lo + C, "ln_mlp_x3: lo planes (x_lo=%d y_lo=%d)", x_lo, y_lo);
    PGT_CHECK(x != y, "ln_mlp_x3: y must not alias x (x is re-read for the residual)");
    RowChainP p{};
    p.x = (const char*)x; p.y = (char*)y; p.w = (const char*)w2;
    p.b0 = b_fc1; p.b1 = b_fc2; p.eps = eps; p.ldx = ldx; p.ldy = ldy; p.M = rows; p.ncol = 512;
    if (int rc = rc_common_checks("ln_mlp_x3", p, 128)) return rc;
    return rc_launch_x3<2, 1>(p, x_lo, y_lo, (hipStream_t)stream);
}

namespace {
// mean[which][f][c] = (sum over the frame's sample tiles and their 8 waves, in that order) / S   (the sampled pass's partial sums)
__global__ __launch_bounds__(512) void sample_partial_mean_kernel(const float* __restrict__ stats, int tpf, int frames, int S,
                                                                  float* __restrict__ mean_ln, float* __restrict__ mean_hid) {
    const int f = blockIdx.x, which = threadIdx.x >> 8, c = threadIdx.x & 255;
    float t = 0.f;
    for (int k = 0; k < tpf * 8; ++k) t += stats[(((long)f * tpf * 8 + k) * 2 + which) * 256 + c];
    (which ? mean_hid : mean_ln)[(long)f * 256 + c] = t / (float)S;
}
}  // namespace

// The sampled pass of pgt_attn_proj_mlp, for the weight-rounding compensation of fc1 / fc2: the inputs of
// those two layers never reach HBM in the fused launch, so their per-frame channel means are taken here, by running the chain
// up to the hidden row on the library's pixel sample of every frame (pgt_sampled_pixel: <= 1024 rows of each HW-row frame).
//   mean_ln [f][c]  = mean over the sample of half(xhat)        (the operand of fc1: the normalised x1, x1 = proj + shortcut)
//   mean_hid[f][c]  = mean over the sample of half(GELU(fc1))   (the operand of fc2; fc1 with the PLAIN bias b_fc1)
// b_proj: (256) or, with b_proj_per_frame, (frames, 256).  workspace: pgt_attn_proj_mlp_sample_workspace_bytes(frames, HW).
extern "C" size_t pgt_attn_proj_mlp_sample_workspace_bytes(int32_t frames, int32_t HW) {
    int run, cells, cell;
    if (frames < 1 || HW < 1) return 0;
    mean_sample_geometry(HW, &run, &cells, &cell);
    return (size_t)frames * ((cells * run + 127) / 128) * 8 * 2 * 256 * sizeof(float);
}

extern "C" int pgt_attn_proj_mlp_sample(int32_t dtype, const void* attn, int32_t lda, const void* shortcut, int32_t lds, int32_t frames,
                                        int32_t HW, int32_t C, const void* w3, const float* b_proj, int32_t b_proj_per_frame,
                                        const float* b_fc1, float eps, void* workspace, float* mean_ln, float* mean_hid,
                                        pgt_stream_t stream) {
    PGT_CHECK(dtype == PGT_F16 && C == 256, "attn_proj_mlp_sample: PGT_F16, C = 256 (dtype %d, C %d)", dtype, C);
    PGT_CHECK(attn && shortcut && w3 && b_proj && b_fc1 && workspace && mean_ln && mean_hid && frames >= 1 && HW >= 1, "attn_proj_mlp_sample: null argument");
    PGT_CHECK(lda % 8 == 0 && lds % 8 == 0 && lda >= C && lds >= C && (((uintptr_t)attn | (uintptr_t)shortcut | (uintptr_t)w3 | (uintptr_t)workspace) & 15) == 0,
              "attn_proj_mlp_sample: misaligned argument");
    int run, cells, cell;
    mean_sample_geometry(HW, &run, &cells, &cell);
    const int S = cells * run, tpf = (S + 127) / 128;
    RowChainP p{};
    p.x = (const char*)attn; p.res = (const char*)shortcut; p.w = (const char*)w3;
    p.b0 = b_proj; p.b0_rows = b_proj_per_frame ? 1 : 0; p.b1 = b_fc1; p.eps = eps;
    p.ldx = lda; p.ldr = lds; p.M = (int)((long)frames * HW); p.ncol = 512; p.HW = HW; p.stats = (float*)workspace;
    PGT_CHECK((long)frames * HW < (1L << 31), "attn_proj_mlp_sample: too many rows");
    hipStream_t st = (hipStream_t)stream;
    if (int rc = rc_launch<3, 1, 8>(p, st, frames * tpf)) return rc;
    hipLaunchKernelGGL(sample_partial_mean_kernel, dim3(frames), dim3(512), 0, st, (const float*)workspace, tpf, frames, S, mean_ln, mean_hid);
    PGT_LAUNCH_CHECK();
    return 0;
}

// The tail of a window-attention block in one launch: x1 = attn Wproj^T + b_proj + shortcut; y = x1 + fc2(GELU(fc1(LN2(x1))))
// (modules/rstt_layers.py:230-232 proj, :329 shortcut add, :335-337 norm2 + Mlp + residual; Mlp :126-132 with mlp_ratio = 1,
// archs/tdcrqvae3_arch.py:499).  w3 = the three (256, 256) weights stacked [Wproj; Wfc1'; Wfc2], K-major half rows, Wfc1' / b_fc1
// carrying norm2's gamma / beta (pgt_fold_layernorm).
extern "C" int pgt_attn_proj_mlp(int32_t dtype, const void* attn, int32_t lda, const void* shortcut, int32_t lds, int32_t rows,
                                 int32_t C, const void* w3, const float* b_proj, const float* b_fc1, const float* b_fc2,
                                 int32_t bias_rows, float eps, void* y, int32_t ldy, pgt_stream_t stream) {
    PGT_CHECK(dtype == PGT_F16, "attn_proj_mlp: dtype %d (PGT_F16 only)", dtype);
    PGT_CHECK(C == 256, "attn_proj_mlp: C=%d (256)", C);
    PGT_CHECK(shortcut && b_fc1 && b_fc2 && lds % 8 == 0 && ((uintptr_t)shortcut & 15) == 0 && lds >= C && lda >= C && ldy >= C,
              "attn_proj_mlp: null / misaligned argument");
    RowChainP p{};
    p.x = (const char*)attn; p.res = (const char*)shortcut; p.y = (char*)y; p.w = (const char*)w3;
    p.b0 = b_proj; p.b0_rows = bias_rows; p.b1 = b_fc1; p.b2 = b_fc2; p.eps = eps;
    p.ldx = lda; p.ldr = lds; p.ldy = ldy; p.M = rows; p.ncol = 768;
    const char* e = getenv("PGT_RC_MLP");              // w8 | w4 pins the variant (tuning)
    const bool w4 = e && !strcmp(e, "w4");
    if (int rc = rc_common_checks("attn_proj_mlp", p, w4 ? 64 : 128)) return rc;
    return w4 ? rc_launch<1, 1, 4>(p, (hipStream_t)stream) : rc_launch<1, 1, 8>(p, (hipStream_t)stream);
}
