// Epilogue shared by the 8-wave kernels whose waves own 128 x 64 outputs as 4 x 2 accumulators of 32x32 (igemm4.hip,
// igemm5.hip): output rows are the consecutive pixels m0 .. m0 + WR*128 - 1, columns n0 .. n0 + WC*64 - 1.
#pragma once
#include "igemm_common.h"

#ifndef PGT_PROBE
#define PGT_PROBE 0
#endif
// probe bits of tools/igemm4_probe.hip inside the epilogue: 64 no global stores, 128 no staging writes, 256 no staging reads

namespace {

template <int WR, int WC> constexpr int epi_stage_bytes() { return WR * 64 * (WC * 64 + 4) * 4; }

// smem: at least epi_stage_bytes<WR, WC>() bytes, no DMA in flight, all waves past their last fragment read.
// X3: split-half output (hi at n, lo at n + p.ylo), split residual (p.rlo) or split SFT operands (p.dlo / p.slo).
// GN: also reduce the GroupNorm statistics of the tile's outputs (ConvP::gn_*).
// T: 16-bit storage type of residual / SFT operands / output (bf16_t, or half_t for PGT_F16 launches; X3 is bf16).
// NJ: 32-column accumulators per wave that hold outputs (2; 1 for the exact-weight form, whose waves own 128 x 32 outputs after
// acc[.][0] += acc[.][1] / 2048: the tile then has WC * 32 columns, n0 = its first OUTPUT channel).
template <int WR, int WC, bool X3 = false, bool GN = false, typename T = bf16_t, int NJ = 2>
__device__ __forceinline__ void epilogue_128x64(const ConvP& p, const f32x16 (&acc)[4][2], char* smem, int m0, int n0,
                                                int tid, int lane, int wr, int wc) {
    constexpr int BN = WC * 32 * NJ;
    static_assert(NJ == 2 || (NJ == 1 && !X3), "one or two accumulator columns per wave");
    const int hh = lane >> 5;
    // ---- epilogue in two passes (ih = 0, 1): every wave stages its 64 x 64 half (acc + bias, fp32) in LDS, then each
    //      thread finishes CPT chunks of 8 channels of one pixel: activation, residual / SFT, 16-byte store.  The
    //      residual (dec, shift) chunks of a pass are requested before its staging writes so that their latency
    //      overlaps the LDS round trip.
    constexpr int SROW = BN + 4, SROWS = WR * 64, CPT = SROWS * (BN / 8) / 512;
    static_assert(SROWS * (BN / 8) % 512 == 0, "chunks per thread");
    float* stage = reinterpret_cast<float*>(smem);
    static_assert(!X3 || sizeof(T) == 2, "16-bit storage");
    // half launches: the SFT operands are not prefetched - the second 32-register prefetch array pushed these
    // instantiations over 256 VGPRs (scratch reloads inside the store loop: the K = 256 linears ran 25 % behind bf16)
    constexpr bool kLateSft = !X3 && sizeof(T) == 2 && !__is_same(T, bf16_t);
    const T* res = reinterpret_cast<const T*>(p.res);
    const T* dec = reinterpret_cast<const T*>(p.dec);
    const T* shf = reinterpret_cast<const T*>(p.shift);
    float bv[2] = {0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int n = n0 + wc * (32 * NJ) + j * 32 + (lane & 31);
        bv[j] = (p.bias && n < p.Cout) ? bias_of(p, m0)[n] : 0.f;
    }
    float gs[16];   // GN: per-thread sum / sum of squares of the 8 channels of this thread's chunk column
#pragma unroll
    for (int e = 0; e < 16; ++e) gs[e] = 0.f;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        uint4 pre0[CPT], pre1[CPT];
#pragma unroll
        for (int c = 0; c < CPT; ++c) {
            const int cidx = tid + 512 * c;
            const int rl = cidx / (BN / 8), n = n0 + (cidx % (BN / 8)) * 8;
            const int m = m0 + (rl >> 6) * 128 + pass * 64 + (rl & 63);
            pre0[c] = pre1[c] = make_uint4(0, 0, 0, 0);
            if (m < p.M && n < p.Cout) {
                if (X3) {
                    if (p.epi == 1) {            // split SFT operands are fetched where they are used (4 x 16 bytes per chunk)
                    } else if (res && p.res_f32) {      // 8 floats = 2 x 16 bytes
                        const float* rf = reinterpret_cast<const float*>(p.res) + (long)m * p.ldr + n;
                        pre0[c] = *reinterpret_cast<const uint4*>(rf);
                        pre1[c] = *reinterpret_cast<const uint4*>(rf + 4);
                    } else if (res) {
                        pre0[c] = *reinterpret_cast<const uint4*>(res + (long)m * p.ldr + n);
                        pre1[c] = *reinterpret_cast<const uint4*>(res + (long)m * p.ldr + p.rlo + n);
                    }
                } else if (p.epi == 1) {
                    if constexpr (!kLateSft) {
                        pre0[c] = *reinterpret_cast<const uint4*>(dec + (long)m * p.ld_dec + n);
                        pre1[c] = *reinterpret_cast<const uint4*>(shf + (long)m * p.ld_shift + n);
                    }
                } else if (res) {
                    pre0[c] = *reinterpret_cast<const uint4*>(res + (long)m * p.ldr + n);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int cl = wc * (32 * NJ) + j * 32 + (lane & 31);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int rl = wr * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
                    if (!(PGT_PROBE & 128)) stage[rl * SROW + cl] = acc[pass * 2 + i][j][e] + bv[j];
                }
        }
        __syncthreads();
        if (X3 && p.x3 == 2) {     // folded 64-channel form: columns 64.. hold the x_hi * w_lo products of channels 0..63
#pragma unroll 1
            for (int c = 0; c < CPT; ++c) {
                const int cidx = tid + 512 * c;
                const int c8 = (cidx % (BN / 8)) * 8;
                if (c8 < 64) {
                    float* sp = stage + (cidx / (BN / 8)) * SROW + c8;
#pragma unroll
                    for (int e = 0; e < 8; e += 4) {
                        float4 a = *reinterpret_cast<const float4*>(sp + e);
                        const float4 b = *reinterpret_cast<const float4*>(sp + 64 + e);
                        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
                        *reinterpret_cast<float4*>(sp + e) = a;
                    }
                }
            }
            __syncthreads();       // the activation below rewrites chunks other threads have just read
        }
        if (p.act != ACT_NONE) {   // in place on the thread's own chunks, in a ROLLED loop: one copy of the switch
#pragma unroll 1
            for (int c = 0; c < CPT; ++c) {
                const int cidx = tid + 512 * c;
                float* sp = stage + (cidx / (BN / 8)) * SROW + (cidx % (BN / 8)) * 8;
                float v[8];
                *reinterpret_cast<float4*>(v) = *reinterpret_cast<const float4*>(sp);
                *reinterpret_cast<float4*>(v + 4) = *reinterpret_cast<const float4*>(sp + 4);
                apply_act8(v, p.act);
                *reinterpret_cast<float4*>(sp) = *reinterpret_cast<const float4*>(v);
                *reinterpret_cast<float4*>(sp + 4) = *reinterpret_cast<const float4*>(v + 4);
            }
        }
#pragma unroll
        for (int c = 0; c < CPT; ++c) {
            const int cidx = tid + 512 * c;
            const int rl = cidx / (BN / 8), c8 = (cidx % (BN / 8)) * 8;
            const int m = m0 + (rl >> 6) * 128 + pass * 64 + (rl & 63), n = n0 + c8;
            if (m >= p.M || n >= p.Cout) continue;
            float v[8];
            if (PGT_PROBE & 256) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (float)(c8 + e);
            } else {
            *reinterpret_cast<float4*>(v) = *reinterpret_cast<const float4*>(stage + rl * SROW + c8);
            *reinterpret_cast<float4*>(v + 4) = *reinterpret_cast<const float4*>(stage + rl * SROW + c8 + 4);
            }
            if ((PGT_PROBE & 64) && v[0] != 12345.f) continue;
            if (X3) {
                if (p.epi == 1) {    // out = dec + w * (dec * scale + shift) on split operands (reference: pgtformer_arch.py:478-479)
                    const T* dp = dec + (long)m * p.ld_dec + n;
                    const T* sp2 = shf + (long)m * p.ld_shift + n;
                    float d[8], sh[8];
                    merge8(*reinterpret_cast<const uint4*>(dp), *reinterpret_cast<const uint4*>(dp + p.dlo), d);
                    merge8(*reinterpret_cast<const uint4*>(sp2), *reinterpret_cast<const uint4*>(sp2 + p.slo), sh);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = d[e] + p.sft_w * (d[e] * v[e] + sh[e]);
                } else if (res) {
                    float r[8];
                    if (p.res_f32) {
                        Vec16<float>::unpack(pre0[c], r);
                        Vec16<float>::unpack(pre1[c], r + 4);
                    } else {
                        merge8(pre0[c], pre1[c], r);
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += r[e];
                }
                if (p.post_relu) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
                }
                if constexpr (GN) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) { gs[e] += v[e]; gs[8 + e] += v[e] * v[e]; }
                }
                if (!p.out_f32) {
                    uint4 hi, lo;
                    split8(v, hi, lo);
                    T* yp = reinterpret_cast<T*>(p.y) + out_row(p, m) * p.ldy + n;
                    *reinterpret_cast<uint4*>(yp) = hi;
                    *reinterpret_cast<uint4*>(yp + p.ylo) = lo;
                    continue;
                }
            } else if (p.epi == 1) {
                float d[8], sh[8];
                if constexpr (kLateSft) {   // (4 launches per forward: fetched where they are used, no second prefetch array)
                    Vec16<T>::unpack(*reinterpret_cast<const uint4*>(dec + (long)m * p.ld_dec + n), d);
                    Vec16<T>::unpack(*reinterpret_cast<const uint4*>(shf + (long)m * p.ld_shift + n), sh);
                } else {
                    Vec16<T>::unpack(pre0[c], d);
                    Vec16<T>::unpack(pre1[c], sh);
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = d[e] + p.sft_w * (d[e] * v[e] + sh[e]);
            } else {
                if (res) {
                    float r[8];
                    Vec16<T>::unpack(pre0[c], r);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += r[e];
                }
                if (p.post_relu) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
                }
            }
            if constexpr (GN && !X3) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { gs[e] += v[e]; gs[8 + e] += v[e] * v[e]; }
            }
            if (p.out_f32) store8<float>(reinterpret_cast<float*>(p.y) + out_row(p, m) * p.ldy + n, v);
            else store8<T>(reinterpret_cast<T*>(p.y) + out_row(p, m) * p.ldy + n, v);
        }
        if (pass == 0) __syncthreads();
    }
    if constexpr (GN) gn_tile_reduce<BN / 8, 8>(p, gs, stage, tid, m0, n0, WR * 128);
}

}  // namespace
