// Flash attention on MFMA for the code-prediction transformer (K8: L=3072, 8 heads, hd=64, bf16).
//
// Swapped formulation so that every per-query quantity lives in ONE lane:
//   S^T = K . Q^T        (v_mfma_f32_32x32x16_bf16: A = K tile rows from LDS, B = Q^T held in registers)
//   O^T += V^T . P^T     (A = V^T from a transposed LDS image, B = P^T built in registers from S^T)
// In the 32x32 accumulator layout lane l owns column (l & 31) = one query, so the running max / sum and
// the rescale factor are per-lane scalars and the only cross-lane traffic per tile is one lane<->lane+32
// exchange of the tile maximum.  The P^T operand is formed directly from the S^T accumulator
// registers: accumulator registers 8s..8s+7 of a 32-key block are exactly the 8 keys lane-half h
// contributes to k-step s if the keys inside each 16-key group are taken in the order
// {4h..4h+3, 8+4h..8+4h+3}; V^T fragments are read with the same key order (two ds_read_b64), so the
// contraction is unchanged and P never round-trips through LDS.
// Workgroup = NW waves x 32 queries (NW = 8 for long sequences: the K / V staging - global loads, the V transpose through
// 4-byte LDS stores, the barrier - is paid once per 256 queries instead of once per 128; measured on the L = 3072 layer:
// staging 34 % of the kernel at NW = 4); K/V streamed in 64-key tiles through a DOUBLE-buffered LDS stage: tile t+1 is
// fetched into registers while tile t is multiplied and written to the other stage afterwards - one barrier per tile.
#include <cstdlib>

#include "common.h"
#include "pgt_internal.h"

namespace {

constexpr int HD = 64;
constexpr int BKV = 64;
constexpr int KSTR = 144;   // K tile row stride in bytes (128 + 16 pad): conflict-free ds_read_b128
constexpr int VSTR = 136;   // V^T row stride in bytes (64 keys * 2 + 8): conflict-free ds_read_b64

// 16-bit operand type of a launch: bf16 (plain bf16 rows), or the half planes of split rows (X3; common.h x3p_t)
template <bool H> struct P16 {
    static __device__ __forceinline__ uint32_t pack2(float lo, float hi) {      // {lo, hi} packed
        if constexpr (H) return f2h2_nosat(lo, hi);                                // P <= 2^8, outputs are convex combinations of V rows
        else return f2bf2(lo, hi);                                               // v_cvt_pk_bf16_f32
    }
    static __device__ __forceinline__ float lo_of(uint32_t w) {
        if constexpr (H) return (float)__builtin_bit_cast(halfx2, w).x;
        else return __uint_as_float(w << 16);
    }
    static __device__ __forceinline__ float hi_of(uint32_t w) {
        if constexpr (H) return (float)__builtin_bit_cast(halfx2, w).y;
        else return __uint_as_float(w & 0xffff0000u);
    }
    static __device__ __forceinline__ f32x16 mma(const uint4& a, const uint4& b, const f32x16& c) {
        if constexpr (H) return mma16<half_t>(a, b, c);
        else return mma16<bf16_t>(a, b, c);
    }
};
// raw v_exp_f32 (2^x): arguments here are <= 8 and results feed a bf16 operand, no range fix-up needed
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// X3: q, k, v and out are split rows (two half planes), lo planes qlo / klo / vlo / olo elements after the hi planes; S^T and O^T
// take three MFMAs per product (hi*hi + lo*hi + hi*lo), P is split in registers after the exp.
// (P on its hi plane only in P.V - 5 products per key tile - was measured in round 5 and rejected: logits error 1.7e-5 -> 3.5e-5,
// profiles/r5_mha_p_single_plane_study.jsonl; the switch is gone.)
// (183 VGPRs at NW = 8: one workgroup per CU.  Forcing 128 registers for two workgroups per CU spills 244 bytes per lane into the tile
// loop and runs 2.1x slower - 4448 against 2106 us at 32 windows, profiles/r6_g_mha_occupancy_experiment.jsonl)
template <bool X3, int NW = 4>
__global__ __launch_bounds__(64 * NW) void mha_mfma_kernel(const uint16_t* __restrict__ q, int ldq, const uint16_t* __restrict__ k,
                                                       int ldk, const uint16_t* __restrict__ v, int ldv,
                                                       uint16_t* __restrict__ out, int ldo, int L, float c /* scale*log2(e) */,
                                                       int qlo, int klo, int vlo, int olo) {
    using E = P16<X3>;                                      // split rows live on half planes
    constexpr int NP = X3 ? 2 : 1;
    constexpr int PLANE = BKV * KSTR + HD * VSTR;
    constexpr int STAGE = NP * PLANE;                       // one K / V^T tile (hi [+ lo] planes)
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5;
    // XCD-aware order over the flat (batch, head, query block) index: the query blocks of one (batch, head) stream the
    // same K / V rows - a contiguous run of ids per XCD keeps them behind one L2 (dispatch is round-robin over the 8 XCDs)
    int qblk, head, b;
    {
        const int nqb = gridDim.x, nh = gridDim.y, nblk = nqb * nh * gridDim.z;
        const int b0 = blockIdx.x + nqb * (blockIdx.y + nh * blockIdx.z);
        const int xcd = b0 & 7, q8 = nblk >> 3, r8 = nblk & 7;
        int id = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (b0 >> 3);
        qblk = id % nqb; id /= nqb;
        head = id % nh;
        b = id / nh;
    }
    const int qi = qblk * (32 * NW) + wave * 32 + (lane & 31);
    const long rowbase = (long)b * L;

    // Q^T fragments (B operand of S^T): 4 k-steps of 16 head-dims, this half's 8 dims each
    uint4 qf[4], qfl[X3 ? 4 : 1];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        qf[s] = make_uint4(0, 0, 0, 0);
        if (qi < L) qf[s] = *reinterpret_cast<const uint4*>(q + (rowbase + qi) * ldq + head * HD + s * 16 + h * 8);
        if constexpr (X3) {
            qfl[s] = make_uint4(0, 0, 0, 0);
            if (qi < L) qfl[s] = *reinterpret_cast<const uint4*>(q + (rowbase + qi) * ldq + qlo + head * HD + s * 16 + h * 8);
        }
    }
    // staging roles.  K: 64 rows x 8 16-byte chunks = 512 pieces; V: 32 key pairs x 8 head-dim chunks, transposed on the
    // way into LDS (dword = {V[2kp][d], V[2kp+1][d]} -> Vt[d][2kp..2kp+1]).  NW = 4: two K pieces and one whole V chunk
    // per thread; NW = 8: one K piece and half a V chunk (4 head dims) per thread.
    constexpr int KPT = 8 / NW;                     // K pieces per thread
    constexpr int VH = NW / 4;                      // V chunk halves: 1 (8 dims per thread) or 2 (4 dims per thread)
    const int k_r0 = tid >> 3, k_cc = tid & 7;      // K: rows k_r0 (+ 32 for the second piece at NW = 4); chunk k_cc
    const int v_item = tid / VH, v_half = tid % VH;
    const int v_kp = v_item >> 3, v_c = v_item & 7; // V: key pair (2kp, 2kp+1), head-dim chunk 8c..8c+7 (half v_half)
    uint4 rk[NP][KPT], rv[NP][2];                   // rv: for VH = 2 only .x/.y (4 dims) are used
    auto gload = [&](int k0) {
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) {
#pragma unroll
            for (int i = 0; i < KPT; ++i) {
                const int kj = k0 + k_r0 + 32 * i;
                rk[pl][i] = make_uint4(0, 0, 0, 0);
                if (kj < L) rk[pl][i] = *reinterpret_cast<const uint4*>(k + (rowbase + kj) * ldk + pl * klo + head * HD + k_cc * 8);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int vj = k0 + 2 * v_kp + i;
                rv[pl][i] = make_uint4(0, 0, 0, 0);
                if (vj < L) {
                    const uint16_t* vp = v + (rowbase + vj) * ldv + pl * vlo + head * HD + v_c * 8 + v_half * 4;
                    if constexpr (VH == 1) {
                        rv[pl][i] = *reinterpret_cast<const uint4*>(vp);
                    } else {
                        const uint2 t2 = *reinterpret_cast<const uint2*>(vp);
                        rv[pl][i].x = t2.x;
                        rv[pl][i].y = t2.y;
                    }
                }
            }
        }
    };
    auto sstore = [&](int buf) {
        char* Ks = smem + buf * STAGE;
        char* Vt = Ks + BKV * KSTR;
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) {
#pragma unroll
            for (int i = 0; i < KPT; ++i) *reinterpret_cast<uint4*>(Ks + pl * PLANE + (k_r0 + 32 * i) * KSTR + k_cc * 16) = rk[pl][i];
            // transpose V: dword = {V[2kp][d], V[2kp+1][d]} -> Vt[d][2kp..2kp+1]
            const uint32_t a[4] = {rv[pl][0].x, rv[pl][0].y, rv[pl][0].z, rv[pl][0].w};
            const uint32_t bb[4] = {rv[pl][1].x, rv[pl][1].y, rv[pl][1].z, rv[pl][1].w};
#pragma unroll
            for (int j = 0; j < 4 / VH; ++j) {
                const uint32_t lo = (a[j] & 0xffffu) | (bb[j] << 16);
                const uint32_t hi = (a[j] >> 16) | (bb[j] & 0xffff0000u);
                const int d0 = v_c * 8 + v_half * 4 + 2 * j;
                *reinterpret_cast<uint32_t*>(Vt + pl * PLANE + d0 * VSTR + v_kp * 4) = lo;
                *reinterpret_cast<uint32_t*>(Vt + pl * PLANE + (d0 + 1) * VSTR + v_kp * 4) = hi;
            }
        }
    };

    f32x16 o[2];
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[d][e] = 0.f;
    float m = -INFINITY, l = 0.f;

    const int nt = (L + BKV - 1) / BKV;
    gload(0);
    sstore(0);
    __syncthreads();
    for (int t = 0; t < nt; ++t) {
        const int k0 = t * BKV;
        const bool more = t + 1 < nt;
        const char* k_rd = smem + (t & 1) * STAGE + (lane & 31) * KSTR + h * 16;
        const char* v_rd = smem + (t & 1) * STAGE + BKV * KSTR + (lane & 31) * VSTR + h * 8;
        if (more) gload(k0 + BKV);
        // ---- S^T = K Q^T : 2 key blocks x 4 k-steps
        f32x16 s[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int e = 0; e < 16; ++e) s[kb][e] = 0.f;
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                const uint4 a = *reinterpret_cast<const uint4*>(k_rd + kb * 32 * KSTR + st * 32);
                if constexpr (X3) {   // small terms first
                    const uint4 al = *reinterpret_cast<const uint4*>(k_rd + PLANE + kb * 32 * KSTR + st * 32);
                    s[kb] = E::mma(al, qf[st], s[kb]);
                    s[kb] = E::mma(a, qfl[st], s[kb]);
                }
                s[kb] = E::mma(a, qf[st], s[kb]);
            }
        }
        if (k0 + BKV > L) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    if (k0 + kb * 32 + (e & 3) + 8 * (e >> 2) + 4 * h >= L) s[kb][e] = -INFINITY;
        }
        // ---- online softmax, per lane = per query
        float mx = s[0][0];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int e = 0; e < 16; ++e) mx = fmaxf(mx, s[kb][e]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        // Lazy rescale, exp2 domain (m = reference exponent of this query, scale folded in): the running sum and the
        // accumulators are rescaled only when some query of the wave would exceed the reference by 2^8 - P <= 256 is as
        // precise in bf16 / split-half as P <= 1, and after the first tiles the 32 + 2 multiplies per tile disappear.
        const float tmx = mx * c;
        if (__any(tmx > m + 8.0f)) {
            const float mnew = fmaxf(m, tmx);
            const float alpha = fast_exp2(m - mnew);     // first tile: m = -inf -> 0 (l and o are 0)
            l *= alpha;
#pragma unroll
            for (int d = 0; d < 2; ++d)
#pragma unroll
                for (int e = 0; e < 16; ++e) o[d][e] *= alpha;
            m = mnew;
        }
        float lsum = 0.f;
        uint4 pf[2][2], pfl[X3 ? 2 : 1][2];   // P^T B-operands: [key block][k-step]
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            float p[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                p[e] = fast_exp2(__builtin_fmaf(s[kb][e], c, -m));
                lsum += p[e];
            }
            pf[kb][0] = make_uint4(E::pack2(p[0], p[1]), E::pack2(p[2], p[3]), E::pack2(p[4], p[5]), E::pack2(p[6], p[7]));
            pf[kb][1] = make_uint4(E::pack2(p[8], p[9]), E::pack2(p[10], p[11]), E::pack2(p[12], p[13]), E::pack2(p[14], p[15]));
            if constexpr (X3) {
                x3_opaque(pf[kb][0]);      // the lo plane is taken against the packed hi bits (common.h)
                x3_opaque(pf[kb][1]);
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const uint32_t w4[4] = {pf[kb][s2].x, pf[kb][s2].y, pf[kb][s2].z, pf[kb][s2].w};
                    uint32_t r4[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        r4[j] = E::pack2(p[s2 * 8 + 2 * j] - E::lo_of(w4[j]),
                                      p[s2 * 8 + 2 * j + 1] - E::hi_of(w4[j]));
                    pfl[kb][s2] = make_uint4(r4[0], r4[1], r4[2], r4[3]);
                }
            }
        }
        l += lsum;
        // ---- O^T += V^T P^T : 2 head-dim blocks x (2 key blocks x 2 k-steps)
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const char* pa = v_rd + d * 32 * VSTR + (kb * 32 + s2 * 16) * 2;
                    const uint2 lo = *reinterpret_cast<const uint2*>(pa);        // keys 4h..4h+3 of the group
                    const uint2 hi = *reinterpret_cast<const uint2*>(pa + 16);   // keys 8+4h..8+4h+3
                    const uint4 a = make_uint4(lo.x, lo.y, hi.x, hi.y);
                    if constexpr (X3) {
                        const uint2 llo = *reinterpret_cast<const uint2*>(pa + PLANE);
                        const uint2 lhi = *reinterpret_cast<const uint2*>(pa + PLANE + 16);
                        const uint4 al = make_uint4(llo.x, llo.y, lhi.x, lhi.y);
                        o[d] = E::mma(al, pf[kb][s2], o[d]);
                        o[d] = E::mma(a, pfl[kb][s2], o[d]);
                    }
                    o[d] = E::mma(a, pf[kb][s2], o[d]);
                }
        // the other stage was last read in tile t-1, which every wave left through the barrier below
        if (more) sstore((t + 1) & 1);
        __syncthreads();
    }
    l += __shfl_xor(l, 32, 64);
    if (qi < L) {
        const float inv = 1.0f / l;
        uint16_t* orow = out + (rowbase + qi) * ldo + head * HD;
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int dd = d * 32 + 8 * g4 + 4 * h;   // rows (e&3) + 8*(e>>2) + 4h of the accumulator
                const float v0 = o[d][4 * g4 + 0] * inv, v1 = o[d][4 * g4 + 1] * inv;
                const float v2 = o[d][4 * g4 + 2] * inv, v3 = o[d][4 * g4 + 3] * inv;
                uint2 w;
                w.x = E::pack2(v0, v1);
                w.y = E::pack2(v2, v3);
                if constexpr (X3) { x3_opaque(w.x); x3_opaque(w.y); }
                *reinterpret_cast<uint2*>(orow + dd) = w;
                if constexpr (X3) {
                    uint2 wl;
                    wl.x = E::pack2(v0 - E::lo_of(w.x), v1 - E::hi_of(w.x));
                    wl.y = E::pack2(v2 - E::lo_of(w.y), v3 - E::hi_of(w.y));
                    *reinterpret_cast<uint2*>(orow + olo + dd) = wl;
                }
            }
    }
}

}  // namespace

int pgt_mha_mfma_bf16(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, void* out, int ldo, int B,
                      int L, int heads, float scale, hipStream_t st, int x3, int qlo, int klo, int vlo, int olo) {
    PGT_CHECK(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && ldo % 4 == 0, "mha: row strides must be multiples of 8");
    PGT_CHECK((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) & 15) == 0 && ((uintptr_t)out & 7) == 0, "mha: misaligned pointer");
    PGT_CHECK(!x3 || (qlo % 8 == 0 && klo % 8 == 0 && vlo % 8 == 0 && olo % 4 == 0), "mha: misaligned lo planes");
    const float c = scale * 1.44269504088896340736f;
    const int nw = L >= 512 ? 8 : 4;            // 256 queries per workgroup once the sequence is long enough to fill the chip
    const dim3 grid((L + 32 * nw - 1) / (32 * nw), heads, B);
#define MHA_GO(X3_, NW_, QLO, KLO, VLO, OLO)                                                                                  \
    hipLaunchKernelGGL((mha_mfma_kernel<X3_, NW_>), grid, dim3(64 * NW_), 0, st, (const uint16_t*)q, ldq, (const uint16_t*)k, \
                       ldk, (const uint16_t*)v, ldv, (uint16_t*)out, ldo, L, c, QLO, KLO, VLO, OLO)
    if (x3) {
        if (nw == 8) MHA_GO(true, 8, qlo, klo, vlo, olo);
        else MHA_GO(true, 4, qlo, klo, vlo, olo);
    } else {
        if (nw == 8) MHA_GO(false, 8, 0, 0, 0, 0);
        else MHA_GO(false, 4, 0, 0, 0, 0);
    }
#undef MHA_GO
    PGT_LAUNCH_CHECK();
    return 0;
}
