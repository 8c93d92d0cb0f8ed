// (T,Wh,Ww)-window attention on MFMA (K5 of SURVEY.md §2.3; reference: modules/rstt_layers.py:195-234 with
// window_partition/reverse :55-88, torch.roll :307-327 and the shift mask :552-568; the same kernel covers the
// Video-Swin parametrisation of modules/swin.py:85-167 — e.g. 3x8x8 windows, N = 192 tokens, C = 512).
//
// One workgroup per (window, head); one wavefront per 48 queries (N = 48 -> 1 wave, N = 192 -> 4 waves).
// Swapped formulation on 16x16 tiles so that a query lives in ONE lane column:
//   S^T[key, q]  = K . Q^T        v_mfma_f32_16x16x32_bf16, both operands are 16-byte row gathers straight
//                                 from the (rolled, partitioned) token rows in HBM — no LDS, no copies
//   O^T[d, q]   += V^T . P^T      v_mfma_f32_16x16x16_bf16: the 4 accumulator registers of an S^T tile
//                                 (keys 4g..4g+3 of lane group g) ARE the P^T operand after exp/convert;
//                                 V^T comes from an LDS image transposed while staging
// Roll / partition / reverse are address arithmetic (tok[]), the relative-position bias is a dense
// (heads,N,N) fp32 table read as float4, the 9-region mask is two region-id compares.  Online softmax over
// 48-key groups (one cross-lane-group max per group), exp2 domain.
#include "common.h"
#include "pgt_internal.h"

namespace {

typedef short short4v __attribute__((ext_vector_type(4)));
typedef _Float16 half8v __attribute__((ext_vector_type(8)));
typedef _Float16 half4v __attribute__((ext_vector_type(4)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));

// 16-bit element type of the kernel: bf16 (F16 = false) or IEEE half (F16 = true; BASELINE.json configs[4]: "fp16 MFMA")
template <bool F16> struct El {
    static __device__ __forceinline__ uint32_t pack2(float lo, float hi) {
        if constexpr (F16) {
            const half2v v = {(_Float16)lo, (_Float16)hi};
            return __builtin_bit_cast(uint32_t, v);
        } else {
            return f2bf2(lo, hi);
        }
    }
    static __device__ __forceinline__ float lo_of(uint32_t w) {      // the two values of a packed dword
        if constexpr (F16) return (float)__builtin_bit_cast(half2v, w).x;
        else return __uint_as_float(w << 16);
    }
    static __device__ __forceinline__ float hi_of(uint32_t w) {
        if constexpr (F16) return (float)__builtin_bit_cast(half2v, w).y;
        else return __uint_as_float(w & 0xffff0000u);
    }
    static __device__ __forceinline__ f32x4 mma32(const uint4& a, const uint4& b, f32x4 c) {   // 16x16x32
        if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8v, a), __builtin_bit_cast(half8v, b), c, 0, 0, 0);
        else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x4 mma16(const uint2& a, const uint2& b, f32x4 c) {   // 16x16x16
        if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(half4v, a), __builtin_bit_cast(half4v, b), c, 0, 0, 0);
        else return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(short4v, a), __builtin_bit_cast(short4v, b), c, 0, 0, 0);
    }
};

// X3: qkv and out are split rows on two half planes (hi plane at the usual columns, lo plane qlo / olo elements further): every
// product is taken as hi*hi + lo*hi + hi*lo (S^T from q, k; O^T from P, V with P split in registers after the exp).
// Windows are (wd, wh, ww) blocks of the (D, H, W) token grid with a cyclic shift (sd, sh, sw) and the 27-region mask of
// modules/swin.py:311-323; the PGTFormer layers use wd = D, sd = 0 (all frames of a spatial window in one group).
#ifndef PGT_WATTN_OCC
#define PGT_WATTN_OCC 4      // waves per SIMD asked of the compiler for the 48-token, 32-wide-head form (tools/bench_wattn.py A/Bs)
#endif
// HPW heads of one window per workgroup (each on its own NW waves): the workgroup then asks for HPW adjacent 64 / 128-byte
// slices of every token row at the same moment instead of leaving them to HPW workgroups that run whenever they are scheduled.
template <int HD, int NW, bool X3 = false, bool F16 = false, int HPW = 1>
__global__ __launch_bounds__(64 * NW * HPW, (HD == 32 && NW == 1 && HPW == 1) ? PGT_WATTN_OCC : 1) void window_attn_mfma_kernel(const uint16_t* __restrict__ qkv, int ldqkv,
                                                                   uint16_t* __restrict__ out, int ldo,
                                                                   const float* __restrict__ bias, int T_, int H, int W,
                                                                   int C, int heads, int wh, int ww, int sh, int sw,
                                                                   int qlo, int olo, int wd, int sd, int p2) {
    static_assert(!(X3 && F16), "X3 selects the split rows (half planes), F16 plain half rows");
    typedef El<F16 || X3> EL;
    constexpr int N = 48 * NW;
    constexpr int VSTR = N * 2 + 8;   // V^T row stride in bytes (keys contiguous, 8-byte pad)
    constexpr int KS = HD / 32;       // k-steps of the S^T MFMA
    constexpr int DT = HD / 16;       // 16-wide head-dim tiles of O^T
    constexpr int NP = X3 ? 2 : 1;    // operand planes
    constexpr float LOG2E = 1.44269504088896340736f;
    // Windows of more than one 48-key group (N = 96 .. 192: the Video-Swin form, BASELINE config 5): the K rows are staged in LDS
    // ONCE per workgroup next to V^T (round 5).  Before, every wave fetched every key group's K fragments from global memory
    // inside the group loop - (NW - 1) dependent L2 / HBM round trips per wave that nothing overlapped (config 5: 0.25 of HBM).
    constexpr bool KLDS = NW > 1;
    constexpr int KSTR = HD * 2 + 16;   // K row stride in bytes (dims contiguous, 16-byte pad: conflict-free ds_read_b128 over 16 keys)
    __shared__ __attribute__((aligned(16))) char k_all[KLDS ? NP * N * KSTR : 16];
    __shared__ __attribute__((aligned(16))) char vt_all[HPW * NP * HD * VSTR];
    __shared__ int tok[N];
    __shared__ int reg[N];

    const int hsub = HPW > 1 ? __builtin_amdgcn_readfirstlane(threadIdx.x / (64 * NW)) : 0;   // which of the workgroup's heads
    const int tid = HPW > 1 ? threadIdx.x - hsub * (64 * NW) : threadIdx.x;                   // thread within that head's waves
    char* const vt = vt_all + hsub * (NP * HD * VSTR);
    // XCD-aware order: consecutive workgroup ids go round-robin to the 8 XCDs (each with its own L2), but the heads of one
    // window read ADJACENT 64 / 128-byte slices of the same token rows - give every XCD a contiguous run of (window, head)
    // ids so that those slices meet in one L2 (bijective for any grid size, as in the conv kernels)
    int bid;
    {
        const int nblk = gridDim.x, b0 = blockIdx.x, xcd = b0 & 7, q8 = nblk >> 3, r8 = nblk & 7;
        bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (b0 >> 3);
    }
    // p2 (host: pgt_window_attn_mfma): ww, wh, W/ww, H/wh and heads are powers of two and wd == D - bit 30 set, their log2 in
    // 5-bit fields.  The index arithmetic is then shifts and masks: with one wave per workgroup and ~1100 instructions per
    // window the dozen integer divisions of the general form were a third of the kernel's issue slots (round 4).
    const bool fast = (p2 >> 30) & 1;
    const int l_ww = p2 & 31, l_wh = (p2 >> 5) & 31, l_nwx = (p2 >> 10) & 31, l_nwy = (p2 >> 15) & 31, l_hd = (p2 >> 20) & 31;
    int head, wx, wy, wz, b;
    const int hgroups = heads / HPW;      // (host: heads % HPW == 0; fast => a power of two)
    if (fast) {
        head = (bid & (hgroups - 1)) * HPW + hsub; bid >>= l_hd;
        wx = bid & ((1 << l_nwx) - 1); bid >>= l_nwx;
        wy = bid & ((1 << l_nwy) - 1); bid >>= l_nwy;
        wz = 0; b = bid;
    } else {
        const int nwx = W / ww, nwy = H / wh;
        head = (bid % hgroups) * HPW + hsub; bid /= hgroups;
        wx = bid % nwx; bid /= nwx;
        wy = bid % nwy; bid /= nwy;
        const int nwz = T_ / wd;
        wz = bid % nwz;
        b = bid / nwz;
    }
    const bool shifted = (sh > 0) || (sw > 0) || (sd > 0);

    for (int i = threadIdx.x; i < N; i += 64 * NW * HPW) {
        int s, r, dd;
        if (fast) {
            s = i & (ww - 1); r = (i >> l_ww) & (wh - 1); dd = i >> (l_ww + l_wh);
        } else {
            s = i % ww; r = (i / ww) % wh; dd = i / (ww * wh);
        }
        const int ds = wz * wd + dd, ys = wy * wh + r, xs = wx * ww + s;      // coordinates in the rolled volume
        int d = ds + sd, y = ys + sh, x = xs + sw;                            // source token (roll by -shift; shifts < sizes)
        d = d >= T_ ? d - T_ : d; y = y >= H ? y - H : y; x = x >= W ? x - W : x;
        tok[i] = ((b * T_ + d) * H + y) * W + x;
        // img_mask regions (rstt_layers.py:552-563; swin.py:313-318): only equality inside one window matters
        const int rd = ds < T_ - wd ? 0 : (ds < T_ - sd ? 1 : 2);
        const int rh = ys < H - wh ? 0 : (ys < H - sh ? 1 : 2);
        const int rw = xs < W - ww ? 0 : (xs < W - sw ? 1 : 2);
        reg[i] = (rd * 3 + rh) * 3 + rw;
    }
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6, g = lane >> 4, col = lane & 15;
    const float scale = X3 ? 1.0f / sqrtf((float)HD) : rsqrtf((float)HD);
    // Q^T fragments of this wave's 3 query tiles and the K fragments of the first 48-key group: requested BEFORE the V^T staging
    // below, so that the three operands travel together - the kernel is a chain of dependent HBM round trips otherwise (V rows ->
    // LDS -> barrier -> Q -> K: measured 3.3 TB/s at 16 waves per CU, round 4)
    uint4 qf[3][KS], ql[X3 ? 3 : 1][KS];
    uint4 kf0[KLDS ? 1 : 3][KS], kl0[(X3 && !KLDS) ? 3 : 1][KS];
    int qidx[3], rq[3];
#pragma unroll
    for (int qt = 0; qt < 3; ++qt) {
        qidx[qt] = wave * 48 + qt * 16 + col;
        rq[qt] = reg[qidx[qt]];
        const uint16_t* qrow = qkv + (long)tok[qidx[qt]] * ldqkv + head * HD + g * 8;
        const uint16_t* krow = qkv + (long)tok[qt * 16 + col] * ldqkv + C + head * HD + g * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            qf[qt][ks] = *reinterpret_cast<const uint4*>(qrow + ks * 32);
            if constexpr (!KLDS) kf0[qt][ks] = *reinterpret_cast<const uint4*>(krow + ks * 32);
            if constexpr (X3) {
                ql[qt][ks] = *reinterpret_cast<const uint4*>(qrow + qlo + ks * 32);
                if constexpr (!KLDS) kl0[qt][ks] = *reinterpret_cast<const uint4*>(krow + qlo + ks * 32);
            }
        }
    }
    if constexpr (KLDS) {   // ---- K image: work item = (plane, key, 16-byte chunk of its head slice); all loads first, then the LDS writes
        constexpr int CH = HD / 8, ITEMS = NP * N * CH, PER = (ITEMS + 64 * NW - 1) / (64 * NW);
        uint4 kv[PER];
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int it = tid + u * 64 * NW;
            kv[u] = make_uint4(0, 0, 0, 0);
            if (it < ITEMS) {
                const int pl = it / (N * CH), it2 = it % (N * CH), key = it2 / CH, c = it2 % CH;
                kv[u] = *reinterpret_cast<const uint4*>(qkv + (long)tok[key] * ldqkv + pl * qlo + C + head * HD + c * 8);
            }
        }
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int it = tid + u * 64 * NW;
            if (it < ITEMS) {
                const int pl = it / (N * CH), it2 = it % (N * CH), key = it2 / CH, c = it2 % CH;
                *reinterpret_cast<uint4*>(k_all + (pl * N + key) * KSTR + c * 16) = kv[u];
            }
        }
    }
    // ---- V^T image: work item = (key pair, 8-channel chunk); dword = {V[2kp][d], V[2kp+1][d]}
    for (int it = tid; it < NP * (N / 2) * (HD / 8); it += 64 * NW) {
        const int pl = it / ((N / 2) * (HD / 8)), it2 = it % ((N / 2) * (HD / 8));   // plane (0 = hi, 1 = lo)
        const int kp = it2 / (HD / 8), c = it2 % (HD / 8);
        const uint16_t* vb = qkv + pl * qlo + 2 * C + head * HD + c * 8;
        const uint4 v0 = *reinterpret_cast<const uint4*>(vb + (long)tok[2 * kp] * ldqkv);
        const uint4 v1 = *reinterpret_cast<const uint4*>(vb + (long)tok[2 * kp + 1] * ldqkv);
        const uint32_t a[4] = {v0.x, v0.y, v0.z, v0.w};
        const uint32_t bb[4] = {v1.x, v1.y, v1.z, v1.w};
        char* vp = vt + pl * HD * VSTR;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            *reinterpret_cast<uint32_t*>(vp + (c * 8 + 2 * j) * VSTR + kp * 4) = (a[j] & 0xffffu) | (bb[j] << 16);
            *reinterpret_cast<uint32_t*>(vp + (c * 8 + 2 * j + 1) * VSTR + kp * 4) = (a[j] >> 16) | (bb[j] & 0xffff0000u);
        }
    }
    __syncthreads();

    float m[3], l[3];
    f32x4 o[3][DT];
#pragma unroll
    for (int qt = 0; qt < 3; ++qt) {
        m[qt] = -INFINITY;
        l[qt] = 0.f;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) o[qt][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    // BPRE: the NEXT key group's bias rows requested while this group multiplies.  36 more live registers: with 64-wide heads the
    // kernel then needs 336 VGPRs = one workgroup per CU instead of two (236) - not taken there; the second workgroup hides the
    // ~1 us of an L2 round trip better than the prefetch does
    constexpr bool BPRE = KLDS && HD == 32;
    float4 bvs_next[BPRE ? 3 : 1][3];
    if constexpr (BPRE) {
#pragma unroll
        for (int qt = 0; qt < 3; ++qt)
#pragma unroll
            for (int kt = 0; kt < 3; ++kt)
                bvs_next[qt][kt] = *reinterpret_cast<const float4*>(bias + ((long)head * N + qidx[qt]) * N + kt * 16 + 4 * g);
    }
    for (int kg = 0; kg < NW; ++kg) {   // groups of 48 keys
        uint4 kf[3][KS], kl[X3 ? 3 : 1][KS];
#pragma unroll
        for (int kt = 0; kt < 3; ++kt) {
            const uint16_t* krow = qkv + (long)tok[kg * 48 + kt * 16 + col] * ldqkv + C + head * HD + g * 8;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                if constexpr (KLDS) {                // fragment = 16 keys x 32 dims: key row (kg*48 + kt*16 + col), dims ks*32 + g*8 .. +8
                    const char* kr = k_all + (kg * 48 + kt * 16 + col) * KSTR + ks * 64 + g * 16;
                    kf[kt][ks] = *reinterpret_cast<const uint4*>(kr);
                    if constexpr (X3) kl[kt][ks] = *reinterpret_cast<const uint4*>(kr + N * KSTR);
                } else if (kg == 0) {                // (requested before the V^T staging)
                    kf[kt][ks] = kf0[kt][ks];
                    if constexpr (X3) kl[kt][ks] = kl0[kt][ks];
                } else {
                    kf[kt][ks] = *reinterpret_cast<const uint4*>(krow + ks * 32);
                    if constexpr (X3) kl[kt][ks] = *reinterpret_cast<const uint4*>(krow + qlo + ks * 32);
                }
            }
        }
        int rk[3][4];
#pragma unroll
        for (int kt = 0; kt < 3; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) rk[kt][r] = reg[kg * 48 + kt * 16 + 4 * g + r];
        float4 bvs[3][3];        // the group's bias rows requested together, ahead of the products that want them (-4 %, round 4)
        if constexpr (BPRE) {
#pragma unroll
            for (int qt = 0; qt < 3; ++qt)
#pragma unroll
                for (int kt = 0; kt < 3; ++kt) {
                    bvs[qt][kt] = bvs_next[qt][kt];
                    if (kg + 1 < NW)
                        bvs_next[qt][kt] = *reinterpret_cast<const float4*>(bias + ((long)head * N + qidx[qt]) * N + (kg + 1) * 48 + kt * 16 + 4 * g);
                }
        } else {
#pragma unroll
            for (int qt = 0; qt < 3; ++qt)
#pragma unroll
                for (int kt = 0; kt < 3; ++kt)
                    bvs[qt][kt] = *reinterpret_cast<const float4*>(bias + ((long)head * N + qidx[qt]) * N + kg * 48 + kt * 16 + 4 * g);
        }
        __builtin_amdgcn_sched_barrier(0);      // (hipcc otherwise sinks every load to its use)
#pragma unroll
        for (int qt = 0; qt < 3; ++qt) {
            f32x4 s[3];
            float mx = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < 3; ++kt) {
                s[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    if constexpr (X3) {   // small terms first
                        s[kt] = EL::mma32(kl[kt][ks], qf[qt][ks], s[kt]);
                        s[kt] = EL::mma32(kf[kt][ks], ql[qt][ks], s[kt]);
                    }
                    s[kt] = EL::mma32(kf[kt][ks], qf[qt][ks], s[kt]);
                }
                const float4 bv = bvs[qt][kt];
                const float bvv[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = s[kt][r] * scale + bvv[r];
                    if (shifted && rk[kt][r] != rq[qt]) v += -100.0f;
                    v *= LOG2E;
                    s[kt][r] = v;
                    mx = fmaxf(mx, v);
                }
            }
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float mnew = fmaxf(m[qt], mx);
            const float alpha = __builtin_amdgcn_exp2f(m[qt] - mnew);
            m[qt] = mnew;
            float lsum = 0.f;
            uint2 pf[3], pl2[X3 ? 3 : 1];
#pragma unroll
            for (int kt = 0; kt < 3; ++kt) {
                float p[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    p[r] = __builtin_amdgcn_exp2f(s[kt][r] - mnew);
                    lsum += p[r];
                }
                pf[kt] = make_uint2(EL::pack2(p[0], p[1]), EL::pack2(p[2], p[3]));
                if constexpr (X3) {
                    x3_opaque(pf[kt].x);      // the lo plane is taken against the packed hi bits (common.h)
                    x3_opaque(pf[kt].y);
                    const float r0 = p[0] - EL::lo_of(pf[kt].x), r1 = p[1] - EL::hi_of(pf[kt].x);
                    const float r2 = p[2] - EL::lo_of(pf[kt].y), r3 = p[3] - EL::hi_of(pf[kt].y);
                    pl2[kt] = make_uint2(EL::pack2(r0, r1), EL::pack2(r2, r3));
                }
            }
            l[qt] = l[qt] * alpha + lsum;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
#pragma unroll
                for (int r = 0; r < 4; ++r) o[qt][dt][r] *= alpha;
#pragma unroll
                for (int kt = 0; kt < 3; ++kt) {
                    const uint2 a = *reinterpret_cast<const uint2*>(vt + (dt * 16 + col) * VSTR + (kg * 48 + kt * 16 + 4 * g) * 2);
                    if constexpr (X3) {
                        const uint2 al = *reinterpret_cast<const uint2*>(vt + HD * VSTR + (dt * 16 + col) * VSTR + (kg * 48 + kt * 16 + 4 * g) * 2);
                        o[qt][dt] = EL::mma16(al, pf[kt], o[qt][dt]);
                        o[qt][dt] = EL::mma16(a, pl2[kt], o[qt][dt]);
                    }
                    o[qt][dt] = EL::mma16(a, pf[kt], o[qt][dt]);
                }
            }
        }
    }
#pragma unroll
    for (int qt = 0; qt < 3; ++qt) {
        float lt = l[qt];
        lt += __shfl_xor(lt, 16, 64);
        lt += __shfl_xor(lt, 32, 64);
        const float inv = 1.0f / lt;
        uint16_t* orow = out + (long)tok[qidx[qt]] * ldo + head * HD + 4 * g;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const float v0 = o[qt][dt][0] * inv, v1 = o[qt][dt][1] * inv, v2 = o[qt][dt][2] * inv, v3 = o[qt][dt][3] * inv;
            uint2 w2 = make_uint2(EL::pack2(v0, v1), EL::pack2(v2, v3));
            if constexpr (X3) { x3_opaque(w2.x); x3_opaque(w2.y); }
            *reinterpret_cast<uint2*>(orow + dt * 16) = w2;
            if constexpr (X3) {
                const uint2 wl = make_uint2(EL::pack2(v0 - EL::lo_of(w2.x), v1 - EL::hi_of(w2.x)),
                                            EL::pack2(v2 - EL::lo_of(w2.y), v3 - EL::hi_of(w2.y)));
                *reinterpret_cast<uint2*>(orow + olo + dt * 16) = wl;
            }
        }
    }
}

constexpr int kDefaultHpw = 8;      // measured on MI355X: 1 -> 8 heads per workgroup -5 ... -8 % (profiles/r4_g_window_attention_hpw.jsonl)

}  // namespace

// mode 0: bf16, 1: split-half (lo planes qlo / olo elements after the hi planes), 2: fp16.  Windows (wd, wh, ww) with shift
// (sd, sh, sw) on a (B, D, H, W) token grid; N = wd*wh*ww in {48, 96, 144, 192}; hd in {32, 64}.
// Returns 1 when the shape is not covered (caller falls back or reports).
int pgt_window_attn_mfma(int mode, const void* qkv, int ldqkv, void* out, int ldo, const float* bias, int B, int D, int H,
                         int W, int C, int heads, int wd, int wh, int ww, int sd, int sh, int sw, hipStream_t st, int qlo,
                         int olo) {
    const int N = wd * wh * ww, hd = C / heads;
    if (N % 48 != 0 || N > 192 || (hd != 32 && hd != 64) || wd <= 0 || D % wd != 0) return 1;
    if (ldqkv % 8 != 0 || ldo % 4 != 0 || ((uintptr_t)qkv & 15) || ((uintptr_t)out & 7) || ((uintptr_t)bias & 15)) return 1;
    if (mode == 1 && (qlo % 8 != 0 || olo % 4 != 0)) return 1;
    const int nw = N / 48;
    // heads per workgroup (48-token windows of 32-wide heads: the model's layers): PGT_WATTN_HPW = 1 | 2 | 4 | 8 for A/Bs
    static const int env_hpw = [] { const char* e = getenv("PGT_WATTN_HPW"); return e ? atoi(e) : 0; }();
    int hpw = env_hpw > 0 ? env_hpw : kDefaultHpw;
    if (nw != 1 || hd != 32 || (hpw != 2 && hpw != 4 && hpw != 8) || heads % hpw != 0) hpw = 1;
    const int grid = B * (D / wd) * (H / wh) * (W / ww) * (heads / hpw);
    sd %= D; sh %= H; sw %= W;      // (the kernel rolls by one conditional subtraction)
    auto lg = [](int v) { int l = 0; while ((1 << l) < v) ++l; return (1 << l) == v ? l : -1; };
    int p2 = 0;
    if (lg(ww) >= 0 && lg(wh) >= 0 && lg(W / ww) >= 0 && lg(H / wh) >= 0 && lg(heads) >= 0 && wd == D)
        p2 = (1 << 30) | lg(ww) | (lg(wh) << 5) | (lg(W / ww) << 10) | (lg(H / wh) << 15) | (lg(heads / hpw) << 20);
#define WAM(HD_, NW_, X3_, F16_)                                                                                              \
    hipLaunchKernelGGL((window_attn_mfma_kernel<HD_, NW_, X3_, F16_>), dim3(grid), dim3(64 * NW_), 0, st, (const uint16_t*)qkv, \
                       ldqkv, (uint16_t*)out, ldo, bias, D, H, W, C, heads, wh, ww, sh, sw, qlo, olo, wd, sd, p2)
#define WAMH(X3_, F16_, HPW_)                                                                                                 \
    hipLaunchKernelGGL((window_attn_mfma_kernel<32, 1, X3_, F16_, HPW_>), dim3(grid), dim3(64 * HPW_), 0, st, (const uint16_t*)qkv, \
                       ldqkv, (uint16_t*)out, ldo, bias, D, H, W, C, heads, wh, ww, sh, sw, qlo, olo, wd, sd, p2)
#define WAM_ALL(X3_, F16_)                                                                                                    \
    do {                                                                                                                      \
        if (hpw == 2) { WAMH(X3_, F16_, 2); break; }                                                                           \
        if (hpw == 4) { WAMH(X3_, F16_, 4); break; }                                                                           \
        if (hpw == 8) { WAMH(X3_, F16_, 8); break; }                                                                           \
        if (hd == 32) {                                                                                                       \
            switch (nw) { case 1: WAM(32, 1, X3_, F16_); break; case 2: WAM(32, 2, X3_, F16_); break;                          \
                          case 3: WAM(32, 3, X3_, F16_); break; default: WAM(32, 4, X3_, F16_); }                              \
        } else {                                                                                                              \
            switch (nw) { case 1: WAM(64, 1, X3_, F16_); break; case 2: WAM(64, 2, X3_, F16_); break;                          \
                          case 3: WAM(64, 3, X3_, F16_); break; default: WAM(64, 4, X3_, F16_); }                              \
        }                                                                                                                     \
    } while (0)
    if (mode == 1) WAM_ALL(true, false);
    else if (mode == 2) WAM_ALL(false, true);
    else WAM_ALL(false, false);
#undef WAM_ALL
#undef WAMH
#undef WAM
    PGT_LAUNCH_CHECK();
    return 0;
}
