// Nearest-code look-up of the residual quantiser with the arg-min INSIDE the distance GEMM (K10 of SURVEY.md §2.3;
// reference: VQEmbedding.compute_distances / find_nearest_embedding, archs/tdcrqvae3_arch.py:100-126, and the same
// distance + arg-min of VectorQuantizer.forward, archs/vqgan_arch.py:48-54).
//
//   codes[t] = first argmin_j ( (|x_t|^2 + |e_j|^2) - 2 <x_t, e_j> )        (addmm(alpha=-2) association, lowest index on ties)
//
// The two-kernel form (distance GEMM with fp32 output -> row arg-min) writes and re-reads an Ntok x K fp32 matrix
// (1.07 GB at Ntok = 262144, K = 1024) for 0.27 GB of algorithmic traffic.  Here the dot products never leave the
// accumulators.  Swapped formulation, as in the attention kernels, so that a token lives in ONE lane column:
//   S^T[code, token] = E . X^T      v_mfma_f32_32x32x16_bf16, A = 32 codebook rows from LDS, B = the wave's 32 tokens,
//                                   held in registers for the whole kernel (D/16 fragments of 16 bytes per lane)
// In the 32x32 accumulator layout lane l owns column (l & 31) = one token and 16 of the 32 codes of the block (rows
// (e&3) + 8(e>>2) + 4(l>>5)), so the running (distance, index) minimum is a per-lane scalar pair updated in ascending
// code order (strict <: the lowest index wins ties); the only cross-lane step is one lane <-> lane+32 exchange at the
// very end.  Workgroup = 4 waves x 32 tokens; the codebook (K x D bf16, L2-resident) streams through LDS in blocks of
// 32 codes, the next block prefetched into registers while the current one is multiplied.  The k order of the
// contraction (ascending 16-element steps, fp32 accumulation) is that of the implicit-GEMM kernels, so the dot
// products - and therefore the codes - are bit-identical to the two-kernel form.
#include <atomic>

#include "common.h"
#include "pgt_internal.h"
#include "igemm_common.h"

namespace {

template <int D>
__global__ __launch_bounds__(256) void rq_nearest_mfma_kernel(const uint16_t* __restrict__ x, int ldx,
                                                              const uint16_t* __restrict__ book /* (K, D) bf16 */,
                                                              const float* __restrict__ xnorm, const float* __restrict__ enorm,
                                                              int rows, int K, int* __restrict__ codes) {
    constexpr int KS = D / 16;                  // k-steps of the 32x32x16 MFMA
    constexpr int RSTR = D * 2 + 16;            // LDS row stride in bytes (+16: conflict-free ds_read_b128 down a column)
    constexpr int CPT = 32 * (D / 8) / 256;     // 16-byte chunks of a 32-code block staged per thread
    static_assert(D % 64 == 0 && CPT >= 1, "embedding dim");
    __shared__ __attribute__((aligned(16))) char Es[32 * RSTR];
    __shared__ __attribute__((aligned(16))) float en_s[32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5;
    const int t = blockIdx.x * 128 + wave * 32 + (lane & 31);
    const bool tok_ok = t < rows;

    // X^T fragments (B operand): this lane's token, k-step s covers dims 16s + 8h .. +7
    uint4 xf[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        xf[s] = make_uint4(0, 0, 0, 0);
        if (tok_ok) xf[s] = *reinterpret_cast<const uint4*>(x + (long)t * ldx + s * 16 + h * 8);
    }
    const float x2 = tok_ok ? xnorm[t] : 0.f;

    // staging roles: chunk c = tid + 256 i of the block: code row c / (D/8), 16-byte column c % (D/8)
    uint4 re[CPT];
    auto gload = [&](int cb) {
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            const int c = tid + 256 * i;
            const int r = c / (D / 8), cc = c % (D / 8);
            const int code = cb * 32 + r;
            re[i] = make_uint4(0, 0, 0, 0);
            if (code < K) re[i] = *reinterpret_cast<const uint4*>(book + (long)code * D + cc * 8);
        }
    };
    auto sstore = [&](int cb) {
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            const int c = tid + 256 * i;
            *reinterpret_cast<uint4*>(Es + (c / (D / 8)) * RSTR + (c % (D / 8)) * 16) = re[i];
        }
        if (tid < 32) en_s[tid] = (cb * 32 + tid < K) ? enorm[cb * 32 + tid] : INFINITY;   // codes >= K never win
    };

    float best = INFINITY;
    int bi = 0x7fffffff;
    const int nb = (K + 31) / 32;
    gload(0);
    sstore(0);
    __syncthreads();
    const char* e_rd = Es + (lane & 31) * RSTR + h * 16;
    for (int cb = 0; cb < nb; ++cb) {
        const bool more = cb + 1 < nb;
        if (more) gload(cb + 1);
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const uint4 a = *reinterpret_cast<const uint4*>(e_rd + s * 32);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, xf[s]),
                                                          acc, 0, 0, 0);
        }
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const float4 en4 = *reinterpret_cast<const float4*>(en_s + 8 * g4 + 4 * h);
            const float en[4] = {en4.x, en4.y, en4.z, en4.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                // the reference's association: (|x|^2 + |e|^2) + (-2) * <x, e>
                const float v = (x2 + en[r]) - 2.0f * acc[4 * g4 + r];
                const int j = cb * 32 + 8 * g4 + 4 * h + r;
                if (v < best) { best = v; bi = j; }   // ascending j per lane: strict < keeps the first minimum
            }
        }
        __syncthreads();
        if (more) {
            sstore(cb + 1);
            __syncthreads();
        }
    }
    // the two lane halves hold disjoint code subsets of the same token
    const float ob = __shfl_xor(best, 32, 64);
    const int oi = __shfl_xor(bi, 32, 64);
    if (ob < best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    if (tok_ok && h == 0) codes[t] = bi;
}

// D = 512 (the model's embedding width): one codebook row is exactly one 1-KiB LDS-DMA (64 lanes x 16 bytes), so the
// codebook streams global -> LDS without passing through registers: the next 64 codes are in flight while the current 64 are
// multiplied, one barrier per 64 codes, no staging stores.  |e|^2 of all codes sits in LDS for the whole kernel.
// 8 waves x 32 tokens per workgroup (the tokens live in registers as MFMA B fragments, 128 VGPRs), so the codebook
// crosses L2 -> LDS once per 256 tokens.  Each wave multiplies TWO 32-code blocks at a time: an MFMA whose accumulator
// is the previous MFMA's stalls ~43 cycles as soon as anything (here: the LDS reads of the next fragments) is issued
// between the two, alternating two accumulators hides that -- and leaves every dot product's k order untouched.
constexpr int DMA_WAVES = 8, DMA_TOK = DMA_WAVES * 32, DMA_RSTR = 512 * 2 + 16, DMA_BUF = 32 * DMA_RSTR;
constexpr int DMA_MAXK = 4096;

__global__ __launch_bounds__(DMA_WAVES * 64) void rq_nearest_dma512_kernel(const uint16_t* __restrict__ x, int ldx,
                                                                          const uint16_t* __restrict__ book,
                                                                          const float* __restrict__ xnorm,
                                                                          const float* __restrict__ enorm, int rows, int K,
                                                                          int npair, int* __restrict__ codes) {
    constexpr int D = 512, KS = D / 16, RSTR = DMA_RSTR, BUF = DMA_BUF, RPW = 64 / DMA_WAVES;
    extern __shared__ __attribute__((aligned(1024))) char smem[];   // 2 x (64 code rows), then npair * 64 floats |e|^2
    float* en_all = reinterpret_cast<float*>(smem + 4 * BUF);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5;
    const int t = blockIdx.x * DMA_TOK + wave * 32 + (lane & 31);
    const bool tok_ok = t < rows;
    const unsigned lds0 = lds_addr(smem);

    auto issue = [&](int pair) {   // this wave's 8 rows of code rows 64*pair .. 64*pair+63 -> stage pair & 1
#pragma unroll
        for (int r0 = 0; r0 < RPW; ++r0) {
            const int r = wave * RPW + r0;
            int code = pair * 64 + r;
            code = code < K ? code : K - 1;             // ragged tail: any valid row, masked by |e|^2 = +inf below
            glds16(book + (long)code * D + lane * 8, lds0 + (pair & 1) * 2 * BUF + r * RSTR);
        }
    };
    issue(0);
    for (int j = tid; j < npair * 64; j += DMA_WAVES * 64) en_all[j] = j < K ? enorm[j] : INFINITY;
    uint4 xf[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        xf[s] = make_uint4(0, 0, 0, 0);
        if (tok_ok) xf[s] = *reinterpret_cast<const uint4*>(x + (long)t * ldx + s * 16 + h * 8);
    }
    const float x2 = tok_ok ? xnorm[t] : 0.f;
    float best = INFINITY;
    int bi = 0x7fffffff;
    // rows 8*g4 + 4*h + r of a finished 32x32 tile; ascending code index per lane, so strict < keeps the first minimum
    auto argmin4 = [&](const f32x16& a, int cb, int g4) {
        const float4 en4 = *reinterpret_cast<const float4*>(en_all + cb * 32 + 8 * g4 + 4 * h);
        const float en[4] = {en4.x, en4.y, en4.z, en4.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float v = (x2 + en[r]) - 2.0f * a[4 * g4 + r];
            const bool lt = v < best;
            best = lt ? v : best;
            bi = lt ? cb * 32 + 8 * g4 + 4 * h + r : bi;
        }
    };
    // the arg-min over a finished pair of tiles (160 VALU operations) is spread between the NEXT pair's MFMAs, in the
    // shadow of the matrix pipe.  Before the first pair: dot products -inf, distances +inf, nothing is kept.
    f32x16 p0, p1;
#pragma unroll
    for (int e = 0; e < 16; ++e) p0[e] = p1[e] = -INFINITY;
    for (int i = 0; i < npair; ++i) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's rows of pair i have landed
        __syncthreads();                                   // ... everybody's have, and everybody has left pair i-1's stage
        if (i + 1 < npair) issue(i + 1);
        const char* e_rd = smem + (i & 1) * 2 * BUF + (lane & 31) * RSTR + h * 16;
        f32x16 acc0, acc1;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc0[e] = acc1[e] = 0.f;
        // fragments are read one group (4 k-steps x 2 blocks) ahead of their MFMAs (hipcc otherwise sinks every read
        // next to its MFMA: read, wait, multiply)
        uint4 fa[2][4][2];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            fa[0][j][0] = *reinterpret_cast<const uint4*>(e_rd + j * 32);
            fa[0][j][1] = *reinterpret_cast<const uint4*>(e_rd + BUF + j * 32);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < KS / 4; ++g) {
            if (g + 1 < KS / 4) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    fa[(g + 1) & 1][j][0] = *reinterpret_cast<const uint4*>(e_rd + ((g + 1) * 4 + j) * 32);
                    fa[(g + 1) & 1][j][1] = *reinterpret_cast<const uint4*>(e_rd + BUF + ((g + 1) * 4 + j) * 32);
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[g & 1][j][0]),
                                                               __builtin_bit_cast(bf16x8, xf[g * 4 + j]), acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[g & 1][j][1]),
                                                               __builtin_bit_cast(bf16x8, xf[g * 4 + j]), acc1, 0, 0, 0);
            }
            const int pc = i > 0 ? 2 * i - 2 : 0;
            if (g < 4) argmin4(p0, pc, g);
            else argmin4(p1, pc + 1, g - 4);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // one MFMA
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // one read of the next group
                __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);   // a slice of the previous pair's arg-min
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        p0 = acc0;
        p1 = acc1;
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) argmin4(p0, 2 * npair - 2, g);
#pragma unroll
    for (int g = 0; g < 4; ++g) argmin4(p1, 2 * npair - 1, g);
    const float ob = __shfl_xor(best, 32, 64);
    const int oi = __shfl_xor(bi, 32, 64);
    if (ob < best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    if (tok_ok && h == 0) codes[t] = bi;
}

}  // namespace

// bf16 tokens x (rows, D) and codebook (K, D) bf16; xnorm / enorm fp32.  D in {64, 128, 256, 512}.
extern "C" int pgt_rq_nearest(int32_t dtype, const void* x, int32_t ldx, const void* book, const float* xnorm,
                              const float* enorm, int32_t rows, int32_t K, int32_t D, int32_t* codes, pgt_stream_t stream) {
    PGT_CHECK(x && book && xnorm && enorm && codes && rows > 0 && K > 0, "rq_nearest: bad argument");
    PGT_CHECK(dtype == PGT_BF16, "rq_nearest: the fused kernel takes bf16 operands (fp32: distance GEMM + pgt_rq_argmin)");
    PGT_CHECK(ldx % 8 == 0 && ((((uintptr_t)x) | ((uintptr_t)book)) & 15) == 0, "rq_nearest: rows must be 16-byte aligned");
    const dim3 grid((rows + 127) / 128), blk(256);
    hipStream_t st = (hipStream_t)stream;
    if (D == 512 && K <= DMA_MAXK) {      // LDS-DMA streaming form
        const int npair = (K + 63) / 64;
        const int lds = 4 * DMA_BUF + npair * 64 * (int)sizeof(float);
        static std::atomic<unsigned long long> attr_set{0};
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (!((attr_set.load(std::memory_order_acquire) >> (dev & 63)) & 1ull)) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&rq_nearest_dma512_kernel),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 4 * DMA_BUF + DMA_MAXK * 4);
            if (e != hipSuccess) { pgt_set_error("rq_nearest: cannot reserve LDS: %s", hipGetErrorString(e)); return -12; }
            attr_set.fetch_or(1ull << (dev & 63), std::memory_order_release);
        }
        hipLaunchKernelGGL(rq_nearest_dma512_kernel, dim3((rows + DMA_TOK - 1) / DMA_TOK), dim3(DMA_WAVES * 64), lds, st,
                           (const uint16_t*)x, ldx, (const uint16_t*)book, xnorm, enorm, rows, K, npair, codes);
        PGT_LAUNCH_CHECK();
        return 0;
    }
#define RQN(D_)                                                                                                   \
    hipLaunchKernelGGL((rq_nearest_mfma_kernel<D_>), grid, blk, 0, st, (const uint16_t*)x, ldx, (const uint16_t*)book, \
                       xnorm, enorm, rows, K, codes)
    switch (D) {
        case 64: RQN(64); break;
        case 128: RQN(128); break;
        case 256: RQN(256); break;
        case 512: RQN(512); break;
        default: PGT_CHECK(false, "rq_nearest: D=%d unsupported (64, 128, 256, 512)", D);
    }
#undef RQN
    PGT_LAUNCH_CHECK();
    return 0;
}
