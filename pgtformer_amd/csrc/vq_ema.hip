// Training-side quantiser: the EMA codebook update of VQEmbedding (reference: archs/tdcrqvae3_arch.py:138-186), fp32.
//
//   pgt_vq_cluster_stats   one batch's statistics per code: sum of the vectors assigned to it and their count
//                          (_update_buffers :139-158: one_hot @ vectors, one_hot.sum(1)), into ONE flat buffer
//                          [K*D sums | K counts] so that data-parallel ranks combine them with a single all-reduce
//   pgt_vq_ema_update      EMA of both (:160-161), restart of codes whose EMA count fell below 1 (:163-177), codebook =
//                          embed_ema / normalised cluster size (_update_embedding :179-186)
//
// The sums are deterministic: one workgroup per code walks the batch in row order (the code indices of the batch stay in
// L2: K x rows x 4 bytes of L2 reads, 1 GB at 262144 rows) and adds the vectors assigned to its code in ascending row order -
// no floating-point atomics, the same bits on every run and for every grid size.
#include "common.h"
#include "pgt_internal.h"

namespace {

constexpr int MAXD = 2048;   // 256 threads x float2 x 4

__global__ __launch_bounds__(256) void vq_cluster_stats_kernel(const float* __restrict__ x, int ldx,
                                                               const int* __restrict__ codes, int rows, int K, int D,
                                                               float* __restrict__ stats) {
    __shared__ int list[256];
    __shared__ int wcnt[4];
    const int k = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float2 acc[MAXD / 512];
#pragma unroll
    for (int i = 0; i < MAXD / 512; ++i) acc[i] = make_float2(0.f, 0.f);
    int count = 0;
    for (int base = 0; base < rows; base += 256) {
        const int r = base + tid;
        const bool hit = r < rows && codes[r] == k;
        const unsigned long long b = __ballot(hit);
        if (lane == 0) wcnt[wave] = __popcll(b);
        __syncthreads();
        int off = 0, total = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            off += w < wave ? wcnt[w] : 0;
            total += wcnt[w];
        }
        if (hit) list[off + __popcll(b & ((1ull << lane) - 1ull))] = r;   // ascending row order
        __syncthreads();
        for (int i = 0; i < total; ++i) {
            const float* row = x + (long)list[i] * ldx;
#pragma unroll
            for (int j = 0; j < MAXD / 512; ++j) {
                const int d = j * 512 + tid * 2;
                if (d < D) {
                    const float2 v = *reinterpret_cast<const float2*>(row + d);
                    acc[j].x += v.x;
                    acc[j].y += v.y;
                }
            }
        }
        count += total;
        __syncthreads();   // the list is rewritten by the next chunk
    }
#pragma unroll
    for (int j = 0; j < MAXD / 512; ++j) {
        const int d = j * 512 + tid * 2;
        if (d < D) *reinterpret_cast<float2*>(stats + (long)k * D + d) = acc[j];
    }
    if (tid == 0) stats[(long)K * D + k] = (float)count;
}

// EMA + restart for code k (one workgroup); the new cluster_size_ema[k] is needed by every code's normalisation, hence the
// second kernel.
__global__ __launch_bounds__(256) void vq_ema_kernel(float* __restrict__ cs_ema, float* __restrict__ embed_ema,
                                                     const float* __restrict__ stats, const float* __restrict__ restart,
                                                     int K, int D, float decay, float alpha) {
    const int k = blockIdx.x;
    // cluster_size_ema.mul_(decay).add_(cluster_size, alpha = 1 - decay)
    const float cs = fmaf(alpha, stats[(long)K * D + k], cs_ema[k] * decay);
    const bool dead = restart && !(cs >= 1.f);   // usage = (cluster_size_ema >= 1)
    for (int d = threadIdx.x; d < D; d += 256) {
        const long o = (long)k * D + d;
        const float e = fmaf(alpha, stats[o], embed_ema[o] * decay);
        embed_ema[o] = dead ? restart[o] : e;   // embed_ema * usage + random * (1 - usage), usage in {0, 1}
    }
    if (threadIdx.x == 0) cs_ema[k] = dead ? 1.f : cs;   // cs * usage + (1 - usage)
}

// n = sum(cluster_size_ema) (fixed-order tree, every workgroup the same bits); weight[k] = embed_ema[k] / (n (cs_k + eps) /
// (n + K eps))
__global__ __launch_bounds__(256) void vq_normalise_kernel(const float* __restrict__ cs_ema, const float* __restrict__ embed_ema,
                                                           float* __restrict__ weight, int ldw, int K, int D, float eps) {
    __shared__ float red[256];
    float s = 0.f;
    for (int j = threadIdx.x; j < K; j += 256) s += cs_ema[j];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    const float n = red[0];
    const int k = blockIdx.x;
    const float norm = n * (cs_ema[k] + eps) / (n + (float)K * eps);
    for (int d = threadIdx.x; d < D; d += 256) weight[(long)k * ldw + d] = embed_ema[(long)k * D + d] / norm;
}

}  // namespace

extern "C" int pgt_vq_cluster_stats(const float* x, int32_t ldx, const int32_t* codes, int32_t rows, int32_t K, int32_t D,
                                    float* stats, pgt_stream_t stream) {
    PGT_CHECK(x && codes && stats && rows > 0 && K > 0, "vq_cluster_stats: bad argument");
    PGT_CHECK(D > 0 && D <= MAXD && D % 2 == 0 && ldx % 2 == 0 && ((uintptr_t)x & 7) == 0 && ((uintptr_t)stats & 7) == 0,
              "vq_cluster_stats: D=%d must be even and <= %d, rows 8-byte aligned", D, MAXD);
    hipLaunchKernelGGL(vq_cluster_stats_kernel, dim3(K), dim3(256), 0, (hipStream_t)stream, x, ldx, codes, rows, K, D, stats);
    PGT_LAUNCH_CHECK();
    return 0;
}

extern "C" int pgt_vq_ema_update(float* cluster_size_ema, float* embed_ema, const float* stats, const float* restart,
                                 float* weight, int32_t ldw, int32_t K, int32_t D, float decay, float one_minus_decay,
                                 float eps, pgt_stream_t stream) {
    PGT_CHECK(cluster_size_ema && embed_ema && stats && weight && K > 0 && D > 0 && ldw >= D, "vq_ema_update: bad argument");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(vq_ema_kernel, dim3(K), dim3(256), 0, st, cluster_size_ema, embed_ema, stats, restart, K, D, decay,
                       one_minus_decay);
    PGT_LAUNCH_CHECK();
    hipLaunchKernelGGL(vq_normalise_kernel, dim3(K), dim3(256), 0, st, cluster_size_ema, embed_ema, weight, ldw, K, D, eps);
    PGT_LAUNCH_CHECK();
    return 0;
}
