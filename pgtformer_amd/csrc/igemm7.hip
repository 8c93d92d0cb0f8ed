// Linear layers with K = 256 on very many rows (the q|k|v / proj / fc1 / fc2 GEMMs of the 128x128 and 64x64 window-attention
// blocks: 0.4 - 1.6 M token rows, 256 -> 256 or 256 -> 768, bf16 or IEEE half).  At K = 256 a row costs 512 B in and 512 B out
// (+ 512 B of residual) for 131 kFLOP: the layer is HBM-bound, and on the 256x256 tiles of igemm4.hip a workgroup spends a
// third of its time in its 4-K-tile main loop between an address set-up and a 128-KiB epilogue, one workgroup per CU with
// nothing to overlap them (2.8 - 3.0 TB/s).
//
// Design - a streaming GEMM, the weights never move:
//   * a workgroup (8 waves) owns 256 output columns; wave w owns columns 32 w .. 32 w + 31 and keeps its whole B operand -
//     32 columns x 256 k = 16 k-steps of the 32x32x16 MFMA - in 64 VGPRs, loaded once from the K-major weight rows.
//   * rows stream through LDS in blocks of 32: one 16-KiB image (32 rows x 512 B, XOR-swizzled 16-byte chunks) per block,
//     double-buffered, filled by LDS-DMA (buffer_load ... lds) one block ahead; every wave reads the whole image
//     (16 ds_read_b128) for its 16 MFMAs.
//   * the 32 x 256 fp32 results are staged in LDS and leave as whole 512-byte rows (bias - per frame where asked -,
//     activation, residual, one rounding), 16 bytes per thread.
//   * 64 KiB of LDS and <= 128 VGPRs: TWO workgroups per CU, so one streams its stores while the other multiplies; the
//     workgroups are persistent over the row blocks (grid = 2 x CUs).
//
// Preconditions (caller): bf16 / half, 1x1, Cin == 256, Cout % 256 == 0, plain epilogue (no SFT, no statistics, no placed
// rows), 16-bit output, 16-byte-aligned rows (ldx, ldy, ldr multiples of 8), input < 2 GiB.
#include "common.h"
#include "pgt_internal.h"
#include "igemm_common.h"

namespace {

constexpr int kRB = 32;                        // rows per block
constexpr int kABuf = kRB * 512;               // bytes of one A image
constexpr int kStageRow = 256;                 // floats per staged row
constexpr int kLds7 = 2 * kABuf + kRB * kStageRow * 4;   // 64 KiB

template <typename T>
__global__ __launch_bounds__(512, 4) void linear_k256_kernel(ConvP p, int nblk) {
    constexpr unsigned kOob = 0x80000000u;
    __shared__ __attribute__((aligned(1024))) char smem[kLds7];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hh = lane >> 5, r32 = lane & 31;
    const int n0 = blockIdx.y * 256;
    const unsigned lds0 = lds_addr(smem);
    const v4i rsrc_x = make_rsrc(p.x, (unsigned)((long)p.M * p.ldx * 2));
    float* stage = reinterpret_cast<float*>(smem + 2 * kABuf);

    // ---- B operand of this wave's 32 columns: (k-step ks) = w[n][ks*16 + hh*8 .. +8]
    uint4 breg[16];
    {
        const int n = n0 + wave * 32 + r32;
        const uint4* wp = reinterpret_cast<const uint4*>(p.w + ((long)n * 256 + hh * 8) * 2);
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) breg[ks] = wp[ks * 2];        // 16 elements = 2 x 16 bytes per k-step
    }
    // DMA role: piece q = 2 wave + i (i < 2) of a block = rows 2q, 2q + 1; lane -> row 2q + (lane >> 5), slot lane & 31
    auto dma = [&](int blk, int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int q = 2 * wave + i;
            const int row = 2 * q + hh;
            const int c = r32 ^ (row & 7);                       // the chunk that lives in this slot (XOR swizzle)
            const long m = (long)blk * kRB + row;
            const unsigned off = m < p.M ? (unsigned)((m * p.ldx + c * 8) * 2) : kOob;
            bufdma16(off, rsrc_x, 0, lds0 + buf * kABuf + q * 1024);
        }
    };
    const T* res = reinterpret_cast<const T*>(p.res);
    T* y = reinterpret_cast<T*>(p.y);

    int blk = blockIdx.x;
    if (blk < nblk) dma(blk, 0);
    for (int it = 0; blk < nblk; blk += gridDim.x, ++it) {
        const int buf = it & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                       // image `buf` is complete; the stage of the previous block has been drained
        if (blk + (int)gridDim.x < nblk) dma(blk + gridDim.x, buf ^ 1);
        const long m0 = (long)blk * kRB;
        // ---- 16 MFMAs: A fragment of k-step ks = row r32, chunk 2 ks + hh
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
        const char* arow = smem + buf * kABuf + r32 * 512;
#pragma unroll
        for (int k4 = 0; k4 < 4; ++k4) {          // four fragments in flight at a time: 64 VGPRs of weights leave room for no more
            uint4 a[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) a[j] = *reinterpret_cast<const uint4*>(arow + (((2 * (4 * k4 + j) + hh) ^ (r32 & 7)) << 4));
#pragma unroll
            for (int j = 0; j < 4; ++j) acc = mma16<T>(a[j], breg[4 * k4 + j], acc);
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- stage (+ bias: one column per lane; per frame where asked - a block of 32 rows lies inside one frame):
        //      accumulator register e of lane (hh, r32) = row (e & 3) + 8 (e >> 2) + 4 hh, column 32 wave + r32
        const float bv = p.bias ? bias_of(p, (int)m0)[n0 + wave * 32 + r32] : 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) stage[((e & 3) + 8 * (e >> 2) + 4 * hh) * kStageRow + wave * 32 + r32] = acc[e] + bv;
        __syncthreads();
        // ---- rows out: 32 rows x 32 chunks of 8 channels = 1024 chunks, 2 per thread
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = tid + 512 * i;
            const int row = idx >> 5, c8 = (idx & 31) * 8;
            const long m = m0 + row;
            if (m >= p.M) continue;
            float v[8];
            *reinterpret_cast<float4*>(v) = *reinterpret_cast<const float4*>(stage + row * kStageRow + c8);
            *reinterpret_cast<float4*>(v + 4) = *reinterpret_cast<const float4*>(stage + row * kStageRow + c8 + 4);
            apply_act8(v, p.act);
            if (res) {
                float r[8];
                load8<T>(res + m * p.ldr + n0 + c8, r);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += r[e];
            }
            if (p.post_relu) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
            }
            store8<T>(y + m * p.ldy + n0 + c8, v);
        }
    }
}

}  // namespace

// See the preconditions at the top of the file; the caller checks them.
int pgt_igemm7_launch(const void* pv, hipStream_t st) {
    const ConvP& p = *reinterpret_cast<const ConvP*>(pv);
    const int nblk = (p.M + kRB - 1) / kRB;
    static int n_cu = 0;
    if (n_cu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
            pgt_set_error("igemm7: cannot query the device");
            return -5;
        }
        n_cu = prop.multiProcessorCount;
    }
    const int ny = p.Cout / 256;
    int gx = (2 * n_cu + ny - 1) / ny;
    if (gx > nblk) gx = nblk;
    if (p.f16) hipLaunchKernelGGL(linear_k256_kernel<half_t>, dim3(gx, ny), dim3(512), 0, st, p, nblk);
    else hipLaunchKernelGGL(linear_k256_kernel<bf16_t>, dim3(gx, ny), dim3(512), 0, st, p, nblk);
    PGT_LAUNCH_CHECK();
    return 0;
}
