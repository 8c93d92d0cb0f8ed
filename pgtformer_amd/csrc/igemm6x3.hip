// 3x3 convolution, 64 input channels, <= 64 output channels on SPLIT-half operands (PGT_F16X3): the full-resolution
// levels of the encoder (64 -> 64 at 512x512, the code-prediction branch).  On the 256- / 128-wide tiles of igemm4.hip a
// 64-channel layer idles half of the tile or, in the folded form, spends a fourth product on a zero quadrant; either way
// the operands stream through LDS once per filter tap.
//
// Design (igemm6.hip's, re-cut for three products per reference product):
//   * MFMA 16x16x32: a wave owns 16 OUTPUT CHANNELS and all 64 pixels of a tile (4 accumulators of 16x16).  Its whole
//     weight set - 9 taps x 64 input channels x {w_hi, w_lo} x 16 channels - lives in REGISTERS: 9 x 2 k-steps x 2 planes
//     B fragments = 144 VGPRs, loaded once per (persistent) workgroup straight from the standard split weight matrix
//     ([w_hi | w_hi | w_lo] per tap: pgt_pack_conv_weight).  4 waves = 64 channels.
//   * per 64-pixel tile the three input rows-with-halo images (ky = 0, 1, 2) of BOTH planes (x_hi, x_lo) are DMA'd into LDS
//     once (6 images of <= 9 KiB, XOR-swizzled: reads shifted by kx stay conflict-free) and serve all 9 taps and all 3 products:
//     acc += x_lo * w_hi,  acc += x_hi * w_lo,  acc += x_hi * w_hi  (small terms first) - exactly three MFMAs per product.
//   * no operand streaming inside the K loop: DMA -> vmcnt(0) -> barrier -> 216 MFMAs per wave -> LDS-staged epilogue
//     (bias, activation, split residual, split8 -> hi / lo planes).  54 KiB of LDS and <= 256 VGPRs per 4-wave workgroup:
//     two workgroups share a CU, one computes while the other waits for its DMA or stores its tile.
//
// Preconditions (caller): PGT_F16X3 with split or fp32 output (then an fp32 residual), KH = KW = 3, stride 1, pad 1, Cin == 64, Cout in {16, 32, 48, 64},
// Ho == H, Wo == W, W and H powers of two, W >= 32, H*W >= 64, input < 2 GiB, plain epilogue (no SFT, no statistics).
#include "common.h"
#include "pgt_internal.h"
#include "igemm_common.h"

namespace {

typedef float f32x4v __attribute__((ext_vector_type(4)));

constexpr int kPieces = 9;                     // 1-KiB pieces per image (8 rows each): E <= 68 rows
constexpr int kImgX = kPieces * 1024;          // bytes of one image
constexpr int kLdsX = 6 * kImgX;               // 54 KiB: (ky = 0..2) x (hi, lo); re-used as the fp32 epilogue stage (17 KiB)
constexpr int kSRowX = 64 + 4;
static_assert(64 * kSRowX * 4 <= kLdsX, "epilogue stage must fit");

__global__ __launch_bounds__(256, 2) void conv3x3_c64_x3_kernel(ConvP p, int ntiles) {
    constexpr unsigned kOob = 0x80000000u;
    __shared__ __attribute__((aligned(1024))) char smem[kLdsX];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // = the wave's 16-channel group
    const int col = lane & 15, kg = lane >> 4;                   // MFMA 16x16x32: column / row index, k group of 8
    const unsigned lds0 = lds_addr(smem);
    const v4i rsrc_x = make_rsrc(p.x, (unsigned)((long)p.N * p.H * p.W * p.ldx * 2));

    // ---- weights of output channel n = 16 wave + col: B fragment of (tap, ks, plane) = w[n][tap*192 + plane_off + ks*32 + kg*8 .. +8]
    //      (standard split matrix: per tap [w_hi (64) | w_hi (64) | w_lo (64)])
    uint4 whi[9][2], wlo[9][2];
    {
        const int n = wave * 16 + col;
        const uint4* wp = reinterpret_cast<const uint4*>(p.w + ((long)n * p.K + kg * 8) * 2);
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                uint4 vh = make_uint4(0, 0, 0, 0), vl = make_uint4(0, 0, 0, 0);
                if (n < p.Cout) {
                    vh = wp[t * 24 + ks * 4];          // (t*192 + ks*32) elements = (t*24 + ks*4) x 16 bytes
                    vl = wp[t * 24 + 16 + ks * 4];     // + 128 elements: the w_lo block
                }
                whi[t][ks] = vh;
                wlo[t][ks] = vl;
            }
    }

    const int S = p.W < 64 ? p.W : 64;
    const int s_shift = p.W < 64 ? p.wo_shift : 6;
    const int S2 = S + 2, segs = 64 >> s_shift, E = 64 + 2 * segs;
    const int row_bytes = p.W * p.ldx * 2;
    int e0[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const int r = mt * 16 + col;
        e0[mt] = (r >> s_shift) * S2 + (r & (S - 1));
    }
    const x3p_t* res = reinterpret_cast<const x3p_t*>(p.res);
    float* stage = reinterpret_cast<float*>(smem);

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int m0 = tile * 64;
        __syncthreads();   // the previous tile's epilogue has left the LDS
        // ---- the six images: piece q = wave + 4 i (i < 3) of image (ky, plane) = rows 8q + (lane >> 3), chunk lane & 7
#pragma unroll 1
        for (int i = 0; i < 3; ++i) {
            const int q = wave + 4 * i;
            if (q >= kPieces) break;           // (wave-uniform) 9 pieces cover the <= 68 rows of an image
            const int e = 8 * q + (lane >> 3);
            const int c = (lane & 7) ^ ((e >> 1) & 7);
            int seg = 0;
            for (int k = 1; k < segs; ++k) seg += e >= k * S2 ? 1 : 0;
            const int xx = e - seg * S2;
            const int mseg = m0 + (seg << s_shift);
            int pix = 0, oy = 0;
            bool ok = false;
            if (e < E && mseg < p.M) {
                const int ox0 = mseg & (p.W - 1);
                const int t = mseg >> p.wo_shift;
                oy = t & (p.H - 1);
                const int img = t >> p.ho_shift;
                const int ix = ox0 - p.pad_l + xx;
                ok = (unsigned)ix < (unsigned)p.W;
                pix = (((img * p.H + oy - p.pad_t) * p.W + ix) * p.ldx + c * 8) * 2;
            }
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const bool v = ok && (unsigned)(oy - p.pad_t + ky) < (unsigned)p.H;
                const unsigned off = v ? (unsigned)(pix + ky * row_bytes) : kOob;
                bufdma16(off, rsrc_x, 0, lds0 + (2 * ky) * kImgX + q * 1024);                // x_hi
                bufdma16(off, rsrc_x, p.xlo * 2, lds0 + (2 * ky + 1) * kImgX + q * 1024);    // x_lo: xlo elements further
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();

        f32x4v acc[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) acc[mt] = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int t = ky * 3 + kx;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    uint4 ah[4], al[4];
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) {
                        const int e = e0[mt] + kx;
                        const int off = e * 128 + (((ks * 4 + kg) ^ ((e >> 1) & 7)) << 4);
                        ah[mt] = *reinterpret_cast<const uint4*>(smem + (2 * ky) * kImgX + off);
                        al[mt] = *reinterpret_cast<const uint4*>(smem + (2 * ky + 1) * kImgX + off);
                    }
                    // small terms first: x_lo * w_hi, x_hi * w_lo, then x_hi * w_hi; the four M tiles interleaved so that
                    // consecutive MFMAs never share an accumulator
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt)
                        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(halfx8, al[mt]), __builtin_bit_cast(halfx8, whi[t][ks]), acc[mt], 0, 0, 0);
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt)
                        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(halfx8, ah[mt]), __builtin_bit_cast(halfx8, wlo[t][ks]), acc[mt], 0, 0, 0);
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt)
                        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(halfx8, ah[mt]), __builtin_bit_cast(halfx8, whi[t][ks]), acc[mt], 0, 0, 0);
                }
            }
        __syncthreads();   // every wave is done with the images

        // ---- epilogue: accumulators staged as fp32 (64 pixels x 64 channels): D[4 kg + r][col] of M tile mt
        {
            const int cl = wave * 16 + col;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) stage[(mt * 16 + 4 * kg + r) * kSRowX + cl] = acc[mt][r];
        }
        __syncthreads();
        for (int cidx = tid; cidx < 64 * 8; cidx += 256) {
            const int rl = cidx >> 3, c8 = (cidx & 7) * 8;
            const int m = m0 + rl;
            if (m >= p.M || c8 >= p.Cout) continue;
            float v[8];
            *reinterpret_cast<float4*>(v) = *reinterpret_cast<const float4*>(stage + rl * kSRowX + c8);
            *reinterpret_cast<float4*>(v + 4) = *reinterpret_cast<const float4*>(stage + rl * kSRowX + c8 + 4);
            if (p.bias) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += p.bias[c8 + e];
            }
            apply_act8(v, p.act);
            if (res) {
                float r[8];
                if (p.res_f32) {      // split arithmetic on fp32-stored tensors (BiSeNet's BasicBlocks)
                    load8<float>(reinterpret_cast<const float*>(p.res) + (long)m * p.ldr + c8, r);
                } else {
                    const x3p_t* rp = res + (long)m * p.ldr + c8;
                    merge8(*reinterpret_cast<const uint4*>(rp), *reinterpret_cast<const uint4*>(rp + p.rlo), r);
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += r[e];
            }
            if (p.post_relu) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
            }
            if (p.out_f32) {
                store8<float>(reinterpret_cast<float*>(p.y) + (long)m * p.ldy + c8, v);
                continue;
            }
            uint4 hi, lo;
            split8(v, hi, lo);
            x3p_t* yp = reinterpret_cast<x3p_t*>(p.y) + (long)m * p.ldy + c8;
            *reinterpret_cast<uint4*>(yp) = hi;
            *reinterpret_cast<uint4*>(yp + p.ylo) = lo;
        }
    }
}

}  // namespace

// See the preconditions at the top of the file; the caller checks them.
int pgt_igemm6x3_launch(const void* pv, hipStream_t st) {
    ConvP p = *reinterpret_cast<const ConvP*>(pv);
    p.wo_shift = __builtin_ctz(p.Wo);
    p.ho_shift = __builtin_ctz(p.Ho);
    const int ntiles = (p.M + 63) / 64;
    static int n_cu = 0;
    if (n_cu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
            pgt_set_error("igemm6x3: cannot query the device");
            return -5;
        }
        n_cu = prop.multiProcessorCount;
    }
    const int grid = ntiles < 2 * n_cu ? ntiles : 2 * n_cu;
    hipLaunchKernelGGL(conv3x3_c64_x3_kernel, dim3(grid), dim3(256), 0, st, p, ntiles);
    PGT_LAUNCH_CHECK();
    return 0;
}
