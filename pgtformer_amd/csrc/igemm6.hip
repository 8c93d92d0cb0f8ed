// 3x3 convolution, 64 input channels, <= 64 output channels (bf16): the full-resolution layers of the decoder / encoder
// (64 -> 64 and 64 -> 3 at 512x512).  A 256-wide implicit-GEMM tile wastes 3/4 of its MFMAs on these layers and the
// 64-wide tiles of igemm.hip re-fetch the input for each of the 9 taps (0.41 PFLOP/s, 1.4 TB/s of a 1.6 GB launch).
//
// Design: the whole weight tensor of a wave's 32 output channels (9 taps x 64 channels) lives in REGISTERS (144 VGPRs,
// loaded once per workgroup), workgroups are persistent and walk over 128-pixel output tiles; per tile the three input
// rows-with-halo images (ky = 0, 1, 2; the igemm5.hip image: pixels of S = min(W, 128)-pixel segments with one extra
// pixel on both sides, XOR-swizzled so that reads shifted by kx are bank-conflict free) are DMA'd into LDS once and
// serve all 9 taps.  No operand streaming inside the K loop, hence no hand-placed waits: DMA -> vmcnt(0) -> barrier ->
// 72 MFMAs per wave fed by 8 ds_read_b128 per tap -> LDS-staged epilogue.  Two 4-wave workgroups share a CU (60 KiB of
// LDS, <= 256 VGPRs each), so one computes while the other waits for its DMA or stores its tile.
//
// Preconditions (caller): bf16, KH = KW = 3, stride 1, pad 1, no up-sampling, Cin == 64, Cout <= 64, Ho == H, Wo == W,
// W and H powers of two, W >= 32, input < 2 GiB; a 16-byte-illegal epilogue (Cout % 8 != 0) takes a scalar path.
#include "common.h"
#include "pgt_internal.h"
#include "igemm_common.h"

namespace {

constexpr int kImgPieces = 20;                 // 1-KiB pieces reserved per image (>= 17 = 136 rows / 8)
constexpr int kImg = kImgPieces * 1024;        // bytes of one image
constexpr int kLds6 = 3 * kImg;                // 60 KiB: three images, re-used as the fp32 epilogue stage (34.8 KiB)
constexpr int kSRow = 64 + 4;
static_assert(128 * kSRow * 4 <= kLds6, "epilogue stage must fit");

__device__ __forceinline__ int swz6(int e, int c) { return e * 128 + ((c ^ ((e >> 1) & 7)) << 4); }

template <typename T>   // 16-bit operand type: bf16_t or half_t (PGT_F16)
__global__ __launch_bounds__(256, 2) void conv3x3_c64_kernel(ConvP p, int ntiles) {
    constexpr unsigned kOob = 0x80000000u;
    __shared__ __attribute__((aligned(1024))) char smem[kLds6];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;   // 64 pixels x 32 output channels per wave
    const int hh = lane >> 5;
    const unsigned lds0 = lds_addr(smem);
    const v4i rsrc_x = make_rsrc(p.x, (unsigned)((long)p.N * p.H * p.W * p.ldx * 2));

    // ---- weights of this wave's 32 output channels: B fragment of (tap, ks) = w[n][tap*64 + ks*16 + hh*8 .. +8]
    uint4 wreg[9][4];
    {
        const int n = wc * 32 + (lane & 31);
        const uint4* wp = reinterpret_cast<const uint4*>(p.w + ((long)n * p.K + hh * 8) * 2);
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                uint4 v = make_uint4(0, 0, 0, 0);
                if (n < p.Cout) v = wp[t * 8 + ks * 2];   // (t*64 + ks*16) elements = (t*8 + ks*2) x 16 bytes
                wreg[t][ks] = v;
            }
    }
    const bool has_bias = p.bias && wc * 32 + (lane & 31) < p.Cout;

    const int S = p.W < 128 ? p.W : 128;
    const int s_shift = p.W < 128 ? p.wo_shift : 7;
    const int S2 = S + 2, segs = 128 >> s_shift, E = 128 + 2 * segs;
    const int row_bytes = p.W * p.ldx * 2;
    int e0[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = wr * 64 + i * 32 + (lane & 31);
        e0[i] = (r >> s_shift) * S2 + (r & (S - 1));
    }
    const T* res = reinterpret_cast<const T*>(p.res);
    const T* dec = reinterpret_cast<const T*>(p.dec);
    const T* shf = reinterpret_cast<const T*>(p.shift);
    float* stage = reinterpret_cast<float*>(smem);

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int m0 = tile * 128;
        const float bv = has_bias ? bias_of(p, m0)[wc * 32 + (lane & 31)] : 0.f;   // (per frame with bias_rows)
        __syncthreads();   // the previous tile's epilogue has left the LDS
        // ---- the three images: piece q = wave + 4 i (i < 5) of image ky = rows 8q + (lane >> 3), chunk lane & 7
#pragma unroll 1
        for (int i = 0; i < 5; ++i) {
            const int q = wave + 4 * i;
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
                bufdma16(v ? (unsigned)(pix + ky * row_bytes) : kOob, rsrc_x, 0, lds0 + ky * kImg + q * 1024);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();

        f32x16 acc[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                int ab[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int e = e0[i] + kx;
                    ab[i] = ky * kImg + e * 128 + ((hh ^ ((e >> 1) & 7)) << 4);
                }
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const uint4 fa = *reinterpret_cast<const uint4*>(smem + (ab[i] ^ (ks << 5)));
                        acc[i] = mma16<T>(fa, wreg[ky * 3 + kx][ks], acc[i]);
                    }
            }
        __syncthreads();   // every wave is done with the images

        // ---- epilogue: acc + bias staged as fp32 (128 rows x 64 channels), then 8 channels of one pixel per thread
        {
            const int cl = wc * 32 + (lane & 31);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int rl = wr * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
                    stage[rl * kSRow + cl] = acc[i][e] + bv;
                }
        }
        __syncthreads();
        if (!p.vec_epi) {   // Cout % 8 != 0 or unaligned operands (the 64 -> 3 output conv): one value per thread and trip
            for (int cidx = tid; cidx < 128 * p.Cout; cidx += 256) {
                const int rl = cidx / p.Cout, n = cidx - rl * p.Cout;
                const int m = m0 + rl;
                if (m >= p.M) continue;
                float v = apply_act(stage[rl * kSRow + n], p.act);
                if (p.epi == 1) {
                    const float d = ldf(dec + (long)m * p.ld_dec + n), sh = ldf(shf + (long)m * p.ld_shift + n);
                    v = d + p.sft_w * (d * v + sh);
                } else {
                    if (res) v += ldf(res + (long)m * p.ldr + n);
                    if (p.post_relu) v = v > 0.f ? v : 0.f;
                }
                if (p.out_f32) reinterpret_cast<float*>(p.y)[(long)m * p.ldy + n] = v;
                else stf(reinterpret_cast<T*>(p.y) + (long)m * p.ldy + n, v);
            }
            continue;
        }
        for (int cidx = tid; cidx < 128 * 8; cidx += 256) {
            const int rl = cidx >> 3, c8 = (cidx & 7) * 8;
            const int m = m0 + rl;
            if (m >= p.M || c8 >= p.Cout) continue;
            float v[8];
            *reinterpret_cast<float4*>(v) = *reinterpret_cast<const float4*>(stage + rl * kSRow + c8);
            *reinterpret_cast<float4*>(v + 4) = *reinterpret_cast<const float4*>(stage + rl * kSRow + c8 + 4);
            apply_act8(v, p.act);
            if (p.epi == 1) {
                float d[8], s[8];
                load8<T>(dec + (long)m * p.ld_dec + c8, d);
                load8<T>(shf + (long)m * p.ld_shift + c8, s);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = d[e] + p.sft_w * (d[e] * v[e] + s[e]);
            } else {
                if (res) {
                    float r[8];
                    load8<T>(res + (long)m * p.ldr + c8, r);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += r[e];
                }
                if (p.post_relu) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
                }
            }
            if (p.out_f32) store8<float>(reinterpret_cast<float*>(p.y) + (long)m * p.ldy + c8, v);
            else store8<T>(reinterpret_cast<T*>(p.y) + (long)m * p.ldy + c8, v);
        }
    }
}

}  // namespace

// See the preconditions at the top of the file; the caller checks them.
int pgt_igemm6_launch(const void* pv, hipStream_t st) {
    ConvP p = *reinterpret_cast<const ConvP*>(pv);
    p.wo_shift = __builtin_ctz(p.Wo);
    p.ho_shift = __builtin_ctz(p.Ho);
    const int ntiles = (p.M + 127) / 128;
    static int n_cu = 0;
    if (n_cu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
            pgt_set_error("igemm6: cannot query the device");
            return -5;
        }
        n_cu = prop.multiProcessorCount;
    }
    const int grid = ntiles < 2 * n_cu ? ntiles : 2 * n_cu;
    if (p.f16) hipLaunchKernelGGL(conv3x3_c64_kernel<half_t>, dim3(grid), dim3(256), 0, st, p, ntiles);
    else hipLaunchKernelGGL(conv3x3_c64_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, p, ntiles);
    PGT_LAUNCH_CHECK();
    return 0;
}
