// Shared between the implicit-GEMM kernels (igemm.hip: register-staged v1; igemm2.hip: LDS-DMA v2).
#pragma once
#include "common.h"

namespace {

// 8 consecutive elements <-> floats with 16-byte accesses
template <typename T> __device__ __forceinline__ void load8(const T* p, float* f);
template <> __device__ __forceinline__ void load8<bf16_t>(const bf16_t* p, float* f) {
    Vec16<bf16_t>::unpack(*reinterpret_cast<const uint4*>(p), f);
}
template <> __device__ __forceinline__ void load8<half_t>(const half_t* p, float* f) {
    Vec16<half_t>::unpack(*reinterpret_cast<const uint4*>(p), f);
}
template <> __device__ __forceinline__ void load8<float>(const float* p, float* f) {
    *reinterpret_cast<float4*>(f) = *reinterpret_cast<const float4*>(p);
    *reinterpret_cast<float4*>(f + 4) = *reinterpret_cast<const float4*>(p + 4);
}
template <typename T> __device__ __forceinline__ void store8(T* p, const float* f);
template <> __device__ __forceinline__ void store8<bf16_t>(bf16_t* p, const float* f) {
    *reinterpret_cast<uint4*>(p) = Vec16<bf16_t>::pack(f);
}
template <> __device__ __forceinline__ void store8<half_t>(half_t* p, const float* f) {
    *reinterpret_cast<uint4*>(p) = Vec16<half_t>::pack(f);
}
template <> __device__ __forceinline__ void store8<float>(float* p, const float* f) {
    *reinterpret_cast<float4*>(p) = *reinterpret_cast<const float4*>(f);
    *reinterpret_cast<float4*>(p + 4) = *reinterpret_cast<const float4*>(f + 4);
}

struct ConvP {
    const char* x;
    const char* w;
    const float* bias;
    const char* res;
    const char* dec;
    const char* shift;
    char* y;
    int N, H, W, Cin, ldx, ups, KH, KW, stride, pad_t, pad_l, Ho, Wo, Cout, ldy;
    int act, post_relu, ldr, epi, ld_dec, ld_shift, out_f32, vec_epi;
    float sft_w;
    int M, K, nbm, nbn;
    int splitk, kt_per_split;   // > 1: K is cut in `splitk` slices of `kt_per_split` K tiles each
    int wo_shift, ho_shift;     // igemm4: log2(Wo), log2(Ho) when both are powers of two, else -1 (set by its launcher)
    int orow_mul, orow_xmul, orow_off;   // output row of pixel m (orow_mul = 0: m), see pgt_conv_desc
    int x3;                     // split-half operands: x = [hi | lo] planes, K runs over [x_hi | x_lo | x_hi] per tap
    int xlo, ylo, rlo;          // element offset of the lo plane inside a pixel row of x / y / residual
    int res_f32;                // x3 with fp32 output: the residual is fp32 too (ldr counts floats)
    int f16;                    // 16-bit operands are IEEE half (PGT_F16) instead of bf16
    int dlo, slo;               // x3 with the SFT epilogue: element offsets of the lo planes of dec / shift
    int bias_rows;              // > 0: bias is a (M / bias_rows, Cout) matrix - one vector per bias_rows consecutive output pixels (a frame)
    int out_split;              // fp32 kernel: y is stored as split-half planes [hi | lo] (lo plane ylo elements further)
    int nw;                     // rows of the weight matrix = GEMM columns (Cout; 128 for the folded 64-channel x3 form, x3 == 2)
    // GroupNorm statistics of the OUTPUT from the epilogue (gn_part != nullptr): every workgroup tile writes the sum and
    // sum of squares of its outputs per channel group to gn_part[((img * gn_maxblk + k) * gn_G + g) * 2 + {0, 1}], k = the
    // tile's row block inside the image (gn_hw output pixels per image in this launch, a multiple of the tile rows);
    // thread 0 of workgroup 0 records the tile rows in *gn_hdr for the finalising kernel (norms.hip).
    float* gn_part;
    float* gn_hdr;
    int gn_cpg, gn_G, gn_maxblk, gn_hw;
    // the operand is in_act(x * in_scale[n][c] + in_shift[n][c]) (fp32 (N, Cin) each): the GroupNorm apply + SiLU of the
    // Normalize that precedes the conv, fused into the operand load (pgt_conv2d_affine_in; igemm8.hip only)
    const float* in_scale;
    const float* in_shift;
    int in_act;
    int w2;                     // exact-weight form (pgt_conv_desc::w2): w = per 32 output channels [32 rows w_hi | 32 rows w_lo * 2048], nw rows
};

// exact-weight layers: y = acc_hi + acc_lo * kW2Inv (the lo plane is stored scaled by 2048 = 2^11: |w - w_hi| <= 2^-11 |w|)
constexpr float kW2Inv = 1.f / 2048.f;

// bias vector of output pixel m: shared, or the one of m's frame (pgt_conv_desc::bias_rows; a workgroup tile never straddles
// two frames: bias_rows is a multiple of 512 rows)
__device__ __forceinline__ const float* bias_of(const ConvP& p, int m) {
    return p.bias_rows ? p.bias + (long)(m / p.bias_rows) * p.Cout : p.bias;
}

// Cross-thread part of the epilogue statistics.  Every thread holds s[0..7] / s[8..15] = sum / sum of squares of the 8
// channels of ITS chunk column over the tile rows it finished; threads whose (tid % NCOL) agree share a column.
// `red`: >= NWAVES * NCOL * 16 floats of LDS nobody else uses any more (the epilogue stage after a barrier).
template <int NCOL, int NWAVES>
__device__ __forceinline__ void gn_tile_reduce(const ConvP& p, float (&s)[16], float* red, int tid, int m0, int n0, int bm) {
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int off = 32; off >= NCOL; off >>= 1)
#pragma unroll
        for (int e = 0; e < 16; ++e) s[e] += __shfl_xor(s[e], off, 64);
    __syncthreads();
    if (lane < NCOL) {
#pragma unroll
        for (int e = 0; e < 16; ++e) red[(wave * NCOL + lane) * 16 + e] = s[e];
    }
    __syncthreads();
    const int cpg = p.gn_cpg;
    const int ngl = (NCOL * 8) / cpg;                      // channel groups of this tile
    if (tid < ngl) {
        const int c0 = tid * cpg;
        float a = 0.f, b = 0.f;
        for (int w = 0; w < NWAVES; ++w)                   // fixed order: deterministic
            for (int c = c0; c < c0 + cpg; ++c) {
                const float* r = red + (w * NCOL + (c >> 3)) * 16 + (c & 7);
                a += r[0];
                b += r[8];
            }
        if (n0 + c0 < p.Cout) {
            const int img = m0 / p.gn_hw;
            const int k = (m0 - img * p.gn_hw) / bm;
            float* o = p.gn_part + (((long)img * p.gn_maxblk + k) * p.gn_G + (n0 + c0) / cpg) * 2;
            o[0] = a;
            o[1] = b;
        }
    }
    if (tid == 0 && blockIdx.x == 0) *p.gn_hdr = (float)bm;
}

// row index of output pixel m in y (dense, or the sub-pixel placement of pgt_conv_desc::orow_*)
__device__ __forceinline__ long out_row(const ConvP& p, int m) {
    return p.orow_mul ? (long)p.orow_mul * m + p.orow_xmul * (m % p.Wo) + p.orow_off : (long)m;
}

}  // namespace

// igemm2.hip: bf16, Cin % 64 == 0, 16-byte epilogue legal.  Returns 0 on success.


namespace {
// LDS-DMA of 16 bytes per lane (1 KiB per wave) as ONE opaque statement: dst = wave-uniform LDS byte address (lane l
// lands at dst + 16 l), src = the lane's global pointer.  Issued through inline asm on purpose: hipcc does not model
// the transfer, so it neither drains it (s_waitcnt vmcnt(0)) ahead of the next ds_read nor at barriers; completion is
// the caller's counted `s_waitcnt vmcnt(N)` followed by a workgroup barrier.  M0 is saved/restored inside the string.
__device__ __forceinline__ void glds16(const void* src, unsigned dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
}
// The same through a buffer descriptor: source = rsrc.base + voff (per lane) + soff (wave-uniform); a lane whose
// voff is outside [0, num_records) transfers zeros (used for padding and ragged edges).
typedef int v4i __attribute__((ext_vector_type(4)));
__device__ __forceinline__ v4i make_rsrc(const void* base, unsigned bytes) {
    const unsigned long b = (unsigned long)base;
    v4i r;
    r.x = (int)(unsigned)b;
    r.y = (int)(unsigned)((b >> 32) & 0xffffu);   // stride 0: raw buffer, range-checked in bytes
    r.z = (int)bytes;
    r.w = 0x00020000;
    return r;
}
__device__ __forceinline__ void bufdma16(unsigned voff, v4i rsrc, int soff, unsigned dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(soff), "s"(dst) : "memory");
}
__device__ __forceinline__ unsigned lds_addr(const void* p) {
    return (unsigned)(unsigned long)(__attribute__((address_space(3))) const void*)p;
}

// Byte offset of (row, 16-byte chunk c) inside an UNPADDED tile of 128-byte rows, XOR-swizzled so that
// ds_read_b128 fragment reads (32 rows, same chunk) and 16-byte staging writes are bank-conflict free:
// a 256-byte super row holds two tile rows (16 slots); slot' = slot ^ (superrow & 15).
__device__ __forceinline__ int swz128(int row, int c) {
    const int sr = row >> 1;
    return sr * 256 + (((((row & 1) << 3) | c) ^ (sr & 15)) << 4);
}
}  // namespace
