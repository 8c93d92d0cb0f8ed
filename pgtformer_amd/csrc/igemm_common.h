// Shared between the implicit-GEMM kernels (igemm.hip: register-staged v1; igemm2.hip: LDS-DMA v2).
#pragma once
#include "common.h"

namespace {

// 8 consecutive elements <-> floats with 16-byte accesses
template <typename T> __device__ __forceinline__ void load8(const T* p, float* f);
template <> __device__ __forceinline__ void load8<bf16_t>(const bf16_t* p, float* f) {
    Vec16<bf16_t>::unpack(*reinterpret_cast<const uint4*>(p), f);
}
template <> __device__ __forceinline__ void load8<float>(const float* p, float* f) {
    *reinterpret_cast<float4*>(f) = *reinterpret_cast<const float4*>(p);
    *reinterpret_cast<float4*>(f + 4) = *reinterpret_cast<const float4*>(p + 4);
}
template <typename T> __device__ __forceinline__ void store8(T* p, const float* f);
template <> __device__ __forceinline__ void store8<bf16_t>(bf16_t* p, const float* f) {
    *reinterpret_cast<uint4*>(p) = Vec16<bf16_t>::pack(f);
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
};

}  // namespace

// igemm2.hip: bf16, Cin % 64 == 0, 16-byte epilogue legal.  Returns 0 on success.


namespace {
// Byte offset of (row, 16-byte chunk c) inside an UNPADDED tile of 128-byte rows, XOR-swizzled so that
// ds_read_b128 fragment reads (32 rows, same chunk) and 16-byte staging writes are bank-conflict free:
// a 256-byte super row holds two tile rows (16 slots); slot' = slot ^ (superrow & 15).
__device__ __forceinline__ int swz128(int row, int c) {
    const int sr = row >> 1;
    return sr * 256 + (((((row & 1) << 3) | c) ^ (sr & 15)) << 4);
}
}  // namespace
