// Shared device helpers for the PGTFormer gfx950 kernels.  CDNA4 only: 64-wide wavefronts, MFMA.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct bf16_t {
    uint16_t v;
};

__device__ __forceinline__ float bf2f(uint16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// round-to-nearest-even fp32 -> bf16: the __bf16 casts lower to gfx950's v_cvt_pk_bf16_f32
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint16_t f2bf(float f) {
    const __bf16 b = (__bf16)f;
    return __builtin_bit_cast(uint16_t, b);
}
__device__ __forceinline__ uint32_t f2bf2(float lo, float hi) {   // {lo, hi} packed, one instruction
    const bf16x2_t v = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(uint32_t, v);
}

template <typename T> struct ElemIO;
template <> struct ElemIO<float> {
    static __device__ __forceinline__ float ld(const float* p) { return *p; }
    static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct ElemIO<bf16_t> {
    static __device__ __forceinline__ float ld(const bf16_t* p) { return bf2f(p->v); }
    static __device__ __forceinline__ void st(bf16_t* p, float v) { p->v = f2bf(v); }
};
// IEEE half storage (PGT_F16): the decoder-side activation / weight type of the default precision mode (11 significand
// bits on the same 16-bit MFMA rate as bf16).  fp32 -> half conversions SATURATE at +-65504 instead of producing inf.
struct half_t { _Float16 v; };
typedef _Float16 halfx2 __attribute__((ext_vector_type(2)));
typedef _Float16 halfx8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ float sat_half(float f) { return __builtin_amdgcn_fmed3f(f, -65504.f, 65504.f); }   // one v_med3_f32
__device__ __forceinline__ uint32_t f2h2(float lo, float hi) {   // {lo, hi} packed, round to nearest even, saturating
    const halfx2 v = {(_Float16)sat_half(lo), (_Float16)sat_half(hi)};
    return __builtin_bit_cast(uint32_t, v);
}
template <> struct ElemIO<half_t> {
    static __device__ __forceinline__ float ld(const half_t* p) { return (float)p->v; }
    static __device__ __forceinline__ void st(half_t* p, float v) { p->v = (_Float16)sat_half(v); }
};
template <typename T> __device__ __forceinline__ float ldf(const T* p) { return ElemIO<T>::ld(p); }
template <typename T> __device__ __forceinline__ void stf(T* p, float v) { ElemIO<T>::st(p, v); }

// 16-byte vector of T unpacked to floats (8 bf16 or 4 f32) and back
template <typename T> struct Vec16;
template <> struct Vec16<float> {
    static constexpr int N = 4;
    static __device__ __forceinline__ void unpack(const uint4& q, float* f) {
        f[0] = __uint_as_float(q.x); f[1] = __uint_as_float(q.y);
        f[2] = __uint_as_float(q.z); f[3] = __uint_as_float(q.w);
    }
    static __device__ __forceinline__ uint4 pack(const float* f) {
        return make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3]));
    }
};
template <> struct Vec16<bf16_t> {
    static constexpr int N = 8;
    static __device__ __forceinline__ void unpack(const uint4& q, float* f) {
        f[0] = __uint_as_float(q.x << 16); f[1] = __uint_as_float(q.x & 0xffff0000u);
        f[2] = __uint_as_float(q.y << 16); f[3] = __uint_as_float(q.y & 0xffff0000u);
        f[4] = __uint_as_float(q.z << 16); f[5] = __uint_as_float(q.z & 0xffff0000u);
        f[6] = __uint_as_float(q.w << 16); f[7] = __uint_as_float(q.w & 0xffff0000u);
    }
    static __device__ __forceinline__ uint4 pack(const float* f) {
        return make_uint4(f2bf2(f[0], f[1]), f2bf2(f[2], f[3]), f2bf2(f[4], f[5]), f2bf2(f[6], f[7]));
    }
};

template <> struct Vec16<half_t> {
    static constexpr int N = 8;
    static __device__ __forceinline__ void unpack(const uint4& q, float* f) {
        const halfx2 a = __builtin_bit_cast(halfx2, q.x), b = __builtin_bit_cast(halfx2, q.y);
        const halfx2 c = __builtin_bit_cast(halfx2, q.z), d = __builtin_bit_cast(halfx2, q.w);
        f[0] = (float)a.x; f[1] = (float)a.y; f[2] = (float)b.x; f[3] = (float)b.y;
        f[4] = (float)c.x; f[5] = (float)c.y; f[6] = (float)d.x; f[7] = (float)d.y;
    }
    static __device__ __forceinline__ uint4 pack(const float* f) {
        return make_uint4(f2h2(f[0], f[1]), f2h2(f[2], f[3]), f2h2(f[4], f[5]), f2h2(f[6], f[7]));
    }
};

// 16-bit MFMA, 32x32x16, fp32 accumulate: bf16 or IEEE half operands (8 elements of K per lane, identical layouts)
template <typename T> __device__ __forceinline__ f32x16 mma16(const uint4& a, const uint4& b, const f32x16& c);
template <> __device__ __forceinline__ f32x16 mma16<bf16_t>(const uint4& a, const uint4& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
template <> __device__ __forceinline__ f32x16 mma16<half_t>(const uint4& a, const uint4& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(halfx8, a), __builtin_bit_cast(halfx8, b), c, 0, 0, 0);
}

// Split ("x3") storage: value = hi + lo on two 16-bit planes, hi = rn16(v), lo = rn16(v - hi).  A product of two split
// numbers is taken as hi*hi + lo*hi + hi*lo on the 16-bit MFMA (the lo*lo term is below the planes' resolution), accumulated
// in fp32.  The plane type is IEEE half (x3p_t): 22 significand bits at the same three MFMAs per product that two bf16 planes
// (16 bits, rounds 2-3a) take - the logit error of the code-prediction branch drops 5x, to the level of fp32 summation-order
// noise (profiles/r3_psnr_sweep.md section 3).  Range: hi saturates at +-65504 (no inf); below 2^-3 the lo plane is subnormal
// and the resolution is absolute (6e-8) instead of relative - the f16 MFMA keeps subnormal operands (measured).
using x3p_t = half_t;
// The lo plane must be taken against the hi bits that are STORED: the compiler is free to form "the half of f" twice in
// two ways (v_cvt_pk_f16_f32 of the fp32 value for the packed store, v_fma_mixlo_f16 fused with the producing multiply for
// the subtraction) which round differently at near-ties - seen in the window-attention epilogue, hi + lo off by one half
// ulp on 2 of 73 728 outputs.  x3_opaque() hides the packed register from that folding (no instruction is emitted).
__device__ __forceinline__ void x3_opaque(uint32_t& w) { asm volatile("" : "+v"(w)); }
__device__ __forceinline__ void x3_opaque(uint4& q) { asm volatile("" : "+v"(q.x), "+v"(q.y), "+v"(q.z), "+v"(q.w)); }
// {lo, hi} packed to half, round to nearest even, NOT saturating (one v_cvt_pk_f16_f32): for values known to fit
__device__ __forceinline__ uint32_t f2h2_nosat(float lo, float hi) {
    const halfx2 v = {(_Float16)lo, (_Float16)hi};
    return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ void split8(const float* f, uint4& hi, uint4& lo) {
    float c[8], h[8], r[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) c[e] = sat_half(f[e]);       // one clamp per value: hi cannot overflow, |c - hi| <= ulp / 2
    hi = make_uint4(f2h2_nosat(c[0], c[1]), f2h2_nosat(c[2], c[3]), f2h2_nosat(c[4], c[5]), f2h2_nosat(c[6], c[7]));
    x3_opaque(hi);
    Vec16<x3p_t>::unpack(hi, h);
#pragma unroll
    for (int e = 0; e < 8; ++e) r[e] = c[e] - h[e];
    lo = make_uint4(f2h2_nosat(r[0], r[1]), f2h2_nosat(r[2], r[3]), f2h2_nosat(r[4], r[5]), f2h2_nosat(r[6], r[7]));
}
__device__ __forceinline__ void merge8(const uint4& hi, const uint4& lo, float* f) {
    float l[8];
    Vec16<x3p_t>::unpack(hi, f);
    Vec16<x3p_t>::unpack(lo, l);
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] += l[e];
}
// two values -> one dword of each plane ({a, b} packed)
__device__ __forceinline__ void split2(float a, float b, uint32_t& hi, uint32_t& lo) {
    a = sat_half(a);
    b = sat_half(b);
    hi = f2h2_nosat(a, b);
    x3_opaque(hi);
    const halfx2 h = __builtin_bit_cast(halfx2, hi);
    lo = f2h2_nosat(a - (float)h.x, b - (float)h.y);
}
__device__ __forceinline__ float x3_hi_of(float f) {      // value of the hi plane of f (opaque: see x3_opaque)
    uint32_t w = (uint32_t)__builtin_bit_cast(uint16_t, (_Float16)sat_half(f));
    x3_opaque(w);
    return (float)__builtin_bit_cast(_Float16, (uint16_t)w);
}

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_GELU = 2, ACT_SILU = 3, ACT_LEAKY02 = 4, ACT_SIGMOID = 5 };

__device__ __forceinline__ float apply_act(float v, int act) {
    switch (act) {
        case ACT_RELU: return v > 0.f ? v : 0.f;
        case ACT_GELU: return 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));  // exact erf GELU
        case ACT_SILU: return v / (1.f + expf(-v));
        case ACT_LEAKY02: return v > 0.f ? v : 0.2f * v;
        case ACT_SIGMOID: return 1.f / (1.f + expf(-v));
        default: return v;
    }
}

// Same on 8 values with the (uniform) switch OUTSIDE the element loop: a per-element switch in a fully unrolled
// accumulator epilogue multiplies the code size by the number of cases (I-cache misses dominate the epilogue).
__device__ __forceinline__ void apply_act8(float* v, int act) {
    switch (act) {
        case ACT_NONE: break;
        case ACT_RELU:
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
            break;
        case ACT_GELU:
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = 0.5f * v[e] * (1.f + erff(v[e] * 0.70710678118654752440f));
            break;
        case ACT_SILU:
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = v[e] / (1.f + expf(-v[e]));
            break;
        case ACT_LEAKY02:
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = v[e] > 0.f ? v[e] : 0.2f * v[e];
            break;
        case ACT_SIGMOID:
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = 1.f / (1.f + expf(-v[e]));
            break;
        default: break;
    }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// error reporting shared by the C-ABI translation units
void pgt_set_error(const char* fmt, ...);
#define PGT_CHECK(cond, ...)            \
    do {                                \
        if (!(cond)) {                  \
            pgt_set_error(__VA_ARGS__); \
            return -22;                 \
        }                               \
    } while (0)
#define PGT_LAUNCH_CHECK()                                                  \
    do {                                                                    \
        hipError_t e_ = hipGetLastError();                                  \
        if (e_ != hipSuccess) {                                             \
            pgt_set_error("HIP launch failed: %s", hipGetErrorString(e_)); \
            return -5;                                                      \
        }                                                                   \
    } while (0)
