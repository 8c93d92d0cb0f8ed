// Internal umbrella: public C-ABI + helpers shared by the translation units of libpgt_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/pgt_hip.h"

// attention_mfma.hip: bf16 MFMA flash attention (hd = 64)
int pgt_mha_mfma_bf16(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, void* out, int ldo, int B,
                      int L, int heads, float scale, hipStream_t st, int x3 = 0, int qlo = 0, int klo = 0, int vlo = 0,
                      int olo = 0);

// igemm2.hip: LDS-DMA implicit GEMM (bf16, Cin % 64 == 0, 16-byte epilogue legal). `conv_p` is a ConvP.
int pgt_igemm2_launch(const void* conv_p, int bn, int stages, hipStream_t st);

// window_attn_mfma.hip: MFMA window attention, mode 0 bf16 / 1 split-half / 2 fp16; (wd, wh, ww) windows with shift
// (sd, sh, sw) on a (B, D, H, W) token grid; returns 1 if the shape is not covered
int pgt_window_attn_mfma(int mode, const void* qkv, int ldqkv, void* out, int ldo, const float* bias, int B, int D, int H,
                         int W, int C, int heads, int wd, int wh, int ww, int sd, int sh, int sw, hipStream_t st,
                         int qlo = 0, int olo = 0);

// igemm4.hip: phase-interleaved 8-wave schedule; bn = 256 -> 256x256 tiles, bn = 128 -> 512x128 tiles; 1 = tile not built
int pgt_igemm4_launch(const void* conv_p, int bn, hipStream_t st);
// igemm5.hip: igemm4's 256x256 schedule with one LDS input image shared by the three horizontal taps (3-wide filters)
int pgt_igemm5_launch(const void* conv_p, hipStream_t st);
// igemm6.hip: 3x3, Cin == 64, Cout <= 64: weights in registers, persistent workgroups, one halo image per filter row
int pgt_igemm6_launch(const void* conv_p, hipStream_t st);
// igemm6x3.hip: the same layers on split-half operands (Cout % 16 == 0): hi / lo weights in registers, MFMA 16x16x32, three products
int pgt_igemm6x3_launch(const void* conv_p, hipStream_t st);
// igemm7.hip: streaming linear for K = 256 on many rows (weights in registers, rows through LDS, two workgroups per CU)
int pgt_igemm7_launch(const void* conv_params, hipStream_t st);
// igemm8.hip: the same layers as igemm6 for W >= 128: a ring of row images in LDS (every input row staged once), weights as the
// MFMA A operand, register epilogue; optionally the preceding GroupNorm apply + activation fused into the operand (ConvP::in_scale)
int pgt_igemm8_launch(const void* conv_p, hipStream_t st);
