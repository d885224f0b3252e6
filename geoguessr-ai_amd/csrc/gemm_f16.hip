// fp16 MFMA GEMM  C[M,N] = epilogue(A[M,K] * B[N,K]^T): the fp16 twin of gg_gemm_nt for the CLIP tower's fp16 mode (BASELINE config c4:
// "CLIP ViT-B/32 embedder ... MFMA fp16").  Same kernel text as the bf16 GEMM -- gemm.hip is compiled a second time here with the 16-bit
// operand type and the MFMA instruction swapped (v_mfma_f32_16x16x32_f16; fp32 accumulation either way): tiling, LDS swizzle, buffer-load
// staging and the epilogue classes are type-independent (16-byte chunks of 8 elements).  Everything lives in namespace gg_f16, so the two
// builds' kernels and parameter structs are distinct symbols; only gg_gemm_nt_f16 is exported.  The BatchNorm-fused, two-source and split-K
// forms (TinyViT training only) are refused by this build: their staging code manipulates bf16 bit patterns.
#include "common.h"
#include <stdlib.h>
#include "../../include/gg.h"

typedef _Float16 gg_f16_t;
typedef gg_f16_t gg_f16x8_t __attribute__((ext_vector_type(8)));
typedef gg_f16_t gg_f16x4_t __attribute__((ext_vector_type(4)));
typedef gg_f16_t gg_f16x2_t __attribute__((ext_vector_type(2)));
#define GG_GEMM_SECOND_TYPE 1
#define GG_GEMM_NT_NAME gg_gemm_nt_f16
#define bf16 gg_f16_t
#define bf16x8 gg_f16x8_t
#define bf16x4 gg_f16x4_t
#define bf16x2 gg_f16x2_t
#define __builtin_amdgcn_mfma_f32_16x16x32_bf16 __builtin_amdgcn_mfma_f32_16x16x32_f16
namespace gg_f16 {
#include "gemm.hip"
}  // namespace gg_f16
