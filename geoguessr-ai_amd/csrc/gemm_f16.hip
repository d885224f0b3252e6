// fp16 MFMA GEMM  C[M,N] = epilogue(A[M,K] * B[N,K]^T): the fp16 twin of gg_gemm_nt for the CLIP tower's fp16 mode (BASELINE config c4:
// "CLIP ViT-B/32 embedder ... MFMA fp16").  gemm.hip names its element type, vector types and MFMA instruction in one place (ge_t / ge8_t / ge_mfma); this
// translation unit compiles it a second time with GG_GEMM_ELEM_F16 (fp16 elements, v_mfma_f32_16x16x32_f16; fp32 accumulation either way) inside namespace
// gg_f16, so the two builds' kernels and parameter structs are distinct symbols and only gg_gemm_nt_f16 is exported.  The BatchNorm-fused, two-source and
// split-K forms (TinyViT training only) are refused by this build: their staging code manipulates bf16 bit patterns.
#include "common.h"
#include <stdlib.h>
#include <type_traits>
#include <algorithm>
#include "../../include/gg.h"

#define GG_GEMM_ELEM_F16 1
#define GG_GEMM_SECOND_TYPE 1
#define GG_GEMM_NT_NAME gg_gemm_nt_f16
namespace gg_f16 {
#include "gemm.hip"
}  // namespace gg_f16
