// Shared between gemm.hip (fp32 MFMA) and gemm_split.hip (bf16 / bf16x3 MFMA).
#pragma once
#include <hip/hip_runtime.h>

namespace t2h {

enum : int { F_RELU_A = 1, F_RELU_B = 2, F_RELU_OUT = 4, F_ACCUM = 8 };

struct GemmArgs {
    const float *A, *B;
    float *C;
    const float *bias;     // [N] or null
    const float *mask;     // [M, ldm] or null: result *= (mask > 0)
    const float *addend;   // [M, ldadd] or null: added to the finished result (after ReLU / mask), fp32 kernels only
    int ldadd;
    float *colsum;         // TN only: per-split column sums of A (i.e. sum over k of A(m,k)), [splits][M] or null
    int M, N, K;
    int lda, ldb, ldc, ldm;
    int flags;
    int k_chunk;           // reduction range per split
    long long slab_stride; // C offset per split (floats)
};

// gemm_split.hip: 128x128 tiles, mode 1 = bf16, 2 = bf16x3
int launch_gemm_split(int mode, bool a_kc, bool b_kc, const GemmArgs &a, int splits, hipStream_t s, const char *what);

// gemm_dma.hip: LDS-DMA staged 128x128x16 fp32 kernels for the big row-streaming layers (NT: b_kc, NN: !b_kc)
bool gemm_dma_applicable(bool b_kc, const GemmArgs &a);
int launch_gemm_dma(bool b_kc, const GemmArgs &a, hipStream_t s, const char *what);
bool gemm_dma_tn_applicable(const GemmArgs &a);
int launch_gemm_dma_tn(const GemmArgs &a, int splits, hipStream_t s, const char *what);

}  // namespace t2h
