// LDS-DMA staged variant of the fp32 MFMA forward GEMM (nn.Linear forward, the "NT" form of gemm.hip: Y = X W^T) for the
// big layers: 128 x 128 x 16 tiles, K a multiple of 16.
//
// Both operands are k-contiguous rows (X [M, K], W [N, K]) and go global -> LDS with global_load_lds_dwordx4, no VGPR
// staging and no ds_write: the VGPR -> LDS store path (~64 B/clk/CU) was the largest single cost of the register-staged
// loop (ablation in DESIGN.md section 4).  An LDS-DMA writes 64 lanes x 16 B contiguously, so the image is lane-linear
// [row][16 floats] without padding; bank conflicts are avoided by an XOR swizzle applied through the per-lane SOURCE
// address (chunk c of row r lands in slot c ^ ((r >> 2) & 3)), and fragments are read with ds_read_b128.  The MFMA
// k-pair index is free as long as A and B agree, so lane half h takes k = 4 (2g + h) .. + 3 in group g in {0, 1}: one
// 16-byte read serves four consecutive MFMAs.  Rows past M / N are clamped to the last valid row (their results are never
// stored).  Numerics: still an exact fp32 fma chain per output, k visited in the permuted order inside each 16-slab.
// Measured (N = 131072 points): 1024 -> 512 forward 1.105 -> 1.05 ms (131 TF = 83 % of the matrix peak), 256 -> 512
// 307 -> 296 us.  The data gradient ("NN": dY rows as above, W [k][n] as a k-major DMA image read by ds_read_b32 in the
// same k order) needs every fragment read of a slab issued before its MFMAs
// (interleaving the W reads with the MFMAs, or W through registers, lost 2-8 %).
#include "t2h_common.h"
#include "gemm_args.h"
#include "gemm_tile.h"

namespace t2h {
namespace {

constexpr int BM = 128, BN = 128, BK = 16, NT = 256, TM = 2, TN = 2;
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

__global__ __launch_bounds__(NT, 4) void gemm_dma_kernel(GemmArgs p) {
    constexpr int IMG = BM * BK;                                     // floats per operand image
    constexpr int STG = 2 * IMG;
    constexpr int LDS_MIN = (NT / 64) * 32 * 36;
    constexpr int LDS_FLOATS = 2 * STG > LDS_MIN ? 2 * STG : LDS_MIN;
    __shared__ __attribute__((aligned(1024))) float lds[LDS_FLOATS];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    // XCD-aware work order (gemm.hip)
    const unsigned nb = gridDim.x * gridDim.y;
    const unsigned bid = blockIdx.y * gridDim.x + blockIdx.x;
    const unsigned q = nb / 8, r = nb % 8, x = bid % 8, i8 = bid / 8;
    const unsigned t = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i8;
    const int tile_n = t % gridDim.x, tile_m = t / gridDim.x;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const bool relu_a = p.flags & F_RELU_A;
    const int nk = p.K / BK;

    // DMA source of this lane: two 16-row blocks per wave and operand, chunk swizzled by the row
    const float *src_a[2], *src_b[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = (wave * 2 + j) * 16 + (lane >> 2), c = (lane & 3) ^ ((row >> 2) & 3);
        src_a[j] = p.A + (size_t)min(m0 + row, p.M - 1) * p.lda + c * 4;
        src_b[j] = p.B + (size_t)min(n0 + row, p.N - 1) * p.ldb + c * 4;
    }
    auto issue = [&](int kt, float *stage) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            __builtin_amdgcn_global_load_lds((glb_void *)(src_a[j] + kt * BK), (lds_void *)(stage + (wave * 2 + j) * 256), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((glb_void *)(src_b[j] + kt * BK), (lds_void *)(stage + IMG + (wave * 2 + j) * 256), 16, 0, 0);
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int z = 0; z < 16; ++z) acc[i][j][z] = 0.0f;

    if (nk > 0) issue(0, lds);
    float4 bias_pre[TN];
    load_bias_fragments<TN>(p.bias, p.N, n0 + wn * (TN * 32), lane, bias_pre);
    __syncthreads();
    const int half = lane >> 5, m = lane & 31, sw = (m >> 2) & 3;
    for (int kt = 0; kt < nk; ++kt) {
        float *cur = lds + (kt & 1) * STG, *nxt = lds + ((kt & 1) ^ 1) * STG;
        if (kt + 1 < nk) issue(kt + 1, nxt);
        const float *ar = cur + (wm * 64 + m) * BK, *br = cur + IMG + (wn * 64 + m) * BK;
        float4 a4[2][TM], b4[2][TN];
#pragma unroll
        for (int g = 0; g < 2; ++g) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                float4 v = *reinterpret_cast<const float4 *>(ar + i * 32 * BK + (((2 * g + half) ^ sw) * 4));
                if (relu_a) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                a4[g][i] = v;
            }
#pragma unroll
            for (int j = 0; j < TN; ++j)
                b4[g][j] = *reinterpret_cast<const float4 *>(br + j * 32 * BK + (((2 * g + half) ^ sw) * 4));
        }
        __builtin_amdgcn_sched_barrier(0);          // all eight fragment reads in flight before the first MFMA
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const float av = e == 0 ? a4[g][i].x : e == 1 ? a4[g][i].y : e == 2 ? a4[g][i].z : a4[g][i].w;
                        const float bv = e == 0 ? b4[g][j].x : e == 1 ? b4[g][j].y : e == 2 ? b4[g][j].z : b4[g][j].w;
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][j], 0, 0, 0);
                    }
        __syncthreads();      // also drains the LDS-DMA of the next stage (vmcnt(0))
    }

    EpilogueArgs e;
    e.C = p.C; e.bias = p.bias; e.mask = p.mask; e.M = p.M; e.N = p.N; e.ldc = p.ldc; e.ldm = p.ldm;
    e.accum = p.flags & F_ACCUM; e.relu_out = p.flags & F_RELU_OUT; e.addend = p.addend; e.ldadd = p.ldadd;
    store_tiles_f32<TM, TN>(acc, lds + wave * (32 * 36), lane, m0 + wm * (TM * 32), n0 + wn * (TN * 32), e, bias_pre);
}

// Data gradient ("NN"): A = dY rows (row image as above), B(k, n) = W[k][n] k-major: DMA image [16][128] (two k rows per
// wave instruction), fragments by ds_read_b32 in the same permuted k order, all reads of a slab issued before its MFMAs.
__global__ __launch_bounds__(NT, 4) void gemm_dma_nn_kernel(GemmArgs p) {
    constexpr int IMG = BM * BK;
    constexpr int STG = 2 * IMG;
    constexpr int LDS_MIN = (NT / 64) * 32 * 36;
    constexpr int LDS_FLOATS = 2 * STG > LDS_MIN ? 2 * STG : LDS_MIN;
    __shared__ __attribute__((aligned(1024))) float lds[LDS_FLOATS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const unsigned nb = gridDim.x * gridDim.y;
    const unsigned bid = blockIdx.y * gridDim.x + blockIdx.x;
    const unsigned q = nb / 8, r = nb % 8, x = bid % 8, i8 = bid / 8;
    const unsigned t = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i8;
    const int tile_n = t % gridDim.x, tile_m = t / gridDim.x;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int nk = p.K / BK;
    const float *src_a[2], *src_b[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = (wave * 2 + j) * 16 + (lane >> 2), c = (lane & 3) ^ ((row >> 2) & 3);
        src_a[j] = p.A + (size_t)min(m0 + row, p.M - 1) * p.lda + c * 4;
        src_b[j] = p.B + (size_t)((wave * 2 + j) * 2 + (lane >> 5)) * p.ldb + min(n0 + (lane & 31) * 4, p.N - 4);
    }
    auto issue = [&](int kt, float *stage) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            __builtin_amdgcn_global_load_lds((glb_void *)(src_a[j] + kt * BK), (lds_void *)(stage + (wave * 2 + j) * 256), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((glb_void *)(src_b[j] + (size_t)kt * BK * p.ldb),
                                             (lds_void *)(stage + IMG + (wave * 2 + j) * 256), 16, 0, 0);
        }
    };
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int z = 0; z < 16; ++z) acc[i][j][z] = 0.0f;
    if (nk > 0) issue(0, lds);
    __syncthreads();
    const int half = lane >> 5, m = lane & 31, sw = (m >> 2) & 3;
    for (int kt = 0; kt < nk; ++kt) {
        float *cur = lds + (kt & 1) * STG, *nxt = lds + ((kt & 1) ^ 1) * STG;
        if (kt + 1 < nk) issue(kt + 1, nxt);
        const float *ar = cur + (wm * 64 + m) * BK;
        const float *bb = cur + IMG + (4 * half) * BN + wn * 64 + m;
        float4 a4[2][TM];
        float b[8][TN];
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int i = 0; i < TM; ++i)
                a4[g][i] = *reinterpret_cast<const float4 *>(ar + i * 32 * BK + (((2 * g + half) ^ sw) * 4));
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int j = 0; j < TN; ++j) b[s][j] = bb[(8 * (s >> 2) + (s & 3)) * BN + j * 32];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const int g = s >> 2, e = s & 3;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const float av = e == 0 ? a4[g][i].x : e == 1 ? a4[g][i].y : e == 2 ? a4[g][i].z : a4[g][i].w;
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b[s][j], acc[i][j], 0, 0, 0);
                }
        }
        __syncthreads();
    }
    EpilogueArgs e;
    e.C = p.C; e.bias = p.bias; e.mask = p.mask; e.M = p.M; e.N = p.N; e.ldc = p.ldc; e.ldm = p.ldm;
    e.accum = p.flags & F_ACCUM; e.relu_out = p.flags & F_RELU_OUT; e.addend = p.addend; e.ldadd = p.ldadd;
    store_tiles_f32<TM, TN>(acc, lds + wave * (32 * 36), lane, m0 + wm * (TM * 32), n0 + wn * (TN * 32), e);
}

// Weight gradient ("TN": dW[n, j] = sum_m dY[m, n] X[m, j], reduction over the rows m split across grid.z): both
// operands are k-major, so both images are [16 k][128] DMA images read by ds_read_b32 in the permuted k order, all 32
// fragment values of a slab requested before its MFMAs.  Rows past the split's end are clamped to valid memory for the
// DMA and zeroed in the A fragments (last slab only); the bias gradient (column sums of dY) is read back from the A
// image by the workgroups of the first column tile.
__global__ __launch_bounds__(NT, 4) void gemm_dma_tn_kernel(GemmArgs p) {
    constexpr int IMG = BK * BM;
    constexpr int STG = 2 * IMG;
    constexpr int LDS_MIN = (NT / 64) * 32 * 36;
    constexpr int LDS_FLOATS = 2 * STG > LDS_MIN ? 2 * STG : LDS_MIN;
    __shared__ __attribute__((aligned(1024))) float lds[LDS_FLOATS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const unsigned nb = gridDim.x * gridDim.y * gridDim.z;
    const unsigned bid = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    const unsigned q = nb / 8, r = nb % 8, x = bid % 8, i8 = bid / 8;
    const unsigned t = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i8;
    const int tile_n = t % gridDim.x, tile_m = (t / gridDim.x) % gridDim.y, split = t / (gridDim.x * gridDim.y);
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int kbeg = split * p.k_chunk, kend = min(p.K, kbeg + p.k_chunk);
    const int nk = (kend - kbeg + BK - 1) / BK;
    const bool relu_b = p.flags & F_RELU_B;
    const bool do_colsum = p.colsum != nullptr && tile_n == 0;

    // DMA: wave instruction j of wave w covers k rows 2 (2w + j) + (lane >> 5), columns 4 (lane & 31) .. + 3
    const int col_a = min(m0 + (lane & 31) * 4, p.M - 4), col_b = min(n0 + (lane & 31) * 4, p.N - 4);
    auto issue = [&](int k0, float *stage) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int krow = min(k0 + (wave * 2 + j) * 2 + (lane >> 5), p.K - 1);
            __builtin_amdgcn_global_load_lds((glb_void *)(p.A + (size_t)krow * p.lda + col_a), (lds_void *)(stage + (wave * 2 + j) * 256), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((glb_void *)(p.B + (size_t)krow * p.ldb + col_b), (lds_void *)(stage + IMG + (wave * 2 + j) * 256),
                                             16, 0, 0);
        }
    };
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int z = 0; z < 16; ++z) acc[i][j][z] = 0.0f;
    float4 csum = make_float4(0.f, 0.f, 0.f, 0.f);
    if (nk > 0) issue(kbeg, lds);
    __syncthreads();
    const int half = lane >> 5, m = lane & 31;
    for (int kt = 0; kt < nk; ++kt) {
        float *cur = lds + (kt & 1) * STG, *nxt = lds + ((kt & 1) ^ 1) * STG;
        const int k0 = kbeg + kt * BK;
        if (kt + 1 < nk) issue(k0 + BK, nxt);
        const float *ab = cur + (4 * half) * BM + wm * 64 + m;
        const float *bb = cur + IMG + (4 * half) * BN + wn * 64 + m;
        float a[8][TM], b[8][TN];
#pragma unroll
        for (int s = 0; s < 8; ++s) {
#pragma unroll
            for (int i = 0; i < TM; ++i) a[s][i] = ab[(8 * (s >> 2) + (s & 3)) * BM + i * 32];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[s][j] = bb[(8 * (s >> 2) + (s & 3)) * BN + j * 32];
        }
        if (do_colsum) {     // this thread: column group tid & 31, k rows tid >> 5 and (tid >> 5) + 8 of the A image
            const int kr = tid >> 5;
            const float4 v0 = *reinterpret_cast<const float4 *>(cur + kr * BM + (tid & 31) * 4);
            const float4 v1 = *reinterpret_cast<const float4 *>(cur + (kr + 8) * BM + (tid & 31) * 4);
            if (k0 + kr < kend) { csum.x += v0.x; csum.y += v0.y; csum.z += v0.z; csum.w += v0.w; }
            if (k0 + kr + 8 < kend) { csum.x += v1.x; csum.y += v1.y; csum.z += v1.z; csum.w += v1.w; }
        }
        if (k0 + BK > kend) {      // partial last slab: rows past the end were clamped for the DMA, drop them here
#pragma unroll
            for (int s = 0; s < 8; ++s)
                if (k0 + 8 * (s >> 2) + 4 * half + (s & 3) >= kend) {
#pragma unroll
                    for (int i = 0; i < TM; ++i) a[s][i] = 0.0f;
                }
        }
        if (relu_b) {
#pragma unroll
            for (int s = 0; s < 8; ++s)
#pragma unroll
                for (int j = 0; j < TN; ++j) b[s][j] = fmaxf(b[s][j], 0.0f);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s][i], b[s][j], acc[i][j], 0, 0, 0);
        __syncthreads();
    }
    if (do_colsum) {       // fixed-order reduction over the 8 threads that share a column group (as gemm.hip)
        float4 *red = reinterpret_cast<float4 *>(lds);
        red[tid] = csum;
        __syncthreads();
        if (tid < 32) {
            float4 tt = red[tid];
            for (int j = tid + 32; j < NT; j += 32) { tt.x += red[j].x; tt.y += red[j].y; tt.z += red[j].z; tt.w += red[j].w; }
            float *dst = p.colsum + (size_t)split * p.M + m0 + tid * 4;
            const float tv[4] = {tt.x, tt.y, tt.z, tt.w};
#pragma unroll
            for (int z = 0; z < 4; ++z)
                if (m0 + tid * 4 + z < p.M) dst[z] = tv[z];
        }
        __syncthreads();
    }
    EpilogueArgs e;
    e.C = p.C + (size_t)split * p.slab_stride;
    e.bias = nullptr; e.mask = nullptr; e.M = p.M; e.N = p.N; e.ldc = p.ldc; e.ldm = 0;
    e.accum = false; e.relu_out = false;
    store_tiles_f32<TM, TN>(acc, lds + wave * (32 * 36), lane, m0 + wm * (TM * 32), n0 + wn * (TN * 32), e);
}

}  // namespace

bool gemm_dma_applicable(bool b_kc, const GemmArgs &a) {
    // NN (data gradient): with the epilogue's reads requested a pass ahead (gemm_tile.h) it is at least as fast as the
    // register-staged kernel at every shape of the decoder (256->128: 124 -> 116 us, 512->256: 338 -> 332 us, others
    // within 1 %); before that change it won only for long reductions
    if (!b_kc && !(a.N % 4 == 0 && a.N >= 4 && !(a.flags & F_RELU_A))) return false;
    return a.K % BK == 0 && a.K >= BK && a.N > 64 && a.M >= 1 && a.lda % 4 == 0 && a.ldb % 4 == 0 &&
           !(a.flags & F_RELU_B) && a.k_chunk >= a.K;
}

int launch_gemm_dma(bool b_kc, const GemmArgs &a, hipStream_t s, const char *what) {
    dim3 grid((a.N + BN - 1) / BN, (a.M + BM - 1) / BM, 1);
    if (grid.y > 65535) return fail(T2H_ERR_ARG, "%s: grid too large", what);
    if (b_kc) hipLaunchKernelGGL(gemm_dma_kernel, grid, dim3(NT), 0, s, a);
    else hipLaunchKernelGGL(gemm_dma_nn_kernel, grid, dim3(NT), 0, s, a);
    note_kernel(b_kc ? "gemm_dma_kernel" : "gemm_dma_nn_kernel");
    return check_launch(what);
}

// TN (weight gradient), 128 x 128 tiles, reduction rows split over grid.z
bool gemm_dma_tn_applicable(const GemmArgs &a) {
    return a.M % 4 == 0 && a.N % 4 == 0 && a.M >= 4 && a.N >= 4 && a.K >= 1 && a.lda % 4 == 0 && a.ldb % 4 == 0 && a.k_chunk % BK == 0;
}

int launch_gemm_dma_tn(const GemmArgs &a, int splits, hipStream_t s, const char *what) {
    dim3 grid((a.N + BN - 1) / BN, (a.M + BM - 1) / BM, splits);
    if (grid.y > 65535 || grid.z > 65535) return fail(T2H_ERR_ARG, "%s: grid too large", what);
    hipLaunchKernelGGL(gemm_dma_tn_kernel, grid, dim3(NT), 0, s, a);
    note_kernel("gemm_dma_tn_kernel");
    return check_launch(what);
}

}  // namespace t2h
