// Per-point MLP GEMMs in exact fp32 on the matrix cores (v_mfma_f32_32x32x2_f32): the dense side of the hot
// path -- nn.Linear forward / input-gradient / weight-gradient for pointnet.py:36-40, resnet.py:26-31 and
// alto.py:63-69,164-170, with the surrounding elementwise work (bias, ReLU, ReLU masks, residual accumulate,
// bias gradient) fused into the prologue / epilogue so no separate elementwise launches remain.
//
//   linear_fwd    Y[M,N]  = [Y +] act_out( act_in(X)[M,K] . W[N,K]^T + b )                 "NT"
//   linear_dgrad  dX[M,K] = [dX +] ( dY[M,N] . W[N,K] ) * (mask > 0)                        "NN"
//   linear_wgrad  dW[N,K] = [dW +] dY[M,N]^T . act_in(X)[M,K];  db[N] = [db +] colsum(dY)   "TN", split over M
//
// M = points of the tile batch (1e5..1e6), N/K = feature widths (8..2048).  One workgroup = BM x BN output tile,
// BK = 16 reduction slab, operands staged global -> registers -> LDS (k-major, rows padded by 4 floats so both the
// transposing ds_write_b32 and the fragment ds_read_b32 stay (nearly) conflict free), LDS double-buffered with
// the next slab's global loads in flight during the MFMAs; 4 waves/SIMD; float4 epilogue through a wave-private LDS
// patch; XCD-aware work order.  MFMA numerics are an fp32 fma chain in k order, so results agree with a scalar fp32
// reference to rounding.  Measured (N = 131072 points): 512<->1024 layers 106-119 TFLOP/s = 67-76 % of the 157.3 TF
// matrix peak, i.e. 90-97 % of what this loop reaches with its global loads removed.
//
// The weight gradient reduces over M: the grid's z dimension splits M, every split writes its partial tile to a
// slab in the caller's workspace, and reduce_slabs_kernel sums the slabs in split order (no atomics =>
// deterministic).
#include <stdlib.h>
#include <mutex>
#include <string>
#include <vector>
#include "t2h_common.h"
#include "gemm_args.h"
#include "gemm_tile.h"

namespace t2h {

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

template <int BM, int BN, int WAVES_M, int WAVES_N, bool A_KC, bool B_KC, int BK = 16, bool XCD = true, int MINW = 1,
          int MODE = 0>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N, MINW) void gemm_kernel(GemmArgs p) {
    constexpr int NT = 64 * WAVES_M * WAVES_N;
    constexpr int TM = BM / (32 * WAVES_M), TN = BN / (32 * WAVES_N);
    constexpr int SA = BM + kPad, SB = BN + kPad;
    static_assert(TM >= 1 && TN >= 1, "tile too small for the wave grid");
    constexpr int LDS_MIN = (NT / 64) * 32 * 36 > 4 * NT ? (NT / 64) * 32 * 36 : 4 * NT;   // epilogue patches / colsum
    constexpr int LDS_FLOATS = 2 * BK * (SA + SB) > LDS_MIN ? 2 * BK * (SA + SB) : LDS_MIN;
    __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
    constexpr int BUF = BK * (SA + SB);            // floats per LDS buffer: A slab then B slab

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    int tile_m = blockIdx.y, tile_n = blockIdx.x, split = blockIdx.z;
    if (XCD) {
        // Workgroups are dealt round-robin over the 8 XCDs (b and b+8 share an L2).  Give every XCD one contiguous
        // run of (split, m-tile, n-tile) work items, so the n-tiles that re-read one A row-tile -- and, for the
        // weight gradient, all tiles of one M-split -- hit the same L2 instead of 8 different ones (measured:
        // 2.75 GB of fabric traffic per 512->1024 launch against 0.81 GB algorithmic without this).  Placement
        // only affects speed, never results.
        const unsigned nb = gridDim.x * gridDim.y * gridDim.z;
        const unsigned b = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        const unsigned q = nb / 8, r = nb % 8, x = b % 8, i = b / 8;
        const unsigned t = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
        tile_n = t % gridDim.x;
        tile_m = (t / gridDim.x) % gridDim.y;
        split = t / (gridDim.x * gridDim.y);
    }
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int kbeg = split * p.k_chunk;
    const int kend = min(p.K, kbeg + p.k_chunk);
    const bool relu_a = p.flags & F_RELU_A, relu_b = p.flags & F_RELU_B;

    TileLoader<BM, NT, A_KC, BK> la;
    TileLoader<BN, NT, B_KC, BK> lb;
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.0f;

    // TN bias gradient: every thread's float4 of the (direct-layout) A tile covers fixed columns m0 + ic*4..+3
    float4 csum = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool do_colsum = !A_KC && p.colsum != nullptr && tile_n == 0;

    const int nk = (kend - kbeg + BK - 1) / BK;
    if (nk > 0) {
        la.load(p.A, p.lda, m0, p.M, kbeg, kend, tid, relu_a);
        lb.load(p.B, p.ldb, n0, p.N, kbeg, kend, tid, relu_b);
        la.store(lds, tid);
        lb.store(lds + BK * SA, tid);
    }
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (do_colsum) {
#pragma unroll
            for (int f = 0; f < TileLoader<BM, NT, A_KC, BK>::PER; ++f) {
                csum.x += la.r[f].x; csum.y += la.r[f].y; csum.z += la.r[f].z; csum.w += la.r[f].w;
            }
        }
        if (kt + 1 < nk) {
            la.load(p.A, p.lda, m0, p.M, kbeg + (kt + 1) * BK, kend, tid, relu_a);
            lb.load(p.B, p.ldb, n0, p.N, kbeg + (kt + 1) * BK, kend, tid, relu_b);
        }
        const float *a_base = lds + cur * BUF + (lane >> 5) * SA + wm * (TM * 32) + (lane & 31);
        const float *b_base = lds + cur * BUF + BK * SA + (lane >> 5) * SB + wn * (TN * 32) + (lane & 31);
        if (MODE == 1) {
            // bf16 operands, fp32 accumulate (BASELINE.json configs[2]): the slab stays fp32 in LDS; each lane gathers
            // the 8 k-values of its row/column (A[row][8h+j], B[8h+j][col]), rounds them to bf16 (RNE,
            // v_cvt_pk_bf16_f32) and issues one 32x32x16 MFMA per 16-deep slab.
#pragma unroll
            for (int ks = 0; ks < BK / 16; ++ks) {
                bf16x8 a[TM], b[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int q = 0; q < 8; ++q)
                        a[i][q] = (__bf16)(lds[cur * BUF + (ks * 16 + (lane >> 5) * 8 + q) * SA + wm * (TM * 32) + i * 32 + (lane & 31)]);
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int q = 0; q < 8; ++q)
                        b[j][q] = (__bf16)(lds[cur * BUF + BK * SA + (ks * 16 + (lane >> 5) * 8 + q) * SB + wn * (TN * 32) + j * 32 + (lane & 31)]);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
            }
        } else if (MODE == 2) {
            // fp32-grade product from bf16 MFMAs: every fp32 operand is split EXACTLY into three bf16 pieces
            // v = v1 + v2 + v3 (+ <= 2^-24 |v|: v1 = bf16(v), v2 = bf16(v - v1), v3 = bf16(v - v1 - v2), the
            // subtractions are exact in fp32).  bf16 x bf16 products are exact in fp32, so summing the six piece
            // products with i + j <= 4 reproduces a*b to ~2^-23 relative (the dropped 2-3, 3-2, 3-3 terms), i.e. at
            // the level of fp32 rounding itself; accumulation is the same fp32 chain as the native path.  Six
            // 32-cycle MFMAs replace eight 64-cycle fp32 ones per 16-deep slab.  Small terms are added first.
#pragma unroll
            for (int ks = 0; ks < BK / 16; ++ks) {
                bf16x8 a1[TM], a2[TM], a3[TM], b1[TN], b2[TN], b3[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        float v = lds[cur * BUF + (ks * 16 + (lane >> 5) * 8 + q) * SA + wm * (TM * 32) + i * 32 + (lane & 31)];
                        __bf16 p1 = (__bf16)v; float r1 = v - (float)p1;
                        __bf16 p2 = (__bf16)r1; float r2 = r1 - (float)p2;
                        a1[i][q] = p1; a2[i][q] = p2; a3[i][q] = (__bf16)r2;
                    }
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        float v = lds[cur * BUF + BK * SA + (ks * 16 + (lane >> 5) * 8 + q) * SB + wn * (TN * 32) + j * 32 + (lane & 31)];
                        __bf16 p1 = (__bf16)v; float r1 = v - (float)p1;
                        __bf16 p2 = (__bf16)r1; float r2 = r1 - (float)p2;
                        b1[j][q] = p1; b2[j][q] = p2; b3[j][q] = (__bf16)r2;
                    }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        f32x16 c = acc[i][j];
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[i], b1[j], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[i], b3[j], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2[i], b2[j], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2[i], b1[j], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[i], b2[j], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[i], b1[j], c, 0, 0, 0);
                        acc[i][j] = c;
                    }
            }
        } else {
            mfma_slab_f32<TM, TN, SA, SB, BK>(a_base, b_base, acc);     // software-pipelined LDS fragment reads
        }
        if (kt + 1 < nk) {
            la.store(lds + (cur ^ 1) * BUF, tid);
            lb.store(lds + (cur ^ 1) * BUF + BK * SA, tid);
        }
        __syncthreads();
    }

    if (do_colsum) {   // fixed-order reduction over the threads that share a column group
        float4 *red = reinterpret_cast<float4 *>(lds);
        red[tid] = csum;
        __syncthreads();
        constexpr int GROUPS = BM / 4;
        if (tid < GROUPS) {
            float4 t = red[tid];
            for (int j = tid + GROUPS; j < NT; j += GROUPS) { t.x += red[j].x; t.y += red[j].y; t.z += red[j].z; t.w += red[j].w; }
            float *dst = p.colsum + (size_t)split * p.M + m0 + tid * 4;
            float tv[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (m0 + tid * 4 + q < p.M) dst[q] = tv[q];
        }
        __syncthreads();       // the epilogue reuses this LDS
    }

    // epilogue.  C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
    // Each wave transposes one 32x32 tile at a time through its private LDS patch (32 x 36 floats) so that global
    // traffic (store, mask load, accumulate load) is 16 bytes per lane: 4 vector-memory instructions per tile
    // instead of 16 dword ones.
    float *C = p.C + (size_t)split * p.slab_stride;
    const bool accum = p.flags & F_ACCUM, relu_out = p.flags & F_RELU_OUT;
    constexpr int EP = 36;
    float *patch = lds + wave * (32 * EP);
    const int er = lane >> 3, ec = (lane & 7) * 4;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int row0 = m0 + wm * (TM * 32) + i * 32, col0 = n0 + wn * (TN * 32) + j * 32;
#pragma unroll
            for (int q = 0; q < 16; ++q)
                patch[((q & 3) + 8 * (q >> 2) + 4 * (lane >> 5)) * EP + (lane & 31)] = acc[i][j][q];
            const int col = col0 + ec;
            float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p.bias && col < p.N) bv = *reinterpret_cast<const float4 *>(p.bias + col);
#pragma unroll
            for (int pass = 0; pass < 4; ++pass) {
                const int row = row0 + pass * 8 + er;
                float4 v = *reinterpret_cast<const float4 *>(patch + (pass * 8 + er) * EP + ec);
                if (row < p.M && col < p.N) {
                    v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
                    if (p.mask) {
                        float4 mk = *reinterpret_cast<const float4 *>(p.mask + (size_t)row * p.ldm + col);
                        v.x = mk.x > 0.f ? v.x : 0.f; v.y = mk.y > 0.f ? v.y : 0.f;
                        v.z = mk.z > 0.f ? v.z : 0.f; v.w = mk.w > 0.f ? v.w : 0.f;
                    }
                    if (relu_out) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                    float4 *dst = reinterpret_cast<float4 *>(C + (size_t)row * p.ldc + col);
                    if (accum) { float4 o = *dst; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
                    if (p.addend) {
                        float4 o = *reinterpret_cast<const float4 *>(p.addend + (size_t)row * p.ldadd + col);
                        v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
                    }
                    *dst = v;
                }
            }
        }
}

// out[e] = [out[e] +] sum_z slabs[z][e].  Per workgroup 16 element-lanes x 16 split-lanes: split-lane q sums the splits
// z = q, q+16, ... in order (4 loads in flight), the 16 lane sums are then added in lane order through LDS: a fixed
// summation tree, so the result is deterministic, and the chain per thread is splits/16 loads deep.  VEC = 4: every
// element-lane owns a float4 (64 consecutive elements = 256-byte row segments per workgroup); VEC = 1 for shapes
// that are not multiples of 4.  Workgroups past the matrix part reduce the bias-gradient slabs
// (col_slabs [splits][rows] -> col_out[rows]) in the same launch, one element per lane.
template <int VEC>
__device__ inline void reduce_slabs_body(float *red, unsigned block, const float *__restrict__ slabs, int splits, long long stride,
                                         int rows, int cols, int ld_out, int accumulate, float *__restrict__ out,
                                         const float *__restrict__ col_slabs, float *__restrict__ col_out,
                                         unsigned matrix_blocks, int col_splits, int col_rows) {
    const int el = threadIdx.x & 15, q = threadIdx.x >> 4;
    if (block >= matrix_blocks) {
        const long long e = (long long)(block - matrix_blocks) * 16 + el;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        if (e < col_rows) {
            int z = q;
            for (; z + 48 < col_splits; z += 64) {
                s0 += col_slabs[(size_t)z * col_rows + e];
                s1 += col_slabs[(size_t)(z + 16) * col_rows + e];
                s2 += col_slabs[(size_t)(z + 32) * col_rows + e];
                s3 += col_slabs[(size_t)(z + 48) * col_rows + e];
            }
            for (; z < col_splits; z += 16) s0 += col_slabs[(size_t)z * col_rows + e];
        }
        red[threadIdx.x] = (s0 + s1) + (s2 + s3);
        __syncthreads();
        if (q == 0 && e < col_rows) {
            float t = red[el];
            for (int k = 1; k < 16; ++k) t += red[k * 16 + el];
            col_out[e] = accumulate ? col_out[e] + t : t;
        }
        return;
    }
    using V = Vec<VEC>;
    const long long total = (long long)rows * cols;
    const long long e = ((long long)block * 16 + el) * VEC;
    V s0, s1, s2, s3;
#pragma unroll
    for (int c = 0; c < VEC; ++c) s0.v[c] = s1.v[c] = s2.v[c] = s3.v[c] = 0.f;
    if (e < total) {
        int z = q;
        for (; z + 48 < splits; z += 64) {
            V a0 = V::load(slabs + (size_t)z * stride + e), a1 = V::load(slabs + (size_t)(z + 16) * stride + e);
            V a2 = V::load(slabs + (size_t)(z + 32) * stride + e), a3 = V::load(slabs + (size_t)(z + 48) * stride + e);
#pragma unroll
            for (int c = 0; c < VEC; ++c) { s0.v[c] += a0.v[c]; s1.v[c] += a1.v[c]; s2.v[c] += a2.v[c]; s3.v[c] += a3.v[c]; }
        }
        for (; z < splits; z += 16) {
            V a0 = V::load(slabs + (size_t)z * stride + e);
#pragma unroll
            for (int c = 0; c < VEC; ++c) s0.v[c] += a0.v[c];
        }
    }
#pragma unroll
    for (int c = 0; c < VEC; ++c) red[threadIdx.x * VEC + c] = (s0.v[c] + s1.v[c]) + (s2.v[c] + s3.v[c]);
    __syncthreads();
    if (q == 0 && e < total) {
        V t;
#pragma unroll
        for (int c = 0; c < VEC; ++c) {
            float acc = red[el * VEC + c];
            for (int k = 1; k < 16; ++k) acc += red[(k * 16 + el) * VEC + c];
            t.v[c] = acc;
        }
        float *dst = out + (size_t)(e / cols) * ld_out + (e % cols);
        if (accumulate) {
            V o = V::load(dst);
#pragma unroll
            for (int c = 0; c < VEC; ++c) t.v[c] += o.v[c];
        }
        t.store(dst);
    }
}

template <int VEC>
__global__ __launch_bounds__(256) void reduce_slabs_kernel(const float *__restrict__ slabs, int splits, long long stride,
                                                          int rows, int cols, int ld_out, int accumulate,
                                                          float *__restrict__ out, const float *__restrict__ col_slabs,
                                                          float *__restrict__ col_out, unsigned matrix_blocks, int col_splits,
                                                          int col_rows) {
    __shared__ float red[256 * VEC];
    reduce_slabs_body<VEC>(red, blockIdx.x, slabs, splits, stride, rows, cols, ld_out, accumulate, out, col_slabs, col_out,
                           matrix_blocks, col_splits, col_rows);
}

// Many slab reductions in ONE launch (t2h_reduce_capture_*): the weight / bias gradients of a tile's backward are not read before
// the end of the backward, so their reductions (one per layer: ~50 launches of 4-20 us per tile-step) are recorded while the
// capture is active and run together.  Same body, same summation tree per output: bit-identical to the separate launches.
constexpr int kBatchSegs = 24;
struct ReduceSeg {
    const float *slabs; float *out; const float *col_slabs; float *col_out;
    long long stride;
    int splits, rows, cols, ld_out, accumulate, col_splits, col_rows, vec;
    unsigned matrix_blocks, blocks;
};
struct ReduceBatch { int n; unsigned first[kBatchSegs + 1]; ReduceSeg seg[kBatchSegs]; };

__global__ __launch_bounds__(256) void reduce_slabs_batch_kernel(ReduceBatch b) {
    __shared__ float red[256 * 4];
    int s = 0;
    while (s + 1 < b.n && blockIdx.x >= b.first[s + 1]) ++s;
    const ReduceSeg &g = b.seg[s];
    const unsigned block = blockIdx.x - b.first[s];
    if (g.vec == 4)
        reduce_slabs_body<4>(red, block, g.slabs, g.splits, g.stride, g.rows, g.cols, g.ld_out, g.accumulate, g.out, g.col_slabs,
                             g.col_out, g.matrix_blocks, g.col_splits, g.col_rows);
    else
        reduce_slabs_body<1>(red, block, g.slabs, g.splits, g.stride, g.rows, g.cols, g.ld_out, g.accumulate, g.out, g.col_slabs,
                             g.col_out, g.matrix_blocks, g.col_splits, g.col_rows);
}

// ---- small-K fallback (fc_pos: K = 3): plain VALU, memory bound --------------------------------------------
__global__ __launch_bounds__(256) void linear_smallk_fwd_kernel(const float *__restrict__ x, int ldx,
                                                               const float *__restrict__ w, const float *__restrict__ bias,
                                                               int M, int K, int N, int ldy, int flags,
                                                               float *__restrict__ y) {
    long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long long)M * N) return;
    int m = (int)(e / N), n = (int)(e % N);
    float acc = bias ? bias[n] : 0.0f;
    for (int k = 0; k < K; ++k) {
        float v = x[(size_t)m * ldx + k];
        if (flags & F_RELU_A) v = fmaxf(v, 0.0f);
        acc = fmaf(v, w[(size_t)n * K + k], acc);
    }
    if (flags & F_RELU_OUT) acc = fmaxf(acc, 0.0f);
    float *dst = y + (size_t)m * ldy + n;
    *dst = (flags & F_ACCUM) ? *dst + acc : acc;
}

constexpr int kSmallKMax = 8;
constexpr int kSmallKRows = 128;   // rows per workgroup (1024 workgroups at 131072 points)
// slab[z][n][k] = sum over the workgroup's rows of dy[m][n] * x[m][k];  colslab[z][n] = sum dy[m][n]
__global__ __launch_bounds__(256) void wgrad_smallk_kernel(const float *__restrict__ dy, int lddy,
                                                          const float *__restrict__ x, int ldx, int M, int K, int N,
                                                          int relu_x, float *__restrict__ slab, float *__restrict__ colslab) {
    __shared__ float red[256 * (kSmallKMax + 1)];
    int tid = threadIdx.x;
    int lanes_n = min(N, 256);                 // threads along n
    int slots = 256 / lanes_n;                 // parallel row slots
    int n_local = tid % lanes_n, slot = tid / lanes_n;
    int m_begin = blockIdx.x * kSmallKRows, m_end = min(M, m_begin + kSmallKRows);
    for (int nb = 0; nb < N; nb += lanes_n) {
        int n = nb + n_local;
        float acc[kSmallKMax + 1];
#pragma unroll
        for (int k = 0; k <= kSmallKMax; ++k) acc[k] = 0.0f;
        if (n < N && slot < slots) {
            // four rows of dY in flight per lane; accumulation stays in ascending row order
            for (int m = m_begin + slot; m < m_end; m += 4 * slots) {
                float g[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) g[u] = m + u * slots < m_end ? dy[(size_t)(m + u * slots) * lddy + n] : 0.0f;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (m + u * slots >= m_end) break;
                    acc[kSmallKMax] += g[u];
#pragma unroll
                    for (int k = 0; k < kSmallKMax; ++k)
                        if (k < K) {
                            float v = x[(size_t)(m + u * slots) * ldx + k];
                            if (relu_x) v = fmaxf(v, 0.0f);
                            acc[k] = fmaf(g[u], v, acc[k]);
                        }
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k <= kSmallKMax; ++k) red[k * 256 + tid] = acc[k];
        __syncthreads();
        if (slot == 0 && n < N) {
#pragma unroll
            for (int k = 0; k <= kSmallKMax; ++k) {
                if (k < K || k == kSmallKMax) {
                    float t = 0.0f;
                    for (int s = 0; s < slots; ++s) t += red[k * 256 + s * lanes_n + n_local];
                    if (k == kSmallKMax) { if (colslab) colslab[(size_t)blockIdx.x * N + n] = t; }
                    else slab[((size_t)blockIdx.x * N + n) * K + k] = t;
                }
            }
        }
    }
}

// Few rows and a long reduction (the grid-side products of the deferred ALTO point update: 1024 .. 16384 pixel rows against
// 256 .. 2560 stacked source columns): a workgroup per output tile walking the whole reduction is a chain of dependent
// global -> LDS -> MFMA round trips with nothing to overlap them.  Here one workgroup = one 32 x 32 output tile and its four
// waves split the reduction: wave w takes the 32-deep slabs w, w + 4, ... with its own double-buffered LDS pair (no
// workgroup barrier in the loop), four times the loads in flight and a quarter of the chain.  The four partial tiles are
// summed in wave order through LDS -- a fixed tree, deterministic -- and the usual epilogue (bias, mask, ReLU, accumulate,
// addend) follows on 16 bytes per lane.
template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(256) void gemm_kwaves_kernel(GemmArgs p) {
    constexpr int BK = 32, S = 32 + kPad, BUF = 2 * BK * S;          // floats per wave buffer: A slab, B slab
    __shared__ __attribute__((aligned(16))) float lds[4 * 2 * BUF];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned nb = gridDim.x * gridDim.y;
    const unsigned b = blockIdx.y * gridDim.x + blockIdx.x;
    const unsigned q8 = nb / 8, r8 = nb % 8, x8 = b % 8, i8 = b / 8;
    const unsigned t = (x8 < r8 ? x8 * (q8 + 1) : r8 * (q8 + 1) + (x8 - r8) * q8) + i8;
    const int tile_n = t % gridDim.x, tile_m = t / gridDim.x;
    const int m0 = tile_m * 32, n0 = tile_n * 32;
    const bool relu_a = p.flags & F_RELU_A, relu_b = p.flags & F_RELU_B;
    const int nk = (p.K + BK - 1) / BK;
    float *mine = lds + wave * 2 * BUF;

    TileLoader<32, 64, A_KC, BK> la;
    TileLoader<32, 64, B_KC, BK> lb;
    f32x16 acc[1][1];
#pragma unroll
    for (int z = 0; z < 16; ++z) acc[0][0][z] = 0.0f;
    int kt = wave;
    if (kt < nk) {
        la.load(p.A, p.lda, m0, p.M, kt * BK, p.K, lane, relu_a);
        lb.load(p.B, p.ldb, n0, p.N, kt * BK, p.K, lane, relu_b);
        la.store(mine, lane);
        lb.store(mine + BK * S, lane);
    }
    int cur = 0;
    for (; kt < nk; kt += 4) {
        const bool more = kt + 4 < nk;
        if (more) {
            la.load(p.A, p.lda, m0, p.M, (kt + 4) * BK, p.K, lane, relu_a);
            lb.load(p.B, p.ldb, n0, p.N, (kt + 4) * BK, p.K, lane, relu_b);
        }
        __builtin_amdgcn_wave_barrier();       // LDS operations of one wave execute in order: its stores above are visible
        const float *a_base = mine + cur * BUF + (lane >> 5) * S + (lane & 31);
        mfma_slab_f32<1, 1, S, S, BK>(a_base, a_base + BK * S, acc);
        if (more) {
            la.store(mine + (cur ^ 1) * BUF, lane);
            lb.store(mine + (cur ^ 1) * BUF + BK * S, lane);
        }
        cur ^= 1;
    }
    __builtin_amdgcn_wave_barrier();
    constexpr int EP = 36;
#pragma unroll
    for (int z = 0; z < 16; ++z) mine[((z & 3) + 8 * (z >> 2) + 4 * (lane >> 5)) * EP + (lane & 31)] = acc[0][0][z];
    __syncthreads();
    const int row = m0 + (tid >> 3), col = n0 + (tid & 7) * 4;
    if (row >= p.M || col >= p.N) return;
    float4 v = *reinterpret_cast<const float4 *>(lds + (tid >> 3) * EP + (tid & 7) * 4);
#pragma unroll
    for (int w = 1; w < 4; ++w) {
        const float4 u = *reinterpret_cast<const float4 *>(lds + w * 2 * BUF + (tid >> 3) * EP + (tid & 7) * 4);
        v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
    }
    if (p.bias) { const float4 bv = *reinterpret_cast<const float4 *>(p.bias + col); v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w; }
    if (p.mask) {
        const float4 mk = *reinterpret_cast<const float4 *>(p.mask + (size_t)row * p.ldm + col);
        v.x = mk.x > 0.f ? v.x : 0.f; v.y = mk.y > 0.f ? v.y : 0.f; v.z = mk.z > 0.f ? v.z : 0.f; v.w = mk.w > 0.f ? v.w : 0.f;
    }
    if (p.flags & F_RELU_OUT) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    float4 *dst = reinterpret_cast<float4 *>(p.C + (size_t)row * p.ldc + col);
    if (p.flags & F_ACCUM) { const float4 o = *dst; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
    if (p.addend) {
        const float4 o = *reinterpret_cast<const float4 *>(p.addend + (size_t)row * p.ldadd + col);
        v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
    }
    *dst = v;
}

template <bool A_KC, bool B_KC>
static int launch_gemm_kwaves(const GemmArgs &a, hipStream_t s, const char *what) {
    dim3 grid((a.N + 31) / 32, (a.M + 31) / 32);
    if (grid.y > 65535) return fail(T2H_ERR_ARG, "%s: grid too large", what);
    hipLaunchKernelGGL((gemm_kwaves_kernel<A_KC, B_KC>), grid, dim3(256), 0, s, a);
    note_kernel(!A_KC ? "gemm_kwaves_kernel<false,false>" : (B_KC ? "gemm_kwaves_kernel<true,true>" : "gemm_kwaves_kernel<true,false>"));
    return check_launch(what);
}
// few output tiles (every CU would not even get one 64 x 64 tile's worth) and a reduction long enough to split
static bool kwaves_applicable(long long rows, int cols, int depth) {
    static const int min_k = [] { const char *e = getenv("T2H_KWAVES_MIN_K"); return e ? atoi(e) : 256; }();
    static const long long max_tiles = [] { const char *e = getenv("T2H_KWAVES_MAX_TILES"); return e ? atoll(e) : 1024ll; }();
    return depth >= min_k && ((rows + 31) / 32) * ((cols + 31) / 32) <= max_tiles;
}

template <int BM, int BN, int WM, int WN, bool A_KC, bool B_KC, int BK = 16, bool XCD = true, int MINW = 1, int MODE = 0>
static int launch_gemm(const GemmArgs &a, int splits, hipStream_t s, const char *what) {
    dim3 grid((a.N + BN - 1) / BN, (a.M + BM - 1) / BM, splits);
    if (grid.y > 65535 || grid.z > 65535) return fail(T2H_ERR_ARG, "%s: grid too large", what);
    hipLaunchKernelGGL((gemm_kernel<BM, BN, WM, WN, A_KC, B_KC, BK, XCD, MINW, MODE>), grid, dim3(64 * WM * WN), 0, s, a);
    static const std::string name = "gemm_kernel<" + std::to_string(BM) + "," + std::to_string(BN) + "," + std::to_string(WM) + "," +
                                    std::to_string(WN) + "," + (A_KC ? "true" : "false") + "," + (B_KC ? "true" : "false") + "," +
                                    std::to_string(BK) + "," + (XCD ? "true" : "false") + "," + std::to_string(MINW) + "," +
                                    std::to_string(MODE) + ">";
    note_kernel(name.c_str());
    return check_launch(what);
}

// pick the N tile for row-streaming GEMMs (M huge).  128x128x16 with 4 waves won the A/B against BK = 32 and
// 256x128 tiles (both lose occupancy: 2 resp. 1 waves/SIMD instead of 3-4).
template <bool B_KC, int MODE>
static int launch_rows_p(const GemmArgs &a, hipStream_t s, const char *what) {
    constexpr int MW = MODE == 0 ? 4 : (MODE == 1 ? 3 : 2);     // register budget: 128 / 168 / 256 per lane
    if (a.N > 64) return launch_gemm<128, 128, 2, 2, true, B_KC, 16, true, MW, MODE>(a, 1, s, what);
    // 33 .. 64 output columns from a long reduction (the level products of the deferred point update at r = 256: 65536 pixel
    // rows x 2560 stacked columns -> 64): the A operand is streamed once, so smaller row tiles = more workgroups in flight
    // (64 x 64 x 32: 222 -> 207 us; 64-deep slabs or 32-row tiles: 262 / 332 us)
    static const bool skinny = !(getenv("T2H_SKINNY") && getenv("T2H_SKINNY")[0] == '0');
    if (MODE == 0 && a.N > 32 && a.K >= 512 && skinny) return launch_gemm<64, 64, 2, 2, true, B_KC, 32, true, 1, 0>(a, 1, s, what);
    if (a.N > 32) return launch_gemm<128, 64, 2, 2, true, B_KC, 16, true, 1, MODE>(a, 1, s, what);
    return launch_gemm<128, 32, 4, 1, true, B_KC, 16, true, 1, MODE>(a, 1, s, what);
}
static int mode_of(int flags) { return (flags & T2H_BF16X3) ? 2 : ((flags & T2H_BF16) ? 1 : 0); }
template <bool B_KC>
static int launch_rows(const GemmArgs &a, int mode, hipStream_t s, const char *what) {
    // bf16: conversion-at-staging kernel (gemm_split.hip).  bf16x3 forward / data gradient: the in-loop split below is
    // faster (0.87 vs 1.0 ms at 512 -> 1024: its 768-cycle MFMA phase hides less of the load latency there).
    if (mode == 1 && a.N > 64) return launch_gemm_split(mode, true, B_KC, a, 1, s, what);
    if (mode == 2) return launch_rows_p<B_KC, 2>(a, s, what);
    if (mode == 1) return launch_rows_p<B_KC, 1>(a, s, what);
    // Few rows (the grid-side products of the deferred ALTO point update: 1024 .. 16384 pixel rows; the composed weight
    // products): 128 x 128 tiles would leave most of the 256 CUs idle behind a long serial reduction -- the launch is
    // latency-bound, so take the smallest tiles that still give every CU a workgroup (results do not depend on the tiling:
    // every output is the same k-ordered fma chain)
    if (mode == 0 && a.N > 64) {
        const long long t128 = (long long)((a.M + 127) / 128) * ((a.N + 127) / 128);
        if (t128 < 192) {
            const long long t64 = (long long)((a.M + 63) / 64) * ((a.N + 63) / 64);
            static const int small_bk = [] { const char *e = getenv("T2H_SMALLM_BK"); return e ? atoi(e) : 64; }();
            if (kwaves_applicable(a.M, a.N, a.K)) return launch_gemm_kwaves<true, B_KC>(a, s, what);
            if (t64 >= 192) {
                if (small_bk >= 32) return launch_gemm<64, 64, 2, 2, true, B_KC, 32, true, 1, 0>(a, 1, s, what);
                return launch_gemm<64, 64, 2, 2, true, B_KC, 16, true, 1, 0>(a, 1, s, what);
            }
            if (small_bk >= 64) return launch_gemm<32, 32, 1, 1, true, B_KC, 64, true, 1, 0>(a, 1, s, what);
            if (small_bk >= 32) return launch_gemm<32, 32, 1, 1, true, B_KC, 32, true, 1, 0>(a, 1, s, what);
            return launch_gemm<32, 32, 1, 1, true, B_KC, 16, true, 1, 0>(a, 1, s, what);
        }
    }
    static const bool use_dma = !(getenv("T2H_GEMM_DMA") && getenv("T2H_GEMM_DMA")[0] == '0');
    // (r02 A/B: 128 x 64 tiles for the 128- / 256-wide layers -- twice the workgroups, VERDICT r01 item 6 -- lose 5-14 % on
    // the forward and gain 10 % only on the data gradient into 256 columns from a 128-long reduction; not adopted)
    if (use_dma && gemm_dma_applicable(B_KC, a)) return launch_gemm_dma(B_KC, a, s, what);
    return launch_rows_p<B_KC, 0>(a, s, what);
}

static bool aligned4(const void *p, int ld) { return ((uintptr_t)p % 16 == 0) && (ld % 4 == 0); }

// ---- capture of deferrable slab reductions (process-wide, explicit: see t2h_reduce_capture_begin in include/t2h.h) ----------------
static std::mutex g_cap_mutex;
static bool g_cap_active = false;
static std::vector<ReduceSeg> g_cap;

static int flush_captured_locked(hipStream_t s) {
    size_t i = 0;
    while (i < g_cap.size()) {
        ReduceBatch b{};
        unsigned blocks = 0;
        for (; i < g_cap.size() && b.n < kBatchSegs; ++i) {
            b.first[b.n] = blocks;
            b.seg[b.n] = g_cap[i];
            blocks += g_cap[i].blocks;
            ++b.n;
        }
        b.first[b.n] = blocks;
        hipLaunchKernelGGL(reduce_slabs_batch_kernel, dim3(blocks), dim3(256), 0, s, b);
    }
    g_cap.clear();
    return check_launch("reduce_slabs_batch");
}

int launch_reduce_slabs(const float *slabs, int splits, long long stride, int rows, int cols, int ld_out, int accumulate,
                        float *out, const float *col_slabs, float *col_out, hipStream_t s, int col_splits, int col_rows,
                        bool deferrable) {
    // col_slabs [col_splits][col_rows] -> col_out[col_rows]; by default the matrix's split count and row count
    if (col_splits <= 0) col_splits = splits;
    if (col_rows <= 0) col_rows = rows;
    const long long total = (long long)rows * cols;
    const bool vec = cols % 4 == 0 && ld_out % 4 == 0 && stride % 4 == 0 && (uintptr_t)slabs % 16 == 0 && (uintptr_t)out % 16 == 0;
    const unsigned mb = (unsigned)((total + (vec ? 63 : 15)) / (vec ? 64 : 16));
    const unsigned cb = col_out ? (unsigned)((col_rows + 15) / 16) : 0u;
    if (deferrable) {
        std::lock_guard<std::mutex> lock(g_cap_mutex);
        if (g_cap_active) {
            // two reductions into the same output in one batch would race: run what is queued first
            for (const ReduceSeg &q : g_cap)
                if (q.out == out || (col_out && q.col_out == col_out)) {
                    if (int rc = flush_captured_locked(s)) return rc;
                    break;
                }
            ReduceSeg g{};
            g.slabs = slabs; g.out = out; g.col_slabs = col_slabs; g.col_out = col_out; g.stride = stride; g.splits = splits;
            g.rows = rows; g.cols = cols; g.ld_out = ld_out; g.accumulate = accumulate; g.col_splits = col_splits;
            g.col_rows = col_rows; g.vec = vec ? 4 : 1; g.matrix_blocks = mb; g.blocks = mb + cb;
            g_cap.push_back(g);
            return T2H_OK;
        }
    }
    if (vec)
        hipLaunchKernelGGL(reduce_slabs_kernel<4>, dim3(mb + cb), dim3(256), 0, s, slabs, splits, stride, rows, cols, ld_out,
                           accumulate, out, col_slabs, col_out, mb, col_splits, col_rows);
    else
        hipLaunchKernelGGL(reduce_slabs_kernel<1>, dim3(mb + cb), dim3(256), 0, s, slabs, splits, stride, rows, cols, ld_out,
                           accumulate, out, col_slabs, col_out, mb, col_splits, col_rows);
    return check_launch("reduce_slabs");
}

}  // namespace t2h

using namespace t2h;

T2H_API int t2h_reduce_capture_begin(void) {
    std::lock_guard<std::mutex> lock(g_cap_mutex);
    if (g_cap_active) return fail(T2H_ERR_ARG, "reduce_capture_begin: a capture is already active");
    g_cap_active = true;
    g_cap.clear();
    return T2H_OK;
}

T2H_API int t2h_reduce_capture_pending(void) {
    std::lock_guard<std::mutex> lock(g_cap_mutex);
    return g_cap_active ? (int)g_cap.size() : -1;
}

T2H_API int t2h_reduce_capture_end(t2h_stream_t stream) {
    std::lock_guard<std::mutex> lock(g_cap_mutex);
    if (!g_cap_active) return fail(T2H_ERR_ARG, "reduce_capture_end: no capture is active");
    g_cap_active = false;
    return flush_captured_locked(as_stream(stream));
}

static int map_flags(int f) {
    int o = 0;
    if (f & T2H_RELU_IN) o |= F_RELU_A;
    if (f & T2H_RELU_OUT) o |= F_RELU_OUT;
    if (f & T2H_ACCUM) o |= F_ACCUM;
    return o;
}

T2H_API int t2h_linear_fwd_add(const float *x, int ldx, const float *w, const float *bias, const float *addend, int ldadd,
                               float *y, int ldy, int M, int K, int N, int flags, t2h_stream_t stream) {
    if (!x || !w || !y) return fail(T2H_ERR_ARG, "linear_fwd: null pointer");
    if (M < 0 || K < 1 || N < 1 || ldx < K || ldy < N) return fail(T2H_ERR_ARG, "linear_fwd: bad shape");
    if (M == 0) return T2H_OK;
    hipStream_t s = as_stream(stream);
    const bool vec_ok = K % 4 == 0 && N % 4 == 0 && aligned4(x, ldx) && aligned4(w, K) && aligned4(y, ldy) &&
                        (!bias || (uintptr_t)bias % 16 == 0);
    if (addend && (!vec_ok || !aligned4(addend, ldadd) || ldadd < N || mode_of(flags) != 0))
        return fail(T2H_ERR_ARG, "linear_fwd_add: the addend needs 16-byte rows (multiples of 4) and the fp32 kernels");
    if (!vec_ok) {   // fc_pos (K = 3) and 1-channel heads: plain VALU kernel
        if (K > 64) return fail(T2H_ERR_ARG, "linear_fwd: K=%d, N=%d need 16-byte rows (multiples of 4) beyond K=64", K, N);
        long long total = (long long)M * N;
        hipLaunchKernelGGL(linear_smallk_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, ldx, w, bias,
                           M, K, N, ldy, map_flags(flags), y);
        return check_launch("linear_fwd(small K)");
    }
    GemmArgs a{};
    a.A = x; a.lda = ldx; a.B = w; a.ldb = K; a.C = y; a.ldc = ldy; a.bias = bias; a.addend = addend; a.ldadd = ldadd;
    a.M = M; a.N = N; a.K = K; a.flags = map_flags(flags); a.k_chunk = K; a.slab_stride = 0;
    return launch_rows<true>(a, mode_of(flags), s, "linear_fwd");
}

T2H_API int t2h_linear_fwd(const float *x, int ldx, const float *w, const float *bias, float *y, int ldy, int M, int K,
                           int N, int flags, t2h_stream_t stream) {
    return t2h_linear_fwd_add(x, ldx, w, bias, nullptr, 0, y, ldy, M, K, N, flags, stream);
}

T2H_API int t2h_linear_dgrad(const float *dy, int lddy, const float *w, float *dx, int lddx, int M, int K, int N,
                             const float *mask, int ldmask, int flags, t2h_stream_t stream) {
    if (!dy || !w || !dx) return fail(T2H_ERR_ARG, "linear_dgrad: null pointer");
    if (M < 0 || K < 1 || N < 1 || lddy < N || lddx < K) return fail(T2H_ERR_ARG, "linear_dgrad: bad shape");
    if (N % 4 != 0 || K % 4 != 0 || !aligned4(dy, lddy) || !aligned4(w, K) || !aligned4(dx, lddx) ||
        (mask && !aligned4(mask, ldmask)))
        return fail(T2H_ERR_ARG, "linear_dgrad: N=%d, K=%d must be multiples of 4 (16-byte rows)", N, K);
    if (M == 0) return T2H_OK;
    GemmArgs a{};
    // dX[M,K] = dY[M,N] . W[N,K]: reduction over N; B(k=n, j) = W[n*K + j] is j-contiguous (direct layout)
    a.A = dy; a.lda = lddy; a.B = w; a.ldb = K; a.C = dx; a.ldc = lddx; a.mask = mask; a.ldm = ldmask;
    a.M = M; a.N = K; a.K = N; a.flags = map_flags(flags & T2H_ACCUM); a.k_chunk = N; a.slab_stride = 0;
    return launch_rows<false>(a, mode_of(flags), as_stream(stream), "linear_dgrad");
}

namespace {
struct WgradPlan { bool smallk; int bm, bn, tiles, splits, k_chunk; };
WgradPlan wgrad_plan(int M, int K, int N) {
    WgradPlan p{};
    p.smallk = (K % 4 != 0);
    if (p.smallk) { p.splits = (M + kSmallKRows - 1) / kSmallKRows; return p; }
    p.bm = N > 64 ? 128 : (N > 32 ? 64 : 32);
    p.bn = K > 64 ? 128 : (K > 32 ? 64 : 32);
    p.tiles = ((N + p.bm - 1) / p.bm) * ((K + p.bn - 1) / p.bn);
    // two-tile layers (128 <-> 256): 256 splits of 512 rows beat 512 of 256 (91 -> 83 us: half the slab traffic, twice the
    // main loop per exposed epilogue); eight tiles and more want the full resident wave (768: -3 %, 512: -10 %)
    const int target = p.tiles <= 2 ? 512 : 1024;
    int want = target / p.tiles;                                  // <= 1024 workgroups = one resident wave (4 per CU): one
                                                                // more would run alone after all others finish
    if (want > 512) want = 512;
    int max_splits = (M + 16 * kMinBK - 1) / (16 * kMinBK);     // at least 16 reduction slabs per workgroup
    int splits = want < 1 ? 1 : (want > max_splits ? max_splits : want);
    if (splits < 1) splits = 1;
    int chunk = (M + splits - 1) / splits;
    chunk = (chunk + 31) / 32 * 32;
    p.k_chunk = chunk;
    p.splits = (M + chunk - 1) / chunk;
    return p;
}
}  // namespace

T2H_API size_t t2h_linear_wgrad_workspace_bytes(int M, int K, int N) {
    if (M < 1 || K < 1 || N < 1) return 0;
    WgradPlan p = wgrad_plan(M, K, N);
    return (size_t)p.splits * ((size_t)N * K + N) * sizeof(float);
}

T2H_API int t2h_linear_wgrad(const float *dy, int lddy, const float *x, int ldx, int M, int K, int N, int flags,
                             float *dw, float *db, void *workspace, size_t workspace_bytes, t2h_stream_t stream) {
    if (!dy || !x || !dw) return fail(T2H_ERR_ARG, "linear_wgrad: null pointer");
    if (M < 1 || K < 1 || N < 1 || lddy < N || ldx < K) return fail(T2H_ERR_ARG, "linear_wgrad: bad shape");
    size_t need = t2h_linear_wgrad_workspace_bytes(M, K, N);
    if (!workspace || workspace_bytes < need)
        return fail(T2H_ERR_WORKSPACE, "linear_wgrad: workspace %zu < %zu bytes", workspace_bytes, need);
    hipStream_t s = as_stream(stream);
    WgradPlan p = wgrad_plan(M, K, N);
    float *slab = static_cast<float *>(workspace);
    float *colslab = slab + (size_t)p.splits * N * K;
    int accumulate = (flags & T2H_ACCUM) ? 1 : 0;
    // one split, plain store, no bias gradient (the grid-side products of the deferred point update): the kernel writes dW
    // itself and the slab reduction launch is skipped (same values: the reduction of one slab is a copy)
    const bool direct = !p.smallk && p.splits == 1 && !accumulate && !db && aligned4(dw, K);
    if (direct) slab = dw;
    if (p.smallk) {
        if (K > kSmallKMax) return fail(T2H_ERR_ARG, "linear_wgrad: K=%d must be a multiple of 4 (or <= %d)", K, kSmallKMax);
        hipLaunchKernelGGL(wgrad_smallk_kernel, dim3(p.splits), dim3(256), 0, s, dy, lddy, x, ldx, M, K, N,
                           (flags & T2H_RELU_IN) ? 1 : 0, slab, db ? colslab : nullptr);
    } else {
        if (N % 4 != 0 || !aligned4(dy, lddy) || !aligned4(x, ldx))
            return fail(T2H_ERR_ARG, "linear_wgrad: N=%d must be a multiple of 4 (16-byte rows)", N);
        GemmArgs a{};
        // dW[N,K] = sum_m dY[m,n] X[m,k]: A(i=n, k=m) = dY[m*lddy + n] (direct), B(k=m, j) = X[m*ldx + j] (direct)
        a.A = dy; a.lda = lddy; a.B = x; a.ldb = ldx; a.C = slab; a.ldc = K; a.colsum = db ? colslab : nullptr;
        a.M = N; a.N = K; a.K = M; a.flags = (flags & T2H_RELU_IN) ? F_RELU_B : 0;
        a.k_chunk = p.k_chunk; a.slab_stride = (long long)N * K;
        int rc;
        const int mode = mode_of(flags);
        // the grid-side products of the deferred point update at r = 32 (1024 pixel rows, no bias gradient; 4096 rows and more: the
        // split launch + slab reduction below is faster, 70 vs 87 us at [2176, 256]): one launch, the
        // workgroup's four waves split the rows (gemm_kwaves_kernel), dW stored (or accumulated) directly
        static const int kwaves_wgrad_max_rows = [] { const char *e = getenv("T2H_KWAVES_WGRAD_MAX_ROWS"); return e ? atoi(e) : 1024; }();
        if (mode == 0 && !db && M <= kwaves_wgrad_max_rows && aligned4(dw, K) && kwaves_applicable(N, K, M)) {
            a.C = dw; a.colsum = nullptr; a.k_chunk = M; a.slab_stride = 0;
            if (accumulate) a.flags |= F_ACCUM;
            return launch_gemm_kwaves<false, false>(a, s, "linear_wgrad");
        }
#define T2H_WG(BM_, BN_, WM_, WN_, MW_)                                                                                \
    (mode == 2 ? launch_gemm<BM_, BN_, WM_, WN_, false, false, 16, true, (MW_ > 2 ? 2 : MW_), 2>(a, p.splits, s, "linear_wgrad") \
     : mode == 1 ? launch_gemm<BM_, BN_, WM_, WN_, false, false, 16, true, (MW_ > 3 ? 3 : MW_), 1>(a, p.splits, s, "linear_wgrad") \
                 : launch_gemm<BM_, BN_, WM_, WN_, false, false, 16, true, MW_, 0>(a, p.splits, s, "linear_wgrad"))
        static const bool tn_dma = !(getenv("T2H_GEMM_DMA") && getenv("T2H_GEMM_DMA")[0] == '0');
        if (p.bm == 128 && p.bn == 128 && mode == 0 && tn_dma && gemm_dma_tn_applicable(a))
            rc = launch_gemm_dma_tn(a, p.splits, s, "linear_wgrad");
        else if (p.bm == 128 && p.bn == 128)
            rc = mode != 0 ? launch_gemm_split(mode, false, false, a, p.splits, s, "linear_wgrad") : T2H_WG(128, 128, 2, 2, 4);
        else if (p.bm == 128 && p.bn == 64) rc = T2H_WG(128, 64, 2, 2, 1);
        else if (p.bm == 128 && p.bn == 32) rc = T2H_WG(128, 32, 4, 1, 1);
        else if (p.bm == 64 && p.bn == 128) rc = T2H_WG(64, 128, 2, 2, 1);
        else if (p.bm == 64 && p.bn == 64) rc = T2H_WG(64, 64, 2, 2, 1);
        else if (p.bm == 64 && p.bn == 32) rc = T2H_WG(64, 32, 2, 1, 1);
        else if (p.bm == 32 && p.bn == 128) rc = T2H_WG(32, 128, 1, 4, 1);
        else if (p.bm == 32 && p.bn == 64) rc = T2H_WG(32, 64, 1, 2, 1);
        else rc = T2H_WG(32, 32, 1, 1, 1);
#undef T2H_WG
        if (rc) return rc;
    }
    if (direct) return T2H_OK;
    return launch_reduce_slabs(slab, p.splits, (long long)N * K, N, K, K, accumulate, dw, colslab, db, s, 0, 0,
                               (flags & T2H_DEFER_REDUCE) != 0);
}
