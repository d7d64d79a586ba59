// Building blocks shared by the fp32 MFMA kernels (gemm.hip: per-point GEMMs; conv.hip: implicit-GEMM 3x3 convs):
// the global -> register -> LDS tile loader, one 16-deep MFMA slab and the float4 epilogue.
#pragma once
#include <hip/hip_runtime.h>

namespace t2h {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int kPad = 4;
constexpr int kMinBK = 16;

// A(m,k): A_KC ? A[m*lda + k] : A[k*lda + m];   B(k,n): B_KC ? B[n*ldb + k] : B[k*ldb + n]
template <int ROWS, int NT, bool KC, int BK>
struct TileLoader {
    static constexpr int TOTAL = ROWS * BK / 4;             // float4s per tile
    static constexpr int PER = (TOTAL + NT - 1) / NT;
    float4 r[PER];

    __device__ inline void load(const float *__restrict__ src, int ld, int row0, int rows, int k0, int kend, int tid,
                                bool relu) {
#pragma unroll
        for (int f = 0; f < PER; ++f) {
            int idx = tid + f * NT;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (TOTAL % NT == 0 || idx < TOTAL) {
                if (KC) {
                    int i = idx / (BK / 4), kc = idx % (BK / 4);
                    int m = row0 + i, k = k0 + kc * 4;
                    if (m < rows && k < kend) v = *reinterpret_cast<const float4 *>(src + (size_t)m * ld + k);
                } else {
                    int k = idx / (ROWS / 4), ic = idx % (ROWS / 4);
                    int m = row0 + ic * 4, kk = k0 + k;
                    if (m < rows && kk < kend) v = *reinterpret_cast<const float4 *>(src + (size_t)kk * ld + m);
                }
            }
            r[f] = v;
        }
        relu_pending = relu;        // applied in store(): touching the values here would wait for the loads at once
    }
    bool relu_pending = false;
    __device__ inline void store(float *__restrict__ lds, int tid) {   // lds: [BK][ROWS + kPad]
#pragma unroll
        for (int f = 0; f < PER; ++f) {
            int idx = tid + f * NT;
            if (relu_pending) { r[f].x = fmaxf(r[f].x, 0.f); r[f].y = fmaxf(r[f].y, 0.f); r[f].z = fmaxf(r[f].z, 0.f); r[f].w = fmaxf(r[f].w, 0.f); }
            if (TOTAL % NT == 0 || idx < TOTAL) {
                if (KC) {
                    int i = idx / (BK / 4), kc = idx % (BK / 4);
                    float *p = lds + (kc * 4) * (ROWS + kPad) + i;
                    p[0] = r[f].x; p[ROWS + kPad] = r[f].y; p[2 * (ROWS + kPad)] = r[f].z; p[3 * (ROWS + kPad)] = r[f].w;
                } else {
                    int k = idx / (ROWS / 4), ic = idx % (ROWS / 4);
                    *reinterpret_cast<float4 *>(lds + k * (ROWS + kPad) + ic * 4) = r[f];
                }
            }
        }
    }
};


// the bias fragments store_tiles_f32 will need (zeros without a bias), to be requested before the main loop
template <int TN>
__device__ inline void load_bias_fragments(const float *bias, int N, int col_base, int lane, float4 (&out)[TN]) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = col_base + j * 32 + (lane & 7) * 4;
        out[j] = (bias && col < N) ? *reinterpret_cast<const float4 *>(bias + col) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

// One BK-deep slab of v_mfma_f32_32x32x2_f32 on k-major LDS tiles (row strides SA / SB floats).  a_base / b_base
// already point at this lane's first element: slab + (lane >> 5) * S + wave offset + (lane & 31).
template <int TM, int TN, int SA, int SB, int BK>
__device__ inline void mfma_slab_f32(const float *a_base, const float *b_base, f32x16 (&acc)[TM][TN]) {
    // software pipelined: the fragments of k-pair kp + 1 are requested from LDS before the MFMAs of kp are issued, so
    // the LDS latency runs under 4 x 64 MFMA cycles instead of stalling the wave between MFMA groups
    float a[2][TM], b[2][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) a[0][i] = a_base[i * 32];
#pragma unroll
    for (int j = 0; j < TN; ++j) b[0][j] = b_base[j * 32];
#pragma unroll
    for (int kp = 0; kp < BK / 2; ++kp) {
        const int cur = kp & 1, nxt = cur ^ 1;
        if (kp + 1 < BK / 2) {
#pragma unroll
            for (int i = 0; i < TM; ++i) a[nxt][i] = a_base[(kp + 1) * 2 * SA + i * 32];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[nxt][j] = b_base[(kp + 1) * 2 * SB + j * 32];
        }
        __builtin_amdgcn_sched_barrier(0);      // keep the LDS requests above the MFMAs (the scheduler sinks them otherwise)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][i], b[cur][j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// max(x, 0) as the one hardware instruction (v_max_f32 returns the other operand for a NaN, as fmaxf does)
__device__ inline float relu1(float x) {
    float y;
    asm("v_max_f32 %0, 0, %1" : "=v"(y) : "v"(x));
    return y;
}

struct EpilogueArgs {
    float *C;
    const float *bias, *mask;
    int M, N, ldc, ldm;
    bool accum, relu_out;
    const float *addend = nullptr;     // [M, ldadd] or null: added last (residual connections)
    int ldadd = 0;
};

// C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).  Each wave
// transposes one 32x32 tile at a time through its private LDS patch (32 x 36 floats) so that global traffic (store,
// mask load, accumulate load) is 16 bytes per lane.  row_base / col_base: first row / column of this wave's tiles.
// bias_pre (optional): the bias fragments of this lane's TN column tiles, loaded by the caller BEFORE its main loop -- the
// epilogue otherwise opens every tile with a dependent global load whose latency nothing hides any more.
template <int TM, int TN>
__device__ inline void store_tiles_f32(f32x16 (&acc)[TM][TN], float *patch, int lane, int row_base, int col_base,
                                       const EpilogueArgs &e, const float4 *bias_pre = nullptr) {
    constexpr int EP = 36;
    const int er = lane >> 3, ec = (lane & 7) * 4;
    if (!e.mask && !e.accum && !e.addend) {       // bias / ReLU only (every forward layer): nothing to read, no branches per pass
        // A wave in its epilogue runs beside three waves per SIMD that issue MFMAs back to back and gets an issue slot only
        // when none of them has one ready (profiles/mfma_corun_lab.hip): its duration is its instruction count.  So: one
        // instruction per ReLU (relu1: fmaxf costs two, it first quiets the operand), the row pointer
        // formed once per tile, and no bounds tests for tiles that lie inside the matrix.
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int row0 = row_base + i * 32, col = col_base + j * 32 + ec;
#pragma unroll
                for (int q = 0; q < 16; ++q)
                    patch[((q & 3) + 8 * (q >> 2) + 4 * (lane >> 5)) * EP + (lane & 31)] = acc[i][j][q];
                float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
                if (bias_pre) bv = bias_pre[j];
                else if (e.bias && col < e.N) bv = *reinterpret_cast<const float4 *>(e.bias + col);
                float *rowp = e.C + (size_t)(row0 + er) * e.ldc + col;
                const size_t step = (size_t)8 * e.ldc;
                float4 v[4];
#pragma unroll
                for (int pass = 0; pass < 4; ++pass) {
                    v[pass] = *reinterpret_cast<const float4 *>(patch + (pass * 8 + er) * EP + ec);
                    v[pass].x += bv.x; v[pass].y += bv.y; v[pass].z += bv.z; v[pass].w += bv.w;
                }
                if (e.relu_out) {
#pragma unroll
                    for (int pass = 0; pass < 4; ++pass) {
                        v[pass].x = relu1(v[pass].x); v[pass].y = relu1(v[pass].y);
                        v[pass].z = relu1(v[pass].z); v[pass].w = relu1(v[pass].w);
                    }
                }
                if (row0 + 32 <= e.M && col_base + j * 32 + 32 <= e.N) {       // wave-uniform: the tile lies inside the matrix
#pragma unroll
                    for (int pass = 0; pass < 4; ++pass) *reinterpret_cast<float4 *>(rowp + pass * step) = v[pass];
                } else {
#pragma unroll
                    for (int pass = 0; pass < 4; ++pass)
                        if (row0 + pass * 8 + er < e.M && col < e.N) *reinterpret_cast<float4 *>(rowp + pass * step) = v[pass];
                }
            }
        return;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int row0 = row_base + i * 32, col0 = col_base + j * 32;
            const int col = col0 + ec;
            // the global reads of a pass (ReLU mask, accumulate target, addend) are requested one pass ahead -- those of
            // pass 0 before the tile goes through the LDS patch -- so their latency runs under the LDS traffic and the
            // previous pass instead of stalling every 8-row pass (one pass ahead, not all four: 128-VGPR budget)
            float4 mk[2], old[2], add[2];
            auto fetch = [&](int pass) {
                const int row = row0 + pass * 8 + er;
                const bool ok = row < e.M && col < e.N;
                if (e.mask && ok) mk[pass & 1] = *reinterpret_cast<const float4 *>(e.mask + (size_t)row * e.ldm + col);
                if (e.accum && ok) old[pass & 1] = *reinterpret_cast<const float4 *>(e.C + (size_t)row * e.ldc + col);
                if (e.addend && ok) add[pass & 1] = *reinterpret_cast<const float4 *>(e.addend + (size_t)row * e.ldadd + col);
            };
            fetch(0);
#pragma unroll
            for (int q = 0; q < 16; ++q)
                patch[((q & 3) + 8 * (q >> 2) + 4 * (lane >> 5)) * EP + (lane & 31)] = acc[i][j][q];
            float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (bias_pre) bv = bias_pre[j];
            else if (e.bias && col < e.N) bv = *reinterpret_cast<const float4 *>(e.bias + col);
#pragma unroll
            for (int pass = 0; pass < 4; ++pass) {
                const int row = row0 + pass * 8 + er;
                if (pass + 1 < 4) fetch(pass + 1);
                float4 v = *reinterpret_cast<const float4 *>(patch + (pass * 8 + er) * EP + ec);
                if (row < e.M && col < e.N) {
                    v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
                    if (e.mask) {
                        const float4 m = mk[pass & 1];
                        v.x = m.x > 0.f ? v.x : 0.f; v.y = m.y > 0.f ? v.y : 0.f;
                        v.z = m.z > 0.f ? v.z : 0.f; v.w = m.w > 0.f ? v.w : 0.f;
                    }
                    if (e.relu_out) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                    if (e.accum) { const float4 o = old[pass & 1]; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
                    if (e.addend) { const float4 o = add[pass & 1]; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
                    *reinterpret_cast<float4 *>(e.C + (size_t)row * e.ldc + col) = v;
                }
            }
        }
}

// reduce_slabs_kernel of gemm.hip: out[r, c] = [out +] sum_z slabs[z][r * cols + c] in a fixed order, and in the same
// launch col_out[r] = [col_out +] sum_z col_slabs[z][r] (the bias gradient; col_out may be null)
// `deferrable` (entry points: flag T2H_DEFER_REDUCE): while a t2h_reduce_capture is active the reduction is recorded instead of
// launched and runs with all other recorded ones in ONE launch at t2h_reduce_capture_end
int launch_reduce_slabs(const float *slabs, int splits, long long stride, int rows, int cols, int ld_out, int accumulate,
                        float *out, const float *col_slabs, float *col_out, hipStream_t s, int col_splits = 0, int col_rows = 0,
                        bool deferrable = false);

// reduce_rows_epilogue_kernel of conv.hip: out[r, c] = [out +] act( sum_z slabs[z][r, c] + bias[c] ) * (mask > 0), splits in order
int launch_reduce_rows_epilogue(const float *slabs, int splits, long long stride, long long M, int N, const EpilogueArgs &e,
                                hipStream_t s);

}  // namespace t2h
