// Operator-level seam: torch_scatter.scatter_max in the reference's own layout (see include/t2h.h).
//
//   scatter_max(src [B, C, N], index [B, 1, N], dim_size = R*R) -> (out [B, C, R*R], arg int64 [B, C, R*R])   pointnet.py:95
//   its backward: the gradient of `out` routed to `arg` only (pytorch-scatter: zeros[.., N + 1].scatter_(arg, grad).narrow(N))
//
// The packaged modules never come here (they pool inside the trunk's loaders on sorted rows); this is what a caller of the
// reference's OPERATORS gets.  The points stay where they are: the tile index built from the cell ids supplies, per cell of
// the plane, the run of (sorted) rows that fall into it and `perm` maps a sorted row back to the caller's row, so a cell's
// maximum is a short sequential scan over gathered 4 C-byte rows -- no atomics, and the stable sort makes "first row of
// the scan that holds the maximum" the first point in the caller's order (pytorch-scatter's CPU tie-break).
// HBM-bound: reads 4 C N + 8 N, writes 12 C R^2 (value + int64 arg) through an LDS transpose so that the channel-major
// output goes out in 32-byte (values) / 64-byte (args) runs of the plane's rows.
#include <float.h>

#include "t2h_common.h"

namespace t2h {

constexpr int kSmThreads = 256;
constexpr int kSmCells = 64;      // Morton-consecutive cells per workgroup: an 8 x 8 block of the plane (nbits >= 3)
constexpr int kSmChunk = 16;      // channels per pass: 4 lanes x float4 per cell

__global__ __launch_bounds__(kSmThreads) void scatter_max_fwd_kernel(const float *__restrict__ feat, int ld,
                                                                     const int32_t *__restrict__ perm,
                                                                     const int32_t *__restrict__ off0, int N, int nbits,
                                                                     int64_t ncells, int C, float *__restrict__ val,
                                                                     int64_t *__restrict__ arg) {
    __shared__ float s_val[kSmChunk][kSmCells + 1];
    __shared__ int s_arg[kSmChunk][kSmCells + 1];
    const int tid = threadIdx.x;
    const int64_t cell0 = (int64_t)blockIdx.x * kSmCells;
    const int64_t plane = (int64_t)1 << (2 * nbits);
    const int R = 1 << nbits;
    // scan role: 4 lanes per cell, lane q owns channels c0 + 4 q .. + 3 of the pass
    const int lc = tid >> 2, q = tid & 3;
    const int64_t cid = cell0 + lc;
    int s = 0, e = 0;
    size_t rowbase = 0;
    if (cid < ncells) {
        s = off0[cid]; e = off0[cid + 1];
        rowbase = (size_t)(cid >> (2 * nbits)) * N;              // perm holds the row inside its tile
    }
    // store role: lane l of a wave = cell (l & 7, l >> 3) of the 8 x 8 block in ROW-major order, wave w = channel w + 4 k
    const int l = tid & 63, wv = tid >> 6;
    const int ml = (int)morton2((uint32_t)(l & 7), (uint32_t)(l >> 3));       // its slot in the Morton-ordered LDS tile
    const int64_t ocid = cell0 + ml;
    const bool ook = ocid < ncells;
    const int64_t ob = ocid >> (2 * nbits);
    const uint32_t om = (uint32_t)(ocid & (plane - 1));
    const int64_t opix = (int64_t)compact1by1(om >> 1) * R + compact1by1(om);

    for (int c0 = 0; c0 < C; c0 += kSmChunk) {
        const int c = c0 + 4 * q;
        float best[4] = {-FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX};
        int who[4] = {-1, -1, -1, -1};
        if (c < C) {
            const int live = min(4, C - c);
            for (int n = s; n < e; ++n) {
                const int src = perm[n];
                const float *row = feat + (rowbase + (size_t)src) * ld + c;
                float v[4];
                if (live == 4 && ((ld | c) & 3) == 0) {
                    float4 t = *reinterpret_cast<const float4 *>(row);
                    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = j < live ? row[j] : -FLT_MAX;
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (v[j] > best[j]) { best[j] = v[j]; who[j] = src; }      // strict >: the first point wins, NaN never does
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            s_val[4 * q + j][lc] = who[j] < 0 ? 0.0f : best[j];
            s_arg[4 * q + j][lc] = who[j] < 0 ? N : who[j];                    // untouched cell: value 0, arg = N
        }
        __syncthreads();
        if (ook) {
#pragma unroll
            for (int k = 0; k < kSmChunk / 4; ++k) {
                const int ch = c0 + wv + 4 * k;
                if (ch < C) {
                    const size_t o = ((size_t)ob * C + ch) * plane + opix;
                    val[o] = s_val[wv + 4 * k][ml];
                    arg[o] = (int64_t)s_arg[wv + 4 * k][ml];
                }
            }
        }
        __syncthreads();
    }
}

// gsrc [B, N, C] (point-major, zeroed by the caller): gsrc[b, arg[b, c, p], c] = gval[b, c, p] for every touched cell.  A point
// lies in one cell, so no two elements write the same address: no atomics, deterministic.
__global__ __launch_bounds__(kSmThreads) void scatter_max_bwd_kernel(const float *__restrict__ gval,
                                                                     const int64_t *__restrict__ arg, int64_t total,
                                                                     int64_t plane, int C, int N, float *__restrict__ gsrc) {
    int64_t i = (int64_t)blockIdx.x * kSmThreads + threadIdx.x;
    if (i >= total) return;
    const int64_t a = arg[i];
    if (a < 0 || a >= N) return;
    const int64_t bc = i / plane;
    const int64_t b = bc / C, c = bc - b * C;
    gsrc[((size_t)b * N + (size_t)a) * C + c] = gval[i];
}

}  // namespace t2h

using namespace t2h;

T2H_API int t2h_scatter_max_fwd(const float *feat, int ld, const int32_t *perm, const int32_t *off0, int B, int N,
                                int nbits, int C, float *val, int64_t *arg, t2h_stream_t stream) {
    if (!feat || !perm || !off0 || !val || !arg) return fail(T2H_ERR_ARG, "t2h_scatter_max_fwd: null pointer");
    if (B < 1 || N < 0 || C < 1 || ld < C || nbits < 1 || nbits > 10)
        return fail(T2H_ERR_ARG, "t2h_scatter_max_fwd: bad sizes (B=%d N=%d C=%d ld=%d nbits=%d); equal-N batches only", B, N,
                    C, ld, nbits);
    const int64_t ncells = (int64_t)B << (2 * nbits);
    const unsigned blocks = (unsigned)((ncells + kSmCells - 1) / kSmCells);
    note_kernel("t2h::scatter_max_fwd_kernel");
    hipLaunchKernelGGL(scatter_max_fwd_kernel, dim3(blocks), dim3(kSmThreads), 0, as_stream(stream), feat, ld, perm, off0, N,
                       nbits, ncells, C, val, arg);
    return check_launch("t2h_scatter_max_fwd");
}

T2H_API int t2h_scatter_max_bwd(const float *gval, const int64_t *arg, int B, int C, int N, int64_t cells, float *gsrc,
                                t2h_stream_t stream) {
    if (!gval || !arg || !gsrc) return fail(T2H_ERR_ARG, "t2h_scatter_max_bwd: null pointer");
    if (B < 1 || C < 1 || N < 0 || cells < 1) return fail(T2H_ERR_ARG, "t2h_scatter_max_bwd: bad sizes");
    if (N == 0) return T2H_OK;
    hipError_t e = hipMemsetAsync(gsrc, 0, (size_t)B * N * C * sizeof(float), as_stream(stream));
    if (e != hipSuccess) return fail(T2H_ERR_LAUNCH, "t2h_scatter_max_bwd: memset: %s", hipGetErrorString(e));
    const int64_t total = (int64_t)B * C * cells;
    note_kernel("t2h::scatter_max_bwd_kernel");
    hipLaunchKernelGGL(scatter_max_bwd_kernel, dim3((unsigned)((total + kSmThreads - 1) / kSmThreads)), dim3(kSmThreads), 0,
                       as_stream(stream), gval, arg, total, cells, C, N, gsrc);
    return check_launch("t2h_scatter_max_bwd");
}
