// Small grid-side pieces of the deferred ALTO point update (tomosar2height_amd/deferred.py; reference alto.py:76-88, 130, 255):
//
//   t2h_cell_counts     points per cell of an ALTO level as a row-major [B, r, r] float plane, straight from the tile CSR
//   t2h_mean_bias_fwd   raster[p, :] = acc[p, :] / max(cnt[p], 1) + [cnt[p] > 0] * cvec[:]
//                       -- scatter_mean's division and its "empty cell = 0" rule applied to the per-cell SUMS' product, plus the
//                       composed bias of the deferred features, in one pass instead of three elementwise launches
//   t2h_mean_bias_bwd   dacc[p, :] = g[p, :] / max(cnt[p], 1);  dcvec[:] = sum over the non-empty cells of g[p, :]
//                       (fixed-order two-stage column sum: deterministic, no atomics)
#include "t2h_common.h"
#include "gemm_tile.h"

namespace t2h {
namespace {

__global__ __launch_bounds__(256) void cell_counts_kernel(const int32_t *__restrict__ off0, int B, int nbits, int level,
                                                         float *__restrict__ cnt) {
    const int rbits = nbits - level, r = 1 << rbits;
    const int64_t total = (int64_t)B << (2 * rbits);
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int x = (int)(i & (r - 1)), y = (int)((i >> rbits) & (r - 1));
    const int64_t b = i >> (2 * rbits);
    const size_t obase = ((size_t)b << (2 * nbits)) + ((size_t)morton2((uint32_t)x, (uint32_t)y) << (2 * level));
    cnt[i] = (float)(off0[obase + ((size_t)1 << (2 * level))] - off0[obase]);
}

__global__ __launch_bounds__(256) void mean_bias_fwd_kernel(const float *__restrict__ acc, const float *__restrict__ cnt,
                                                           const float *__restrict__ cvec, int64_t total4, int C4,
                                                           float *__restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= total4) return;
    const int64_t p = t / C4;
    const int c4 = (int)(t % C4);
    const float n = cnt[p];
    const float den = n > 0.f ? n : 1.f, ne = n > 0.f ? 1.f : 0.f;
    const float4 a = reinterpret_cast<const float4 *>(acc)[t];
    const float4 cv = reinterpret_cast<const float4 *>(cvec)[c4];
    float4 o;
    o.x = __fadd_rn(__fdiv_rn(a.x, den), __fmul_rn(ne, cv.x)); o.y = __fadd_rn(__fdiv_rn(a.y, den), __fmul_rn(ne, cv.y));
    o.z = __fadd_rn(__fdiv_rn(a.z, den), __fmul_rn(ne, cv.z)); o.w = __fadd_rn(__fdiv_rn(a.w, den), __fmul_rn(ne, cv.w));
    reinterpret_cast<float4 *>(out)[t] = o;
}

// rows per workgroup of the backward (its column sums go to one slab row): 64, more for long matrices so that at most 256
// slab rows are left for the second stage -- one workgroup per 64 columns sums them, a chain of `slabs / 64` dependent loads
// (65536 rows x 64 columns: 1024 slabs of 64 rows made that chain the whole cost, 19 of 22 us)
static int mb_rows(int64_t P) {
    int64_t r = (P + 255) / 256;
    r = (r + 63) / 64 * 64;
    return (int)(r < 64 ? 64 : (r > 4096 ? 4096 : r));
}
// dacc = g / max(cnt, 1); slab[blockIdx.x][:] = sum over this block's non-empty rows of g.  Threads: C / 4 float4 columns x
// `slots` row slots (rows dealt round-robin); the slots are combined through LDS in slot order -> a fixed summation order.
__global__ __launch_bounds__(256) void mean_bias_bwd_kernel(const float *__restrict__ g, const float *__restrict__ cnt,
                                                           int64_t P, int C, float *__restrict__ dacc,
                                                           float *__restrict__ slab, int kMbRows) {
    __shared__ float4 red[256];
    const int C4 = C / 4;
    const int cols = min(C4, 256), slots = 256 / cols;
    const int col = threadIdx.x % cols, slot = threadIdx.x / cols;
    const int64_t p0 = (int64_t)blockIdx.x * kMbRows;
    const int rows = (int)min((int64_t)kMbRows, P - p0);
    // (uniform trip count: every thread reaches both barriers of every pass, also when C / 4 is not a multiple of `cols`)
    for (int c4base = 0; c4base < C4; c4base += cols) {
        const int c4 = c4base + col;
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        if (slot < slots && c4 < C4) {
            for (int i = slot; i < rows; i += slots) {
                const float n = cnt[p0 + i];
                const float4 v = reinterpret_cast<const float4 *>(g + (size_t)(p0 + i) * C)[c4];
                if (dacc) {
                    const float den = n > 0.f ? n : 1.f;
                    reinterpret_cast<float4 *>(dacc + (size_t)(p0 + i) * C)[c4] =
                        make_float4(__fdiv_rn(v.x, den), __fdiv_rn(v.y, den), __fdiv_rn(v.z, den), __fdiv_rn(v.w, den));
                }
                if (n > 0.f) { s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
            }
        }
        __syncthreads();
        red[threadIdx.x] = s;
        __syncthreads();
        if (slab && slot == 0 && c4 < C4) {
            float4 t = red[col];
            for (int q = 1; q < slots; ++q) {
                const float4 u = red[q * cols + col];
                t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
            }
            reinterpret_cast<float4 *>(slab + (size_t)blockIdx.x * C)[c4] = t;
        }
    }
}

}  // namespace
}  // namespace t2h

using namespace t2h;

T2H_API int t2h_cell_counts(const int32_t *off0, int B, int nbits, int level, float *cnt, t2h_stream_t stream) {
    if (!off0 || !cnt) return fail(T2H_ERR_ARG, "cell_counts: null pointer");
    if (B < 1 || nbits < 1 || nbits > T2H_MAX_NBITS || level < 0 || level > nbits) return fail(T2H_ERR_ARG, "cell_counts: bad level");
    const int64_t total = (int64_t)B << (2 * (nbits - level));
    hipLaunchKernelGGL(cell_counts_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), off0, B, nbits,
                       level, cnt);
    return check_launch("cell_counts");
}

static bool al16(const void *p) { return ((uintptr_t)p & 15) == 0; }

T2H_API int t2h_mean_bias_fwd(const float *acc, const float *cnt, const float *cvec, int64_t P, int C, float *out,
                              t2h_stream_t stream) {
    if (!acc || !cnt || !cvec || !out) return fail(T2H_ERR_ARG, "mean_bias_fwd: null pointer");
    if (P < 0 || C < 4 || C % 4 != 0 || !al16(acc) || !al16(cvec) || !al16(out))
        return fail(T2H_ERR_ARG, "mean_bias_fwd: needs C %% 4 == 0 and 16-byte aligned rows");
    if (P == 0) return T2H_OK;
    const int64_t total4 = P * (C / 4);
    hipLaunchKernelGGL(mean_bias_fwd_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, as_stream(stream), acc, cnt,
                       cvec, total4, C / 4, out);
    return check_launch("mean_bias_fwd");
}

T2H_API size_t t2h_mean_bias_bwd_workspace_bytes(int64_t P, int C) {
    if (P < 1 || C < 1) return 0;
    const int kMbRows = mb_rows(P);
    return (size_t)((P + kMbRows - 1) / kMbRows) * C * sizeof(float);
}

T2H_API int t2h_mean_bias_bwd(const float *g, const float *cnt, int64_t P, int C, float *dacc, float *dcvec, void *workspace,
                              size_t workspace_bytes, t2h_stream_t stream) {
    if (!g || !cnt) return fail(T2H_ERR_ARG, "mean_bias_bwd: null pointer");
    if (P < 1 || C < 4 || C % 4 != 0 || !al16(g) || (dacc && !al16(dacc)) || (dcvec && !al16(dcvec)))
        return fail(T2H_ERR_ARG, "mean_bias_bwd: needs P >= 1, C %% 4 == 0 and 16-byte aligned rows");
    const size_t need = t2h_mean_bias_bwd_workspace_bytes(P, C);
    if (dcvec && (!workspace || workspace_bytes < need || !al16(workspace)))
        return fail(T2H_ERR_WORKSPACE, "mean_bias_bwd: workspace %zu < %zu bytes", workspace_bytes, need);
    if (!dacc && !dcvec) return T2H_OK;
    const int kMbRows = mb_rows(P);
    const int blocks = (int)((P + kMbRows - 1) / kMbRows);
    float *slab = dcvec ? static_cast<float *>(workspace) : nullptr;
    hipLaunchKernelGGL(mean_bias_bwd_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), g, cnt, P, C, dacc, slab, kMbRows);
    if (int rc = check_launch("mean_bias_bwd")) return rc;
    if (!dcvec) return T2H_OK;
    return launch_reduce_slabs(slab, blocks, C, 1, C, C, 0, dcvec, nullptr, nullptr, as_stream(stream));
}
