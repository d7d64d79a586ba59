// DSM mosaic of the test path (reference: generator.py:149-157): per tile
//     dsm[t:t+H, l:l+W] += flip_rows(height) * patch_weight;   weight[t:t+H, l:l+W] += patch_weight
// and at the end  dsm = maximum(dsm / weight, 0)  (0/0 stays NaN = no data).  float64 accumulators as in the
// reference; tiles overlap by half a patch, so the per-tile launches are stream-ordered (one tile at a time).
#include <math.h>

#include "t2h_common.h"

namespace t2h {

__global__ __launch_bounds__(256) void mosaic_accumulate_kernel(const float *__restrict__ height, int H, int W,
                                                                const double *__restrict__ patch_weight,
                                                                double *__restrict__ dsm, double *__restrict__ weight,
                                                                int rows, int cols, int t_row, int l_col, int flip) {
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    int r = t_row + y, c = l_col + x;
    if (r < 0 || r >= rows || c < 0 || c >= cols) return;
    int sy = flip ? H - 1 - y : y;                        // generator.py:147 .flip(1) of the [1,H,W,1] output
    double w = patch_weight[(size_t)y * W + x];
    size_t o = (size_t)r * cols + c;
    dsm[o] += (double)height[(size_t)sy * W + x] * w;
    weight[o] += w;
}

__global__ __launch_bounds__(256) void mosaic_finalize_kernel(double *__restrict__ dsm, const double *__restrict__ weight,
                                                              long long n) {
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double v = dsm[i] / weight[i];
    dsm[i] = isnan(v) ? v : fmax(v, 0.0);                 // torch.maximum propagates NaN
}

}  // namespace t2h

using namespace t2h;

T2H_API int t2h_mosaic_accumulate(const float *height, int H, int W, const double *patch_weight, double *dsm,
                                  double *weight, int rows, int cols, int t_row, int l_col, int flip_rows,
                                  t2h_stream_t stream) {
    if (!height || !patch_weight || !dsm || !weight) return fail(T2H_ERR_ARG, "mosaic_accumulate: null pointer");
    if (H < 1 || W < 1 || rows < 1 || cols < 1 || H > 65535) return fail(T2H_ERR_ARG, "mosaic_accumulate: bad shape");
    hipLaunchKernelGGL(mosaic_accumulate_kernel, dim3((W + 255) / 256, H), dim3(256), 0, as_stream(stream), height, H, W,
                       patch_weight, dsm, weight, rows, cols, t_row, l_col, flip_rows);
    return check_launch("mosaic_accumulate");
}

T2H_API int t2h_mosaic_finalize(double *dsm, const double *weight, int64_t n, t2h_stream_t stream) {
    if (!dsm || !weight || n < 0) return fail(T2H_ERR_ARG, "mosaic_finalize: bad argument");
    if (n == 0) return T2H_OK;
    hipLaunchKernelGGL(mosaic_finalize_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), dsm,
                       weight, (long long)n);
    return check_launch("mosaic_finalize");
}
