// Grid-side kernels of the "grid-feature decoder" (decoder/pixel.py:94-125) and layout glue.
//
//   upsample_bilinear   F.interpolate(size, mode='bilinear', align_corners=True)   pixel.py:107,110
//   nchw <-> nhwc       [B,C,P] <-> [B,P,C] through a padded LDS tile
//
// Pure HBM streaming: x-fastest thread mapping for coalesced NCHW rows; the backward is a gather over the
// (at most ~2/scale + 2) output rows/cols that reference an input pixel, so it needs no atomics.
#include "t2h_common.h"

namespace t2h {

// ATen upsample_bilinear2d index math (align_corners=True): src = scale*dst, i0 = (int)src,
// i1 = i0 + (i0 < in-1), lambda1 = src - i0, lambda0 = 1 - lambda1.
struct Lerp { int i0, i1; float l0, l1; };
__device__ inline Lerp lerp_index(float scale, int dst, int in_size) {
    float real = __fmul_rn(scale, (float)dst);
    int a = min((int)real, in_size - 1);
    Lerp L;
    L.i0 = a;
    L.i1 = a + (a < in_size - 1 ? 1 : 0);
    float lam = fminf(fmaxf(__fsub_rn(real, (float)a), 0.0f), 1.0f);
    L.l1 = lam;
    L.l0 = __fsub_rn(1.0f, lam);
    return L;
}

__global__ __launch_bounds__(256) void upsample_fwd_kernel(const float *__restrict__ in, const float *__restrict__ addend,
                                                          int h, int w, int H, int W, float sh, float sw,
                                                          float *__restrict__ out) {
    int x = blockIdx.x * blockDim.x + threadIdx.x;
    int y = blockIdx.y;
    size_t bc = blockIdx.z;
    if (x >= W) return;
    Lerp ly = lerp_index(sh, y, h), lx = lerp_index(sw, x, w);
    const float *src = in + bc * h * w;
    float top = __fadd_rn(__fmul_rn(lx.l0, src[(size_t)ly.i0 * w + lx.i0]), __fmul_rn(lx.l1, src[(size_t)ly.i0 * w + lx.i1]));
    float bot = __fadd_rn(__fmul_rn(lx.l0, src[(size_t)ly.i1 * w + lx.i0]), __fmul_rn(lx.l1, src[(size_t)ly.i1 * w + lx.i1]));
    float v = __fadd_rn(__fmul_rn(ly.l0, top), __fmul_rn(ly.l1, bot));
    size_t o = (bc * H + y) * W + x;
    if (addend) v = __fadd_rn(v, addend[o]);
    out[o] = v;
}

__device__ inline void source_range(float scale, int i, int out_size, int &lo, int &hi) {
    // outputs whose taps can touch input index i satisfy scale*dst in (i-1, i+1); be generous by one.
    if (scale <= 0.0f) { lo = 0; hi = out_size - 1; return; }
    lo = max(0, (int)floorf((float)(i - 1) / scale) - 1);
    hi = min(out_size - 1, (int)ceilf((float)(i + 1) / scale) + 1);
}

__global__ __launch_bounds__(256) void upsample_bwd_kernel(const float *__restrict__ gout, int h, int w, int H, int W,
                                                          float sh, float sw, float *__restrict__ gin) {
    int ix = blockIdx.x * blockDim.x + threadIdx.x;
    int iy = blockIdx.y;
    size_t bc = blockIdx.z;
    if (ix >= w) return;
    int ylo, yhi, xlo, xhi;
    source_range(sh, iy, H, ylo, yhi);
    source_range(sw, ix, W, xlo, xhi);
    const float *src = gout + bc * H * W;
    float acc = 0.0f;
    for (int y = ylo; y <= yhi; ++y) {
        Lerp ly = lerp_index(sh, y, h);
        float wy0 = ly.i0 == iy ? ly.l0 : 0.0f, wy1 = ly.i1 == iy ? ly.l1 : 0.0f;
        if (ly.i0 != iy && ly.i1 != iy) continue;
        for (int x = xlo; x <= xhi; ++x) {
            Lerp lx = lerp_index(sw, x, w);
            if (lx.i0 != ix && lx.i1 != ix) continue;
            float wx0 = lx.i0 == ix ? lx.l0 : 0.0f, wx1 = lx.i1 == ix ? lx.l1 : 0.0f;
            float g = src[(size_t)y * W + x];
            // same four products ATen's backward scatters: l_y * l_x * g per (tap_y, tap_x)
            if (ly.i0 == iy && lx.i0 == ix) acc = __fadd_rn(acc, __fmul_rn(__fmul_rn(wy0, wx0), g));
            if (ly.i0 == iy && lx.i1 == ix) acc = __fadd_rn(acc, __fmul_rn(__fmul_rn(wy0, wx1), g));
            if (ly.i1 == iy && lx.i0 == ix) acc = __fadd_rn(acc, __fmul_rn(__fmul_rn(wy1, wx0), g));
            if (ly.i1 == iy && lx.i1 == ix) acc = __fadd_rn(acc, __fmul_rn(__fmul_rn(wy1, wx1), g));
        }
    }
    gin[(bc * h + iy) * w + ix] = acc;
}

// [B, rows, cols] -> [B, cols, rows] through a 32x33 LDS tile.
__global__ __launch_bounds__(256) void transpose_kernel(const float *__restrict__ in, int rows, int cols,
                                                       float *__restrict__ out) {
    __shared__ float tile[32][33];
    size_t b = blockIdx.z;
    int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    const float *src = in + b * rows * cols;
    float *dst = out + b * rows * cols;
#pragma unroll
    for (int k = 0; k < 32; k += 8) {
        int r = r0 + ty + k, c = c0 + tx;
        if (r < rows && c < cols) tile[ty + k][tx] = src[(size_t)r * cols + c];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 32; k += 8) {
        int c = c0 + ty + k, r = r0 + tx;
        if (r < rows && c < cols) dst[(size_t)c * rows + r] = tile[tx][ty + k];
    }
}

}  // namespace t2h

using namespace t2h;

static int check_up(const char *what, int B, int C, int h, int w, int H, int W) {
    if (B < 1 || C < 1 || h < 1 || w < 1 || H < 1 || W < 1 || (int64_t)B * C > 65535 || H > 65535)
        return fail(T2H_ERR_ARG, "%s: unsupported shape (B=%d C=%d %dx%d -> %dx%d)", what, B, C, h, w, H, W);
    return T2H_OK;
}

T2H_API int t2h_upsample_bilinear_fwd(const float *in, const float *addend, int B, int C, int h, int w, int H, int W,
                                      float *out, t2h_stream_t stream) {
    if (!in || !out) return fail(T2H_ERR_ARG, "upsample_bilinear_fwd: null pointer");
    int rc = check_up("upsample_bilinear_fwd", B, C, h, w, H, W);
    if (rc) return rc;
    float sh = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.0f;
    float sw = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.0f;
    hipLaunchKernelGGL(upsample_fwd_kernel, dim3((W + 255) / 256, H, B * C), dim3(256), 0, as_stream(stream), in, addend, h,
                       w, H, W, sh, sw, out);
    return check_launch("upsample_bilinear_fwd");
}

T2H_API int t2h_upsample_bilinear_bwd(const float *gout, int B, int C, int h, int w, int H, int W, float *gin,
                                      t2h_stream_t stream) {
    if (!gout || !gin) return fail(T2H_ERR_ARG, "upsample_bilinear_bwd: null pointer");
    int rc = check_up("upsample_bilinear_bwd", B, C, h, w, H, W);
    if (rc) return rc;
    if (h > 65535) return fail(T2H_ERR_ARG, "upsample_bilinear_bwd: h too large");
    float sh = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.0f;
    float sw = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.0f;
    hipLaunchKernelGGL(upsample_bwd_kernel, dim3((w + 255) / 256, h, B * C), dim3(256), 0, as_stream(stream), gout, h, w, H,
                       W, sh, sw, gin);
    return check_launch("upsample_bilinear_bwd");
}

static int transpose_launch(const char *what, const float *in, int B, int rows, int cols, float *out, t2h_stream_t stream) {
    if (!in || !out) return fail(T2H_ERR_ARG, "%s: null pointer", what);
    if (B < 1 || rows < 1 || cols < 1 || B > 65535 || (rows + 31) / 32 > 65535)
        return fail(T2H_ERR_ARG, "%s: unsupported shape", what);
    hipLaunchKernelGGL(transpose_kernel, dim3((cols + 31) / 32, (rows + 31) / 32, B), dim3(256), 0, as_stream(stream), in,
                       rows, cols, out);
    return check_launch(what);
}

T2H_API int t2h_nchw_to_nhwc(const float *in, int B, int C, int P, float *out, t2h_stream_t stream) {
    return transpose_launch("nchw_to_nhwc", in, B, C, P, out, stream);
}
T2H_API int t2h_nhwc_to_nchw(const float *in, int B, int C, int P, float *out, t2h_stream_t stream) {
    return transpose_launch("nhwc_to_nchw", in, B, P, C, out, stream);
}
