// sample_mode = 'bicubic' (r06): the one other interpolation torch accepts at the reference's two call sites
//
//   F.interpolate(c, size, mode='bicubic', align_corners=True)                                   decoder/pixel.py:107,110
//   F.grid_sample(c, vgrid, padding_mode='border', align_corners=True, mode='bicubic')           encoder/alto.py:95,204
//
// No shipped config selects it (tomosar2height.yaml:27 says bilinear), so these are plain kernels: one thread per output
// element, cubic convolution with A = -0.75 as ATen's upsample_bicubic2d / grid_sampler_2d, taps outside the plane clamped
// to the border pixel (upsample_get_value_bounded; grid_sampler's get_value_bounded with padding_mode = border).  The
// resampling backward is a gather (no atomics, deterministic); the point-sampling backward adds with atomics like
// t2h_sample_bwd_atomic (the order of the additions is not reproducible: documented in include/t2h.h).
// Layout: element (b, c, y, x) of a plane at b * sb + c * sc + y * sy + x * sx -- NCHW (sc = h w, sy = w, sx = 1) or NHWC
// (sc = 1, sy = w C, sx = C), chosen by `channels_last`.
#include "t2h_common.h"

namespace t2h {
namespace {

__device__ inline float cc1(float x) { const float A = -0.75f; return ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f; }
__device__ inline float cc2(float x) { const float A = -0.75f; return ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A; }
__device__ inline void cubic_coeffs(float t, float (&c)[4]) {
    c[0] = cc2(t + 1.f); c[1] = cc1(t); c[2] = cc1(1.f - t); c[3] = cc2(2.f - t);
}
__device__ inline int clampi(int v, int lo, int hi) { return min(max(v, lo), hi); }

struct Lay { long long sb, sc, sy, sx; };
__host__ __device__ inline Lay layout(int C, int h, int w, int cl) {
    Lay l;
    l.sb = (long long)C * h * w;
    if (cl) { l.sc = 1; l.sx = C; l.sy = (long long)w * C; }
    else { l.sc = (long long)h * w; l.sx = 1; l.sy = w; }
    return l;
}

// one thread per output element; the fastest index follows the layout (x for NCHW, c for NHWC)
__global__ __launch_bounds__(256) void upsample_bicubic_fwd_kernel(const float *__restrict__ in, const float *__restrict__ addend,
                                                                  int B, int C, int h, int w, int H, int W, float sh, float sw,
                                                                  int cl, float *__restrict__ out) {
    const long long total = (long long)B * C * H * W;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    int b, c, y, x;
    if (cl) { c = (int)(t % C); x = (int)((t / C) % W); y = (int)((t / ((long long)C * W)) % H); b = (int)(t / ((long long)C * W * H)); }
    else { x = (int)(t % W); y = (int)((t / W) % H); c = (int)((t / ((long long)W * H)) % C); b = (int)(t / ((long long)W * H * C)); }
    const Lay li = layout(C, h, w, cl), lo = layout(C, H, W, cl);
    const float ry = sh * (float)y, rx = sw * (float)x;
    const int iy = (int)floorf(ry), ix = (int)floorf(rx);
    float cy[4], cx[4];
    cubic_coeffs(ry - (float)iy, cy);
    cubic_coeffs(rx - (float)ix, cx);
    const float *src = in + b * li.sb + c * li.sc;
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float *row = src + clampi(iy - 1 + i, 0, h - 1) * li.sy;
        float r = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) r += row[clampi(ix - 1 + j, 0, w - 1) * li.sx] * cx[j];
        acc += r * cy[i];
    }
    const long long o = b * lo.sb + c * lo.sc + y * lo.sy + x * lo.sx;
    out[o] = addend ? acc + addend[o] : acc;
}

// gather form of the adjoint: input pixel (iy, ix) collects from every output pixel one of whose (clamped) taps is that pixel
__device__ inline void out_range(float scale, int i, int out_size, int &lo, int &hi) {
    if (scale <= 0.f) { lo = 0; hi = out_size - 1; return; }
    lo = max(0, (int)floorf((float)(i - 2) / scale) - 1);          // floor(scale * dst) in [i - 2, i + 1], generous by one
    hi = min(out_size - 1, (int)ceilf((float)(i + 2) / scale) + 1);
}
__device__ inline float tap_weight(float scale, int dst, int in_size, int i) {      // sum of dst's coefficients whose clamped tap is i
    const float r = scale * (float)dst;
    const int f = (int)floorf(r);
    if (f < i - 2 || f > i + 2) return 0.f;
    float c[4];
    cubic_coeffs(r - (float)f, c);
    float wsum = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) wsum += clampi(f - 1 + k, 0, in_size - 1) == i ? c[k] : 0.f;
    return wsum;
}

__global__ __launch_bounds__(256) void upsample_bicubic_bwd_kernel(const float *__restrict__ gout, int B, int C, int h, int w, int H,
                                                                  int W, float sh, float sw, int cl, float *__restrict__ gin) {
    const long long total = (long long)B * C * h * w;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    int b, c, iy, ix;
    if (cl) { c = (int)(t % C); ix = (int)((t / C) % w); iy = (int)((t / ((long long)C * w)) % h); b = (int)(t / ((long long)C * w * h)); }
    else { ix = (int)(t % w); iy = (int)((t / w) % h); c = (int)((t / ((long long)w * h)) % C); b = (int)(t / ((long long)w * h * C)); }
    const Lay li = layout(C, h, w, cl), lo = layout(C, H, W, cl);
    int ylo, yhi, xlo, xhi;
    out_range(sh, iy, H, ylo, yhi);
    out_range(sw, ix, W, xlo, xhi);
    const float *src = gout + b * lo.sb + c * lo.sc;
    float acc = 0.f;
    for (int y = ylo; y <= yhi; ++y) {
        const float wy = tap_weight(sh, y, h, iy);
        if (wy == 0.f) continue;
        float r = 0.f;
        for (int x = xlo; x <= xhi; ++x) {
            const float wx = tap_weight(sw, x, w, ix);
            if (wx != 0.f) r += src[y * lo.sy + x * lo.sx] * wx;
        }
        acc += r * wy;
    }
    gin[b * li.sb + c * li.sc + iy * li.sy + ix * li.sx] = acc;
}

// points [B * N, dim] with x, y in the plane's [0, 1] coordinates (the reference's vgrid = 2 xy - 1 with align_corners=True:
// pixel coordinate = xy (r - 1)); out [B * N, C] point-major.  One thread per (point, channel).
__global__ __launch_bounds__(256) void sample_bicubic_fwd_kernel(const float *__restrict__ plane, const float *__restrict__ pts, int dim,
                                                                long long n_pts, int n_per_b, int r, int C, int cl,
                                                                float *__restrict__ out) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_pts * C) return;
    const long long n = t / C;
    const int c = (int)(t % C), b = (int)(n / n_per_b);
    const Lay l = layout(C, r, r, cl);
    const float fx = pts[n * dim] * (float)(r - 1), fy = pts[n * dim + 1] * (float)(r - 1);
    const int ix = (int)floorf(fx), iy = (int)floorf(fy);
    float cx[4], cy[4];
    cubic_coeffs(fx - (float)ix, cx);
    cubic_coeffs(fy - (float)iy, cy);
    const float *src = plane + b * l.sb + c * l.sc;
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float *row = src + clampi(iy - 1 + i, 0, r - 1) * l.sy;
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) s += row[clampi(ix - 1 + j, 0, r - 1) * l.sx] * cx[j];
        acc += s * cy[i];
    }
    out[t] = acc;
}

__global__ __launch_bounds__(256) void sample_bicubic_bwd_kernel(const float *__restrict__ gout, const float *__restrict__ pts, int dim,
                                                                long long n_pts, int n_per_b, int r, int C, int cl,
                                                                float *__restrict__ gplane) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_pts * C) return;
    const long long n = t / C;
    const int c = (int)(t % C), b = (int)(n / n_per_b);
    const Lay l = layout(C, r, r, cl);
    const float fx = pts[n * dim] * (float)(r - 1), fy = pts[n * dim + 1] * (float)(r - 1);
    const int ix = (int)floorf(fx), iy = (int)floorf(fy);
    float cx[4], cy[4];
    cubic_coeffs(fx - (float)ix, cx);
    cubic_coeffs(fy - (float)iy, cy);
    float *dst = gplane + b * l.sb + c * l.sc;
    const float g = gout[t];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            atomicAdd(dst + clampi(iy - 1 + i, 0, r - 1) * l.sy + clampi(ix - 1 + j, 0, r - 1) * l.sx, g * cx[j] * cy[i]);
}

// mode = 'nearest' (grid_sampler_2d: coordinate clipped to the plane, then nearbyint -- round half to even)
template <bool BWD>
__global__ __launch_bounds__(256) void sample_nearest_kernel(const float *__restrict__ src, const float *__restrict__ pts, int dim,
                                                            long long n_pts, int n_per_b, int r, int C, int cl, float *__restrict__ dst) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_pts * C) return;
    const long long n = t / C;
    const int c = (int)(t % C), b = (int)(n / n_per_b);
    const Lay l = layout(C, r, r, cl);
    const float fx = fminf(fmaxf(pts[n * dim] * (float)(r - 1), 0.f), (float)(r - 1));
    const float fy = fminf(fmaxf(pts[n * dim + 1] * (float)(r - 1), 0.f), (float)(r - 1));
    const long long at = b * l.sb + c * l.sc + (long long)rintf(fy) * l.sy + (long long)rintf(fx) * l.sx;
    if (BWD) atomicAdd(dst + at, src[t]);
    else dst[t] = src[at];
}

}  // namespace
}  // namespace t2h

using namespace t2h;

T2H_API int t2h_sample_nearest_fwd(const float *plane, const float *pts, int dim, int B, int64_t N, int r, int C, int channels_last,
                                   float *out, t2h_stream_t stream) {
    if (!plane || !pts || !out) return fail(T2H_ERR_ARG, "sample_nearest_fwd: null pointer");
    if (B < 1 || N < 0 || r < 1 || C < 1 || dim < 2) return fail(T2H_ERR_ARG, "sample_nearest_fwd: bad shape");
    if (N == 0) return T2H_OK;
    const long long total = (long long)B * N * C;
    hipLaunchKernelGGL(sample_nearest_kernel<false>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), plane, pts,
                       dim, (long long)B * N, (int)N, r, C, channels_last, out);
    return check_launch("sample_nearest_fwd");
}

T2H_API int t2h_sample_nearest_bwd(const float *gout, const float *pts, int dim, int B, int64_t N, int r, int C, int channels_last,
                                   float *gplane, t2h_stream_t stream) {
    if (!gout || !pts || !gplane) return fail(T2H_ERR_ARG, "sample_nearest_bwd: null pointer");
    if (B < 1 || N < 0 || r < 1 || C < 1 || dim < 2) return fail(T2H_ERR_ARG, "sample_nearest_bwd: bad shape");
    if (hipMemsetAsync(gplane, 0, (size_t)B * C * r * r * sizeof(float), as_stream(stream)) != hipSuccess)
        return fail(T2H_ERR_LAUNCH, "sample_nearest_bwd: memset failed");
    if (N == 0) return T2H_OK;
    const long long total = (long long)B * N * C;
    hipLaunchKernelGGL(sample_nearest_kernel<true>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), gout, pts, dim,
                       (long long)B * N, (int)N, r, C, channels_last, gplane);
    return check_launch("sample_nearest_bwd");
}

static int check_planes(const char *what, int B, int C, int h, int w, int H, int W) {
    if (B < 1 || C < 1 || h < 1 || w < 1 || H < 1 || W < 1) return fail(T2H_ERR_ARG, "%s: bad shape", what);
    if ((long long)B * C * H * W >= ((long long)1 << 40)) return fail(T2H_ERR_ARG, "%s: too many elements", what);
    return T2H_OK;
}

T2H_API int t2h_upsample_bicubic_fwd(const float *in, const float *addend, int B, int C, int h, int w, int H, int W, int channels_last,
                                     float *out, t2h_stream_t stream) {
    if (!in || !out) return fail(T2H_ERR_ARG, "upsample_bicubic_fwd: null pointer");
    if (int rc = check_planes("upsample_bicubic_fwd", B, C, h, w, H, W)) return rc;
    const float sh = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f, sw = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
    const long long total = (long long)B * C * H * W;
    hipLaunchKernelGGL(upsample_bicubic_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), in, addend,
                       B, C, h, w, H, W, sh, sw, channels_last, out);
    return check_launch("upsample_bicubic_fwd");
}

T2H_API int t2h_upsample_bicubic_bwd(const float *gout, int B, int C, int h, int w, int H, int W, int channels_last, float *gin,
                                     t2h_stream_t stream) {
    if (!gout || !gin) return fail(T2H_ERR_ARG, "upsample_bicubic_bwd: null pointer");
    if (int rc = check_planes("upsample_bicubic_bwd", B, C, h, w, H, W)) return rc;
    const float sh = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f, sw = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
    const long long total = (long long)B * C * h * w;
    hipLaunchKernelGGL(upsample_bicubic_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), gout, B, C,
                       h, w, H, W, sh, sw, channels_last, gin);
    return check_launch("upsample_bicubic_bwd");
}

T2H_API int t2h_sample_bicubic_fwd(const float *plane, const float *pts, int dim, int B, int64_t N, int r, int C, int channels_last,
                                   float *out, t2h_stream_t stream) {
    if (!plane || !pts || !out) return fail(T2H_ERR_ARG, "sample_bicubic_fwd: null pointer");
    if (B < 1 || N < 0 || r < 1 || C < 1 || dim < 2) return fail(T2H_ERR_ARG, "sample_bicubic_fwd: bad shape");
    if (N == 0) return T2H_OK;
    const long long total = (long long)B * N * C;
    hipLaunchKernelGGL(sample_bicubic_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), plane, pts,
                       dim, (long long)B * N, (int)N, r, C, channels_last, out);
    return check_launch("sample_bicubic_fwd");
}

T2H_API int t2h_sample_bicubic_bwd(const float *gout, const float *pts, int dim, int B, int64_t N, int r, int C, int channels_last,
                                   float *gplane, t2h_stream_t stream) {
    if (!gout || !pts || !gplane) return fail(T2H_ERR_ARG, "sample_bicubic_bwd: null pointer");
    if (B < 1 || N < 0 || r < 1 || C < 1 || dim < 2) return fail(T2H_ERR_ARG, "sample_bicubic_bwd: bad shape");
    if (hipMemsetAsync(gplane, 0, (size_t)B * C * r * r * sizeof(float), as_stream(stream)) != hipSuccess)
        return fail(T2H_ERR_LAUNCH, "sample_bicubic_bwd: memset failed");
    if (N == 0) return T2H_OK;
    const long long total = (long long)B * N * C;
    hipLaunchKernelGGL(sample_bicubic_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), gout, pts, dim,
                       (long long)B * N, (int)N, r, C, channels_last, gplane);
    return check_launch("sample_bicubic_bwd");
}
