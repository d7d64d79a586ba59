// 3x3 stride-1 padding-1 convolution with a handful of input channels (the image U-Net's first layer,
// reference encoder/unet.py:112-187: Conv2d(3, 32, 3, padding=1) at 512 x 512) -- the one convolution of the image
// configs (BASELINE.json configs[2] / [4]) that the implicit-GEMM kernels of conv.hip do not take (their reduction runs
// in 16-channel slabs).  With K = 9 * Cin = 27 it is HBM bound (3 MB in, 33.5 MB out at 512^2): plain VALU kernels on
// 16 x 16 pixel tiles, the 18 x 18 x Cin input patch and the weights in LDS.  NHWC activations, weights
// [Cout][3][3][Cin] (the channels_last memory of an nn.Conv2d weight), like conv.hip.
//
//   forward   y[p][co] = act(b[co] + sum_{tap, ci} x[p + tap][ci] * w[co][tap][ci])       (fma chain: tap-major, ci inner)
//   dgrad     dx[p][ci] = sum_{tap, co} dy[p - tap][co] * w[co][tap][ci]                   (only if the input needs it)
//   wgrad     dw[co][tap][ci] = sum_p dy[p][co] * x[p + tap][ci],  db[co] = sum_p dy[p][co]
//             per-workgroup slabs over a fixed pixel-tile schedule, added in slab order: deterministic, no atomics
#include "t2h_common.h"

namespace t2h {
namespace {

constexpr int TS = 16;                 // pixel tile side
constexpr int kMaxCin = 8, kMaxCout = 64;
constexpr int kWgradWgs = 256;         // slabs of the weight gradient

struct SmallArgs {
    const float *x, *w, *bias, *dy;
    float *y, *dx, *slab;
    int B, H, W, Cin, Cout, relu, accumulate;
};

// the (TS + 2)^2 x Cin input patch of tile (b, ty, tx) -> LDS, zero outside the image
__device__ inline void load_patch(const float *__restrict__ x, float *patch, int b, int y0, int x0, int H, int W, int Cin, int tid) {
    const int n = (TS + 2) * (TS + 2) * Cin;
    for (int i = tid; i < n; i += 256) {
        const int ci = i % Cin, pix = i / Cin, px = pix % (TS + 2), py = pix / (TS + 2);
        const int gy = y0 - 1 + py, gx = x0 - 1 + px;
        patch[i] = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? x[(((size_t)b * H + gy) * W + gx) * Cin + ci] : 0.0f;
    }
}

__global__ __launch_bounds__(256) void conv_small_fwd_kernel(SmallArgs a) {
    __shared__ float patch[(TS + 2) * (TS + 2) * kMaxCin];
    __shared__ __attribute__((aligned(16))) float wt[9 * kMaxCin * kMaxCout];      // [tap * Cin + ci][co]
    __shared__ float bs[kMaxCout];
    const int tid = threadIdx.x, K = 9 * a.Cin;
    const int tiles_x = (a.W + TS - 1) / TS, tiles_y = (a.H + TS - 1) / TS;
    const int b = blockIdx.x / (tiles_x * tiles_y), t = blockIdx.x % (tiles_x * tiles_y);
    const int y0 = (t / tiles_x) * TS, x0 = (t % tiles_x) * TS;
    for (int i = tid; i < K * a.Cout; i += 256) { const int co = i / K, k = i % K; wt[k * a.Cout + co] = a.w[i]; }
    if (tid < a.Cout) bs[tid] = a.bias ? a.bias[tid] : 0.0f;
    load_patch(a.x, patch, b, y0, x0, a.H, a.W, a.Cin, tid);
    __syncthreads();
    const int py = tid / TS, px = tid % TS, gy = y0 + py, gx = x0 + px;
    if (gy >= a.H || gx >= a.W) return;
    float *dst = a.y + (((size_t)b * a.H + gy) * a.W + gx) * a.Cout;
    for (int co = 0; co < a.Cout; co += 4) {
        float4 acc = make_float4(bs[co], bs[co + 1], bs[co + 2], bs[co + 3]);
        for (int tap = 0; tap < 9; ++tap) {
            const float *pp = patch + ((py + tap / 3) * (TS + 2) + px + tap % 3) * a.Cin;
            for (int ci = 0; ci < a.Cin; ++ci) {
                const float xv = pp[ci];
                const float4 wv = *reinterpret_cast<const float4 *>(wt + (tap * a.Cin + ci) * a.Cout + co);
                acc.x = fmaf(xv, wv.x, acc.x); acc.y = fmaf(xv, wv.y, acc.y);
                acc.z = fmaf(xv, wv.z, acc.z); acc.w = fmaf(xv, wv.w, acc.w);
            }
        }
        if (a.relu) { acc.x = fmaxf(acc.x, 0.f); acc.y = fmaxf(acc.y, 0.f); acc.z = fmaxf(acc.z, 0.f); acc.w = fmaxf(acc.w, 0.f); }
        *reinterpret_cast<float4 *>(dst + co) = acc;
    }
}

// one thread per input pixel: dx[p][ci] = sum over taps, co of dy[p - tap][co] * w[co][tap][ci]
__global__ __launch_bounds__(256) void conv_small_dgrad_kernel(SmallArgs a) {
    __shared__ float wt[9 * kMaxCin * kMaxCout];                                  // [co][tap][ci] as stored
    const int K = 9 * a.Cin;
    for (int i = threadIdx.x; i < K * a.Cout; i += 256) wt[i] = a.w[i];
    __syncthreads();
    const long long p = (long long)blockIdx.x * 256 + threadIdx.x, P = (long long)a.B * a.H * a.W;
    if (p >= P) return;
    const int gx = (int)(p % a.W), gy = (int)((p / a.W) % a.H), b = (int)(p / ((long long)a.W * a.H));
    float acc[kMaxCin];
#pragma unroll
    for (int ci = 0; ci < kMaxCin; ++ci) acc[ci] = 0.0f;
    for (int tap = 0; tap < 9; ++tap) {
        const int sy = gy - (tap / 3 - 1), sx = gx - (tap % 3 - 1);              // the output pixel that saw p through `tap`
        if (sy < 0 || sy >= a.H || sx < 0 || sx >= a.W) continue;
        const float *g = a.dy + (((size_t)b * a.H + sy) * a.W + sx) * a.Cout;
        for (int co = 0; co < a.Cout; ++co) {
            const float gv = g[co];
#pragma unroll
            for (int ci = 0; ci < kMaxCin; ++ci)
                if (ci < a.Cin) acc[ci] = fmaf(gv, wt[co * K + tap * a.Cin + ci], acc[ci]);
        }
    }
    float *dst = a.dx + (size_t)p * a.Cin;
#pragma unroll
    for (int ci = 0; ci < kMaxCin; ++ci)
        if (ci < a.Cin) dst[ci] = a.accumulate ? dst[ci] + acc[ci] : acc[ci];
}

// slab[wg][co * K + k] (k = tap * Cin + ci), then colslab[wg][co] behind all matrix slabs; tiles wg, wg + gridDim.x, ...
__global__ __launch_bounds__(256) void conv_small_wgrad_kernel(SmallArgs a) {
    __shared__ float patch[(TS + 2) * (TS + 2) * kMaxCin];
    __shared__ __attribute__((aligned(16))) float gt[TS * TS * kMaxCout];         // dy tile [pixel][co]
    const int tid = threadIdx.x, K = 9 * a.Cin, co4n = a.Cout / 4;
    const int tiles_x = (a.W + TS - 1) / TS, tiles_y = (a.H + TS - 1) / TS, n_tiles = a.B * tiles_x * tiles_y;
    // thread -> (4 output channels, one (tap, ci)) pairs: K * Cout / 4 <= 72 * 16 = 1152, up to kPairs per thread
    constexpr int kPairs = (9 * kMaxCin * (kMaxCout / 4) + 255) / 256;
    float4 acc[kPairs];
#pragma unroll
    for (int u = 0; u < kPairs; ++u) acc[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    float bacc = 0.0f;
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int b = tile / (tiles_x * tiles_y), t = tile % (tiles_x * tiles_y);
        const int y0 = (t / tiles_x) * TS, x0 = (t % tiles_x) * TS;
        __syncthreads();
        load_patch(a.x, patch, b, y0, x0, a.H, a.W, a.Cin, tid);
        for (int i = tid; i < TS * TS * co4n; i += 256) {
            const int pix = i / co4n, c4 = i % co4n, gy = y0 + pix / TS, gx = x0 + pix % TS;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gy < a.H && gx < a.W) v = *reinterpret_cast<const float4 *>(a.dy + (((size_t)b * a.H + gy) * a.W + gx) * a.Cout + c4 * 4);
            *reinterpret_cast<float4 *>(gt + pix * a.Cout + c4 * 4) = v;
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < kPairs; ++u) {
            const int pair = tid + u * 256;
            if (pair >= K * co4n) continue;
            const int c4 = pair / K, k = pair % K, tap = k / a.Cin, ci = k % a.Cin;
            float4 s = acc[u];
            for (int pix = 0; pix < TS * TS; ++pix) {                            // pixels in raster order: fixed summation order
                const float xv = patch[((pix / TS + tap / 3) * (TS + 2) + pix % TS + tap % 3) * a.Cin + ci];
                const float4 g = *reinterpret_cast<const float4 *>(gt + pix * a.Cout + c4 * 4);
                s.x = fmaf(g.x, xv, s.x); s.y = fmaf(g.y, xv, s.y); s.z = fmaf(g.z, xv, s.z); s.w = fmaf(g.w, xv, s.w);
            }
            acc[u] = s;
        }
        if (tid < a.Cout)
            for (int pix = 0; pix < TS * TS; ++pix) bacc += gt[pix * a.Cout + tid];
    }
    float *slab = a.slab + (size_t)blockIdx.x * a.Cout * K;
    float *colslab = a.slab + (size_t)gridDim.x * a.Cout * K + (size_t)blockIdx.x * a.Cout;
#pragma unroll
    for (int u = 0; u < kPairs; ++u) {
        const int pair = tid + u * 256;
        if (pair >= K * co4n) continue;
        const int c4 = pair / K, k = pair % K;
        slab[(c4 * 4 + 0) * K + k] = acc[u].x; slab[(c4 * 4 + 1) * K + k] = acc[u].y;
        slab[(c4 * 4 + 2) * K + k] = acc[u].z; slab[(c4 * 4 + 3) * K + k] = acc[u].w;
    }
    if (tid < a.Cout) colslab[tid] = bacc;
}

int check_small(const char *what, int B, int H, int W, int Cin, int Cout) {
    if (B < 1 || H < 1 || W < 1 || Cin < 1 || Cout < 1) return fail(T2H_ERR_ARG, "%s: bad shape", what);
    if (Cin > kMaxCin || Cout > kMaxCout || Cout % 4) return fail(T2H_ERR_ARG, "%s: needs Cin <= %d, Cout <= %d, Cout %% 4 == 0 (got %d, %d)", what, kMaxCin, kMaxCout, Cin, Cout);
    if ((long long)B * H * W > (1LL << 30)) return fail(T2H_ERR_ARG, "%s: more than 2^30 pixels", what);
    return T2H_OK;
}

}  // namespace

int launch_reduce_slabs(const float *slabs, int splits, long long stride, int rows, int cols, int ld_out, int accumulate,
                        float *out, const float *col_slabs, float *col_out, hipStream_t s, int col_splits = 0, int col_rows = 0,
                        bool deferrable = false);

}  // namespace t2h

using namespace t2h;

T2H_API int t2h_conv3x3_smallcin_fwd(const float *x, const float *w, const float *bias, float *y, int B, int H, int W, int Cin,
                                     int Cout, int flags, t2h_stream_t stream) {
    if (!x || !w || !y) return fail(T2H_ERR_ARG, "conv3x3_smallcin_fwd: null pointer");
    if (int rc = check_small("conv3x3_smallcin_fwd", B, H, W, Cin, Cout)) return rc;
    if (((uintptr_t)y & 15)) return fail(T2H_ERR_ARG, "conv3x3_smallcin_fwd: y must be 16-byte aligned");
    SmallArgs a{};
    a.x = x; a.w = w; a.bias = bias; a.y = y; a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.relu = (flags & T2H_RELU_OUT) ? 1 : 0;
    const int tiles = B * ((H + TS - 1) / TS) * ((W + TS - 1) / TS);
    hipLaunchKernelGGL(conv_small_fwd_kernel, dim3(tiles), dim3(256), 0, as_stream(stream), a);
    return check_launch("conv3x3_smallcin_fwd");
}

T2H_API int t2h_conv3x3_smallcin_dgrad(const float *dy, const float *w, float *dx, int B, int H, int W, int Cin, int Cout, int flags,
                                       t2h_stream_t stream) {
    if (!dy || !w || !dx) return fail(T2H_ERR_ARG, "conv3x3_smallcin_dgrad: null pointer");
    if (int rc = check_small("conv3x3_smallcin_dgrad", B, H, W, Cin, Cout)) return rc;
    SmallArgs a{};
    a.dy = dy; a.w = w; a.dx = dx; a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.accumulate = (flags & T2H_ACCUM) ? 1 : 0;
    const long long P = (long long)B * H * W;
    hipLaunchKernelGGL(conv_small_dgrad_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, as_stream(stream), a);
    return check_launch("conv3x3_smallcin_dgrad");
}

T2H_API size_t t2h_conv3x3_smallcin_wgrad_workspace_bytes(int Cin, int Cout) {
    if (Cin < 1 || Cout < 1 || Cin > kMaxCin || Cout > kMaxCout) return 0;
    return (size_t)kWgradWgs * (Cout * 9 * Cin + Cout) * sizeof(float);
}

T2H_API int t2h_conv3x3_smallcin_wgrad(const float *dy, const float *x, float *dw, float *db, int B, int H, int W, int Cin, int Cout,
                                       int flags, void *workspace, size_t workspace_bytes, t2h_stream_t stream) {
    if (!dy || !x || !dw) return fail(T2H_ERR_ARG, "conv3x3_smallcin_wgrad: null pointer");
    if (int rc = check_small("conv3x3_smallcin_wgrad", B, H, W, Cin, Cout)) return rc;
    if (((uintptr_t)dy & 15)) return fail(T2H_ERR_ARG, "conv3x3_smallcin_wgrad: dy must be 16-byte aligned");
    const size_t need = t2h_conv3x3_smallcin_wgrad_workspace_bytes(Cin, Cout);
    if (!workspace || workspace_bytes < need) return fail(T2H_ERR_WORKSPACE, "conv3x3_smallcin_wgrad: workspace %zu < %zu bytes", workspace_bytes, need);
    SmallArgs a{};
    a.dy = dy; a.x = x; a.slab = static_cast<float *>(workspace); a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
    const int tiles = B * ((H + TS - 1) / TS) * ((W + TS - 1) / TS);
    const int wgs = tiles < kWgradWgs ? tiles : kWgradWgs;
    hipStream_t s = as_stream(stream);
    hipLaunchKernelGGL(conv_small_wgrad_kernel, dim3(wgs), dim3(256), 0, s, a);
    if (int rc = check_launch("conv3x3_smallcin_wgrad")) return rc;
    const int K = 9 * Cin;
    return launch_reduce_slabs(a.slab, wgs, (long long)Cout * K, Cout, K, K, (flags & T2H_ACCUM) ? 1 : 0, dw,
                               db ? a.slab + (size_t)wgs * Cout * K : nullptr, db, s);
}
