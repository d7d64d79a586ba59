// Grid-side fusions around the (MIOpen) convolutions of the ALTO U-Net and the pixel decoder -- SURVEY 8f-1, first
// step: remove the elementwise passes between the convs.  All tensors are pixel-major NHWC ([B,H,W,C], channels_last).
//
//   bias_relu_fwd      y = relu(y + bias[c])  in place            conv bias + F.relu   (alto.py:98-99,226-227; pixel.py:27-30)
//   bias_relu_bwd      gm = g * (y > 0); db[c] = sum_p gm           relu backward + conv bias gradient in one pass
//   head1x1_*          out[p] = b + sum_i <w_i, x_i[p,:]>          torch.cat([x,x1,x2,x3]) + conv4 1x1 (pixel.py:31):
//                                                                  the 288-channel concat never materialises
//   upsample_nhwc_*    F.interpolate(bilinear, align_corners=True) on NHWC planes (pixel.py:107)
//
// HBM-bound row streaming: lanes along channels (float4), reductions over pixels as per-workgroup partial rows that
// reduce_rows_kernel adds in a fixed order (deterministic).
#include "t2h_common.h"

namespace t2h {

constexpr int kT = 256;

// ---------------------------------------------------------------------------------------------- bias + relu
__global__ __launch_bounds__(kT) void bias_relu_fwd_kernel(float *__restrict__ y, const float *__restrict__ bias,
                                                           long long n4, int C4, int relu) {
    long long i = (long long)blockIdx.x * kT + threadIdx.x;
    if (i >= n4) return;
    float4 v = reinterpret_cast<float4 *>(y)[i];
    float4 b = reinterpret_cast<const float4 *>(bias)[i % C4];
    v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
    if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    reinterpret_cast<float4 *>(y)[i] = v;
}

// gm = g * (y > 0): the ReLU backward in front of the conv3x3 data / weight gradients (conv.hip), whose weight-gradient
// kernel produces the bias gradient itself
__global__ __launch_bounds__(kT) void relu_mask_kernel(const float *__restrict__ g, const float *__restrict__ y,
                                                       float *__restrict__ gm, long long n4) {
    long long i = (long long)blockIdx.x * kT + threadIdx.x;
    if (i >= n4) return;
    float4 gv = reinterpret_cast<const float4 *>(g)[i];
    float4 yv = reinterpret_cast<const float4 *>(y)[i];
    gv.x = yv.x > 0.f ? gv.x : 0.f; gv.y = yv.y > 0.f ? gv.y : 0.f;
    gv.z = yv.z > 0.f ? gv.z : 0.f; gv.w = yv.w > 0.f ? gv.w : 0.f;
    reinterpret_cast<float4 *>(gm)[i] = gv;
}

// one workgroup = kRowsPerBlock pixels; thread (slot, lane): lane covers float4 channel group(s), slot strides pixels
constexpr int kRowsPerBlock = 256;
__global__ __launch_bounds__(kT) void bias_relu_bwd_kernel(const float *__restrict__ g, const float *__restrict__ y,
                                                           float *__restrict__ gm, long long P, int C, int lg, int relu,
                                                           float *__restrict__ partial /* [blocks][C] */) {
    __shared__ float4 red[kT];
    const int G = 1 << lg, slots = kT >> lg;
    const int lane = threadIdx.x & (G - 1), slot = threadIdx.x >> lg;
    long long p0 = (long long)blockIdx.x * kRowsPerBlock, p1 = min(P, p0 + kRowsPerBlock);
    for (int c = lane * 4; c < C; c += 4 * G) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (long long p = p0 + slot; p < p1; p += slots) {
            float4 gv = *reinterpret_cast<const float4 *>(g + p * C + c);
            if (relu) {
                float4 yv = *reinterpret_cast<const float4 *>(y + p * C + c);
                gv.x = yv.x > 0.f ? gv.x : 0.f; gv.y = yv.y > 0.f ? gv.y : 0.f;
                gv.z = yv.z > 0.f ? gv.z : 0.f; gv.w = yv.w > 0.f ? gv.w : 0.f;
                *reinterpret_cast<float4 *>(gm + p * C + c) = gv;
            }
            acc.x += gv.x; acc.y += gv.y; acc.z += gv.z; acc.w += gv.w;
        }
        __syncthreads();
        red[threadIdx.x] = acc;
        __syncthreads();
        if (slot == 0) {
            float4 t = red[lane];
            for (int s = 1; s < slots; ++s) { float4 u = red[s * G + lane]; t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w; }
            *reinterpret_cast<float4 *>(partial + (size_t)blockIdx.x * C + c) = t;
        }
    }
}

// out[c] = [out[c] +] sum_b partial[b][c], b ascending in 64 interleaved lanes then lane order (fixed tree).  4 channels
// per workgroup: with 16 (and 16 row lanes) the 289-channel head had 19 workgroups walking 64 rows each -- 25 us.
constexpr int kRedCh = 4, kRedLanes = kT / kRedCh;
__global__ __launch_bounds__(kT) void reduce_rows_kernel(const float *__restrict__ partial, int nblocks, int C,
                                                         int accumulate, float *__restrict__ out) {
    __shared__ float red[kT];
    const int el = threadIdx.x & (kRedCh - 1), q = threadIdx.x / kRedCh;
    int c = blockIdx.x * kRedCh + el;
    float s = 0.f;
    if (c < C)
        for (int b = q; b < nblocks; b += kRedLanes) s += partial[(size_t)b * C + c];
    red[threadIdx.x] = s;
    __syncthreads();
    if (q == 0 && c < C) {
        float t = red[el];
        for (int k = 1; k < kRedLanes; ++k) t += red[k * kRedCh + el];
        out[c] = accumulate ? out[c] + t : t;
    }
}

// the head's weight-gradient partials [nblocks][Ctot + 4] -> dw[0 .. Ctot) and dbias (column Ctot), written or accumulated: the
// same fixed tree as reduce_rows_kernel, results straight to their destinations (r04: a scratch row + two device copies)
__global__ __launch_bounds__(kT) void head_finish_kernel(const float *__restrict__ partial, int nblocks, int Ctot, int accumulate,
                                                         float *__restrict__ dw, float *__restrict__ dbias) {
    __shared__ float red[kT];
    const int C = Ctot + 4;
    const int el = threadIdx.x & (kRedCh - 1), q = threadIdx.x / kRedCh;
    const int c = blockIdx.x * kRedCh + el;
    float s = 0.f;
    if (c <= Ctot)
        for (int b = q; b < nblocks; b += kRedLanes) s += partial[(size_t)b * C + c];
    red[threadIdx.x] = s;
    __syncthreads();
    if (q == 0 && c <= Ctot) {
        float t = red[el];
        for (int k = 1; k < kRedLanes; ++k) t += red[k * kRedCh + el];
        float *dst = c < Ctot ? dw + c : dbias;
        if (dst) *dst = accumulate ? *dst + t : t;
    }
}

// ---------------------------------------------------------------------------------------------- concat-free 1x1 head
struct HeadArgs {
    const float *x[4];
    float *dx[4];
    int C[4];
    int off[4];        // channel offset of input i inside the concatenated weight
    int n_in;
    int Ctot;
};

// 16 lanes per pixel
__global__ __launch_bounds__(kT) void head1x1_fwd_kernel(HeadArgs a, const float *__restrict__ w, const float *__restrict__ bias,
                                                         long long P, float *__restrict__ out) {
    long long t = (long long)blockIdx.x * kT + threadIdx.x;
    long long p = t >> 4;
    int lane = threadIdx.x & 15;
    float acc = 0.f;
    if (p < P) {
        for (int i = 0; i < a.n_in; ++i) {
            const float *row = a.x[i] + p * a.C[i];
            const float *wi = w + a.off[i];
            for (int c = lane * 4; c < a.C[i]; c += 64) {
                float4 v = *reinterpret_cast<const float4 *>(row + c);
                float4 k = *reinterpret_cast<const float4 *>(wi + c);
                acc += v.x * k.x + v.y * k.y + v.z * k.z + v.w * k.w;
            }
        }
    }
    acc += __shfl_xor(acc, 8, 16);
    acc += __shfl_xor(acc, 4, 16);
    acc += __shfl_xor(acc, 2, 16);
    acc += __shfl_xor(acc, 1, 16);
    if (p < P && lane == 0) out[p] = acc + (bias ? bias[0] : 0.f);
}

// dx_i[p, c] = [dx_i +] g[p] * w_i[c] (* (x_i[p, c] > 0) when bit 8+i of flags is set); bit 0 of flags = accumulate
__global__ __launch_bounds__(kT) void head1x1_dgrad_kernel(HeadArgs a, const float *__restrict__ w, const float *__restrict__ g,
                                                           long long P, int flags) {
    const int accumulate = flags & 1;
    long long t = (long long)blockIdx.x * kT + threadIdx.x;
    long long p = t >> 4;
    int lane = threadIdx.x & 15;
    if (p >= P) return;
    float gp = g[p];
    for (int i = 0; i < a.n_in; ++i) {
        if (!a.dx[i]) continue;
        float *row = a.dx[i] + p * a.C[i];
        const float *wi = w + a.off[i];
        for (int c = lane * 4; c < a.C[i]; c += 64) {
            float4 k = *reinterpret_cast<const float4 *>(wi + c);
            float4 v = make_float4(gp * k.x, gp * k.y, gp * k.z, gp * k.w);
            if ((flags >> (8 + i)) & 1) {
                float4 xv = *reinterpret_cast<const float4 *>(a.x[i] + p * a.C[i] + c);
                v.x = xv.x > 0.f ? v.x : 0.f; v.y = xv.y > 0.f ? v.y : 0.f;
                v.z = xv.z > 0.f ? v.z : 0.f; v.w = xv.w > 0.f ? v.w : 0.f;
            }
            if (accumulate) { float4 o = *reinterpret_cast<float4 *>(row + c); v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
            *reinterpret_cast<float4 *>(row + c) = v;
        }
    }
}

// partial[blk][c] = sum over the block's pixels of g[p] * x_i[p, c];  partial[blk][Ctot] = sum g[p]
__global__ __launch_bounds__(kT) void head1x1_wgrad_kernel(HeadArgs a, const float *__restrict__ g, long long P,
                                                           float *__restrict__ partial) {
    __shared__ float4 red[kT];
    long long p0 = (long long)blockIdx.x * kRowsPerBlock, p1 = min(P, p0 + kRowsPerBlock);
    const int row_stride = a.Ctot + 4;
    for (int i = 0; i < a.n_in; ++i) {
        const int C4 = a.C[i] / 4;                       // float4 groups of this input
        int lg = 0;
        while ((1 << lg) < C4 && lg < 8) ++lg;
        const int G = 1 << lg, slots = kT >> lg;
        const int lane = threadIdx.x & (G - 1), slot = threadIdx.x >> lg;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (lane < C4)
            for (long long p = p0 + slot; p < p1; p += slots) {
                float gp = g[p];
                float4 v = *reinterpret_cast<const float4 *>(a.x[i] + p * a.C[i] + lane * 4);
                acc.x += gp * v.x; acc.y += gp * v.y; acc.z += gp * v.z; acc.w += gp * v.w;
            }
        __syncthreads();
        red[threadIdx.x] = acc;
        __syncthreads();
        if (slot == 0 && lane < C4) {
            float4 t = red[lane];
            for (int s = 1; s < slots; ++s) { float4 u = red[s * G + lane]; t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w; }
            *reinterpret_cast<float4 *>(partial + (size_t)blockIdx.x * row_stride + a.off[i] + lane * 4) = t;
        }
    }
    // bias gradient: sum of g over the block's pixels
    __syncthreads();
    float s = 0.f;
    for (long long p = p0 + threadIdx.x; p < p1; p += kT) s += g[p];
    reinterpret_cast<float *>(red)[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int k = 0; k < kT; ++k) t += reinterpret_cast<float *>(red)[k];
        partial[(size_t)blockIdx.x * row_stride + a.Ctot] = t;
        partial[(size_t)blockIdx.x * row_stride + a.Ctot + 1] = 0.f;
        partial[(size_t)blockIdx.x * row_stride + a.Ctot + 2] = 0.f;
        partial[(size_t)blockIdx.x * row_stride + a.Ctot + 3] = 0.f;
    }
}

// ---------------------------------------------------------------------------------------------- NHWC bilinear upsample
struct Lerp2 { int i0, i1; float l0, l1; };
// half: align_corners=False source index (ATen area_pixel_compute_source_index: scale * (dst + 0.5) - 0.5, clamped at 0)
__device__ inline Lerp2 lerp_index2(float scale, int dst, int in_size, int half = 0) {
    float real = half ? fmaxf(__fsub_rn(__fmul_rn(scale, __fadd_rn((float)dst, 0.5f)), 0.5f), 0.0f) : __fmul_rn(scale, (float)dst);
    int a = min((int)real, in_size - 1);
    Lerp2 L;
    L.i0 = a; L.i1 = a + (a < in_size - 1 ? 1 : 0);
    float lam = fminf(fmaxf(__fsub_rn(real, (float)a), 0.0f), 1.0f);
    L.l1 = lam; L.l0 = __fsub_rn(1.0f, lam);
    return L;
}

// one lane-group (G lanes x float4) per output pixel
__global__ __launch_bounds__(kT) void upsample_nhwc_fwd_kernel(const float *__restrict__ in, const float *__restrict__ addend,
                                                               int B, int h, int w, int H, int W, int C, int lg, float sh,
                                                               float sw, float *__restrict__ out, int half) {
    long long t = (long long)blockIdx.x * kT + threadIdx.x;
    long long pix = t >> lg;
    if (pix >= (long long)B * H * W) return;
    int x = (int)(pix % W), y = (int)((pix / W) % H), b = (int)(pix / ((long long)W * H));
    Lerp2 ly = lerp_index2(sh, y, h, half), lx = lerp_index2(sw, x, w, half);
    const float *base = in + (size_t)b * h * w * C;
    const float *p00 = base + ((size_t)ly.i0 * w + lx.i0) * C, *p01 = base + ((size_t)ly.i0 * w + lx.i1) * C;
    const float *p10 = base + ((size_t)ly.i1 * w + lx.i0) * C, *p11 = base + ((size_t)ly.i1 * w + lx.i1) * C;
    for (int c = ((int)t & ((1 << lg) - 1)) * 4; c < C; c += 4 << lg) {
        float4 a = *reinterpret_cast<const float4 *>(p00 + c), bq = *reinterpret_cast<const float4 *>(p01 + c);
        float4 cq = *reinterpret_cast<const float4 *>(p10 + c), d = *reinterpret_cast<const float4 *>(p11 + c);
        float4 o;
#define T2H_LERP(F) o.F = __fadd_rn(__fmul_rn(ly.l0, __fadd_rn(__fmul_rn(lx.l0, a.F), __fmul_rn(lx.l1, bq.F))), \
                               __fmul_rn(ly.l1, __fadd_rn(__fmul_rn(lx.l0, cq.F), __fmul_rn(lx.l1, d.F))))
        T2H_LERP(x); T2H_LERP(y); T2H_LERP(z); T2H_LERP(w);
#undef T2H_LERP
        if (addend) {
            float4 e = *reinterpret_cast<const float4 *>(addend + pix * C + c);
            o.x = __fadd_rn(o.x, e.x); o.y = __fadd_rn(o.y, e.y); o.z = __fadd_rn(o.z, e.z); o.w = __fadd_rn(o.w, e.w);
        }
        *reinterpret_cast<float4 *>(out + pix * C + c) = o;
    }
}

__device__ inline void source_range2(float scale, int i, int out_size, int &lo, int &hi) {
    if (scale <= 0.0f) { lo = 0; hi = out_size - 1; return; }
    lo = max(0, (int)floorf((float)(i - 1) / scale) - 1);
    hi = min(out_size - 1, (int)ceilf((float)(i + 1) / scale) + 1);
}

// one lane-group per INPUT pixel: gathers the output pixels whose taps touch it (deterministic, no atomics)
__global__ __launch_bounds__(kT) void upsample_nhwc_bwd_kernel(const float *__restrict__ gout, int B, int h, int w, int H,
                                                               int W, int C, int lg, float sh, float sw,
                                                               float *__restrict__ gin, int half) {
    long long t = (long long)blockIdx.x * kT + threadIdx.x;
    long long pix = t >> lg;
    if (pix >= (long long)B * h * w) return;
    int ix = (int)(pix % w), iy = (int)((pix / w) % h), b = (int)(pix / ((long long)w * h));
    int ylo, yhi, xlo, xhi;
    source_range2(sh, iy, H, ylo, yhi);
    source_range2(sw, ix, W, xlo, xhi);
    const float *base = gout + (size_t)b * H * W * C;
    for (int c = ((int)t & ((1 << lg) - 1)) * 4; c < C; c += 4 << lg) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int y = ylo; y <= yhi; ++y) {
            Lerp2 ly = lerp_index2(sh, y, h, half);
            if (ly.i0 != iy && ly.i1 != iy) continue;
            for (int x = xlo; x <= xhi; ++x) {
                Lerp2 lx = lerp_index2(sw, x, w, half);
                if (lx.i0 != ix && lx.i1 != ix) continue;
                float wgt = 0.f;
                if (ly.i0 == iy && lx.i0 == ix) wgt = __fadd_rn(wgt, __fmul_rn(ly.l0, lx.l0));
                if (ly.i0 == iy && lx.i1 == ix) wgt = __fadd_rn(wgt, __fmul_rn(ly.l0, lx.l1));
                if (ly.i1 == iy && lx.i0 == ix) wgt = __fadd_rn(wgt, __fmul_rn(ly.l1, lx.l0));
                if (ly.i1 == iy && lx.i1 == ix) wgt = __fadd_rn(wgt, __fmul_rn(ly.l1, lx.l1));
                float4 g = *reinterpret_cast<const float4 *>(base + ((size_t)y * W + x) * C + c);
                acc.x = __fadd_rn(acc.x, __fmul_rn(wgt, g.x)); acc.y = __fadd_rn(acc.y, __fmul_rn(wgt, g.y));
                acc.z = __fadd_rn(acc.z, __fmul_rn(wgt, g.z)); acc.w = __fadd_rn(acc.w, __fmul_rn(wgt, g.w));
            }
        }
        *reinterpret_cast<float4 *>(gin + pix * C + c) = acc;
    }
}

// ---------------------------------------------------------------------------------------------- 2x2 max-pool (NHWC)
// nn.MaxPool2d(kernel_size=2, stride=2) of the ALTO down path (alto.py:61,110,136): first maximum in window scan
// order (0,0),(0,1),(1,0),(1,1) wins, like ATen's strict '>' -- the planes are full of exact ties (empty cells = 0).
// which[pixel][c] = 2 * dy + dx of the winner, one byte per output element, consumed by the backward.
__global__ __launch_bounds__(kT) void maxpool2x2_fwd_kernel(const float *__restrict__ in, int B, int H, int W, int C, int lg,
                                                            float *__restrict__ out, uint8_t *__restrict__ which) {
    long long t = (long long)blockIdx.x * kT + threadIdx.x;
    long long pix = t >> lg;
    const int h = H / 2, w = W / 2;
    if (pix >= (long long)B * h * w) return;
    int x = (int)(pix % w), y = (int)((pix / w) % h), b = (int)(pix / ((long long)w * h));
    const float *p00 = in + (((size_t)b * H + 2 * y) * W + 2 * x) * C;
    const float *p10 = p00 + (size_t)W * C;
    for (int c = ((int)t & ((1 << lg) - 1)) * 4; c < C; c += 4 << lg) {
        const float4 v[4] = {*reinterpret_cast<const float4 *>(p00 + c), *reinterpret_cast<const float4 *>(p00 + C + c),
                             *reinterpret_cast<const float4 *>(p10 + c), *reinterpret_cast<const float4 *>(p10 + C + c)};
        float4 best = v[0];
        uchar4 arg = make_uchar4(0, 0, 0, 0);
#pragma unroll
        for (int k = 1; k < 4; ++k) {
            if (v[k].x > best.x || v[k].x != v[k].x) { best.x = v[k].x; arg.x = k; }     // NaN propagates like ATen
            if (v[k].y > best.y || v[k].y != v[k].y) { best.y = v[k].y; arg.y = k; }
            if (v[k].z > best.z || v[k].z != v[k].z) { best.z = v[k].z; arg.z = k; }
            if (v[k].w > best.w || v[k].w != v[k].w) { best.w = v[k].w; arg.w = k; }
        }
        *reinterpret_cast<float4 *>(out + pix * C + c) = best;
        *reinterpret_cast<uchar4 *>(which + pix * C + c) = arg;
    }
}

// gin[2y+dy, 2x+dx, c] = (which == 2 dy + dx) ? g[y, x, c] : 0 -- every input element written exactly once
__global__ __launch_bounds__(kT) void maxpool2x2_bwd_kernel(const float *__restrict__ g, const uint8_t *__restrict__ which, int B,
                                                            int H, int W, int C, int lg, const float *__restrict__ addend,
                                                            int ld_add, float *__restrict__ gin) {
    long long t = (long long)blockIdx.x * kT + threadIdx.x;
    long long pix = t >> lg;
    const int h = H / 2, w = W / 2;
    if (pix >= (long long)B * h * w) return;
    int x = (int)(pix % w), y = (int)((pix / w) % h), b = (int)(pix / ((long long)w * h));
    float *p00 = gin + (((size_t)b * H + 2 * y) * W + 2 * x) * C;
    float *p10 = p00 + (size_t)W * C;
    for (int c = ((int)t & ((1 << lg) - 1)) * 4; c < C; c += 4 << lg) {
        const float4 gv = *reinterpret_cast<const float4 *>(g + pix * C + c);
        const uchar4 a = *reinterpret_cast<const uchar4 *>(which + pix * C + c);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float4 o = make_float4(a.x == k ? gv.x : 0.f, a.y == k ? gv.y : 0.f, a.z == k ? gv.z : 0.f, a.w == k ? gv.w : 0.f);
            float *dst = (k & 2 ? p10 : p00) + (k & 1) * C + c;
            if (addend) {       // the plane's other gradient (the U-Net skip connection)
                // (addend: pixel stride ld_add >= C -- a channel slice of a wider NHWC tensor, e.g. one half of a concatenation's gradient)
                const size_t off = (size_t)(dst - gin);
                const float4 ad = *reinterpret_cast<const float4 *>(addend + (off / C) * ld_add + off % C);
                o.x = __fadd_rn(ad.x, o.x); o.y = __fadd_rn(ad.y, o.y); o.z = __fadd_rn(ad.z, o.z); o.w = __fadd_rn(ad.w, o.w);
            }
            *reinterpret_cast<float4 *>(dst) = o;
        }
    }
}

static int lg_for(int C) { return group_log2(C, 4); }

}  // namespace t2h

using namespace t2h;

T2H_API int t2h_bias_relu_fwd(float *y, const float *bias, int64_t P, int C, int relu, t2h_stream_t stream) {
    if (!y || !bias || P < 0 || C < 4 || C % 4) return fail(T2H_ERR_ARG, "bias_relu_fwd: bad argument (C %% 4 == 0 required)");
    if (P == 0) return T2H_OK;
    long long n4 = (long long)P * C / 4;
    hipLaunchKernelGGL(bias_relu_fwd_kernel, dim3((unsigned)((n4 + kT - 1) / kT)), dim3(kT), 0, as_stream(stream), y, bias, n4,
                       C / 4, relu);
    return check_launch("bias_relu_fwd");
}

T2H_API int t2h_relu_mask(const float *g, const float *y, float *g_masked, int64_t n, t2h_stream_t stream) {
    if (!g || !y || !g_masked) return fail(T2H_ERR_ARG, "relu_mask: null pointer");
    if (n < 0 || n % 4 != 0 || (uintptr_t)g % 16 || (uintptr_t)y % 16 || (uintptr_t)g_masked % 16)
        return fail(T2H_ERR_ARG, "relu_mask: n must be a multiple of 4 and the pointers 16-byte aligned");
    if (n == 0) return T2H_OK;
    long long n4 = n / 4;
    hipLaunchKernelGGL(relu_mask_kernel, dim3((unsigned)((n4 + kT - 1) / kT)), dim3(kT), 0, as_stream(stream), g, y, g_masked, n4);
    return check_launch("relu_mask");
}

T2H_API size_t t2h_bias_relu_bwd_workspace_bytes(int64_t P, int C) {
    if (P < 1 || C < 1) return 0;
    return (size_t)((P + kRowsPerBlock - 1) / kRowsPerBlock) * C * sizeof(float);
}

T2H_API int t2h_bias_relu_bwd(const float *g, const float *y, float *g_masked, int64_t P, int C, int relu, int accumulate,
                              float *dbias, void *workspace, size_t workspace_bytes, t2h_stream_t stream) {
    if (!g || !dbias || (relu && (!y || !g_masked)) || P < 1 || C < 4 || C % 4)
        return fail(T2H_ERR_ARG, "bias_relu_bwd: bad argument");
    if (!workspace || workspace_bytes < t2h_bias_relu_bwd_workspace_bytes(P, C))
        return fail(T2H_ERR_WORKSPACE, "bias_relu_bwd: workspace too small");
    int nblocks = (int)((P + kRowsPerBlock - 1) / kRowsPerBlock);
    float *partial = static_cast<float *>(workspace);
    hipLaunchKernelGGL(bias_relu_bwd_kernel, dim3(nblocks), dim3(kT), 0, as_stream(stream), g, y, g_masked, (long long)P, C,
                       lg_for(C), relu, partial);
    hipLaunchKernelGGL(reduce_rows_kernel, dim3((C + kRedCh - 1) / kRedCh), dim3(kT), 0, as_stream(stream), partial, nblocks, C, accumulate,
                       dbias);
    return check_launch("bias_relu_bwd");
}

static int fill_head(HeadArgs &a, const float *const *x, float *const *dx, const int *C, int n_in, const char *what) {
    if (n_in < 1 || n_in > 4) return fail(T2H_ERR_ARG, "%s: 1..4 inputs", what);
    int off = 0;
    for (int i = 0; i < 4; ++i) { a.x[i] = nullptr; a.dx[i] = nullptr; a.C[i] = 0; a.off[i] = 0; }
    for (int i = 0; i < n_in; ++i) {
        if (C[i] < 4 || C[i] % 4 || C[i] > 1024) return fail(T2H_ERR_ARG, "%s: channel counts must be multiples of 4", what);
        a.x[i] = x ? x[i] : nullptr; a.dx[i] = dx ? dx[i] : nullptr; a.C[i] = C[i]; a.off[i] = off; off += C[i];
    }
    a.n_in = n_in; a.Ctot = off;
    return T2H_OK;
}

T2H_API int t2h_head1x1_fwd(const float *const *x, const int *C, int n_in, const float *w, const float *bias, int64_t P,
                            float *out, t2h_stream_t stream) {
    if (!x || !C || !w || !out || P < 0) return fail(T2H_ERR_ARG, "head1x1_fwd: bad argument");
    HeadArgs a;
    int rc = fill_head(a, x, nullptr, C, n_in, "head1x1_fwd");
    if (rc) return rc;
    if (P == 0) return T2H_OK;
    hipLaunchKernelGGL(head1x1_fwd_kernel, dim3((unsigned)((P * 16 + kT - 1) / kT)), dim3(kT), 0, as_stream(stream), a, w, bias,
                       (long long)P, out);
    return check_launch("head1x1_fwd");
}

T2H_API size_t t2h_head1x1_bwd_workspace_bytes(int64_t P, int Ctot) {
    if (P < 1 || Ctot < 1) return 0;
    return ((size_t)((P + kRowsPerBlock - 1) / kRowsPerBlock) + 1) * (Ctot + 4) * sizeof(float);   // partial rows + result row
}

T2H_API int t2h_head1x1_bwd(const float *const *x, float *const *dx, const int *C, int n_in, const float *w, const float *g,
                            int64_t P, int dx_flags, float *dw, float *dbias, void *workspace, size_t workspace_bytes,
                            t2h_stream_t stream) {
    if (!x || !dx || !C || !w || !g || !dw || P < 1) return fail(T2H_ERR_ARG, "head1x1_bwd: bad argument");
    HeadArgs a;
    int rc = fill_head(a, x, dx, C, n_in, "head1x1_bwd");
    if (rc) return rc;
    if (!workspace || workspace_bytes < t2h_head1x1_bwd_workspace_bytes(P, a.Ctot))
        return fail(T2H_ERR_WORKSPACE, "head1x1_bwd: workspace too small");
    hipStream_t s = as_stream(stream);
    hipLaunchKernelGGL(head1x1_dgrad_kernel, dim3((unsigned)((P * 16 + kT - 1) / kT)), dim3(kT), 0, s, a, w, g, (long long)P,
                       dx_flags);
    int nblocks = (int)((P + kRowsPerBlock - 1) / kRowsPerBlock);
    float *partial = static_cast<float *>(workspace);
    hipLaunchKernelGGL(head1x1_wgrad_kernel, dim3(nblocks), dim3(kT), 0, s, a, g, (long long)P, partial);
    // columns [0, Ctot) -> dw, column Ctot -> dbias (row stride Ctot + 4); dx_flags bit 1: added to what dw / dbias hold
    hipLaunchKernelGGL(head_finish_kernel, dim3((a.Ctot + 1 + kRedCh - 1) / kRedCh), dim3(kT), 0, s, partial, nblocks, a.Ctot,
                       (dx_flags >> 1) & 1, dw, dbias);
    return check_launch("head1x1_bwd");
}

T2H_API int t2h_maxpool2x2_nhwc_fwd(const float *in, int B, int H, int W, int C, float *out, uint8_t *which, t2h_stream_t stream) {
    if (!in || !out || !which || B < 1 || H < 2 || W < 2 || (H & 1) || (W & 1) || C < 4 || C % 4)
        return fail(T2H_ERR_ARG, "maxpool2x2_nhwc_fwd: bad argument (even H, W and C %% 4 == 0 required)");
    int lg = lg_for(C);
    long long threads = ((long long)B * (H / 2) * (W / 2)) << lg;
    hipLaunchKernelGGL(maxpool2x2_fwd_kernel, dim3((unsigned)((threads + kT - 1) / kT)), dim3(kT), 0, as_stream(stream), in, B, H, W,
                       C, lg, out, which);
    return check_launch("maxpool2x2_nhwc_fwd");
}

T2H_API int t2h_maxpool2x2_nhwc_bwd(const float *gout, const uint8_t *which, int B, int H, int W, int C, float *gin,
                                    t2h_stream_t stream) {
    return t2h_maxpool2x2_nhwc_bwd_add(gout, which, B, H, W, C, nullptr, 0, gin, stream);
}

T2H_API int t2h_maxpool2x2_nhwc_bwd_add(const float *gout, const uint8_t *which, int B, int H, int W, int C, const float *addend,
                                        int ld_addend, float *gin, t2h_stream_t stream) {
    if (!gout || !gin || !which || B < 1 || H < 2 || W < 2 || (H & 1) || (W & 1) || C < 4 || C % 4)
        return fail(T2H_ERR_ARG, "maxpool2x2_nhwc_bwd: bad argument (even H, W and C %% 4 == 0 required)");
    if (addend && (((uintptr_t)addend & 15) || ld_addend < C || ld_addend % 4))
        return fail(T2H_ERR_ARG, "maxpool2x2_nhwc_bwd: addend must be 16-byte aligned with a pixel stride >= C, a multiple of 4");
    int lg = lg_for(C);
    long long threads = ((long long)B * (H / 2) * (W / 2)) << lg;
    hipLaunchKernelGGL(maxpool2x2_bwd_kernel, dim3((unsigned)((threads + kT - 1) / kT)), dim3(kT), 0, as_stream(stream), gout, which,
                       B, H, W, C, lg, addend, ld_addend, gin);
    return check_launch("maxpool2x2_nhwc_bwd");
}

T2H_API int t2h_upsample_bilinear_nhwc_fwd(const float *in, const float *addend, int B, int C, int h, int w, int H, int W,
                                           float *out, t2h_stream_t stream) {
    if (!in || !out || B < 1 || C < 4 || C % 4 || h < 1 || w < 1 || H < 1 || W < 1)
        return fail(T2H_ERR_ARG, "upsample_bilinear_nhwc_fwd: bad argument (C %% 4 == 0 required)");
    float sh = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.0f, sw = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.0f;
    int lg = lg_for(C);
    long long threads = ((long long)B * H * W) << lg;
    hipLaunchKernelGGL(upsample_nhwc_fwd_kernel, dim3((unsigned)((threads + kT - 1) / kT)), dim3(kT), 0, as_stream(stream), in,
                       addend, B, h, w, H, W, C, lg, sh, sw, out, 0);
    return check_launch("upsample_bilinear_nhwc_fwd");
}

T2H_API int t2h_upsample_bilinear_nhwc_bwd(const float *gout, int B, int C, int h, int w, int H, int W, float *gin,
                                           t2h_stream_t stream) {
    if (!gout || !gin || B < 1 || C < 4 || C % 4 || h < 1 || w < 1 || H < 1 || W < 1)
        return fail(T2H_ERR_ARG, "upsample_bilinear_nhwc_bwd: bad argument (C %% 4 == 0 required)");
    float sh = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.0f, sw = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.0f;
    int lg = lg_for(C);
    long long threads = ((long long)B * h * w) << lg;
    hipLaunchKernelGGL(upsample_nhwc_bwd_kernel, dim3((unsigned)((threads + kT - 1) / kT)), dim3(kT), 0, as_stream(stream),
                       gout, B, h, w, H, W, C, lg, sh, sw, gin, 0);
    return check_launch("upsample_bilinear_nhwc_bwd");
}

// nn.Upsample(mode='bilinear', scale_factor=2) (align_corners=False): the non-parametric up path of upconv2x2(mode='upsample')
T2H_API int t2h_upsample2x_nhwc_fwd(const float *in, int B, int C, int h, int w, float *out, t2h_stream_t stream) {
    if (!in || !out || B < 1 || C < 4 || C % 4 || h < 1 || w < 1 || h > 16384 || w > 16384)
        return fail(T2H_ERR_ARG, "upsample2x_nhwc_fwd: bad argument (C %% 4 == 0 required)");
    int lg = lg_for(C);
    long long threads = ((long long)B * 2 * h * 2 * w) << lg;
    hipLaunchKernelGGL(upsample_nhwc_fwd_kernel, dim3((unsigned)((threads + kT - 1) / kT)), dim3(kT), 0, as_stream(stream), in,
                       (const float *)nullptr, B, h, w, 2 * h, 2 * w, C, lg, 0.5f, 0.5f, out, 1);
    return check_launch("upsample2x_nhwc_fwd");
}

T2H_API int t2h_upsample2x_nhwc_bwd(const float *gout, int B, int C, int h, int w, float *gin, t2h_stream_t stream) {
    if (!gout || !gin || B < 1 || C < 4 || C % 4 || h < 1 || w < 1 || h > 16384 || w > 16384)
        return fail(T2H_ERR_ARG, "upsample2x_nhwc_bwd: bad argument (C %% 4 == 0 required)");
    int lg = lg_for(C);
    long long threads = ((long long)B * h * w) << lg;
    hipLaunchKernelGGL(upsample_nhwc_bwd_kernel, dim3((unsigned)((threads + kT - 1) / kT)), dim3(kT), 0, as_stream(stream),
                       gout, B, h, w, 2 * h, 2 * w, C, lg, 0.5f, 0.5f, gin, 1);
    return check_launch("upsample2x_nhwc_bwd");
}
