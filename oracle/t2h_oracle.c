/* TEST INFRASTRUCTURE ONLY -- parity oracle, not product code.
 *
 * Plain-C restatement of the operator-level arithmetic on the hot path of
 * zhu-xlab/tomosar2height, in ORIGINAL point order, single-threaded, with the
 * reference CPU path's summation order.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it (via oracle/c_oracle.py).
 *
 * Pin status: checked against the tests/golden/ npz fixtures, which were produced by the
 * reference's own Python modules (tests/golden/make_golden.py).  The
 * scatter_max tie-break restates pytorch-scatter's published CPU semantics
 * (parity unpinned: the reference holds no vector for it).
 *
 * Citations are relative to /root/reference.  Build: oracle/build.py (gcc,
 * -O2 -ffp-contract=off so no FMA contraction changes rounding).
 *
 * Layout conventions: point features are point-major [B,N,C] (the reference's
 * native layout before its .permute(0,2,1) views); planes are [B,C,H,W].
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

#define T2H_ORACLE_API __attribute__((visibility("default")))

/* utils/coordinate.py:12-28 : (x*reso).long() truncates toward zero; idx = ix + reso*iy. */
T2H_ORACLE_API int t2h_oracle_coordinate2index(const float *pts, int stride, int B, int N, int reso,
                                               int64_t *index /* [B,N] */) {
    for (long i = 0; i < (long)B * N; ++i) {
        float fx = pts[i * stride + 0] * (float)reso;
        float fy = pts[i * stride + 1] * (float)reso;
        int64_t ix = (int64_t)fx, iy = (int64_t)fy; /* C cast == trunc, as Tensor.long() */
        index[i] = ix + (int64_t)reso * iy;
    }
    return 0;
}

/* torch_scatter.scatter_max over the last dim (call site pointnet.py:95): running value starts at
 * the lowest float, strict '>' update => first occurrence wins ties; untouched cells -> 0 / arg = N. */
T2H_ORACLE_API int t2h_oracle_scatter_max(const float *feat /* [B,N,C] */, const int64_t *index /* [B,N] */,
                                          int B, int N, int C, int cells, float *out /* [B,C,cells] */,
                                          int64_t *arg /* [B,C,cells] */) {
    for (int b = 0; b < B; ++b) {
        float *o = out + (size_t)b * C * cells;
        int64_t *a = arg + (size_t)b * C * cells;
        for (size_t i = 0; i < (size_t)C * cells; ++i) { o[i] = -FLT_MAX; a[i] = N; }
        for (int n = 0; n < N; ++n) {
            int64_t cell = index[(size_t)b * N + n];
            if (cell < 0 || cell >= cells) return -1;
            const float *row = feat + ((size_t)b * N + n) * C;
            for (int c = 0; c < C; ++c) {
                if (row[c] > o[(size_t)c * cells + cell]) {
                    o[(size_t)c * cells + cell] = row[c];
                    a[(size_t)c * cells + cell] = n;
                }
            }
        }
        for (size_t i = 0; i < (size_t)C * cells; ++i)
            if (a[i] == N) o[i] = 0.0f;
    }
    return 0;
}

/* pointnet.py:92-99 : scatter_max then gather back to every point of the cell. */
T2H_ORACLE_API int t2h_oracle_pool_local_fwd(const float *feat, const int64_t *index, int B, int N, int C,
                                             int cells, float *pooled /* [B,N,C] */,
                                             int64_t *arg /* [B,C,cells] */) {
    float *tmp = (float *)malloc(sizeof(float) * (size_t)B * C * cells);
    if (!tmp) return -2;
    int rc = t2h_oracle_scatter_max(feat, index, B, N, C, cells, tmp, arg);
    if (rc == 0) {
        for (int b = 0; b < B; ++b)
            for (int n = 0; n < N; ++n) {
                int64_t cell = index[(size_t)b * N + n];
                for (int c = 0; c < C; ++c)
                    pooled[((size_t)b * N + n) * C + c] = tmp[((size_t)b * C + c) * cells + cell];
            }
    }
    free(tmp);
    return rc;
}

/* backward of gather (scatter-add of point grads into cells, point order) followed by the backward of
 * scatter_max (cell grad routed to the arg-max point only). */
T2H_ORACLE_API int t2h_oracle_pool_local_bwd(const float *gpooled /* [B,N,C] */, const int64_t *index,
                                             const int64_t *arg, int B, int N, int C, int cells,
                                             float *gfeat /* [B,N,C] */) {
    float *gcell = (float *)calloc((size_t)B * C * cells, sizeof(float));
    if (!gcell) return -2;
    memset(gfeat, 0, sizeof(float) * (size_t)B * N * C);
    for (int b = 0; b < B; ++b) {
        for (int n = 0; n < N; ++n) {
            int64_t cell = index[(size_t)b * N + n];
            for (int c = 0; c < C; ++c)
                gcell[((size_t)b * C + c) * cells + cell] += gpooled[((size_t)b * N + n) * C + c];
        }
        for (int c = 0; c < C; ++c)
            for (int cell = 0; cell < cells; ++cell) {
                int64_t a = arg[((size_t)b * C + c) * cells + cell];
                if (a < N) gfeat[((size_t)b * N + a) * C + c] = gcell[((size_t)b * C + c) * cells + cell];
            }
    }
    free(gcell);
    return 0;
}

/* scatter_mean(out=zeros) (pointnet.py:109; alto.py:85,194): fp32 scatter_add in point order,
 * count clamped to >= 1, one division. */
T2H_ORACLE_API int t2h_oracle_scatter_mean_fwd(const float *feat, const int64_t *index, int B, int N, int C,
                                               int cells, float *plane /* [B,C,cells] */) {
    float *cnt = (float *)malloc(sizeof(float) * (size_t)cells);
    if (!cnt) return -2;
    for (int b = 0; b < B; ++b) {
        float *p = plane + (size_t)b * C * cells;
        memset(p, 0, sizeof(float) * (size_t)C * cells);
        memset(cnt, 0, sizeof(float) * (size_t)cells);
        for (int n = 0; n < N; ++n) {
            int64_t cell = index[(size_t)b * N + n];
            if (cell < 0 || cell >= cells) { free(cnt); return -1; }
            cnt[cell] += 1.0f;
            for (int c = 0; c < C; ++c) p[(size_t)c * cells + cell] += feat[((size_t)b * N + n) * C + c];
        }
        for (int cell = 0; cell < cells; ++cell) {
            float d = cnt[cell] < 1.0f ? 1.0f : cnt[cell];
            for (int c = 0; c < C; ++c) p[(size_t)c * cells + cell] /= d;
        }
    }
    free(cnt);
    return 0;
}

T2H_ORACLE_API int t2h_oracle_scatter_mean_bwd(const float *gplane /* [B,C,cells] */, const int64_t *index,
                                               int B, int N, int C, int cells, float *gfeat /* [B,N,C] */) {
    float *cnt = (float *)malloc(sizeof(float) * (size_t)cells);
    if (!cnt) return -2;
    for (int b = 0; b < B; ++b) {
        memset(cnt, 0, sizeof(float) * (size_t)cells);
        for (int n = 0; n < N; ++n) cnt[index[(size_t)b * N + n]] += 1.0f;
        for (int n = 0; n < N; ++n) {
            int64_t cell = index[(size_t)b * N + n];
            float d = cnt[cell] < 1.0f ? 1.0f : cnt[cell];
            for (int c = 0; c < C; ++c)
                gfeat[((size_t)b * N + n) * C + c] = gplane[((size_t)b * C + c) * cells + cell] / d;
        }
    }
    free(cnt);
    return 0;
}

/* F.grid_sample(plane, 2*xy-1, bilinear, padding_mode='border', align_corners=True) as called at
 * alto.py:90-95,199-205; arithmetic of ATen's grid_sampler_2d CPU kernel:
 *   ix = ((g+1)/2)*(W-1); clip to [0, W-1]; taps floor/floor+1; out-of-range taps contribute 0. */
static inline float unnormalize_clip(float x01, int size) {
    float g = 2.0f * x01 - 1.0f;                       /* alto.py:94 */
    float ix = ((g + 1.0f) / 2.0f) * (float)(size - 1);
    float hi = (float)(size - 1);
    ix = ix < 0.0f ? 0.0f : ix;
    ix = ix > hi ? hi : ix;
    return ix;
}

T2H_ORACLE_API int t2h_oracle_grid_sample_fwd(const float *plane /* [B,C,H,W] */, const float *pts, int stride,
                                              int B, int C, int H, int W, int N, float *out /* [B,N,C] */) {
    for (int b = 0; b < B; ++b)
        for (int n = 0; n < N; ++n) {
            const float *p = pts + ((size_t)b * N + n) * stride;
            float ix = unnormalize_clip(p[0], W), iy = unnormalize_clip(p[1], H);
            float fx = floorf(ix), fy = floorf(iy);
            int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
            float nw = ((float)x1 - ix) * ((float)y1 - iy), ne = (ix - (float)x0) * ((float)y1 - iy);
            float sw = ((float)x1 - ix) * (iy - (float)y0), se = (ix - (float)x0) * (iy - (float)y0);
            for (int c = 0; c < C; ++c) {
                const float *img = plane + ((size_t)b * C + c) * H * W;
                float r = 0.0f;
                if (y0 >= 0 && y0 < H && x0 >= 0 && x0 < W) r += img[(size_t)y0 * W + x0] * nw;
                if (y0 >= 0 && y0 < H && x1 >= 0 && x1 < W) r += img[(size_t)y0 * W + x1] * ne;
                if (y1 >= 0 && y1 < H && x0 >= 0 && x0 < W) r += img[(size_t)y1 * W + x0] * sw;
                if (y1 >= 0 && y1 < H && x1 >= 0 && x1 < W) r += img[(size_t)y1 * W + x1] * se;
                out[((size_t)b * N + n) * C + c] = r;
            }
        }
    return 0;
}

/* grid_sampler_2d_backward w.r.t. the plane only (points carry no grad): 4-tap scatter-add, point order. */
T2H_ORACLE_API int t2h_oracle_grid_sample_bwd(const float *gout /* [B,N,C] */, const float *pts, int stride,
                                              int B, int C, int H, int W, int N,
                                              float *gplane /* [B,C,H,W] */) {
    memset(gplane, 0, sizeof(float) * (size_t)B * C * H * W);
    for (int b = 0; b < B; ++b)
        for (int n = 0; n < N; ++n) {
            const float *p = pts + ((size_t)b * N + n) * stride;
            float ix = unnormalize_clip(p[0], W), iy = unnormalize_clip(p[1], H);
            float fx = floorf(ix), fy = floorf(iy);
            int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
            float nw = ((float)x1 - ix) * ((float)y1 - iy), ne = (ix - (float)x0) * ((float)y1 - iy);
            float sw = ((float)x1 - ix) * (iy - (float)y0), se = (ix - (float)x0) * (iy - (float)y0);
            for (int c = 0; c < C; ++c) {
                float *img = gplane + ((size_t)b * C + c) * H * W;
                float g = gout[((size_t)b * N + n) * C + c];
                if (y0 >= 0 && y0 < H && x0 >= 0 && x0 < W) img[(size_t)y0 * W + x0] += nw * g;
                if (y0 >= 0 && y0 < H && x1 >= 0 && x1 < W) img[(size_t)y0 * W + x1] += ne * g;
                if (y1 >= 0 && y1 < H && x0 >= 0 && x0 < W) img[(size_t)y1 * W + x0] += sw * g;
                if (y1 >= 0 && y1 < H && x1 >= 0 && x1 < W) img[(size_t)y1 * W + x1] += se * g;
            }
        }
    return 0;
}

/* F.interpolate(size=(H,W), mode='bilinear', align_corners=True) (pixel.py:107,110): ATen
 * upsample_bilinear2d: scale = (in-1)/(out-1); src = scale*dst; i0 = (int)src; i1 = i0 + (i0 < in-1). */
static inline void src_index(float scale, int dst, int in_size, int *i0, int *i1, float *l0, float *l1) {
    float real = scale * (float)dst;
    int a = (int)real;
    if (a > in_size - 1) a = in_size - 1;
    int off = (a < in_size - 1) ? 1 : 0;
    float lam = real - (float)a;
    lam = lam < 0.0f ? 0.0f : (lam > 1.0f ? 1.0f : lam);
    *i0 = a; *i1 = a + off; *l1 = lam; *l0 = 1.0f - lam;
}

T2H_ORACLE_API int t2h_oracle_upsample_bilinear_fwd(const float *in /* [B,C,h,w] */, int B, int C, int h, int w,
                                                    int H, int W, float *out /* [B,C,H,W] */) {
    float sh = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.0f;
    float sw = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.0f;
    for (size_t bc = 0; bc < (size_t)B * C; ++bc) {
        const float *src = in + bc * h * w;
        float *dst = out + bc * H * W;
        for (int y = 0; y < H; ++y) {
            int y0, y1; float ly0, ly1;
            src_index(sh, y, h, &y0, &y1, &ly0, &ly1);
            for (int x = 0; x < W; ++x) {
                int x0, x1; float lx0, lx1;
                src_index(sw, x, w, &x0, &x1, &lx0, &lx1);
                dst[(size_t)y * W + x] =
                    ly0 * (lx0 * src[(size_t)y0 * w + x0] + lx1 * src[(size_t)y0 * w + x1]) +
                    ly1 * (lx0 * src[(size_t)y1 * w + x0] + lx1 * src[(size_t)y1 * w + x1]);
            }
        }
    }
    return 0;
}

T2H_ORACLE_API int t2h_oracle_upsample_bilinear_bwd(const float *gout /* [B,C,H,W] */, int B, int C, int h, int w,
                                                    int H, int W, float *gin /* [B,C,h,w] */) {
    float sh = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.0f;
    float sw = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.0f;
    memset(gin, 0, sizeof(float) * (size_t)B * C * h * w);
    for (size_t bc = 0; bc < (size_t)B * C; ++bc) {
        float *dst = gin + bc * h * w;
        const float *src = gout + bc * H * W;
        for (int y = 0; y < H; ++y) {
            int y0, y1; float ly0, ly1;
            src_index(sh, y, h, &y0, &y1, &ly0, &ly1);
            for (int x = 0; x < W; ++x) {
                int x0, x1; float lx0, lx1;
                src_index(sw, x, w, &x0, &x1, &lx0, &lx1);
                float g = src[(size_t)y * W + x];
                dst[(size_t)y0 * w + x0] += ly0 * lx0 * g;
                dst[(size_t)y0 * w + x1] += ly0 * lx1 * g;
                dst[(size_t)y1 * w + x0] += ly1 * lx0 * g;
                dst[(size_t)y1 * w + x1] += ly1 * lx1 * g;
            }
        }
    }
    return 0;
}

/* nn.Linear: y = act_in(x) W^T + b, W is [Nout, K] row-major (torch convention). */
T2H_ORACLE_API int t2h_oracle_linear_fwd(const float *x, const float *w, const float *bias, int M, int K,
                                         int Nout, int relu_in, float *y) {
    for (int m = 0; m < M; ++m)
        for (int o = 0; o < Nout; ++o) {
            float acc = bias ? bias[o] : 0.0f;
            for (int k = 0; k < K; ++k) {
                float v = x[(size_t)m * K + k];
                if (relu_in && v < 0.0f) v = 0.0f;
                acc += v * w[(size_t)o * K + k];
            }
            y[(size_t)m * Nout + o] = acc;
        }
    return 0;
}

/* block/resnet.py:36-54 : y = shortcut(x) + fc_1(relu(fc_0(relu(x)))); shortcut bias-free or identity. */
T2H_ORACLE_API int t2h_oracle_resblock_fwd(const float *x, const float *w0, const float *b0, const float *w1,
                                           const float *b1, const float *ws /* may be NULL */, int M, int Cin,
                                           int Ch, int Cout, float *y) {
    float *hbuf = (float *)malloc(sizeof(float) * (size_t)M * Ch);
    float *dx = (float *)malloc(sizeof(float) * (size_t)M * Cout);
    if (!hbuf || !dx) { free(hbuf); free(dx); return -2; }
    t2h_oracle_linear_fwd(x, w0, b0, M, Cin, Ch, 1, hbuf);
    t2h_oracle_linear_fwd(hbuf, w1, b1, M, Ch, Cout, 1, dx);
    if (ws) {
        t2h_oracle_linear_fwd(x, ws, NULL, M, Cin, Cout, 0, y);
        for (size_t i = 0; i < (size_t)M * Cout; ++i) y[i] += dx[i];
    } else {
        if (Cin != Cout) { free(hbuf); free(dx); return -3; }
        for (size_t i = 0; i < (size_t)M * Cout; ++i) y[i] = x[i] + dx[i];
    }
    free(hbuf); free(dx);
    return 0;
}
