// 3x3 grid convolutions (stride 1, zero padding 1) of the ALTO U-Net and the pixel decoder as implicit GEMMs in exact
// fp32 on the matrix cores -- the planes the point<->grid kernels exchange with the grid side
// (reference: conv3x3 of alto.py:59-61,157-182 with F.relu at alto.py:98-99,226-227; ConvDecoder pixel.py:20-32).
//
// Layout: activations NHWC [B,H,W,C] (torch channels_last), weights [Cout][3][3][Cin] (= the channels_last memory of
// torch's [Cout,Cin,3,3] parameter), so every reduction slab of 16 input channels is one contiguous 64-byte run per
// pixel and no im2col buffer exists anywhere:
//
//   fwd    Y[p, co]  = act( sum_{tap,ci} X[p + off(tap), ci] W[co, tap, ci] + b[co] )        rows = pixels, K = 9 Cin
//   dgrad  dX[p, ci] = [dX +] ( sum_{tap,co} dY[p - off(tap), co] W[co, tap, ci] ) * (mask > 0)          K = 9 Cout
//   wgrad  dW[co, tap, ci] = [dW +] sum_p dY[p, co] X[p + off(tap), ci];  db[co] = [db +] sum_p dY[p, co]   K = pixels
//
// The 2x2 stride-2 transposed convolutions of the ALTO up path (upconv2x2 of alto.py, used at alto.py:175,215-218,236)
// are the same kernels with another pixel geometry: output pixel (2y+dy, 2x+dx) takes tap (dy,dx) of input pixel (y,x),
// weights [Cin][2][2][Cout] (channels_last memory of torch's [Cin,Cout,2,2]):
//
//   up fwd    Y[up(p,tap), co] = sum_ci X[p, ci] W[ci, tap, co] + b[co]       plain rows, N = 4 Cout, scattering epilogue
//   up dgrad  dX[p, ci] = [dX +] sum_{tap,co} dY[up(p,tap), co] W[ci, tap, co]                gathered rows, K = 4 Cout
//   up wgrad  dW[ci, tap, co] = [dW +] sum_p X[p, ci] dY[up(p,tap), co]                         gathered columns
//
// The kernels are gemm.hip's 128 x BN x 16 MFMA loop with a gathering loader on the activation operand (per-row pixel
// coordinates kept in registers, out-of-image taps read as zero).  Small planes with many channels (32^2 x 512: 32
// output tiles) split the reduction over grid.z into slabs in caller workspace, summed in a fixed order by the
// epilogue kernel (bias / ReLU / mask / accumulate applied there) -- deterministic, no atomics.
#include <stdlib.h>
#include <string>
#include "t2h_common.h"
#include "gemm_args.h"
#include "gemm_tile.h"

namespace t2h {
namespace {

constexpr int BK = 16;
constexpr int NT = 256;

struct ConvArgs {
    const float *act;        // gathered operand, NHWC [B,H,W,Ca]
    const float *mat;        // dense operand: weights [Cout][9][Cin] (fwd/dgrad) or dY rows [P, Cout] (wgrad)
    float *C;                // output rows or slab base
    const float *bias, *mask;
    const float *addend;     // transposed-conv forward: tensor of the output's shape added to the result, or null
    float *colsum;           // wgrad: per-split column sums of dY, [splits][Cout] or null
    int H, W, logW, Ca;
    int M, N, K;             // GEMM extents (rows, columns, reduction)
    int ldmat, ldc, ldm;
    int flags;
    int k_chunk;
    long long slab_stride;
};

__device__ inline void xcd_remap(int &tile_m, int &tile_n, int &split) {
    // same work order as gemm.hip: one contiguous run of (split, m-tile, n-tile) items per XCD
    const unsigned nb = gridDim.x * gridDim.y * gridDim.z;
    const unsigned b = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    const unsigned q = nb / 8, r = nb % 8, x = b % 8, i = b / 8;
    const unsigned t = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
    tile_n = t % gridDim.x;
    tile_m = (t / gridDim.x) % gridDim.y;
    split = t / (gridDim.x * gridDim.y);
}

enum : int { G_CONV_FWD = 0, G_CONV_DGRAD = 1, G_UP_FWD = 2, G_UP_DGRAD = 3 };

// scattering epilogue of the transposed convolution: GEMM row = input pixel, column n = tap * Cout + co goes to output
// pixel up(row, tap).  Same LDS-patch transpose as store_tiles_f32 (16 bytes per lane).
template <int TM, int TN>
__device__ inline void store_tiles_up2x2(f32x16 (&acc)[TM][TN], float *patch, int lane, int row_base, int col_base,
                                         const EpilogueArgs &e, int logW, int W, int Cout) {
    constexpr int EP = 36;
    const int er = lane >> 3, ec = (lane & 7) * 4;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int row0 = row_base + i * 32, col = col_base + j * 32 + ec;
#pragma unroll
            for (int q = 0; q < 16; ++q)
                patch[((q & 3) + 8 * (q >> 2) + 4 * (lane >> 5)) * EP + (lane & 31)] = acc[i][j][q];
            const int tap = col / Cout, co = col - tap * Cout;
            const long long tap_off = (long long)(tap >> 1) * 2 * W + (tap & 1);
            float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (e.bias && col < e.N) bv = *reinterpret_cast<const float4 *>(e.bias + co);
#pragma unroll
            for (int pass = 0; pass < 4; ++pass) {
                const int row = row0 + pass * 8 + er;
                float4 v = *reinterpret_cast<const float4 *>(patch + (pass * 8 + er) * EP + ec);
                if (row < e.M && col < e.N) {
                    v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
                    const long long opix = (long long)(row >> logW) * 4 * W + 2 * (row & (W - 1)) + tap_off;
                    float4 *dst = reinterpret_cast<float4 *>(e.C + opix * Cout + co);
                    if (e.accum) { float4 o = *dst; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
                    if (e.addend) {
                        float4 o = *reinterpret_cast<const float4 *>(e.addend + opix * Cout + co);
                        v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
                    }
                    *dst = v;
                }
            }
        }
}

// ---- fwd / dgrad: rows = pixels -------------------------------------------------------------------------------
// A(m, k = tap * Ca + c) = act[src(m, tap)][c]; 128 rows x 16 k per slab, 2 float4 per thread: rows (tid >> 2) and
// (tid >> 2) + 64, channel group tid & 3.  src = pixel(m) + off(tap), zero outside the image (3x3); pixel m itself
// (transposed-conv forward, one "tap"); up(m, tap) (transposed-conv data gradient).
template <int BN, int WAVES_M, int WAVES_N, int GEOM, int MINW>
__global__ __launch_bounds__(NT, MINW) void conv_rows_kernel(ConvArgs p) {
    constexpr bool DGRAD = GEOM == G_CONV_DGRAD;
    constexpr bool IS3X3 = GEOM == G_CONV_FWD || GEOM == G_CONV_DGRAD;
    constexpr bool B_KC = GEOM == G_CONV_FWD || GEOM == G_UP_DGRAD;
    constexpr int BM = 128;
    constexpr int TM = BM / (32 * WAVES_M), TN = BN / (32 * WAVES_N);
    constexpr int SA = BM + kPad, SB = BN + kPad;
    constexpr int LDS_MIN = (NT / 64) * 32 * 36;
    constexpr int LDS_FLOATS = 2 * BK * (SA + SB) > LDS_MIN ? 2 * BK * (SA + SB) : LDS_MIN;
    __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
    constexpr int BUF = BK * (SA + SB);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    int tile_m, tile_n, split;
    xcd_remap(tile_m, tile_n, split);
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int kbeg = split * p.k_chunk;
    const int kend = min(p.K, kbeg + p.k_chunk);
    const int nk = (kend - kbeg) / BK;

    // pixel coordinates of this thread's two gather rows; rows past M get a y that fails every bounds test.
    // 3x3: (gy, gx) = pixel coordinates; transposed conv: gy = 0 / out of range, gx = up(m, tap 0)
    const int kc4 = (tid & 3) * 4;
    int gy[2], gx[2];
#pragma unroll
    for (int f = 0; f < 2; ++f) {
        const int m = m0 + (tid >> 2) + f * 64;
        if (IS3X3) {
            gx[f] = m & (p.W - 1);
            gy[f] = m < p.M ? ((m >> p.logW) & (p.H - 1)) : (1 << 20);
        } else {
            gx[f] = (m >> p.logW) * 4 * p.W + 2 * (m & (p.W - 1));
            gy[f] = m < p.M ? 0 : (1 << 20);
        }
    }
    float4 ra[2];
    TileLoader<BN, NT, B_KC, BK> lb;
    // reduction order: channel slab outer, tap inner -- the 9 (4) taps of one 16-channel slab re-read nearly the same
    // 64-byte pixel chunks back to back, so they hit L1/L2 instead of coming over the fabric once per tap
    constexpr int NTAP = IS3X3 ? 9 : (GEOM == G_UP_DGRAD ? 4 : 1);
    int tap = (kbeg / BK) % NTAP, c0 = (kbeg / BK) / NTAP * BK;     // position of the next slab to load

    auto load_slab = [&]() {
        if (IS3X3) {
            const int ky = (tap * 11) >> 5, kx = tap - 3 * ky;
            const int dy = DGRAD ? 1 - ky : ky - 1, dx = DGRAD ? 1 - kx : kx - 1;
            const long long shift = ((long long)dy * p.W + dx) * p.Ca + c0 + kc4;
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                const bool ok = (unsigned)(gy[f] + dy) < (unsigned)p.H && (unsigned)(gx[f] + dx) < (unsigned)p.W;
                const long long m = m0 + (tid >> 2) + f * 64;
                ra[f] = ok ? *reinterpret_cast<const float4 *>(p.act + m * p.Ca + shift) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        } else {
            const long long tap_off = (long long)(tap >> 1) * 2 * p.W + (tap & 1);
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                const long long src = GEOM == G_UP_FWD ? (long long)(m0 + (tid >> 2) + f * 64) : gx[f] + tap_off;
                ra[f] = gy[f] == 0 ? *reinterpret_cast<const float4 *>(p.act + src * p.Ca + c0 + kc4)
                                   : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        if (B_KC)    // B(k, n) = mat[n][k]: k-contiguous rows (3x3 forward: W[co][tap][ci]; up dgrad: W[ci][tap][co])
            lb.load(p.mat, p.ldmat, n0, p.N, tap * p.Ca + c0, p.K, tid, false);
        else         // B(k, n) = mat[k][tap][n]: direct layout (3x3 dgrad: W[co][tap][ci]; up forward: W[ci][(tap,co)])
            lb.load(p.mat + (size_t)tap * p.N, p.ldmat, n0, p.N, c0, p.Ca, tid, false);
        if (++tap == NTAP) { tap = 0; c0 += BK; }
    };
    auto store_slab = [&](float *buf) {
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            float *q = buf + kc4 * SA + (tid >> 2) + f * 64;
            q[0] = ra[f].x; q[SA] = ra[f].y; q[2 * SA] = ra[f].z; q[3 * SA] = ra[f].w;
        }
        lb.store(buf + BK * SA, tid);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.0f;

    if (nk > 0) { load_slab(); store_slab(lds); }
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) load_slab();
        const float *a_base = lds + cur * BUF + (lane >> 5) * SA + wm * (TM * 32) + (lane & 31);
        const float *b_base = lds + cur * BUF + BK * SA + (lane >> 5) * SB + wn * (TN * 32) + (lane & 31);
        mfma_slab_f32<TM, TN, SA, SB, BK>(a_base, b_base, acc);
        if (kt + 1 < nk) store_slab(lds + (cur ^ 1) * BUF);
        __syncthreads();
    }

    EpilogueArgs e;
    e.C = p.C + (size_t)split * p.slab_stride;
    e.bias = p.bias; e.mask = p.mask; e.M = p.M; e.N = p.N; e.ldc = p.ldc; e.ldm = p.ldm;
    e.accum = p.flags & F_ACCUM; e.relu_out = p.flags & F_RELU_OUT;
    e.addend = GEOM == G_UP_FWD ? p.addend : nullptr;
    if (GEOM == G_UP_FWD)
        store_tiles_up2x2<TM, TN>(acc, lds + wave * (32 * 36), lane, m0 + wm * (TM * 32), n0 + wn * (TN * 32), e, p.logW, p.W,
                                  p.N / 4);
    else
        store_tiles_f32<TM, TN>(acc, lds + wave * (32 * 36), lane, m0 + wm * (TM * 32), n0 + wn * (TN * 32), e);
}

// out[r, c] = [out +] act( sum_z slabs[z][r, c] + bias[c] ) * (mask > 0); one float4 per thread, splits in order
__global__ __launch_bounds__(256) void reduce_rows_epilogue_kernel(const float *__restrict__ slabs, int splits,
                                                                  long long stride, int M, int N, EpilogueArgs e) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const int n4 = N / 4;
    if (idx >= (long long)M * n4) return;
    const int row = (int)(idx / n4), col = (int)(idx % n4) * 4;
    const float *src = slabs + (size_t)row * N + col;
    float4 v = *reinterpret_cast<const float4 *>(src);
    for (int z = 1; z < splits; ++z) {
        float4 t = *reinterpret_cast<const float4 *>(src + (size_t)z * stride);
        v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
    }
    if (e.bias) {
        float4 bv = *reinterpret_cast<const float4 *>(e.bias + col);
        v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
    }
    if (e.mask) {
        float4 mk = *reinterpret_cast<const float4 *>(e.mask + (size_t)row * e.ldm + col);
        v.x = mk.x > 0.f ? v.x : 0.f; v.y = mk.y > 0.f ? v.y : 0.f;
        v.z = mk.z > 0.f ? v.z : 0.f; v.w = mk.w > 0.f ? v.w : 0.f;
    }
    if (e.relu_out) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    float4 *dst = reinterpret_cast<float4 *>(e.C + (size_t)row * e.ldc + col);
    if (e.accum) { float4 o = *dst; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
    *dst = v;
}

// ---- wgrad: rows = output channels, columns = (tap, ci), reduction over pixels ----------------------------------
// A(i = co, k = pixel) = dY[pixel][co] (plain direct-layout loader, column sums = bias gradient);
// B(k = pixel, n = tap * Cin + ci) = X[pixel + off(tap)][ci]: every thread owns one float4 column group (fixed tap
// and ci, Cin % 4 == 0) and walks pixels.
// Transposed conv (UP): A = X[pixel][ci], B(k = pixel, n = tap * Cout + co) = dY[up(pixel, tap)][co].
template <int BM, int BN, int WAVES_M, int WAVES_N, int MINW, bool UP>
__global__ __launch_bounds__(NT, MINW) void conv_wgrad_kernel(ConvArgs p) {
    constexpr int TM = BM / (32 * WAVES_M), TN = BN / (32 * WAVES_N);
    constexpr int SA = BM + kPad, SB = BN + kPad;
    constexpr int LDS_MIN = (NT / 64) * 32 * 36 > 4 * NT ? (NT / 64) * 32 * 36 : 4 * NT;
    constexpr int LDS_FLOATS = 2 * BK * (SA + SB) > LDS_MIN ? 2 * BK * (SA + SB) : LDS_MIN;
    __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
    constexpr int BUF = BK * (SA + SB);
    constexpr int GROUPS = BN / 4;                 // float4 column groups per slab row
    constexpr int KSTEP = NT / GROUPS;             // slab rows covered per pass
    constexpr int PER = (BK + KSTEP - 1) / KSTEP;
    using LoaderA = TileLoader<BM, NT, false, BK>;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    int tile_m, tile_n, split;
    xcd_remap(tile_m, tile_n, split);
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int kbeg = split * p.k_chunk;
    const int kend = min(p.K, kbeg + p.k_chunk);
    const int nk = (kend - kbeg + BK - 1) / BK;

    const int ic = tid % GROUPS, krow = tid / GROUPS;
    const int n = n0 + ic * 4;
    const int tap = n / p.Ca, ci = n - tap * p.Ca;
    const int ky = UP ? tap >> 1 : (tap * 11) >> 5, kx = UP ? tap & 1 : tap - 3 * ky;
    const int dy = n < p.N ? ky - 1 : (1 << 20), dx = kx - 1;      // columns past N fail every bounds test
    const long long shift = UP ? ((long long)ky * 2 * p.W + kx) * p.Ca + ci : ((long long)(ky - 1) * p.W + dx) * p.Ca + ci;

    LoaderA la;
    float4 rb[PER];
    auto load_b = [&](int k0) {
#pragma unroll
        for (int f = 0; f < PER; ++f) {
            const int k = krow + f * KSTEP;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (BK % KSTEP == 0 || k < BK) {
                const int pix = k0 + k;
                if (UP) {
                    const long long up0 = (long long)(pix >> p.logW) * 4 * p.W + 2 * (pix & (p.W - 1));
                    if (pix < kend && n < p.N) v = *reinterpret_cast<const float4 *>(p.act + up0 * p.Ca + shift);
                } else {
                    const int x = pix & (p.W - 1), y = (pix >> p.logW) & (p.H - 1);
                    const bool ok = pix < kend && (unsigned)(y + dy) < (unsigned)p.H && (unsigned)(x + dx) < (unsigned)p.W;
                    if (ok) v = *reinterpret_cast<const float4 *>(p.act + (long long)pix * p.Ca + shift);
                }
            }
            rb[f] = v;
        }
    };
    auto store_b = [&](float *buf) {
#pragma unroll
        for (int f = 0; f < PER; ++f) {
            const int k = krow + f * KSTEP;
            if (BK % KSTEP == 0 || k < BK) *reinterpret_cast<float4 *>(buf + k * SB + ic * 4) = rb[f];
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.0f;

    float4 csum = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool do_colsum = !UP && p.colsum != nullptr && tile_n == 0;
    // transposed conv: the bias gradient is the column sum of dY = the B operand; one row tile per column tile forms it
    float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool do_bsum = UP && p.colsum != nullptr && tile_m == 0;

    if (nk > 0) {
        la.load(p.mat, p.ldmat, m0, p.M, kbeg, kend, tid, false);
        load_b(kbeg);
        la.store(lds, tid);
        store_b(lds + BK * SA);
    }
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (do_colsum) {
#pragma unroll
            for (int f = 0; f < LoaderA::PER; ++f) {
                csum.x += la.r[f].x; csum.y += la.r[f].y; csum.z += la.r[f].z; csum.w += la.r[f].w;
            }
        }
        if (do_bsum) {
#pragma unroll
            for (int f = 0; f < PER; ++f) { bsum.x += rb[f].x; bsum.y += rb[f].y; bsum.z += rb[f].z; bsum.w += rb[f].w; }
        }
        if (kt + 1 < nk) {
            la.load(p.mat, p.ldmat, m0, p.M, kbeg + (kt + 1) * BK, kend, tid, false);
            load_b(kbeg + (kt + 1) * BK);
        }
        const float *a_base = lds + cur * BUF + (lane >> 5) * SA + wm * (TM * 32) + (lane & 31);
        const float *b_base = lds + cur * BUF + BK * SA + (lane >> 5) * SB + wn * (TN * 32) + (lane & 31);
        mfma_slab_f32<TM, TN, SA, SB, BK>(a_base, b_base, acc);
        if (kt + 1 < nk) {
            la.store(lds + (cur ^ 1) * BUF, tid);
            store_b(lds + (cur ^ 1) * BUF + BK * SA);
        }
        __syncthreads();
    }

    if (do_colsum) {   // fixed-order reduction over the threads that share a column group (as gemm.hip)
        float4 *red = reinterpret_cast<float4 *>(lds);
        red[tid] = csum;
        __syncthreads();
        constexpr int CG = BM / 4;
        if (tid < CG) {
            float4 t = red[tid];
            for (int j = tid + CG; j < NT; j += CG) { t.x += red[j].x; t.y += red[j].y; t.z += red[j].z; t.w += red[j].w; }
            float *dst = p.colsum + (size_t)split * p.M + m0 + tid * 4;
            float tv[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (m0 + tid * 4 + q < p.M) dst[q] = tv[q];
        }
        __syncthreads();
    }

    if (do_bsum) {     // threads (krow) that share a column group are summed in krow order: colsum[split][n0 + 4 ic ..]
        float4 *red = reinterpret_cast<float4 *>(lds);
        red[tid] = bsum;
        __syncthreads();
        if (tid < GROUPS) {
            float4 t = red[tid];
            for (int j = tid + GROUPS; j < NT; j += GROUPS) { t.x += red[j].x; t.y += red[j].y; t.z += red[j].z; t.w += red[j].w; }
            float *dst = p.colsum + (size_t)split * p.N + n0 + tid * 4;
            if (n0 + tid * 4 < p.N) *reinterpret_cast<float4 *>(dst) = t;
        }
        __syncthreads();
    }

    EpilogueArgs e;
    e.C = p.C + (size_t)split * p.slab_stride;
    e.bias = nullptr; e.mask = nullptr; e.M = p.M; e.N = p.N; e.ldc = p.ldc; e.ldm = 0;
    e.accum = false; e.relu_out = false;
    store_tiles_f32<TM, TN>(acc, lds + wave * (32 * 36), lane, m0 + wm * (TM * 32), n0 + wn * (TN * 32), e);
}

// ---- host side ---------------------------------------------------------------------------------------------------
int ilog2_exact(int v) {
    if (v < 1 || (v & (v - 1))) return -1;
    int l = 0;
    while ((1 << l) < v) ++l;
    return l;
}

struct RowsPlan { int bn, splits, k_chunk; };
RowsPlan rows_plan(long long M, int N, int K) {
    RowsPlan r{};
    r.bn = N > 64 ? 128 : (N > 32 ? 64 : 32);
    // (narrower column tiles before splitting the reduction were A/B'd for the 32^2-128^2 planes in r02: 5.58 -> 5.61-5.66 ms
    // over the step's 3x3 layers, so the 128-wide tile + split reduction stays; only the transposed-conv forward, which
    // cannot split, narrows its tiles -- rows_plan_for)
    const long long tiles = ((M + 127) / 128) * ((N + r.bn - 1) / r.bn);
    const int nk = K / BK;
    int splits = 1;
    // fewer tiles than two per CU: split the reduction.  ~512 workgroups for planes from 256 x 256 up (768 / 1024 there:
    // +10-30 % slower), ~768 for the smaller ones (128->128 at 128^2: 65 -> 60 us; 256 / 384: slower)
    static const long long tgt_small = getenv("T2H_CONV_ROWS_WGS_SMALL") ? atoll(getenv("T2H_CONV_ROWS_WGS_SMALL")) : 768;
    static const long long tgt_large = getenv("T2H_CONV_ROWS_WGS_LARGE") ? atoll(getenv("T2H_CONV_ROWS_WGS_LARGE")) : 512;
    const long long tgt = M <= 16384 ? tgt_small : tgt_large;
    if (tiles < tgt) {
        splits = (int)((tgt + tiles - 1) / tiles);
        const int max_splits = nk / 8 > 1 ? nk / 8 : 1;   // at least 8 slabs per split
        if (splits > max_splits) splits = max_splits;
        if (splits > 32) splits = 32;
    }
    int chunk = (nk + splits - 1) / splits;
    r.k_chunk = chunk * BK;
    r.splits = (nk + chunk - 1) / chunk;
    return r;
}

template <int GEOM>
int launch_rows(const ConvArgs &a, const RowsPlan &r, hipStream_t s, const char *what) {
    dim3 grid((a.N + r.bn - 1) / r.bn, (a.M + 127) / 128, r.splits);
    if (grid.y > 65535) return fail(T2H_ERR_ARG, "%s: too many pixels", what);
    if (r.bn == 128) hipLaunchKernelGGL((conv_rows_kernel<128, 2, 2, GEOM, 4>), grid, dim3(NT), 0, s, a);
    else if (r.bn == 64) hipLaunchKernelGGL((conv_rows_kernel<64, 2, 2, GEOM, 4>), grid, dim3(NT), 0, s, a);
    else hipLaunchKernelGGL((conv_rows_kernel<32, 4, 1, GEOM, 4>), grid, dim3(NT), 0, s, a);
    static const std::string names[3] = {"conv_rows_kernel<128,2,2," + std::to_string(GEOM) + ",4>",
                                         "conv_rows_kernel<64,2,2," + std::to_string(GEOM) + ",4>",
                                         "conv_rows_kernel<32,4,1," + std::to_string(GEOM) + ",4>"};
    note_kernel(names[r.bn == 128 ? 0 : (r.bn == 64 ? 1 : 2)].c_str());
    return check_launch(what);
}

// GEMM extents of the four row-streaming geometries; H, W = dims of the plane whose pixels are the GEMM rows
struct RowsShape { int Ca, N, K, ldmat; };
template <int GEOM>
RowsShape rows_shape(int Cin, int Cout) {
    if (GEOM == G_CONV_FWD) return {Cin, Cout, 9 * Cin, 9 * Cin};
    if (GEOM == G_CONV_DGRAD) return {Cout, Cin, 9 * Cout, 9 * Cin};
    if (GEOM == G_UP_FWD) return {Cin, 4 * Cout, Cin, 4 * Cout};
    return {Cout, Cin, 4 * Cout, 4 * Cout};
}
template <int GEOM>
RowsPlan rows_plan_for(long long M, int Cin, int Cout) {
    RowsShape sh = rows_shape<GEOM>(Cin, Cout);
    RowsPlan r = rows_plan(M, sh.N, sh.K);
    if (GEOM == G_UP_FWD && r.splits > 1) {
        // scattering epilogue: no slabs, so parallelism can only come from narrower column tiles (a 32 x 32 plane with
        // 512 -> 256 channels is a 1024 x 1024 x 512 GEMM: 64 tiles of 128 x 128 would occupy a quarter of the CUs)
        r.splits = 1; r.k_chunk = sh.K;
        const long long row_tiles = (M + 127) / 128;
        static const long long up_tiles = getenv("T2H_UPCONV_FWD_TILES") ? atoll(getenv("T2H_UPCONV_FWD_TILES")) : 256;
        while (r.bn > 32 && row_tiles * ((sh.N + r.bn - 1) / r.bn) < up_tiles) r.bn >>= 1;
    }
    return r;
}

// shared body of the row-streaming entry points
template <int GEOM>
int conv_rows(const float *act, const float *w, const float *bias, const float *mask, float *out, int B, int H, int W,
              int Cin, int Cout, int flags, void *ws, size_t ws_bytes, hipStream_t s, const char *what,
              const float *addend = nullptr) {
    const RowsShape sh = rows_shape<GEOM>(Cin, Cout);
    const int Ca = sh.Ca, N = sh.N;
    const long long M = (long long)B * H * W;
    RowsPlan r = rows_plan_for<GEOM>(M, Cin, Cout);
    ConvArgs a{};
    a.act = act; a.mat = w; a.H = H; a.W = W; a.logW = ilog2_exact(W); a.Ca = Ca; a.addend = addend;
    a.M = (int)M; a.N = N; a.K = sh.K; a.ldmat = sh.ldmat; a.k_chunk = r.k_chunk;
    EpilogueArgs e{};
    e.C = out; e.bias = bias; e.mask = mask; e.M = (int)M; e.N = N; e.ldc = N; e.ldm = N;
    e.accum = flags & T2H_ACCUM; e.relu_out = flags & T2H_RELU_OUT;
    if (r.splits == 1) {
        a.C = out; a.bias = bias; a.mask = mask; a.ldc = N; a.ldm = N; a.slab_stride = 0;
        a.flags = (e.accum ? F_ACCUM : 0) | (e.relu_out ? F_RELU_OUT : 0);
        return launch_rows<GEOM>(a, r, s, what);
    }
    const size_t need = (size_t)r.splits * M * N * sizeof(float);
    if (!ws || ws_bytes < need) return fail(T2H_ERR_WORKSPACE, "%s: workspace %zu < %zu bytes", what, ws_bytes, need);
    a.C = static_cast<float *>(ws); a.ldc = N; a.slab_stride = M * N; a.flags = 0;
    if (int rc = launch_rows<GEOM>(a, r, s, what)) return rc;
    const long long total = M * (N / 4);
    hipLaunchKernelGGL(reduce_rows_epilogue_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s,
                       static_cast<const float *>(ws), r.splits, M * N, (int)M, N, e);
    return check_launch(what);
}

int check_geometry(const char *what, int B, int H, int W, int Cin, int Cout) {
    if (B < 1 || H < 1 || W < 1 || Cin < 1 || Cout < 1) return fail(T2H_ERR_ARG, "%s: bad shape", what);
    if (ilog2_exact(H) < 0 || ilog2_exact(W) < 0 || H > 32768 || W > 32768)
        return fail(T2H_ERR_ARG, "%s: H=%d, W=%d must be powers of two", what, H, W);
    if ((long long)B * H * W > (1LL << 30)) return fail(T2H_ERR_ARG, "%s: more than 2^30 pixels", what);
    return T2H_OK;
}

struct WgradPlan { int bm, bn, splits, k_chunk; };
// rows = channels of the dense operand, ncols = taps x channels of the gathered one, P = pixels reduced over
WgradPlan wgrad_plan_for(long long P, int rows, int ncols) {
    WgradPlan p{};
    p.bm = rows > 64 ? 128 : (rows > 32 ? 64 : 32);
    p.bn = 128;
    const long long tiles = (long long)((rows + p.bm - 1) / p.bm) * ((ncols + p.bn - 1) / p.bn);
    // planes up to 128 x 128 (short reductions): 512 workgroups beat the full resident wave by 3-12 % (128->128 at 128^2:
    // 70 -> 61 us, 256->256 at 64^2: 70 -> 63) -- half the slabs to write and reduce, twice the main loop per epilogue;
    // from 256 x 256 up 1024 stays (512: +4-9 % slower there; 2048 wins or loses by shape within +-5 %)
    // (T2H_CONV_WGRAD_WGS: the A/B of DESIGN.md section 4 -- fewer workgroups = fewer split slabs = less traffic, more time)
    static const long long forced = getenv("T2H_CONV_WGRAD_WGS") ? atoll(getenv("T2H_CONV_WGRAD_WGS")) : 0;
    const long long target = forced > 0 ? forced : (P <= 16384 ? 512 : 1024);
    long long want = target / tiles;       // <= 1024 workgroups = one resident wave; 1025 would leave one running alone
                                         // (measured: 64->128 at 512^2, 5 tiles: 464 us with 1025 workgroups, 390 with 1020)
    if (want > 512) want = 512;
    long long max_splits = (P + 16 * BK - 1) / (16 * BK);      // at least 16 slabs per workgroup
    long long splits = want < max_splits ? want : max_splits;
    if (splits < 1) splits = 1;
    long long chunk = (P + splits - 1) / splits;
    chunk = (chunk + 31) / 32 * 32;
    p.k_chunk = (int)chunk;
    p.splits = (int)((P + chunk - 1) / chunk);
    return p;
}
WgradPlan conv_wgrad_plan(long long P, int Cin, int Cout) { return wgrad_plan_for(P, Cout, 9 * Cin); }

bool al16(const void *q) { return (uintptr_t)q % 16 == 0; }

}  // namespace

int launch_reduce_rows_epilogue(const float *slabs, int splits, long long stride, long long M, int N, const EpilogueArgs &e,
                                hipStream_t s) {
    const long long total = M * (N / 4);
    hipLaunchKernelGGL(reduce_rows_epilogue_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, slabs, splits, stride,
                       (int)M, N, e);
    return check_launch("reduce_rows_epilogue");
}

}  // namespace t2h

using namespace t2h;

T2H_API size_t t2h_conv3x3_fwd_workspace_bytes(int B, int H, int W, int Cin, int Cout) {
    if (B < 1 || H < 1 || W < 1 || Cin < 1 || Cout < 1) return 0;
    const long long M = (long long)B * H * W;
    RowsPlan r = rows_plan_for<G_CONV_FWD>(M, Cin, Cout);
    return r.splits > 1 ? (size_t)r.splits * M * Cout * sizeof(float) : 0;
}

T2H_API int t2h_conv3x3_fwd(const float *x, const float *w, const float *bias, float *y, int B, int H, int W, int Cin,
                            int Cout, int flags, void *workspace, size_t workspace_bytes, t2h_stream_t stream) {
    if (!x || !w || !y) return fail(T2H_ERR_ARG, "conv3x3_fwd: null pointer");
    if (int rc = check_geometry("conv3x3_fwd", B, H, W, Cin, Cout)) return rc;
    if (Cin % 16 != 0 || Cout % 4 != 0 || !al16(x) || !al16(w) || !al16(y) || (bias && !al16(bias)))
        return fail(T2H_ERR_ARG, "conv3x3_fwd: Cin=%d must be a multiple of 16, Cout=%d of 4, pointers 16-byte aligned", Cin, Cout);
    return conv_rows<G_CONV_FWD>(x, w, bias, nullptr, y, B, H, W, Cin, Cout, flags, workspace, workspace_bytes, as_stream(stream),
                            "conv3x3_fwd");
}

T2H_API size_t t2h_conv3x3_dgrad_workspace_bytes(int B, int H, int W, int Cin, int Cout) {
    if (B < 1 || H < 1 || W < 1 || Cin < 1 || Cout < 1) return 0;
    const long long M = (long long)B * H * W;
    RowsPlan r = rows_plan_for<G_CONV_DGRAD>(M, Cin, Cout);
    return r.splits > 1 ? (size_t)r.splits * M * Cin * sizeof(float) : 0;
}

T2H_API int t2h_conv3x3_dgrad(const float *dy, const float *w, float *dx, const float *mask, int B, int H, int W, int Cin,
                              int Cout, int flags, void *workspace, size_t workspace_bytes, t2h_stream_t stream) {
    if (!dy || !w || !dx) return fail(T2H_ERR_ARG, "conv3x3_dgrad: null pointer");
    if (int rc = check_geometry("conv3x3_dgrad", B, H, W, Cin, Cout)) return rc;
    if (Cout % 16 != 0 || Cin % 4 != 0 || !al16(dy) || !al16(w) || !al16(dx) || (mask && !al16(mask)))
        return fail(T2H_ERR_ARG, "conv3x3_dgrad: Cout=%d must be a multiple of 16, Cin=%d of 4, pointers 16-byte aligned", Cout, Cin);
    return conv_rows<G_CONV_DGRAD>(dy, w, nullptr, mask, dx, B, H, W, Cin, Cout, flags & T2H_ACCUM, workspace, workspace_bytes,
                           as_stream(stream), "conv3x3_dgrad");
}

T2H_API size_t t2h_conv3x3_wgrad_workspace_bytes(int B, int H, int W, int Cin, int Cout) {
    if (B < 1 || H < 1 || W < 1 || Cin < 1 || Cout < 1) return 0;
    WgradPlan p = conv_wgrad_plan((long long)B * H * W, Cin, Cout);
    return (size_t)p.splits * ((size_t)Cout * 9 * Cin + Cout) * sizeof(float);
}

T2H_API int t2h_conv3x3_wgrad(const float *dy, const float *x, float *dw, float *db, int B, int H, int W, int Cin, int Cout,
                              int flags, void *workspace, size_t workspace_bytes, t2h_stream_t stream) {
    if (!dy || !x || !dw) return fail(T2H_ERR_ARG, "conv3x3_wgrad: null pointer");
    if (int rc = check_geometry("conv3x3_wgrad", B, H, W, Cin, Cout)) return rc;
    if (Cin % 4 != 0 || Cout % 4 != 0 || !al16(dy) || !al16(x))
        return fail(T2H_ERR_ARG, "conv3x3_wgrad: Cin=%d and Cout=%d must be multiples of 4, pointers 16-byte aligned", Cin, Cout);
    const size_t need = t2h_conv3x3_wgrad_workspace_bytes(B, H, W, Cin, Cout);
    if (!workspace || workspace_bytes < need)
        return fail(T2H_ERR_WORKSPACE, "conv3x3_wgrad: workspace %zu < %zu bytes", workspace_bytes, need);
    hipStream_t s = as_stream(stream);
    const long long P = (long long)B * H * W;
    WgradPlan p = conv_wgrad_plan(P, Cin, Cout);
    const int Ncols = 9 * Cin;
    float *slab = static_cast<float *>(workspace);
    float *colslab = slab + (size_t)p.splits * Cout * Ncols;
    ConvArgs a{};
    a.act = x; a.mat = dy; a.C = slab; a.colsum = db ? colslab : nullptr;
    a.H = H; a.W = W; a.logW = ilog2_exact(W); a.Ca = Cin;
    a.M = Cout; a.N = Ncols; a.K = (int)P; a.ldmat = Cout; a.ldc = Ncols;
    a.k_chunk = p.k_chunk; a.slab_stride = (long long)Cout * Ncols;
    dim3 grid((Ncols + p.bn - 1) / p.bn, (Cout + p.bm - 1) / p.bm, p.splits);
    if (grid.z > 65535) return fail(T2H_ERR_ARG, "conv3x3_wgrad: too many splits");
    if (p.bm == 128) hipLaunchKernelGGL((conv_wgrad_kernel<128, 128, 2, 2, 4, false>), grid, dim3(NT), 0, s, a);
    else if (p.bm == 64) hipLaunchKernelGGL((conv_wgrad_kernel<64, 128, 2, 2, 4, false>), grid, dim3(NT), 0, s, a);
    else hipLaunchKernelGGL((conv_wgrad_kernel<32, 128, 1, 4, 4, false>), grid, dim3(NT), 0, s, a);
    note_kernel(p.bm == 128 ? "conv_wgrad_kernel<128,128,2,2,4,false>" : (p.bm == 64 ? "conv_wgrad_kernel<64,128,2,2,4,false>" : "conv_wgrad_kernel<32,128,1,4,4,false>"));
    if (int rc = check_launch("conv3x3_wgrad")) return rc;
    const int accumulate = (flags & T2H_ACCUM) ? 1 : 0;
    return launch_reduce_slabs(slab, p.splits, (long long)Cout * Ncols, Cout, Ncols, Ncols, accumulate, dw, colslab, db, s, 0, 0,
                               (flags & T2H_DEFER_REDUCE) != 0);
}

// ---- ConvTranspose2d(kernel_size=2, stride=2): H, W are the INPUT plane's dims, the output is 2H x 2W ----------------
static int check_up(const char *what, int B, int H, int W, int Cin, int Cout) {
    if (int rc = check_geometry(what, B, H, W, Cin, Cout)) return rc;
    if ((long long)B * H * W > (1LL << 28)) return fail(T2H_ERR_ARG, "%s: more than 2^28 input pixels", what);
    if (Cin % 16 != 0 || Cout % 16 != 0) return fail(T2H_ERR_ARG, "%s: Cin=%d and Cout=%d must be multiples of 16", what, Cin, Cout);
    return T2H_OK;
}

T2H_API int t2h_upconv2x2_fwd_add(const float *x, const float *w, const float *bias, const float *addend, float *y, int B,
                                  int H, int W, int Cin, int Cout, int flags, t2h_stream_t stream) {
    if (!x || !w || !y) return fail(T2H_ERR_ARG, "upconv2x2_fwd: null pointer");
    if (int rc = check_up("upconv2x2_fwd", B, H, W, Cin, Cout)) return rc;
    if (!al16(x) || !al16(w) || !al16(y) || (bias && !al16(bias)) || (addend && !al16(addend)))
        return fail(T2H_ERR_ARG, "upconv2x2_fwd: pointers must be 16-byte aligned");
    return conv_rows<G_UP_FWD>(x, w, bias, nullptr, y, B, H, W, Cin, Cout, flags & T2H_ACCUM, nullptr, 0, as_stream(stream),
                               "upconv2x2_fwd", addend);
}

T2H_API int t2h_upconv2x2_fwd(const float *x, const float *w, const float *bias, float *y, int B, int H, int W, int Cin,
                              int Cout, int flags, t2h_stream_t stream) {
    return t2h_upconv2x2_fwd_add(x, w, bias, nullptr, y, B, H, W, Cin, Cout, flags, stream);
}

T2H_API size_t t2h_upconv2x2_dgrad_workspace_bytes(int B, int H, int W, int Cin, int Cout) {
    if (B < 1 || H < 1 || W < 1 || Cin < 1 || Cout < 1) return 0;
    const long long M = (long long)B * H * W;
    RowsPlan r = rows_plan_for<G_UP_DGRAD>(M, Cin, Cout);
    return r.splits > 1 ? (size_t)r.splits * M * Cin * sizeof(float) : 0;
}

T2H_API int t2h_upconv2x2_dgrad(const float *dy, const float *w, float *dx, int B, int H, int W, int Cin, int Cout, int flags,
                                void *workspace, size_t workspace_bytes, t2h_stream_t stream) {
    if (!dy || !w || !dx) return fail(T2H_ERR_ARG, "upconv2x2_dgrad: null pointer");
    if (int rc = check_up("upconv2x2_dgrad", B, H, W, Cin, Cout)) return rc;
    if (!al16(dy) || !al16(w) || !al16(dx)) return fail(T2H_ERR_ARG, "upconv2x2_dgrad: pointers must be 16-byte aligned");
    return conv_rows<G_UP_DGRAD>(dy, w, nullptr, nullptr, dx, B, H, W, Cin, Cout, flags & T2H_ACCUM, workspace, workspace_bytes,
                                 as_stream(stream), "upconv2x2_dgrad");
}

T2H_API size_t t2h_upconv2x2_wgrad_workspace_bytes(int B, int H, int W, int Cin, int Cout) {
    if (B < 1 || H < 1 || W < 1 || Cin < 1 || Cout < 1) return 0;
    WgradPlan p = wgrad_plan_for((long long)B * H * W, Cin, 4 * Cout);
    return (size_t)p.splits * ((size_t)Cin * 4 * Cout + 4 * Cout) * sizeof(float);
}

T2H_API int t2h_upconv2x2_wgrad(const float *dy, const float *x, float *dw, int B, int H, int W, int Cin, int Cout, int flags,
                                void *workspace, size_t workspace_bytes, t2h_stream_t stream) {
    return t2h_upconv2x2_wgrad_bias(dy, x, dw, nullptr, B, H, W, Cin, Cout, flags, workspace, workspace_bytes, stream);
}

T2H_API int t2h_upconv2x2_wgrad_bias(const float *dy, const float *x, float *dw, float *db, int B, int H, int W, int Cin,
                                     int Cout, int flags, void *workspace, size_t workspace_bytes, t2h_stream_t stream) {
    if (!dy || !x || !dw) return fail(T2H_ERR_ARG, "upconv2x2_wgrad: null pointer");
    if (int rc = check_up("upconv2x2_wgrad", B, H, W, Cin, Cout)) return rc;
    if (!al16(dy) || !al16(x)) return fail(T2H_ERR_ARG, "upconv2x2_wgrad: pointers must be 16-byte aligned");
    const size_t need = t2h_upconv2x2_wgrad_workspace_bytes(B, H, W, Cin, Cout);
    if (!workspace || workspace_bytes < need)
        return fail(T2H_ERR_WORKSPACE, "upconv2x2_wgrad: workspace %zu < %zu bytes", workspace_bytes, need);
    hipStream_t s = as_stream(stream);
    const long long P = (long long)B * H * W;
    const int Ncols = 4 * Cout;
    WgradPlan p = wgrad_plan_for(P, Cin, Ncols);
    float *slab = static_cast<float *>(workspace);
    ConvArgs a{};
    float *colslab = slab + (size_t)p.splits * Cin * Ncols;            // [splits][4 taps][Cout]: column sums of dY
    a.act = dy; a.mat = x; a.C = slab; a.colsum = db ? colslab : nullptr;
    a.H = H; a.W = W; a.logW = ilog2_exact(W); a.Ca = Cout;
    a.M = Cin; a.N = Ncols; a.K = (int)P; a.ldmat = Cin; a.ldc = Ncols;
    a.k_chunk = p.k_chunk; a.slab_stride = (long long)Cin * Ncols;
    dim3 grid((Ncols + p.bn - 1) / p.bn, (Cin + p.bm - 1) / p.bm, p.splits);
    if (grid.z > 65535) return fail(T2H_ERR_ARG, "upconv2x2_wgrad: too many splits");
    if (p.bm == 128) hipLaunchKernelGGL((conv_wgrad_kernel<128, 128, 2, 2, 4, true>), grid, dim3(NT), 0, s, a);
    else if (p.bm == 64) hipLaunchKernelGGL((conv_wgrad_kernel<64, 128, 2, 2, 4, true>), grid, dim3(NT), 0, s, a);
    else hipLaunchKernelGGL((conv_wgrad_kernel<32, 128, 1, 4, 4, true>), grid, dim3(NT), 0, s, a);
    note_kernel(p.bm == 128 ? "conv_wgrad_kernel<128,128,2,2,4,true>" : (p.bm == 64 ? "conv_wgrad_kernel<64,128,2,2,4,true>" : "conv_wgrad_kernel<32,128,1,4,4,true>"));
    if (int rc = check_launch("upconv2x2_wgrad")) return rc;
    // the bias gradient sums the column slabs over the splits AND the four taps: [splits * 4][Cout] rows of length Cout
    return launch_reduce_slabs(slab, p.splits, (long long)Cin * Ncols, Cin, Ncols, Ncols, (flags & T2H_ACCUM) ? 1 : 0, dw,
                               db ? colslab : nullptr, db, s, db ? 4 * p.splits : 0, Cout, (flags & T2H_DEFER_REDUCE) != 0);
}
