// 3x3 grid convolutions (stride 1, zero padding 1) of the ALTO U-Net and the pixel decoder as implicit GEMMs in exact
// fp32 on the matrix cores -- the planes the point<->grid kernels exchange with the grid side
// (reference: conv3x3 of alto.py:59-61,157-182 with F.relu at alto.py:98-99,229-230; ConvDecoder pixel.py:20-32).
//
// Layout: activations NHWC [B,H,W,C] (torch channels_last), weights [Cout][3][3][Cin] (= the channels_last memory of
// torch's [Cout,Cin,3,3] parameter), so every reduction slab of 16 input channels is one contiguous 64-byte run per
// pixel and no im2col buffer exists anywhere:
//
//   fwd    Y[p, co]  = act( sum_{tap,ci} X[p + off(tap), ci] W[co, tap, ci] + b[co] )        rows = pixels, K = 9 Cin
//   dgrad  dX[p, ci] = [dX +] ( sum_{tap,co} dY[p - off(tap), co] W[co, tap, ci] ) * (mask > 0)          K = 9 Cout
//   wgrad  dW[co, tap, ci] = [dW +] sum_p dY[p, co] X[p + off(tap), ci];  db[co] = [db +] sum_p dY[p, co]   K = pixels
//
// The kernels are gemm.hip's 128 x BN x 16 MFMA loop with a gathering loader on the activation operand (per-row pixel
// coordinates kept in registers, out-of-image taps read as zero).  Small planes with many channels (32^2 x 512: 32
// output tiles) split the reduction over grid.z into slabs in caller workspace, summed in a fixed order by the
// epilogue kernel (bias / ReLU / mask / accumulate applied there) -- deterministic, no atomics.
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

// ---- fwd / dgrad: rows = pixels -------------------------------------------------------------------------------
// A(m, k = tap * Ca + c) = act[pixel(m) + off(tap)][c], zero outside the image; 128 rows x 16 k per slab, 2 float4 per
// thread: rows (tid >> 2) and (tid >> 2) + 64, channel group tid & 3.
template <int BN, int WAVES_M, int WAVES_N, bool DGRAD, int MINW>
__global__ __launch_bounds__(NT, MINW) void conv_rows_kernel(ConvArgs p) {
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

    // pixel coordinates of this thread's two gather rows; rows past M get a y that fails every bounds test
    const int kc4 = (tid & 3) * 4;
    int gy[2], gx[2];
#pragma unroll
    for (int f = 0; f < 2; ++f) {
        const int m = m0 + (tid >> 2) + f * 64;
        gx[f] = m & (p.W - 1);
        gy[f] = m < p.M ? ((m >> p.logW) & (p.H - 1)) : (1 << 20);
    }
    float4 ra[2];
    TileLoader<BN, NT, !DGRAD, BK> lb;
    int tap = kbeg / p.Ca, c0 = kbeg - tap * p.Ca;     // position of the next slab to load

    auto load_slab = [&]() {
        const int ky = (tap * 11) >> 5, kx = tap - 3 * ky;
        const int dy = DGRAD ? 1 - ky : ky - 1, dx = DGRAD ? 1 - kx : kx - 1;
        const long long shift = ((long long)dy * p.W + dx) * p.Ca + c0 + kc4;
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            const bool ok = (unsigned)(gy[f] + dy) < (unsigned)p.H && (unsigned)(gx[f] + dx) < (unsigned)p.W;
            const long long m = m0 + (tid >> 2) + f * 64;
            ra[f] = ok ? *reinterpret_cast<const float4 *>(p.act + m * p.Ca + shift) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (DGRAD)   // B(k = co, n = ci) = W[co][tap][ci]: direct layout, rows co0.. at stride 9 Cin
            lb.load(p.mat + (size_t)tap * p.N, p.ldmat, n0, p.N, c0, p.Ca, tid, false);
        else         // B(k, n = co) = W[co][k]: k-contiguous rows
            lb.load(p.mat, p.ldmat, n0, p.N, tap * p.Ca + c0, p.K, tid, false);
        c0 += BK;
        if (c0 >= p.Ca) { c0 = 0; ++tap; }
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
template <int BM, int BN, int WAVES_M, int WAVES_N, int MINW>
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
    const int ky = (tap * 11) >> 5, kx = tap - 3 * ky;
    const int dy = n < p.N ? ky - 1 : (1 << 20), dx = kx - 1;      // columns past N fail every bounds test
    const long long shift = ((long long)(ky - 1) * p.W + dx) * p.Ca + ci;

    LoaderA la;
    float4 rb[PER];
    auto load_b = [&](int k0) {
#pragma unroll
        for (int f = 0; f < PER; ++f) {
            const int k = krow + f * KSTEP;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (BK % KSTEP == 0 || k < BK) {
                const int pix = k0 + k;
                const int x = pix & (p.W - 1), y = (pix >> p.logW) & (p.H - 1);
                const bool ok = pix < kend && (unsigned)(y + dy) < (unsigned)p.H && (unsigned)(x + dx) < (unsigned)p.W;
                if (ok) v = *reinterpret_cast<const float4 *>(p.act + (long long)pix * p.Ca + shift);
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
    const bool do_colsum = p.colsum != nullptr && tile_n == 0;

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
    const long long tiles = ((M + 127) / 128) * ((N + r.bn - 1) / r.bn);
    const int nk = K / BK;
    int splits = 1;
    if (tiles < 512) {                               // fewer tiles than two per CU: split the reduction
        splits = (int)((768 + tiles - 1) / tiles);
        const int max_splits = nk / 8 > 1 ? nk / 8 : 1;   // at least 8 slabs per split
        if (splits > max_splits) splits = max_splits;
        if (splits > 32) splits = 32;
    }
    int chunk = (nk + splits - 1) / splits;
    r.k_chunk = chunk * BK;
    r.splits = (nk + chunk - 1) / chunk;
    return r;
}

template <bool DGRAD>
int launch_rows(const ConvArgs &a, const RowsPlan &r, hipStream_t s, const char *what) {
    dim3 grid((a.N + r.bn - 1) / r.bn, (a.M + 127) / 128, r.splits);
    if (grid.y > 65535) return fail(T2H_ERR_ARG, "%s: too many pixels", what);
    if (r.bn == 128) hipLaunchKernelGGL((conv_rows_kernel<128, 2, 2, DGRAD, 4>), grid, dim3(NT), 0, s, a);
    else if (r.bn == 64) hipLaunchKernelGGL((conv_rows_kernel<64, 2, 2, DGRAD, 4>), grid, dim3(NT), 0, s, a);
    else hipLaunchKernelGGL((conv_rows_kernel<32, 4, 1, DGRAD, 4>), grid, dim3(NT), 0, s, a);
    return check_launch(what);
}

// shared body of fwd / dgrad
template <bool DGRAD>
int conv_rows(const float *act, const float *w, const float *bias, const float *mask, float *out, int B, int H, int W,
              int Cin, int Cout, int flags, void *ws, size_t ws_bytes, hipStream_t s, const char *what) {
    const int Ca = DGRAD ? Cout : Cin, N = DGRAD ? Cin : Cout;
    const long long M = (long long)B * H * W;
    RowsPlan r = rows_plan(M, N, 9 * Ca);
    ConvArgs a{};
    a.act = act; a.mat = w; a.H = H; a.W = W; a.logW = ilog2_exact(W); a.Ca = Ca;
    a.M = (int)M; a.N = N; a.K = 9 * Ca; a.ldmat = 9 * Cin; a.k_chunk = r.k_chunk;
    EpilogueArgs e{};
    e.C = out; e.bias = bias; e.mask = mask; e.M = (int)M; e.N = N; e.ldc = N; e.ldm = N;
    e.accum = flags & T2H_ACCUM; e.relu_out = flags & T2H_RELU_OUT;
    if (r.splits == 1) {
        a.C = out; a.bias = bias; a.mask = mask; a.ldc = N; a.ldm = N; a.slab_stride = 0;
        a.flags = (e.accum ? F_ACCUM : 0) | (e.relu_out ? F_RELU_OUT : 0);
        return launch_rows<DGRAD>(a, r, s, what);
    }
    const size_t need = (size_t)r.splits * M * N * sizeof(float);
    if (!ws || ws_bytes < need) return fail(T2H_ERR_WORKSPACE, "%s: workspace %zu < %zu bytes", what, ws_bytes, need);
    a.C = static_cast<float *>(ws); a.ldc = N; a.slab_stride = M * N; a.flags = 0;
    if (int rc = launch_rows<DGRAD>(a, r, s, what)) return rc;
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
WgradPlan conv_wgrad_plan(long long P, int Cin, int Cout) {
    WgradPlan p{};
    const int Ncols = 9 * Cin;
    p.bm = Cout > 64 ? 128 : (Cout > 32 ? 64 : 32);
    p.bn = 128;
    const long long tiles = (long long)((Cout + p.bm - 1) / p.bm) * ((Ncols + p.bn - 1) / p.bn);
    long long want = (1024 + tiles - 1) / tiles;
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

bool al16(const void *q) { return (uintptr_t)q % 16 == 0; }

}  // namespace
}  // namespace t2h

using namespace t2h;

T2H_API size_t t2h_conv3x3_fwd_workspace_bytes(int B, int H, int W, int Cin, int Cout) {
    if (B < 1 || H < 1 || W < 1 || Cin < 1 || Cout < 1) return 0;
    const long long M = (long long)B * H * W;
    RowsPlan r = rows_plan(M, Cout, 9 * Cin);
    return r.splits > 1 ? (size_t)r.splits * M * Cout * sizeof(float) : 0;
}

T2H_API int t2h_conv3x3_fwd(const float *x, const float *w, const float *bias, float *y, int B, int H, int W, int Cin,
                            int Cout, int flags, void *workspace, size_t workspace_bytes, t2h_stream_t stream) {
    if (!x || !w || !y) return fail(T2H_ERR_ARG, "conv3x3_fwd: null pointer");
    if (int rc = check_geometry("conv3x3_fwd", B, H, W, Cin, Cout)) return rc;
    if (Cin % 16 != 0 || Cout % 4 != 0 || !al16(x) || !al16(w) || !al16(y) || (bias && !al16(bias)))
        return fail(T2H_ERR_ARG, "conv3x3_fwd: Cin=%d must be a multiple of 16, Cout=%d of 4, pointers 16-byte aligned", Cin, Cout);
    return conv_rows<false>(x, w, bias, nullptr, y, B, H, W, Cin, Cout, flags, workspace, workspace_bytes, as_stream(stream),
                            "conv3x3_fwd");
}

T2H_API size_t t2h_conv3x3_dgrad_workspace_bytes(int B, int H, int W, int Cin, int Cout) {
    if (B < 1 || H < 1 || W < 1 || Cin < 1 || Cout < 1) return 0;
    const long long M = (long long)B * H * W;
    RowsPlan r = rows_plan(M, Cin, 9 * Cout);
    return r.splits > 1 ? (size_t)r.splits * M * Cin * sizeof(float) : 0;
}

T2H_API int t2h_conv3x3_dgrad(const float *dy, const float *w, float *dx, const float *mask, int B, int H, int W, int Cin,
                              int Cout, int flags, void *workspace, size_t workspace_bytes, t2h_stream_t stream) {
    if (!dy || !w || !dx) return fail(T2H_ERR_ARG, "conv3x3_dgrad: null pointer");
    if (int rc = check_geometry("conv3x3_dgrad", B, H, W, Cin, Cout)) return rc;
    if (Cout % 16 != 0 || Cin % 4 != 0 || !al16(dy) || !al16(w) || !al16(dx) || (mask && !al16(mask)))
        return fail(T2H_ERR_ARG, "conv3x3_dgrad: Cout=%d must be a multiple of 16, Cin=%d of 4, pointers 16-byte aligned", Cout, Cin);
    return conv_rows<true>(dy, w, nullptr, mask, dx, B, H, W, Cin, Cout, flags & T2H_ACCUM, workspace, workspace_bytes,
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
    if (p.bm == 128) hipLaunchKernelGGL((conv_wgrad_kernel<128, 128, 2, 2, 4>), grid, dim3(NT), 0, s, a);
    else if (p.bm == 64) hipLaunchKernelGGL((conv_wgrad_kernel<64, 128, 2, 2, 4>), grid, dim3(NT), 0, s, a);
    else hipLaunchKernelGGL((conv_wgrad_kernel<32, 128, 1, 4, 4>), grid, dim3(NT), 0, s, a);
    if (int rc = check_launch("conv3x3_wgrad")) return rc;
    const int accumulate = (flags & T2H_ACCUM) ? 1 : 0;
    return launch_reduce_slabs(slab, p.splits, (long long)Cout * Ncols, Cout, Ncols, Ncols, accumulate, dw, colslab, db, s);
}
