// Fused PointNet trunk block (reference: encoder/pointnet.py:72-82, 92-99; block/resnet.py:36-54), hidden_dim = 32.
//
// One launch per ResnetBlockFC(64 -> 32), over the cell-sorted rows of a tile:
//
//     X    = [net_prev | pool_local(net_prev)]            (first block: X = fc_pos(points), pointnet.py:72)
//     hr   = relu(fc_0(relu(X)))                          (saved for the backward)
//     out  = shortcut(X) + (fc_1(hr) + b1)                (resnet.py:47-54)
//     c    = fc_c(relu(out))                              (last block only, pointnet.py:81-82)
//
// The reference's scatter_max + gather + torch.cat (pointnet.py:76-78) has no kernel of its own here: the
// segmented max over each point's finest-level cell runs in the LOADER of the block that consumes it.  A workgroup
// owns 128 consecutive sorted rows; cells are contiguous row runs, so the rows of every cell that intersects the
// tile are either in the tile (LDS) or a short run just outside it, which is streamed from global memory (the
// neighbouring tile reads the same rows and derives the same maximum, so nothing is exchanged between workgroups).
// The pooled half therefore never travels through HBM, and `pool_max` disappears from the trace.  Arg-max semantics
// are torch_scatter's CPU ones (strict '>': the first row of a cell wins ties; NaN never wins), recorded as one bit
// per (row, channel) for the backward.
//
// GEMMs: v_mfma_f32_32x32x2_f32 (exact fp32).  The X tile [128][64] and the three weight matrices (20 KB) sit in
// LDS row-major with a 4-float row pad; a lane reads 4 consecutive k of its row with one ds_read_b128 and feeds four
// MFMAs from it (the k order inside an MFMA pair is free as long as A and B agree), conflict-free for both operands.
// Each of the 4 waves owns 32 rows: 64 + 16 (+ 16) MFMAs per tile.
#include <float.h>

#include "t2h_common.h"

namespace t2h {
namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int TR = 128;          // rows per workgroup
constexpr int XS = 68;           // row stride of 64-wide LDS tiles (floats)
constexpr int HS = 36;           // row stride of 32-wide LDS tiles
constexpr int G = 8;             // lanes per 32-channel row (float4 each)
constexpr int NG = 256 / G;      // row groups per pass

struct TrunkFwdArgs {
    const float *pts; int dim; const float *wpos, *bpos;                  // first block
    const float *net_prev; int ld_prev; const int32_t *cell, *off0;       // later blocks
    const float *w0, *b0, *w1, *b1, *ws, *wc, *bc;
    int M;
    float *x_full;       // optional [M, 64]: the block input materialised (tests / unfused backward)
    float *hr;           // [M, 32]
    float *out; int ld_out;
    uint8_t *winner;     // [M, 8]: arg-max bits of net_prev's pooling (later blocks)
    float *c_out;        // [M, 32] (last block)
};

__device__ inline float clean(float x) { return x != x ? -FLT_MAX : x; }

struct Best { float4 v; int4 a; };
__device__ inline void best_init(Best &b) {
    b.v = make_float4(-FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX);
    b.a = make_int4(-1, -1, -1, -1);
}
__device__ inline void best_strict(Best &b, float4 v, int n) {            // rows visited in ascending order
    v.x = clean(v.x); v.y = clean(v.y); v.z = clean(v.z); v.w = clean(v.w);
    if (v.x > b.v.x) { b.v.x = v.x; b.a.x = n; }
    if (v.y > b.v.y) { b.v.y = v.y; b.a.y = n; }
    if (v.z > b.v.z) { b.v.z = v.z; b.a.z = n; }
    if (v.w > b.v.w) { b.v.w = v.w; b.a.w = n; }
}
__device__ inline void best_merge(float &bv, int &ba, float v, int a) {    // order-free: ties keep the smaller index
    if (a >= 0 && (v > bv || (v == bv && (ba < 0 || a < ba)))) { bv = v; ba = a; }
}

// Segmented max of the tile's 32-channel rows (left half of Xs) over finest-level cells -> right half of Xs, winner
// bits to global.  scratch: >= 4096 floats of LDS.  Must be called by all 256 threads; ends with a barrier.
__device__ inline void pool_into_tile(const TrunkFwdArgs &a, float *Xs, float *scratch, int r0, int r1, int tid) {
    float4 *pval = reinterpret_cast<float4 *>(scratch);                   // [NG * G]
    int4 *parg = reinterpret_cast<int4 *>(scratch + 4 * NG * G);          // [NG * G]
    float4 *oval = reinterpret_cast<float4 *>(scratch + 8 * NG * G);      // [2 * G]
    int4 *oarg = reinterpret_cast<int4 *>(scratch + 8 * NG * G + 8 * G);  // [2 * G]
    int *bounds = reinterpret_cast<int *>(scratch + 8 * NG * G + 16 * G); // [4]
    short4 *argv = reinterpret_cast<short4 *>(scratch + 8 * NG * G + 16 * G + 4);   // [TR * G] (8 bytes each)
    const int lane = tid & (G - 1), grp = tid >> 3;
    int segs[TR / NG], sege[TR / NG];
#pragma unroll
    for (int p = 0; p < TR / NG; ++p) {
        const int row = r0 + p * NG + grp;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        segs[p] = sege[p] = -1;
        if (row < r1) {
            const int cid = a.cell[row];
            segs[p] = a.off0[cid]; sege[p] = a.off0[cid + 1];
            v = *reinterpret_cast<const float4 *>(a.net_prev + (size_t)row * a.ld_prev + lane * 4);
        }
        *reinterpret_cast<float4 *>(Xs + (p * NG + grp) * XS + lane * 4) = v;
    }
    if (tid == 0) {
        const int ch = a.cell[r0], ct = a.cell[r1 - 1];
        bounds[0] = a.off0[ch]; bounds[1] = a.off0[ch + 1]; bounds[2] = a.off0[ct]; bounds[3] = a.off0[ct + 1];
    }
    __syncthreads();
    // rows of the border cells that lie outside the tile: [bounds[0], r0) and [r1, bounds[3])
#pragma unroll
    for (int side = 0; side < 2; ++side) {
        const int lo = side == 0 ? bounds[0] : r1, hi = side == 0 ? r0 : bounds[3];
        if (hi <= lo) continue;                                           // uniform over the workgroup
        Best b; best_init(b);
        for (int n = lo + grp; n < hi; n += NG)
            best_strict(b, *reinterpret_cast<const float4 *>(a.net_prev + (size_t)n * a.ld_prev + lane * 4), n);
        pval[grp * G + lane] = b.v; parg[grp * G + lane] = b.a;
        __syncthreads();
        if (grp == 0) {
            Best t; t.v = pval[lane]; t.a = parg[lane];
            for (int g = 1; g < NG; ++g) {
                const float4 v = pval[g * G + lane]; const int4 ar = parg[g * G + lane];
                best_merge(t.v.x, t.a.x, v.x, ar.x); best_merge(t.v.y, t.a.y, v.y, ar.y);
                best_merge(t.v.z, t.a.z, v.z, ar.z); best_merge(t.v.w, t.a.w, v.w, ar.w);
            }
            oval[side * G + lane] = t.v; oarg[side * G + lane] = t.a;
        }
        __syncthreads();
    }
    // the first in-tile row of every cell reduces its cell from LDS, in ascending row order
#pragma unroll
    for (int p = 0; p < TR / NG; ++p) {
        const int row = r0 + p * NG + grp;
        if (row >= r1 || row != max(segs[p], r0)) continue;
        Best b; best_init(b);
        if (segs[p] < r0) {                                               // earlier rows win ties
            const int4 ar = oarg[lane];
            b.v = oval[lane];
            b.a = make_int4(ar.x < 0 ? -1 : -2, ar.y < 0 ? -1 : -2, ar.z < 0 ? -1 : -2, ar.w < 0 ? -1 : -2);
        }
        const int end = min(sege[p], r1);
        for (int n = row; n < end; ++n)
            best_strict(b, *reinterpret_cast<const float4 *>(Xs + (n - r0) * XS + lane * 4), n - r0);
        if (sege[p] > r1) {                                               // later rows: only a larger value wins
            const float4 v = oval[G + lane];
            if (v.x > b.v.x) { b.v.x = v.x; b.a.x = -2; }
            if (v.y > b.v.y) { b.v.y = v.y; b.a.y = -2; }
            if (v.z > b.v.z) { b.v.z = v.z; b.a.z = -2; }
            if (v.w > b.v.w) { b.v.w = v.w; b.a.w = -2; }
        }
        // untouched (all NaN) -> 0, as torch_scatter's fill of cells that no value entered
        *reinterpret_cast<float4 *>(Xs + (row - r0) * XS + 32 + lane * 4) =
            make_float4(b.a.x == -1 ? 0.f : b.v.x, b.a.y == -1 ? 0.f : b.v.y, b.a.z == -1 ? 0.f : b.v.z, b.a.w == -1 ? 0.f : b.v.w);
        argv[(row - r0) * G + lane] = make_short4((short)b.a.x, (short)b.a.y, (short)b.a.z, (short)b.a.w);
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < TR / NG; ++p) {
        const int row = r0 + p * NG + grp;
        if (row >= r1) {
            *reinterpret_cast<float4 *>(Xs + (p * NG + grp) * XS + 32 + lane * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
            continue;
        }
        const int slot = max(segs[p], r0) - r0, me = row - r0;
        const float4 v = *reinterpret_cast<const float4 *>(Xs + slot * XS + 32 + lane * 4);
        const short4 ar = argv[slot * G + lane];
        if (slot != me) *reinterpret_cast<float4 *>(Xs + me * XS + 32 + lane * 4) = v;
        a.winner[(size_t)row * G + lane] = (uint8_t)((ar.x == me) | ((ar.y == me) << 1) | ((ar.z == me) << 2) | ((ar.w == me) << 3));
    }
    __syncthreads();
}

// 32 rows x 32 columns x K of  A[m][k] * B[n][k]  with both operands row-major in LDS (strides SA, SB); RELU_A applies
// max(., 0) to A.  pa / pb already point at this lane's row and k offset 4 * (lane >> 5).
template <int K, bool RELU_A>
__device__ inline void mfma_rows(const float *pa, const float *pb, f32x16 &acc) {
#pragma unroll
    for (int g = 0; g < K / 8; ++g) {
        const float4 x = *reinterpret_cast<const float4 *>(pa + 8 * g);
        const float4 w = *reinterpret_cast<const float4 *>(pb + 8 * g);
        const float xv[4] = {x.x, x.y, x.z, x.w}, wv[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
        for (int j = 0; j < 4; ++j)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(RELU_A ? fmaxf(xv[j], 0.f) : xv[j], wv[j], acc, 0, 0, 0);
    }
}

// Two products over the same A rows in one sweep: acc_r += relu(A) B0^T, acc_s += A B1^T
template <int K>
__device__ inline void mfma_rows_pair(const float *pa, const float *pb0, const float *pb1, f32x16 &acc_r, f32x16 &acc_s) {
#pragma unroll
    for (int g = 0; g < K / 8; ++g) {
        const float4 x = *reinterpret_cast<const float4 *>(pa + 8 * g);
        const float4 w0 = *reinterpret_cast<const float4 *>(pb0 + 8 * g);
        const float4 w1 = *reinterpret_cast<const float4 *>(pb1 + 8 * g);
        const float xv[4] = {x.x, x.y, x.z, x.w}, w0v[4] = {w0.x, w0.y, w0.z, w0.w}, w1v[4] = {w1.x, w1.y, w1.z, w1.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acc_r = __builtin_amdgcn_mfma_f32_32x32x2f32(fmaxf(xv[j], 0.f), w0v[j], acc_r, 0, 0, 0);
            acc_s = __builtin_amdgcn_mfma_f32_32x32x2f32(xv[j], w1v[j], acc_s, 0, 0, 0);
        }
    }
}

// C/D layout of the 32x32 MFMA: col = lane & 31, row = (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5)
__device__ inline int acc_row(int q, int lane) { return (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5); }

// one wave stores its 32 x 32 LDS tile (stride HS) to global rows [row0, row0 + 32) as float4 rows
__device__ inline void store_tile_rows(const float *tile, float *dst, int ld, int row0, int M, int lane) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = i * 8 + (lane >> 3), c = (lane & 7) * 4;
        if (row0 + r < M)
            *reinterpret_cast<float4 *>(dst + (size_t)(row0 + r) * ld + c) = *reinterpret_cast<const float4 *>(tile + r * HS + c);
    }
}

template <bool FIRST, bool LAST>
__global__ __launch_bounds__(256, 2) void trunk_block_fwd_kernel(TrunkFwdArgs a) {
    __shared__ __attribute__((aligned(16))) float Xs[TR * XS];
    __shared__ __attribute__((aligned(16))) float Hsm[TR * HS];
    __shared__ __attribute__((aligned(16))) float W0s[32 * XS];
    __shared__ __attribute__((aligned(16))) float Wss[32 * XS];
    __shared__ __attribute__((aligned(16))) float W1s[32 * HS];
    __shared__ __attribute__((aligned(16))) float Wcs[LAST ? 32 * HS : 4];
    __shared__ float bsm[96];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r0 = blockIdx.x * TR, r1 = min(r0 + TR, a.M);

    // ---- weights -> LDS (row-major [n][k], padded rows)
#pragma unroll
    for (int f = 0; f < 2; ++f) {
        const int idx = tid + f * 256, n = idx >> 4, k4 = (idx & 15) * 4;
        *reinterpret_cast<float4 *>(W0s + n * XS + k4) = *reinterpret_cast<const float4 *>(a.w0 + n * 64 + k4);
        *reinterpret_cast<float4 *>(Wss + n * XS + k4) = *reinterpret_cast<const float4 *>(a.ws + n * 64 + k4);
    }
    {
        const int n = tid >> 3, k4 = (tid & 7) * 4;
        *reinterpret_cast<float4 *>(W1s + n * HS + k4) = *reinterpret_cast<const float4 *>(a.w1 + n * 32 + k4);
        if (LAST) *reinterpret_cast<float4 *>(Wcs + n * HS + k4) = *reinterpret_cast<const float4 *>(a.wc + n * 32 + k4);
    }
    if (tid < 32) { bsm[tid] = a.b0[tid]; bsm[32 + tid] = a.b1[tid]; if (LAST) bsm[64 + tid] = a.bc[tid]; }

    // ---- the block input X -> Xs
    if (FIRST) {
        float *wp = Hsm;                     // [64][3] + [64] staged through the (still unused) hr tile
        if (tid < 192) wp[tid] = a.wpos[tid];
        if (tid < 64) wp[192 + tid] = a.bpos[tid];
        __syncthreads();
        const int row = tid >> 1, c0 = (tid & 1) * 32;
        float p0 = 0.f, p1 = 0.f, p2 = 0.f;
        if (r0 + row < r1) {
            const float *p = a.pts + (size_t)(r0 + row) * a.dim;
            p0 = p[0]; p1 = p[1]; p2 = p[2];
        }
#pragma unroll
        for (int c = 0; c < 32; c += 4) {
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = c0 + c + j;
                float acc = wp[192 + n];                                   // bias first, then k = 0, 1, 2 (linear_smallk_fwd)
                acc = fmaf(p0, wp[n * 3 + 0], acc);
                acc = fmaf(p1, wp[n * 3 + 1], acc);
                acc = fmaf(p2, wp[n * 3 + 2], acc);
                v[j] = r0 + row < r1 ? acc : 0.f;
            }
            *reinterpret_cast<float4 *>(Xs + row * XS + c0 + c) = make_float4(v[0], v[1], v[2], v[3]);
        }
        __syncthreads();
    } else {
        pool_into_tile(a, Xs, Hsm, r0, r1, tid);
    }
    if (a.x_full) {
#pragma unroll
        for (int f = 0; f < 8; ++f) {
            const int idx = tid + f * 256, row = idx >> 4, c = (idx & 15) * 4;
            if (r0 + row < r1)
                *reinterpret_cast<float4 *>(a.x_full + (size_t)(r0 + row) * 64 + c) = *reinterpret_cast<const float4 *>(Xs + row * XS + c);
        }
    }

    // ---- GEMMs: this wave's 32 rows
    const int r = lane & 31, h = lane >> 5;
    const float *xa = Xs + (wave * 32 + r) * XS + 4 * h;
    f32x16 acc_h, acc_s, acc_d;
#pragma unroll
    for (int q = 0; q < 16; ++q) { acc_h[q] = 0.f; acc_s[q] = 0.f; acc_d[q] = 0.f; }
    mfma_rows_pair<64>(xa, W0s + r * XS + 4 * h, Wss + r * XS + 4 * h, acc_h, acc_s);
    float *ht = Hsm + wave * 32 * HS;                                       // this wave's hr tile
    {
        const float b0 = bsm[r];
#pragma unroll
        for (int q = 0; q < 16; ++q) ht[acc_row(q, lane) * HS + r] = fmaxf(acc_h[q] + b0, 0.f);   // relu(fc_0(relu(x)))
    }
    __syncthreads();
    store_tile_rows(ht, a.hr, 32, r0 + wave * 32, a.M, lane);
    mfma_rows<32, false>(ht + r * HS + 4 * h, W1s + r * HS + 4 * h, acc_d);
    float *ot = Xs + wave * 32 * XS;                                        // X rows of this wave are consumed: reuse for out
    {
        const float b1 = bsm[32 + r];
#pragma unroll
        for (int q = 0; q < 16; ++q) ot[acc_row(q, lane) * HS + r] = acc_s[q] + (acc_d[q] + b1);   // x_s + dx (resnet.py:54)
    }
    __syncthreads();
    store_tile_rows(ot, a.out, a.ld_out, r0 + wave * 32, a.M, lane);
    if (LAST) {
        f32x16 acc_c;
#pragma unroll
        for (int q = 0; q < 16; ++q) acc_c[q] = 0.f;
        mfma_rows<32, true>(ot + r * HS + 4 * h, Wcs + r * HS + 4 * h, acc_c);     // fc_c(relu(net))
        __syncthreads();                                                            // hr tile stores have been issued from ht
        const float bc = bsm[64 + r];
#pragma unroll
        for (int q = 0; q < 16; ++q) ht[acc_row(q, lane) * HS + r] = acc_c[q] + bc;
        __syncthreads();
        store_tile_rows(ht, a.c_out, 32, r0 + wave * 32, a.M, lane);
    }
}

}  // namespace
}  // namespace t2h

using namespace t2h;

static bool al16(const void *p) { return ((uintptr_t)p & 15) == 0; }

T2H_API int t2h_trunk_block_fwd(const float *pts, int dim, const float *w_pos, const float *b_pos, const float *net_prev,
                                int ld_prev, const int32_t *cell, const int32_t *off0, const float *w0, const float *b0,
                                const float *w1, const float *b1, const float *ws, const float *wc, const float *bc, int64_t M,
                                float *x_full, float *hr, float *out, int ld_out, uint8_t *winner, float *c_out,
                                t2h_stream_t stream) {
    const bool first = pts != nullptr;
    const bool last = wc != nullptr;
    if (!w0 || !b0 || !w1 || !b1 || !ws || !hr || !out) return fail(T2H_ERR_ARG, "trunk_block_fwd: null pointer");
    if (first ? (!w_pos || !b_pos || dim < 3) : (!net_prev || !cell || !off0 || !winner || ld_prev < 32 || ld_prev % 4))
        return fail(T2H_ERR_ARG, "trunk_block_fwd: bad input description");
    if (first && last) return fail(T2H_ERR_ARG, "trunk_block_fwd: a block is either the first or the last");
    if (last && (!bc || !c_out)) return fail(T2H_ERR_ARG, "trunk_block_fwd: last block needs bc and c_out");
    if (M < 0 || M >= ((int64_t)1 << 31) - TR || ld_out < 32 || ld_out % 4) return fail(T2H_ERR_ARG, "trunk_block_fwd: bad shape");
    if (!al16(w0) || !al16(w1) || !al16(ws) || !al16(hr) || !al16(out) || (x_full && !al16(x_full)) || (net_prev && !al16(net_prev)) ||
        (wc && !al16(wc)) || (c_out && !al16(c_out)))
        return fail(T2H_ERR_ARG, "trunk_block_fwd: pointers must be 16-byte aligned");
    if (M == 0) return T2H_OK;
    TrunkFwdArgs a{};
    a.pts = pts; a.dim = dim; a.wpos = w_pos; a.bpos = b_pos;
    a.net_prev = net_prev; a.ld_prev = ld_prev; a.cell = cell; a.off0 = off0;
    a.w0 = w0; a.b0 = b0; a.w1 = w1; a.b1 = b1; a.ws = ws; a.wc = wc; a.bc = bc;
    a.M = (int)M; a.x_full = x_full; a.hr = hr; a.out = out; a.ld_out = ld_out; a.winner = winner; a.c_out = c_out;
    const dim3 grid((unsigned)((M + TR - 1) / TR));
    hipStream_t s = as_stream(stream);
    if (first) hipLaunchKernelGGL((trunk_block_fwd_kernel<true, false>), grid, dim3(256), 0, s, a);
    else if (last) hipLaunchKernelGGL((trunk_block_fwd_kernel<false, true>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((trunk_block_fwd_kernel<false, false>), grid, dim3(256), 0, s, a);
    note_kernel(first ? "trunk_block_fwd_kernel<true,false>" : (last ? "trunk_block_fwd_kernel<false,true>" : "trunk_block_fwd_kernel<false,false>"));
    return check_launch("trunk_block_fwd");
}
