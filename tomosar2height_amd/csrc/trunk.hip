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
#include <stdlib.h>

#include "t2h_common.h"

namespace t2h {
namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

#ifndef T2H_LAB_TRUNK_PRIO
#define T2H_LAB_TRUNK_PRIO 0     // lab builds: 1 = s_setprio 1 around the one-launch forward's MFMA clusters, 2 = around its pooling
#endif
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
    float *x_full;       // optional [M, 64]: the block input materialised (tests)
    float *pooled;       // optional [M, 32]: the pooled half of the input (what the backward reads), later blocks
    float *hr;           // [M, 32]
    float *out; int ld_out;
    uint8_t *winner;     // [M, 8]: arg-max bits of net_prev's pooling (later blocks)
    float *c_out;        // [M, 32] (last block)
    int loader;          // 1 = r02 loader (per-cell head loop), else the r03 ballot / per-row form (T2H_TRUNK_LOADER, A/B)
};

__device__ inline float clean(float x) { return x != x ? -FLT_MAX : x; }

// `a` where the integer n is positive, else +0 -- without a compare result (a kernel that can share a CU with the split convolutions
// keeps none for a later vector select: DESIGN.md section 8).  The clamp is an opaque v_med3_i32: written as min(max(n, 0), 1) the
// compiler recognises the idiom and emits the very v_cmp -> v_cndmask it was meant to avoid.
__device__ inline float keep_if_positive(int n, float a) {
    int m;
    asm("v_med3_i32 %0, %1, 0, 1" : "=v"(m) : "v"(n));
    return __int_as_float(__float_as_int(a) & -m);
}

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
            const int used = min(NG, hi - lo);                            // groups past the run's length hold empty partials
            for (int g = 1; g < used; ++g) {
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
        int n = row;
        for (; n + 3 < end; n += 4) {                                     // four LDS rows in flight
            const float *xp = Xs + (n - r0) * XS + lane * 4;
            const float4 v0 = *reinterpret_cast<const float4 *>(xp), v1 = *reinterpret_cast<const float4 *>(xp + XS);
            const float4 v2 = *reinterpret_cast<const float4 *>(xp + 2 * XS), v3 = *reinterpret_cast<const float4 *>(xp + 3 * XS);
            best_strict(b, v0, n - r0); best_strict(b, v1, n + 1 - r0); best_strict(b, v2, n + 2 - r0); best_strict(b, v3, n + 3 - r0);
        }
        for (; n < end; ++n)
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

// r03 loader: the same segmented max without the per-row `cell -> off0` hops, without the serial per-cell head loop and without
// the broadcast phase.  The cell ids of the tile's rows go to LDS; a wave ballot over "this row starts a cell" gives two 64-bit
// head masks, from which every row derives its cell's in-tile range with two bit scans (clz / ctz); then EVERY row reduces its
// own cell over the LDS rows in ascending order -- the same strict '>' scan the head row ran, so all rows of a cell arrive at the
// same maximum and arg-max, and each writes its pooled half and winner bits itself.  A cell of L rows costs L^2 LDS row reads
// instead of L, but they run in parallel over the rows (the head loop serialised the workgroup on its longest cell), and
// typical cells hold 2-3 rows.  Only the two border cells still need `off0` (their rows outside the tile, streamed as before).
__device__ inline int mask_prev_head(unsigned long long m0, unsigned long long m1, int me) {     // last head <= me
    if (me < 64) return 63 - __clzll(m0 & (me == 63 ? ~0ull : ((2ull << me) - 1ull)));
    const unsigned long long b1 = m1 & (me == 127 ? ~0ull : ((2ull << (me - 64)) - 1ull));
    return b1 ? 127 - __clzll(b1) : 63 - __clzll(m0);
}
__device__ inline int mask_next_head(unsigned long long m0, unsigned long long m1, int me) {     // first head > me, or TR
    if (me < 64) {
        const unsigned long long rest = me == 63 ? 0ull : (m0 >> (me + 1));
        if (rest) return me + 1 + __ffsll((long long)rest) - 1;
        return m1 ? 64 + __ffsll((long long)m1) - 1 : TR;
    }
    const unsigned long long rest = me == 127 ? 0ull : (m1 >> (me - 64 + 1));
    return rest ? me + 1 + __ffsll((long long)rest) - 1 : TR;
}

__device__ inline void pool_into_tile_v2(const TrunkFwdArgs &a, float *Xs, float *scratch, int r0, int r1, int tid) {
    float4 *pval = reinterpret_cast<float4 *>(scratch);                   // [NG * G]
    int4 *parg = reinterpret_cast<int4 *>(scratch + 4 * NG * G);          // [NG * G]
    float4 *oval = reinterpret_cast<float4 *>(scratch + 8 * NG * G);      // [2 * G]
    int4 *oarg = reinterpret_cast<int4 *>(scratch + 8 * NG * G + 8 * G);  // [2 * G]
    int *bounds = reinterpret_cast<int *>(scratch + 8 * NG * G + 16 * G); // [4]
    int *cells = bounds + 4;                                              // [TR]
    unsigned long long *masks = reinterpret_cast<unsigned long long *>(cells + TR);     // [2] (8-byte aligned: 2308 floats in)
    const int lane = tid & (G - 1), grp = tid >> 3;
#pragma unroll
    for (int p = 0; p < TR / NG; ++p) {
        const int me = p * NG + grp, row = r0 + me;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        int cid = -1 - me;                                                // rows past the end: one-row cells of their own
        if (row < r1) {
            cid = a.cell[row];
            v = *reinterpret_cast<const float4 *>(a.net_prev + (size_t)row * a.ld_prev + lane * 4);
        }
        *reinterpret_cast<float4 *>(Xs + me * XS + lane * 4) = v;
        if (lane == 0) cells[me] = cid;
    }
    if (tid == 0) {
        const int ch = a.cell[r0], ct = a.cell[r1 - 1];
        bounds[0] = a.off0[ch]; bounds[1] = a.off0[ch + 1]; bounds[2] = a.off0[ct]; bounds[3] = a.off0[ct + 1];
    }
    __syncthreads();
    if (tid < TR) {                                                       // waves 0 and 1, all lanes: row = tid
        const bool head = tid == 0 || cells[tid] != cells[tid - 1];
        const unsigned long long m = __ballot(head);
        if ((tid & 63) == 0) masks[tid >> 6] = m;
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
            const int used = min(NG, hi - lo);                            // groups past the run's length hold empty partials
            for (int g = 1; g < used; ++g) {
                const float4 v = pval[g * G + lane]; const int4 ar = parg[g * G + lane];
                best_merge(t.v.x, t.a.x, v.x, ar.x); best_merge(t.v.y, t.a.y, v.y, ar.y);
                best_merge(t.v.z, t.a.z, v.z, ar.z); best_merge(t.v.w, t.a.w, v.w, ar.w);
            }
            oval[side * G + lane] = t.v; oarg[side * G + lane] = t.a;
        }
        __syncthreads();
    }
    const unsigned long long m0 = masks[0], m1 = masks[1];
    const int rows = r1 - r0;
    const bool before = bounds[0] < r0, after = bounds[3] > r1;
#pragma unroll
    for (int p = 0; p < TR / NG; ++p) {
        const int me = p * NG + grp, row = r0 + me;
        if (row >= r1) {
            *reinterpret_cast<float4 *>(Xs + me * XS + 32 + lane * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
            continue;
        }
        const int s = mask_prev_head(m0, m1, me), e = mask_next_head(m0, m1, me);       // the cell's rows inside the tile
        Best b; best_init(b);
        if (s == 0 && before) {                                           // earlier rows win ties
            const int4 ar = oarg[lane];
            b.v = oval[lane];
            b.a = make_int4(ar.x < 0 ? -1 : -2, ar.y < 0 ? -1 : -2, ar.z < 0 ? -1 : -2, ar.w < 0 ? -1 : -2);
        }
        int n = s;
        for (; n + 3 < e; n += 4) {                                       // four LDS rows in flight
            const float *xp = Xs + n * XS + lane * 4;
            const float4 v0 = *reinterpret_cast<const float4 *>(xp), v1 = *reinterpret_cast<const float4 *>(xp + XS);
            const float4 v2 = *reinterpret_cast<const float4 *>(xp + 2 * XS), v3 = *reinterpret_cast<const float4 *>(xp + 3 * XS);
            best_strict(b, v0, n); best_strict(b, v1, n + 1); best_strict(b, v2, n + 2); best_strict(b, v3, n + 3);
        }
        for (; n < e; ++n) best_strict(b, *reinterpret_cast<const float4 *>(Xs + n * XS + lane * 4), n);
        if (e == rows && after) {                                         // later rows: only a larger value wins
            const float4 v = oval[G + lane];
            if (v.x > b.v.x) { b.v.x = v.x; b.a.x = -2; }
            if (v.y > b.v.y) { b.v.y = v.y; b.a.y = -2; }
            if (v.z > b.v.z) { b.v.z = v.z; b.a.z = -2; }
            if (v.w > b.v.w) { b.v.w = v.w; b.a.w = -2; }
        }
        // untouched (all NaN) -> 0, as torch_scatter's fill of cells that no value entered.  Only the left halves of the tile
        // are read above and only right halves written here, so no barrier separates the two
        *reinterpret_cast<float4 *>(Xs + me * XS + 32 + lane * 4) =
            make_float4(b.a.x == -1 ? 0.f : b.v.x, b.a.y == -1 ? 0.f : b.v.y, b.a.z == -1 ? 0.f : b.v.z, b.a.w == -1 ? 0.f : b.v.w);
        a.winner[(size_t)row * G + lane] = (uint8_t)((b.a.x == me) | ((b.a.y == me) << 1) | ((b.a.z == me) << 2) | ((b.a.w == me) << 3));
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

// Orders this wave's LDS writes before its later LDS reads (the LDS executes one wave's accesses in order; the fences keep
// the compiler from moving them) -- for data that only this wave touches, instead of a workgroup barrier.
__device__ inline void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

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
    __shared__ float wps[FIRST ? 256 : 4];                                  // fc_pos: [64][3] weights + [64] bias
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

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

    if (FIRST) {
        if (tid < 192) wps[tid] = a.wpos[tid];
        if (tid < 64) wps[192 + tid] = a.bpos[tid];
    }
    __syncthreads();
    // persistent over tiles (the weights above are staged once; co-resident workgroups drift out of phase, so one's
    // loader runs under the other's MFMAs)
    const int n_tiles = (a.M + TR - 1) / TR;
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int r0 = tile * TR, r1 = min(r0 + TR, a.M);
    // ---- the block input X -> Xs
    if (FIRST) {
        const float *wp = wps;
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
        if (a.loader == 1) pool_into_tile(a, Xs, Hsm, r0, r1, tid);
        else pool_into_tile_v2(a, Xs, Hsm, r0, r1, tid);
    }
    // copy-outs: every wave copies ITS OWN 32 rows (the rows its GEMMs read and, in the LAST variant, later overwrite with
    // relu(out)), so no other wave ever reads rows that their owner may already be rewriting
    if (!FIRST && a.pooled) {
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            const int idx = lane + f * 64, row = wave * 32 + (idx >> 3), c = (idx & 7) * 4;
            if (r0 + row < r1)
                *reinterpret_cast<float4 *>(a.pooled + (size_t)(r0 + row) * 32 + c) = *reinterpret_cast<const float4 *>(Xs + row * XS + 32 + c);
        }
    }
    if (a.x_full) {
#pragma unroll
        for (int f = 0; f < 8; ++f) {
            const int idx = lane + f * 64, row = wave * 32 + (idx >> 4), c = (idx & 15) * 4;
            if (r0 + row < r1)
                *reinterpret_cast<float4 *>(a.x_full + (size_t)(r0 + row) * 64 + c) = *reinterpret_cast<const float4 *>(Xs + row * XS + c);
        }
    }

    // ---- GEMMs: this wave's 32 rows.  From here to the end of the tile a wave touches only its own rows of Xs / Hsm (and
    // the read-only weights), so wave-level ordering suffices -- no workgroup barrier until the next tile's loader -- and
    // results leave straight from the MFMA C/D registers (each store instruction writes two 128-byte row segments).
    const int r = lane & 31, h = lane >> 5;
    const float *xa = Xs + (wave * 32 + r) * XS + 4 * h;
    const int rb = r0 + wave * 32 + 4 * h;                                  // global row of C/D register 0
    f32x16 acc_h, acc_s, acc_d;
#pragma unroll
    for (int q = 0; q < 16; ++q) { acc_h[q] = 0.f; acc_s[q] = 0.f; acc_d[q] = 0.f; }
    mfma_rows_pair<64>(xa, W0s + r * XS + 4 * h, Wss + r * XS + 4 * h, acc_h, acc_s);
    float *ht = Hsm + wave * 32 * HS;                                       // this wave's hr tile (A operand of fc_1)
    {
        const float b0 = bsm[r];
        float *hp = a.hr + (size_t)rb * 32 + r;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int ro = (q & 3) + 8 * (q >> 2);
            const float v = fmaxf(acc_h[q] + b0, 0.f);                      // relu(fc_0(relu(x)))
            ht[(ro + 4 * h) * HS + r] = v;
            if (rb + ro < a.M) hp[ro * 32] = v;
        }
    }
    wave_sync();
    mfma_rows<32, false>(ht + r * HS + 4 * h, W1s + r * HS + 4 * h, acc_d);
    {
        const float b1 = bsm[32 + r];
        float *op = a.out + (size_t)rb * a.ld_out + r;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int ro = (q & 3) + 8 * (q >> 2);
            acc_s[q] = acc_s[q] + (acc_d[q] + b1);                          // x_s + dx (resnet.py:54)
            if (rb + ro < a.M) op[(size_t)ro * a.ld_out] = acc_s[q];
        }
    }
    if (LAST) {
        float *ot = Xs + wave * 32 * XS;                                    // X rows of this wave are consumed: relu(out) as A operand
#pragma unroll
        for (int q = 0; q < 16; ++q) ot[acc_row(q, lane) * HS + r] = acc_s[q];
        wave_sync();
        f32x16 acc_c;
#pragma unroll
        for (int q = 0; q < 16; ++q) acc_c[q] = 0.f;
        mfma_rows<32, true>(ot + r * HS + 4 * h, Wcs + r * HS + 4 * h, acc_c);     // fc_c(relu(net))
        const float bc = bsm[64 + r];
        float *cp = a.c_out + (size_t)rb * 32 + r;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int ro = (q & 3) + 8 * (q >> 2);
            if (rb + ro < a.M) cp[ro * 32] = acc_c[q] + bc;
        }
    }
    __syncthreads();                     // the tiles in LDS are free for the next tile
    }
}


// =====================================================================================================================
// r06: the WHOLE trunk forward in one launch (pointnet.py:72-82): fc_pos -> block 0 -> n x {pool_local, block} -> fc_c.
//
// All poolings of the trunk use the SAME finest-level cells (pointnet.py:70, 76-79), and after the tile sort a cell is a run of
// consecutive rows.  A workgroup that owns WHOLE cells therefore needs nothing from any other workgroup between the blocks: work
// unit k = the cells whose first row lies in [k S, (k + 1) S) (S = `stride` rows, 96 by default), found from `cell` / `off0`
// with two dependent loads per boundary.  A unit of at most TR = 128 rows (every unit unless a cell with more than TR - S + 1
// rows starts near its end) takes the FAST path: X = fc_pos(points) is formed in LDS, every block's output goes back into the
// left half of the X tile from the MFMA registers, the segmented max runs on the tile in LDS (head masks of the unit's cells by
// one ballot, kept in registers for all blocks; each row scans its own cell: the strict '>' scan of pool_into_tile_v2, no border
// cells), and only what the backward reads leaves the chip: hr, out, the pooled half and the winner bits of every block, c.
// Between two blocks: one barrier before the pooling (all waves have written their rows of `out`), one after it (which also covers
// the next block's weights: they are fetched from L2 into LDS beside the pooling).  A longer unit takes the SLOW path inside the same
// launch: block by block over 128-row chunks with the one-block loader (pool_into_tile: its border runs are rows of this unit,
// written by this workgroup and published by a device-scope fence + barrier between the blocks) -- correct for any cell size,
// one workgroup per heavy unit.  Arithmetic and summation order per element are those of trunk_block_fwd_kernel: the two forms
// agree bit for bit (tests/test_hip_trunk.py).
constexpr int kMaxTrunkBlocks = 8;
struct TrunkFusedArgs {
    const float *pts; int dim; const float *wpos, *bpos;
    const float *w0[kMaxTrunkBlocks], *b0[kMaxTrunkBlocks], *w1[kMaxTrunkBlocks], *b1[kMaxTrunkBlocks], *ws[kMaxTrunkBlocks];
    const float *wc, *bc;
    const int32_t *cell, *off0;
    const int2 *bounds;           // (first row, end row) of every unit (t2h_trunk_units_build), or NULL: fixed-stride units looked up per unit
    const int *n_bounds;          // their number (device word written by t2h_trunk_units_build)
    int M, nb, stride, n_units;
    int ablate;          // lab builds (-DT2H_TRUNK_ABLATE) only: bit 0 no pooling, 1 no global stores, 2 weights staged once, 3 no MFMAs
    float *hr[kMaxTrunkBlocks], *out[kMaxTrunkBlocks], *pooled[kMaxTrunkBlocks];
    uint8_t *winner[kMaxTrunkBlocks];
    float *c_out;
};

#ifdef T2H_TRUNK_ABLATE
#define T2H_ABL(a, bit) (((a).ablate >> (bit)) & 1)
#else
#define T2H_ABL(a, bit) 0
#endif

__device__ inline int unit_start(const TrunkFusedArgs &a, int k) {
    const long long row = (long long)k * a.stride;
    if (row >= a.M) return a.M;
    const int c = a.cell[row], st = a.off0[c];
    return st == (int)row ? (int)row : a.off0[c + 1];
}

// block weights -> LDS (row-major [n][k], padded rows); the caller separates this from the GEMMs that read them by a barrier
__device__ inline void stage_block_weights(const TrunkFusedArgs &a, int b, float *W0s, float *Wss, float *W1s, float *Wcs, float *bsm,
                                           int tid) {
#pragma unroll
    for (int f = 0; f < 2; ++f) {
        const int idx = tid + f * 256, n = idx >> 4, k4 = (idx & 15) * 4;
        *reinterpret_cast<float4 *>(W0s + n * XS + k4) = *reinterpret_cast<const float4 *>(a.w0[b] + n * 64 + k4);
        *reinterpret_cast<float4 *>(Wss + n * XS + k4) = *reinterpret_cast<const float4 *>(a.ws[b] + n * 64 + k4);
    }
    const int n = tid >> 3, k4 = (tid & 7) * 4;
    *reinterpret_cast<float4 *>(W1s + n * HS + k4) = *reinterpret_cast<const float4 *>(a.w1[b] + n * 32 + k4);
    if (b == a.nb - 1) *reinterpret_cast<float4 *>(Wcs + n * HS + k4) = *reinterpret_cast<const float4 *>(a.wc + n * 32 + k4);
    if (tid < 32) { bsm[tid] = a.b0[b][tid]; bsm[32 + tid] = a.b1[b][tid]; if (b == a.nb - 1) bsm[64 + tid] = a.bc[tid]; }
}

// the GEMMs of one block on this wave's 32 rows of the X tile (the code of trunk_block_fwd_kernel, same order of operations);
// rows [row0, row_end) of the tile are real.  keep: also write `out` into the left half of this wave's X rows (next block's input)
__device__ inline void fused_block_gemms(float *Xs, float *Hsm, const float *W0s, const float *Wss, const float *W1s, const float *Wcs,
                                         const float *bsm, int row0, int row_end, float *hr, float *out, float *c_out, bool last,
                                         bool keep, int lane, int wave, int abl = 0) {
    const bool no_store = (abl >> 1) & 1, no_mfma = (abl >> 3) & 1;          // (lab builds only: always 0 otherwise)
    const int r = lane & 31, h = lane >> 5;
    const float *xa = Xs + (wave * 32 + r) * XS + 4 * h;
    const int rb = row0 + wave * 32 + 4 * h;
    f32x16 acc_h, acc_s, acc_d;
#pragma unroll
    for (int q = 0; q < 16; ++q) { acc_h[q] = 0.f; acc_s[q] = 0.f; acc_d[q] = 0.f; }
#if T2H_LAB_TRUNK_PRIO == 1
    __builtin_amdgcn_s_setprio(1);
#endif
    if (!no_mfma) mfma_rows_pair<64>(xa, W0s + r * XS + 4 * h, Wss + r * XS + 4 * h, acc_h, acc_s);
#if T2H_LAB_TRUNK_PRIO == 1
    __builtin_amdgcn_s_setprio(0);
#endif
    float *ht = Hsm + wave * 32 * HS;
    {
        const float b0 = bsm[r];
        float *hp = hr + (size_t)rb * 32 + r;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int ro = (q & 3) + 8 * (q >> 2);
            const float v = fmaxf(acc_h[q] + b0, 0.f);
            ht[(ro + 4 * h) * HS + r] = v;
            if (rb + ro < row_end && !no_store) hp[ro * 32] = v;
        }
    }
    wave_sync();
#if T2H_LAB_TRUNK_PRIO == 1
    __builtin_amdgcn_s_setprio(1);
#endif
    if (!no_mfma) mfma_rows<32, false>(ht + r * HS + 4 * h, W1s + r * HS + 4 * h, acc_d);
#if T2H_LAB_TRUNK_PRIO == 1
    __builtin_amdgcn_s_setprio(0);
#endif
    {
        const float b1 = bsm[32 + r];
        float *op = out + (size_t)rb * 32 + r;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int ro = (q & 3) + 8 * (q >> 2);
            acc_s[q] = acc_s[q] + (acc_d[q] + b1);
            if (rb + ro < row_end && !no_store) op[(size_t)ro * 32] = acc_s[q];
        }
    }
    if (last) {
        float *ot = Xs + wave * 32 * XS;
#pragma unroll
        for (int q = 0; q < 16; ++q) ot[acc_row(q, lane) * HS + r] = acc_s[q];
        wave_sync();
        f32x16 acc_c;
#pragma unroll
        for (int q = 0; q < 16; ++q) acc_c[q] = 0.f;
        mfma_rows<32, true>(ot + r * HS + 4 * h, Wcs + r * HS + 4 * h, acc_c);
        const float bc = bsm[64 + r];
        float *cp = c_out + (size_t)rb * 32 + r;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int ro = (q & 3) + 8 * (q >> 2);
            if (rb + ro < row_end) cp[ro * 32] = acc_c[q] + bc;
        }
    } else if (keep) {
        // this wave's rows of X are consumed (only this wave reads them in the GEMM phase): the block's output becomes the left
        // half of the next block's input; rows past the unit hold zeros like the loader's
        float *xt = Xs + (wave * 32) * XS;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int ro = (q & 3) + 8 * (q >> 2);
            xt[(ro + 4 * h) * XS + r] = keep_if_positive(row_end - (rb + ro), acc_s[q]);     // (r06: was a select on sixteen parked compares)
        }
    }
}

// fc_pos of the unit's (or chunk's) points -> the X tile (the loader of the FIRST variant)
__device__ inline void fused_fc_pos(const TrunkFusedArgs &a, float *Xs, const float *wps, int r0, int r1, int tid) {
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
            float acc = wps[192 + n];
            acc = fmaf(p0, wps[n * 3 + 0], acc);
            acc = fmaf(p1, wps[n * 3 + 1], acc);
            acc = fmaf(p2, wps[n * 3 + 2], acc);
            v[j] = keep_if_positive(r1 - (r0 + row), acc);
        }
        *reinterpret_cast<float4 *>(Xs + row * XS + c0 + c) = make_float4(v[0], v[1], v[2], v[3]);
    }
}

// segmented max of the unit's rows (left half of the X tile, whole cells only) -> right half, winner bits and pooled half to
// global.  span[p]: first and one-past-last tile row of the cell of this thread's row p (cs | ce << 8), found once per unit from
// the head masks (the four poolings of a unit share them).
// Each row scans its own cell in ascending order (strict '>': the first maximum wins) -- the L^2 form of pool_into_tile_v2.  (Tried
// in r06: a long cell's rows SHARING its scan -- eight residue classes, partials through LDS, merged by every row after a barrier
// -- costs two barriers per row slot and measured slower, 711 against 633 us per four-tile launch; profiles/r06_trunk_fused.txt.)
__device__ inline void fused_pool_local(float *Xs, const int (&span)[TR / NG], int s, int rows, uint8_t *winner, float *pooled,
                                        int tid, int abl = 0) {
    const int lane = tid & (G - 1), grp = tid >> 3;
#pragma unroll
    for (int p = 0; p < TR / NG; ++p) {
        const int me = p * NG + grp;
        if (me >= rows) {
            *reinterpret_cast<float4 *>(Xs + me * XS + 32 + lane * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
            continue;
        }
        const int cs = span[p] & 255, ce = span[p] >> 8;
        Best b; best_init(b);
        int n = cs;
        for (; n + 3 < ce; n += 4) {
            const float *xp = Xs + n * XS + lane * 4;
            const float4 v0 = *reinterpret_cast<const float4 *>(xp), v1 = *reinterpret_cast<const float4 *>(xp + XS);
            const float4 v2 = *reinterpret_cast<const float4 *>(xp + 2 * XS), v3 = *reinterpret_cast<const float4 *>(xp + 3 * XS);
            best_strict(b, v0, n); best_strict(b, v1, n + 1); best_strict(b, v2, n + 2); best_strict(b, v3, n + 3);
        }
        for (; n < ce; ++n) best_strict(b, *reinterpret_cast<const float4 *>(Xs + n * XS + lane * 4), n);
        // untouched (all NaN) -> 0, as torch_scatter's fill of cells that no value entered.  Only the left halves of the tile are
        // read above and only right halves written here, so no barrier separates the two
        const float4 pv = make_float4(b.a.x == -1 ? 0.f : b.v.x, b.a.y == -1 ? 0.f : b.v.y, b.a.z == -1 ? 0.f : b.v.z,
                                      b.a.w == -1 ? 0.f : b.v.w);
        *reinterpret_cast<float4 *>(Xs + me * XS + 32 + lane * 4) = pv;
        if ((abl >> 1) & 1) continue;
        *reinterpret_cast<float4 *>(pooled + (size_t)(s + me) * 32 + lane * 4) = pv;
        winner[(size_t)(s + me) * G + lane] = (uint8_t)((b.a.x == me) | ((b.a.y == me) << 1) | ((b.a.z == me) << 2) | ((b.a.w == me) << 3));
    }
}

// The same pooling with ONE scan per cell (r06, last day): phase A -- a lane group per CELL scans its rows once (the row-parallel form
// above reads a cell of L rows L times: fine for the 2-3-row cells of a sparse tile, 174 of 633 us when units pack 30-60-row
// cells) and leaves the pooled row in the right half of the cell's FIRST row and the four winners as bytes in `args`; barrier;
// phase B -- a lane group per ROW copies its cell's result into its own right half, writes the global pooled row and its winner
// bits.  Same visiting order per cell (ascending rows, strict >), so values and winners are those of the form above bit for bit.
// cstart[c] = first row of cell c inside the unit, cstart[ncell] = rows.  Phase B writes only non-first rows' right halves and
// reads only first rows', so it needs no barrier inside.
__device__ inline void fused_pool_cells(float *Xs, unsigned *args, const uint8_t *cstart, int ncell, const int (&span)[TR / NG], int s,
                                        int rows, uint8_t *winner, float *pooled, int tid, int abl = 0) {
    const int lane = tid & (G - 1), grp = tid >> 3;
    for (int c = grp; c < ncell; c += NG) {
        const int cs = cstart[c], ce = cstart[c + 1];
        Best b; best_init(b);
        int n = cs;
        for (; n + 3 < ce; n += 4) {
            const float *xp = Xs + n * XS + lane * 4;
            const float4 v0 = *reinterpret_cast<const float4 *>(xp), v1 = *reinterpret_cast<const float4 *>(xp + XS);
            const float4 v2 = *reinterpret_cast<const float4 *>(xp + 2 * XS), v3 = *reinterpret_cast<const float4 *>(xp + 3 * XS);
            best_strict(b, v0, n); best_strict(b, v1, n + 1); best_strict(b, v2, n + 2); best_strict(b, v3, n + 3);
        }
        for (; n < ce; ++n) best_strict(b, *reinterpret_cast<const float4 *>(Xs + n * XS + lane * 4), n);
        // untouched (all NaN) -> 0, as torch_scatter's fill of cells that no value entered
        *reinterpret_cast<float4 *>(Xs + cs * XS + 32 + lane * 4) =
            make_float4(b.a.x == -1 ? 0.f : b.v.x, b.a.y == -1 ? 0.f : b.v.y, b.a.z == -1 ? 0.f : b.v.z, b.a.w == -1 ? 0.f : b.v.w);
        args[cs * G + lane] = (unsigned)(b.a.x & 255) | ((unsigned)(b.a.y & 255) << 8) | ((unsigned)(b.a.z & 255) << 16) |
                              ((unsigned)(b.a.w & 255) << 24);              // (-1 -> 255: no row of a 128-row tile)
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < TR / NG; ++p) {
        const int me = p * NG + grp;
        if (me >= rows) {
            *reinterpret_cast<float4 *>(Xs + me * XS + 32 + lane * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
            continue;
        }
        const int cs = span[p] & 255;
        const float4 pv = *reinterpret_cast<const float4 *>(Xs + cs * XS + 32 + lane * 4);
        const unsigned pk = args[cs * G + lane];
        if (me != cs) *reinterpret_cast<float4 *>(Xs + me * XS + 32 + lane * 4) = pv;
        if ((abl >> 1) & 1) continue;
        *reinterpret_cast<float4 *>(pooled + (size_t)(s + me) * 32 + lane * 4) = pv;
        winner[(size_t)(s + me) * G + lane] = (uint8_t)(((pk & 255u) == (unsigned)me) | ((((pk >> 8) & 255u) == (unsigned)me) << 1) |
                                                        ((((pk >> 16) & 255u) == (unsigned)me) << 2) | (((pk >> 24) == (unsigned)me) << 3));
    }
}

// a unit longer than a tile: chunks of TR rows, block by block, through memory (the one-block loader's border runs are rows of this
// unit, written by this workgroup and published by a device-scope fence + barrier between the blocks).  Expects block 0's weights staged; leaves the last block's.
__device__ inline void fused_slow_unit(const TrunkFusedArgs &a, float *Xs, float *Hsm, float *W0s, float *Wss, float *W1s,
                                             float *Wcs, float *bsm, const float *wps, int s, int e, int tid) {
    const int lane = tid & 63, wave = tid >> 6;
    for (int b = 0; b < a.nb; ++b) {
        const bool last = b == a.nb - 1;
        if (b > 0) {
            __threadfence();                                 // this workgroup's `out` rows of the previous block, device-wide
            __syncthreads();
            stage_block_weights(a, b, W0s, Wss, W1s, Wcs, bsm, tid);
            __syncthreads();
        }
        TrunkFwdArgs t{};
        t.net_prev = b > 0 ? a.out[b - 1] : nullptr; t.ld_prev = 32; t.cell = a.cell; t.off0 = a.off0;
        t.M = a.M; t.pooled = a.pooled[b]; t.winner = a.winner[b]; t.loader = 1;
        for (int c0 = s; c0 < e; c0 += TR) {
            const int c1 = min(c0 + TR, e);
            if (b == 0) { fused_fc_pos(a, Xs, wps, c0, c1, tid); __syncthreads(); }
            else {
                pool_into_tile(t, Xs, Hsm, c0, c1, tid);
                for (int f = 0; f < 4; ++f) {                // the pooled half the backward reads: every wave its own rows
                    const int idx = lane + f * 64, row = wave * 32 + (idx >> 3), c = (idx & 7) * 4;
                    if (c0 + row < c1)
                        *reinterpret_cast<float4 *>(a.pooled[b] + (size_t)(c0 + row) * 32 + c) =
                            *reinterpret_cast<const float4 *>(Xs + row * XS + 32 + c);
                }
            }
            fused_block_gemms(Xs, Hsm, W0s, Wss, W1s, Wcs, bsm, c0, c1, a.hr[b], a.out[b], a.c_out, last, false, lane, wave);
            __syncthreads();
        }
    }
}

__global__ __launch_bounds__(256, 2) void trunk_fused_fwd_kernel(TrunkFusedArgs a) {
    __shared__ __attribute__((aligned(16))) float Xs[TR * XS];
    __shared__ __attribute__((aligned(16))) float Hsm[TR * HS];
    __shared__ __attribute__((aligned(16))) float W0s[32 * XS];
    __shared__ __attribute__((aligned(16))) float Wss[32 * XS];
    __shared__ __attribute__((aligned(16))) float W1s[32 * HS];
    __shared__ __attribute__((aligned(16))) float Wcs[32 * HS];
    __shared__ float bsm[96];
    __shared__ float wps[256];
    __shared__ int unit[2];
    __shared__ unsigned long long heads[2];
    __shared__ uint8_t cstart[TR + 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < 192) wps[tid] = a.wpos[tid];
    if (tid < 64) wps[192 + tid] = a.bpos[tid];
    // fc_pos of the workgroup's FIRST unit reads wps before the unit's first barrier.  r06 (last day): without this barrier a wave
    // that loads none of wps (wave 3) could get there first and compute its 32 rows with whatever the LDS held -- the same values
    // when the previous workgroup on the CU was this kernel's (which is why the kernel alone never showed it), something else beside
    // the split convolutions: 2 of 4 000 launches, one unit each (profiles/coresidency_trunk_fused.py, r06_coresidency.txt section 8)
#ifndef T2H_LAB_NO_PROLOGUE_BARRIER                         // (lab build: shows T2H_POISON_LDS catching it in tests/test_hip_trunk.py)
    __syncthreads();
#endif
#ifdef T2H_TRUNK_ABLATE
    const int abl = a.ablate;
#else
    constexpr int abl = 0;
#endif
    // Latencies taken off the unit's critical path: the bounds of the NEXT unit of this workgroup are requested one unit ahead
    // (a.bounds: one load each; without the array, the two-hop lookup by thread 0), its points and cell ids while this unit's
    // blocks run, and block 0's weights for the next unit right after this unit's last GEMM.
    int k = blockIdx.x;
    int s = 0, e = 0;
    const int n_units = a.bounds ? *a.n_bounds : a.n_units;     // (a local: writing the member would copy the whole argument block to scratch)
    if (k < n_units) {
        if (a.bounds) { const int2 u = a.bounds[k]; s = u.x; e = u.y; }
        else {
            if (tid == 0) { unit[0] = (abl & 16) ? min(k * a.stride, a.M) : unit_start(a, k); unit[1] = (abl & 16) ? min((k + 1) * a.stride, a.M) : unit_start(a, k + 1); }
            __syncthreads();
            s = unit[0]; e = unit[1];
        }
    }
    stage_block_weights(a, 0, W0s, Wss, W1s, Wcs, bsm, tid);
    // this thread's share of a unit's inputs: row tid >> 1 of the points (both threads of a pair read the same point), cell id of row tid
    float p0 = 0.f, p1 = 0.f, p2 = 0.f;
    int cid = 0;
    auto request = [&](int us, int ue) {
        const int row = tid >> 1;
        p0 = p1 = p2 = 0.f;
        if (ue - us <= TR) {                                 // (a long unit reads its points chunk by chunk itself)
            if (us + row < ue) { const float *p = a.pts + (size_t)(us + row) * a.dim; p0 = p[0]; p1 = p[1]; p2 = p[2]; }
            cid = (tid < TR && us + tid < ue) ? a.cell[us + tid] : -1 - tid;
        }
    };
    request(s, e);
    for (; k < n_units; k += gridDim.x) {
        const int kn = k + gridDim.x;
        int sn = a.M, en = a.M;
        if (kn < n_units && a.bounds) { const int2 u = a.bounds[kn]; sn = u.x; en = u.y; }  // (in flight during this unit)
        const int rows = e - s;
        if (rows > 0 && rows <= TR) {
            // ---------------------------------------------------------------- fast path: the unit lives in LDS for all blocks
            int *cells = reinterpret_cast<int *>(Hsm);
            {
                const int row = tid >> 1, c0 = (tid & 1) * 32;
#pragma unroll
                for (int c = 0; c < 32; c += 4) {
                    float v[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int n = c0 + c + j;
                        float acc = wps[192 + n];
                        acc = fmaf(p0, wps[n * 3 + 0], acc);
                        acc = fmaf(p1, wps[n * 3 + 1], acc);
                        acc = fmaf(p2, wps[n * 3 + 2], acc);
                        v[j] = keep_if_positive(rows - row, acc);
                    }
                    *reinterpret_cast<float4 *>(Xs + row * XS + c0 + c) = make_float4(v[0], v[1], v[2], v[3]);
                }
                if (tid < TR) cells[tid] = cid;                  // (Hsm is free until the unit's first GEMM)
            }
            __syncthreads();                                 // X, the cell ids, block 0's weights
            if (tid < TR) {
                const bool head = tid == 0 || cells[tid] != cells[tid - 1];
                const unsigned long long m = __ballot(head);
                if ((tid & 63) == 0) heads[tid >> 6] = m;
            }
            __syncthreads();
            int span[TR / NG];
            int ncell;
            {
                const unsigned long long m0 = heads[0], m1 = heads[1];
#pragma unroll
                for (int p = 0; p < TR / NG; ++p) {
                    const int me = p * NG + (tid >> 3);
                    span[p] = mask_prev_head(m0, m1, me) | (mask_next_head(m0, m1, me) << 8);
                }
                // the unit's cells as a list (rows past the unit carry unique ids: heads that do not count).  Read first behind the
                // barrier that follows the block's GEMMs
                const unsigned long long v0 = rows >= 64 ? ~0ull : ((1ull << rows) - 1);
                const unsigned long long v1 = rows >= 128 ? ~0ull : (rows > 64 ? ((1ull << (rows - 64)) - 1) : 0ull);
                const int n0 = __popcll(m0 & v0);
                ncell = n0 + __popcll(m1 & v1);
                if (tid < rows) {
                    const unsigned long long mine = tid < 64 ? m0 : m1;
                    if ((mine >> (tid & 63)) & 1)
                        cstart[(tid < 64 ? 0 : n0) + __popcll(mine & ((1ull << (tid & 63)) - 1))] = (uint8_t)tid;
                }
                if (tid == 0) cstart[ncell] = (uint8_t)rows;         // (rows == 128 -> 128: fits a byte)
            }
            if (!a.bounds && kn < n_units) {               // (no bounds array: the next unit's lookup, off the critical path)
                if (tid == 0) { unit[0] = (abl & 16) ? min(kn * a.stride, a.M) : unit_start(a, kn); unit[1] = (abl & 16) ? min((kn + 1) * a.stride, a.M) : unit_start(a, kn + 1); }
            }
            for (int b = 0; b < a.nb; ++b) {
                const bool last = b == a.nb - 1;
                if (wave * 32 < rows)
                    fused_block_gemms(Xs, Hsm, W0s, Wss, W1s, Wcs, bsm, s, e, a.hr[b], a.out[b], a.c_out, last, !last, lane, wave, abl);
                else if (!last) {                           // a wave without rows still owes the next loader its zero rows
                    for (int i = lane; i < 32 * 8; i += 64)
                        *reinterpret_cast<float4 *>(Xs + (wave * 32 + (i >> 3)) * XS + (i & 7) * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
                }
                __syncthreads();                             // every wave's `out` rows are in the tile; the block's weights are consumed
                if (b == 0) {                                // the next unit's inputs, requested under this unit's remaining blocks
                    if (!a.bounds && kn < n_units) { sn = unit[0]; en = unit[1]; }
                    request(sn, en);
                }
                if (!(abl & 4) || last) stage_block_weights(a, last ? 0 : b + 1, W0s, Wss, W1s, Wcs, bsm, tid);
                if (last) break;
                if (!(abl & 1)) {
                    if (abl & 32) fused_pool_local(Xs, span, s, rows, a.winner[b + 1], a.pooled[b + 1], tid, abl);
                    else {
#if T2H_LAB_TRUNK_PRIO == 2
                        __builtin_amdgcn_s_setprio(1);
#endif
                        fused_pool_cells(Xs, reinterpret_cast<unsigned *>(Hsm), cstart, ncell, span, s, rows, a.winner[b + 1],
                                         a.pooled[b + 1], tid, abl);
#if T2H_LAB_TRUNK_PRIO == 2
                        __builtin_amdgcn_s_setprio(0);
#endif
                    }
                }
                __syncthreads();
            }
        } else if (rows > TR) {
            // ---------------------------------------------------------------- slow path: chunks of TR rows, block by block, through HBM
            __syncthreads();                                 // block 0's weights (and wps)
            if (!a.bounds && kn < n_units) {
                if (tid == 0) { unit[0] = unit_start(a, kn); unit[1] = unit_start(a, kn + 1); }
            }
            fused_slow_unit(a, Xs, Hsm, W0s, Wss, W1s, Wcs, bsm, wps, s, e, tid);
            if (!a.bounds && kn < n_units) { sn = unit[0]; en = unit[1]; }
            request(sn, en);
            stage_block_weights(a, 0, W0s, Wss, W1s, Wcs, bsm, tid);
        } else {
            // an empty unit (the cell that starts before k S covers the whole window): nothing to do but to move on
            if (!a.bounds && kn < n_units) {
                __syncthreads();
                if (tid == 0) { unit[0] = unit_start(a, kn); unit[1] = unit_start(a, kn + 1); }
                __syncthreads();
                sn = unit[0]; en = unit[1];
            }
            request(sn, en);
        }
        s = sn; e = en;
    }
}

// Work units of t2h_trunk_fused_fwd, packed greedily: a unit takes whole cells while they fit TR rows.  The packing is a
// sequential walk, so the rows are cut into segments of kUnitSeg rows (snapped up to the next cell start) and one thread walks
// each segment: unit k = (start, end) pairs in slot k of the segment's kUnitSlots slots (unused slots: start == end).  A cell of
// more than TR rows is a unit of its own (the block-by-block path).  A unit is shorter than TR - (next cell's rows) only at a
// segment's end, so units average ~TR - half a cell (vs a fixed stride that must leave room for the largest cell).
constexpr int kUnitSeg = 1024, kUnitSlots = 24;       // (<= 2 units per TR + 1 rows in the worst alternation of 1-row and TR-row cells)
__device__ inline int snap_up(const int32_t *cell, const int32_t *off0, long long row, int M) {
    if (row >= M) return M;
    const int c = cell[row], st = off0[c];
    return st == (int)row ? (int)row : off0[c + 1];
}
__global__ __launch_bounds__(64) void trunk_units_kernel(const int32_t *__restrict__ cell, const int32_t *__restrict__ off0, int M,
                                                        int n_seg, int2 *__restrict__ units, int *__restrict__ counts) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_seg) return;
    int u = j == 0 ? 0 : snap_up(cell, off0, (long long)j * kUnitSeg, M);
    const int seg_end = snap_up(cell, off0, (long long)(j + 1) * kUnitSeg, M);
    int2 *slot = units + (size_t)j * kUnitSlots;
    int n = 0;
    while (u < seg_end && n < kUnitSlots) {
        int nxt = seg_end;
        if (seg_end - u > TR) {
            const int c = cell[u + TR], st = off0[c];          // the cell that holds row u + TR
            nxt = st > u ? st : off0[c + 1];                   // (st <= u: the cell at u is longer than TR rows -- a unit of its own)
        }
        if (n == kUnitSlots - 1) nxt = seg_end;                // (cannot happen by the bound above; never drop rows)
        slot[n++] = make_int2(u, nxt);
        u = nxt;
    }
    counts[j] = n;
    for (; n < kUnitSlots; ++n) slot[n] = make_int2(seg_end, seg_end);
}

// the used slots of all segments -> one dense list (segment order = row order) + their number: one workgroup, chunks of 1024 segments
__global__ __launch_bounds__(1024) void trunk_units_compact_kernel(const int2 *__restrict__ raw, const int *__restrict__ counts, int n_seg,
                                                                  int2 *__restrict__ dense, int *__restrict__ total) {
    __shared__ int scan[1024];
    __shared__ int carry;
    const int t = threadIdx.x;
    if (t == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < n_seg; base += 1024) {
        const int j = base + t, c = j < n_seg ? counts[j] : 0;
        scan[t] = c;
        __syncthreads();
        for (int d = 1; d < 1024; d <<= 1) {                 // inclusive scan (Hillis-Steele)
            const int v = t >= d ? scan[t - d] : 0;
            __syncthreads();
            scan[t] += v;
            __syncthreads();
        }
        const int at = carry + scan[t] - c;
        for (int i = 0; i < c; ++i) dense[at + i] = raw[(size_t)j * kUnitSlots + i];
        __syncthreads();
        if (t == 1023) carry += scan[1023];
        __syncthreads();
    }
    if (t == 0) *total = carry;
}

// =====================================================================================================================
// Backward of one block (resnet.py:36-54 + the gather / scatter_max backward of pointnet.py:92-99), one launch:
//
//     g    = g_net + route(sum over the cell of g_pool)       (the pooling's backward, folded into this loader;
//                                                               last block: g = (gc Wc) * (out > 0), the fc_c backward)
//     dhr  = (g W1) * (hr > 0)
//     dX   = g Ws + (dhr W0) * (X > 0)                        -> [d net_prev | d pooled_prev]  (first block: fc_pos wgrad)
//     dW1 += g^T hr, dW0 += dhr^T relu(X), dWs += g^T X, db1 += colsum g, db0 += colsum dhr
//
// Everything a weight gradient or a ReLU mask needs is held in the MFMA C/D register layout (lane = column, registers =
// rows): hr, X and out are loaded from global memory straight into that layout (each load instruction reads two 128-byte
// row segments), dhr leaves its MFMA in it, and the reduction index of a weight gradient (the row) may be enumerated in
// any order as long as both operands agree -- so those products need no LDS at all.  LDS holds only what the two data
// gradients read row-wise (g and dhr, 128 x 32 each) and the transposed weights (23 KB): 60 KB, two workgroups per CU.
// A workgroup walks `tiles_per_wg` consecutive tiles with the weight-gradient accumulators in registers, then writes one
// slab; t2h_reduce_segments adds the slabs in a fixed order (deterministic, no atomics).
struct TrunkBwdArgs {
    const float *g_net; int ld_gn;
    const float *g_pool; int ld_gp; const uint8_t *winner; const int32_t *cell, *off0;
    const float *gc, *wc, *out_last;                                        // last block
    const float *hr, *x_left, *x_right; int ld_xl, ld_xr;                     // X = [x_left | x_right], 32 columns each
    const float *pts; int dim; const float *wpos, *bpos;                    // first block
    const float *w0, *w1, *ws;
    int M, tiles_per_wg;
    float *dx;                                                              // [M, 64] (not for the first block)
    float *slab; int slab_floats;
};

// slab layout (floats): dW0 [32][64] | dWs [32][64] | dW1 [32][32] | db0 [32] | db1 [32] | LAST: dWc [32][32] | dbc [32]
//                                                                              | FIRST: dWpos [64][3] | dbpos [64]
constexpr int SL_W0 = 0, SL_WS = 2048, SL_W1 = 4096, SL_B0 = 5120, SL_B1 = 5152, SL_X = 5184;

// the tile's g rows -> Gs (row-major, stride HS): g_net + (winner bit ? sum over the row's cell of g_pool : 0)
__device__ inline void load_g_tile(const TrunkBwdArgs &a, float *Gs, float *scratch, int r0, int r1, int tid) {
    const int lane = tid & (G - 1), grp = tid >> 3;
    if (!a.g_pool) {
#pragma unroll
        for (int p = 0; p < TR / NG; ++p) {
            const int row = r0 + p * NG + grp;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < r1) v = *reinterpret_cast<const float4 *>(a.g_net + (size_t)row * a.ld_gn + lane * 4);
            *reinterpret_cast<float4 *>(Gs + (p * NG + grp) * HS + lane * 4) = v;
        }
        __syncthreads();
        return;
    }
    float4 *val = reinterpret_cast<float4 *>(scratch);                      // [TR * G]: g_pool rows, then cell sums at head rows
    float4 *pval = reinterpret_cast<float4 *>(Gs);                          // [NG * G]   (Gs is written last)
    float4 *oval = reinterpret_cast<float4 *>(Gs + 4 * NG * G);             // [2 * G]
    int *bounds = reinterpret_cast<int *>(Gs + 4 * NG * G + 8 * G);         // [4]
    int segs[TR / NG], sege[TR / NG];
#pragma unroll
    for (int p = 0; p < TR / NG; ++p) {
        const int row = r0 + p * NG + grp;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        segs[p] = sege[p] = -1;
        if (row < r1) {
            const int cid = a.cell[row];
            segs[p] = a.off0[cid]; sege[p] = a.off0[cid + 1];
            v = *reinterpret_cast<const float4 *>(a.g_pool + (size_t)row * a.ld_gp + lane * 4);
        }
        val[(p * NG + grp) * G + lane] = v;
    }
    if (tid == 0) {
        const int ch = a.cell[r0], ct = a.cell[r1 - 1];
        bounds[0] = a.off0[ch]; bounds[1] = a.off0[ch + 1]; bounds[2] = a.off0[ct]; bounds[3] = a.off0[ct + 1];
    }
    __syncthreads();
#pragma unroll
    for (int side = 0; side < 2; ++side) {
        const int lo = side == 0 ? bounds[0] : r1, hi = side == 0 ? r0 : bounds[3];
        if (hi <= lo) continue;                                             // uniform over the workgroup
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int n = lo + grp; n < hi; n += NG) {
            const float4 v = *reinterpret_cast<const float4 *>(a.g_pool + (size_t)n * a.ld_gp + lane * 4);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        pval[grp * G + lane] = acc;
        __syncthreads();
        if (grp == 0) {
            float4 t = pval[lane];
            const int used = min(NG, hi - lo);                              // later groups hold zeros: adding them changes nothing
            for (int g = 1; g < used; ++g) { const float4 v = pval[g * G + lane]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
            oval[side * G + lane] = t;
        }
        __syncthreads();
    }
    float4 headsum[TR / NG];
#pragma unroll
    for (int p = 0; p < TR / NG; ++p) {
        const int row = r0 + p * NG + grp;
        headsum[p] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row >= r1 || row != max(segs[p], r0)) continue;
        float4 acc = segs[p] < r0 ? oval[lane] : make_float4(0.f, 0.f, 0.f, 0.f);
        const int end = min(sege[p], r1);
        for (int n = row; n < end; ++n) { const float4 v = val[(n - r0) * G + lane]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
        if (sege[p] > r1) { const float4 v = oval[G + lane]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
        headsum[p] = acc;
    }
    __syncthreads();                                                        // every head has read its rows: now overwrite
#pragma unroll
    for (int p = 0; p < TR / NG; ++p) {
        const int row = r0 + p * NG + grp;
        if (row < r1 && row == max(segs[p], r0)) val[(row - r0) * G + lane] = headsum[p];
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < TR / NG; ++p) {
        const int row = r0 + p * NG + grp;
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row < r1) {
            const float4 sum = val[(max(segs[p], r0) - r0) * G + lane];
            const uint8_t bits = a.winner[(size_t)row * G + lane];
            const float4 gn = *reinterpret_cast<const float4 *>(a.g_net + (size_t)row * a.ld_gn + lane * 4);
            o = make_float4(gn.x + ((bits & 1) ? sum.x : 0.f), gn.y + ((bits & 2) ? sum.y : 0.f),
                            gn.z + ((bits & 4) ? sum.z : 0.f), gn.w + ((bits & 8) ? sum.w : 0.f));
        }
        *reinterpret_cast<float4 *>(Gs + (p * NG + grp) * HS + lane * 4) = o;
    }
    __syncthreads();
}

// x > 0 ? a : +0 without a compare: the sign test as an integer clamp of x's bits to {0, 1}, the select as a bitwise AND with
// -that.  (Bit-identical for every non-NaN x.)  The FIRST backward variant uses it for its ReLU masks: a `v_cmp -> SGPR pair ->
// v_cndmask` there was seen to lose lanes 48..63 of the mask beside the split-convolution kernels (r05 / r06:
// profiles/r06_coresidency.txt -- 86 of 2000 launches of a dense replay), and no wait state between the two changes that
// (the clamp is an opaque v_med3_i32: written as min(max(bits, 0), 1) the compiler recognises the idiom and emits the very
// v_cmp_lt_i32 -> v_cndmask it was meant to avoid)
__device__ inline float keep_where_positive(float x, float a) {
    int m;
    asm("v_med3_i32 %0, %1, 0, 1" : "=v"(m) : "v"(__float_as_int(x)));
    return __int_as_float(__float_as_int(a) & -m);
}

// weight gradient tile: acc[n][k] += sum over the 32 rows of a[row][n] * b[row][k], operands in the C/D register layout
__device__ inline void mfma_outer(const float (&av)[16], const float (&bv)[16], f32x16 &acc) {
#pragma unroll
    for (int q = 0; q < 16; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q], bv[q], acc, 0, 0, 0);
}

template <bool FIRST, bool LAST>
__global__ __launch_bounds__(256, 2) void trunk_block_bwd_kernel(TrunkBwdArgs a) {
    __shared__ __attribute__((aligned(16))) float Gs[TR * HS];
    __shared__ __attribute__((aligned(16))) float Ds[TR * HS];
    __shared__ __attribute__((aligned(16))) float W1t[32 * HS];              // [k][n] = W1[n][k]
    __shared__ __attribute__((aligned(16))) float W0t[64 * HS];              // [k][n] = W0[n][k]
    __shared__ __attribute__((aligned(16))) float Wst[64 * HS];
    __shared__ __attribute__((aligned(16))) float Wct[LAST ? 32 * HS : 4];
    __shared__ float Pts[FIRST ? TR * 3 + 256 : 4];                          // tile points, then Wpos [64][3] + bpos [64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;

    // ---- transposed weights -> LDS
#pragma unroll
    for (int f = 0; f < 2; ++f) {
        const int idx = tid + f * 256, n = idx >> 4, k4 = (idx & 15) * 4;    // W [32][64]: n row, 4 consecutive k
        const float4 v0 = *reinterpret_cast<const float4 *>(a.w0 + n * 64 + k4);
        const float4 vs = *reinterpret_cast<const float4 *>(a.ws + n * 64 + k4);
        W0t[(k4 + 0) * HS + n] = v0.x; W0t[(k4 + 1) * HS + n] = v0.y; W0t[(k4 + 2) * HS + n] = v0.z; W0t[(k4 + 3) * HS + n] = v0.w;
        Wst[(k4 + 0) * HS + n] = vs.x; Wst[(k4 + 1) * HS + n] = vs.y; Wst[(k4 + 2) * HS + n] = vs.z; Wst[(k4 + 3) * HS + n] = vs.w;
    }
    {
        const int n = tid >> 3, k4 = (tid & 7) * 4;
        const float4 v1 = *reinterpret_cast<const float4 *>(a.w1 + n * 32 + k4);
        W1t[(k4 + 0) * HS + n] = v1.x; W1t[(k4 + 1) * HS + n] = v1.y; W1t[(k4 + 2) * HS + n] = v1.z; W1t[(k4 + 3) * HS + n] = v1.w;
        if (LAST) {
            const float4 vc = *reinterpret_cast<const float4 *>(a.wc + n * 32 + k4);
            Wct[(k4 + 0) * HS + n] = vc.x; Wct[(k4 + 1) * HS + n] = vc.y; Wct[(k4 + 2) * HS + n] = vc.z; Wct[(k4 + 3) * HS + n] = vc.w;
        }
    }
    if (FIRST) {
        if (tid < 192) Pts[TR * 3 + tid] = a.wpos[tid];
        if (tid < 64) Pts[TR * 3 + 192 + tid] = a.bpos[tid];
    }

    f32x16 acc_w0[2], acc_ws[2], acc_w1, acc_wc;
#pragma unroll
    for (int q = 0; q < 16; ++q) { acc_w0[0][q] = acc_w0[1][q] = acc_ws[0][q] = acc_ws[1][q] = acc_w1[q] = acc_wc[q] = 0.f; }
    float db0 = 0.f, db1 = 0.f, dbc = 0.f;
    float dwp[2][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}}, dbp[2] = {0.f, 0.f};

    const int n_tiles = (a.M + TR - 1) / TR;
    const int t_begin = blockIdx.x * a.tiles_per_wg, t_end = min(t_begin + a.tiles_per_wg, n_tiles);
    for (int tile = t_begin; tile < t_end; ++tile) {
        const int r0 = tile * TR, r1 = min(r0 + TR, a.M);
        const int wrow0 = r0 + wave * 32;                                    // first global row of this wave
        float g_cd[16];
        // register q of the C/D layout holds row rb + ro(q), ro(q) = (q & 3) + 8 (q >> 2): one base, constant offsets
        const int rb = wrow0 + 4 * h;
#define T2H_RO(q) (((q) & 3) + 8 * ((q) >> 2))
        // "row q of this lane lies inside the cloud" (false only in the last workgroup), compared AFRESH at every use (`fresh` hides
        // the row from common-subexpression elimination): left to itself the compiler computes the sixteen compares once per tile and
        // parks them in SGPR pairs across ~1000 instructions, and a parked compare result is what the r05 co-residency fault eats
        // (point_grid.hip's forward walk; profiles/r05_coresidency.txt).  Also 24 VGPRs cheaper here.  The FIRST variant (at the
        // register limit: it spills with this rewrite) has no row masks at all since r06, see below
        auto fresh = [](int v) { if constexpr (!FIRST) asm volatile("" : "+v"(v)); return v; };
#define T2H_ROW_OK(q) (fresh(rb) + T2H_RO(q) < r1)
        // hr in the C/D layout, requested before the g stage: its barriers keep the compiler from hoisting the loads itself
        float hr_cd[16];
        if constexpr (FIRST) {
            // r06: NO row masks in this variant.  A row past the cloud (last tile only) has g = 0 (load_g_tile stores zeros for it)
            // and zero points in LDS, hence dhr = 0 and dX = 0: every product it enters is 0 x (a finite number) whatever X holds
            // for it -- X is left unmasked
            // (buffer loads: the tile's valid rows are the descriptor's range, a row past it reads as 0 -- the range check is the
            // mask -- and the sixteen loads share one offset register)
            const __amdgpu_buffer_rsrc_t hr_rows = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float *>(a.hr + (size_t)r0 * 32), 0, (r1 - r0) * 128, 0x00020000);
            const int voff = ((wave * 32 + 4 * h) * 32 + r) * 4;
#pragma unroll
            for (int q = 0; q < 16; ++q)
                hr_cd[q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(hr_rows, voff + T2H_RO(q) * 128, 0, 0));
        } else {
#pragma unroll
            for (int q = 0; q < 16; ++q) hr_cd[q] = T2H_ROW_OK(q) ? (a.hr + (size_t)rb * 32 + r)[T2H_RO(q) * 32] : 0.f;
        }

        // ---------------------------------------------------------------- g -> Gs (row-major) and g_cd (registers)
        if constexpr (LAST) {
#pragma unroll
            for (int p = 0; p < TR / NG; ++p) {                              // gc tile -> Ds
                const int row = r0 + p * NG + (tid >> 3);
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (row < r1) v = *reinterpret_cast<const float4 *>(a.gc + (size_t)row * 32 + (tid & 7) * 4);
                *reinterpret_cast<float4 *>(Ds + (p * NG + (tid >> 3)) * HS + (tid & 7) * 4) = v;
            }
            __syncthreads();
            f32x16 acc_g;
#pragma unroll
            for (int q = 0; q < 16; ++q) acc_g[q] = 0.f;
            mfma_rows<32, false>(Ds + (wave * 32 + r) * HS + 4 * h, Wct + r * HS + 4 * h, acc_g);     // gc Wc
            float gc_cd[16], o_cd[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                gc_cd[q] = Ds[(wave * 32 + acc_row(q, lane)) * HS + r];
                o_cd[q] = T2H_ROW_OK(q) ? (a.out_last + (size_t)rb * 32 + r)[T2H_RO(q) * 32] : 0.f;
            }
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                g_cd[q] = o_cd[q] > 0.f ? acc_g[q] : 0.f;                    // through relu(net) of pointnet.py:81
                dbc += gc_cd[q];
                o_cd[q] = fmaxf(o_cd[q], 0.f);
            }
            mfma_outer(gc_cd, o_cd, acc_wc);                                 // dWc += gc^T relu(out)
#pragma unroll
            for (int q = 0; q < 16; ++q) Gs[(wave * 32 + acc_row(q, lane)) * HS + r] = g_cd[q];
            __syncthreads();
        } else {
            load_g_tile(a, Gs, Ds, r0, r1, tid);
#pragma unroll
            for (int q = 0; q < 16; ++q) g_cd[q] = Gs[(wave * 32 + acc_row(q, lane)) * HS + r];
        }

        // ---------------------------------------------------------------- dhr = (g W1) * (hr > 0); dW1, db1, db0
        float d_cd[16];
        {
            f32x16 acc_d;
#pragma unroll
            for (int q = 0; q < 16; ++q) acc_d[q] = 0.f;
            mfma_rows<32, false>(Gs + (wave * 32 + r) * HS + 4 * h, W1t + r * HS + 4 * h, acc_d);
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                d_cd[q] = FIRST ? keep_where_positive(hr_cd[q], acc_d[q]) : (hr_cd[q] > 0.f ? acc_d[q] : 0.f);
                db1 += g_cd[q];
                db0 += d_cd[q];
            }
            mfma_outer(g_cd, hr_cd, acc_w1);                                 // dW1 += g^T hr   (hr is already relu'ed)
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) Ds[(wave * 32 + acc_row(q, lane)) * HS + r] = d_cd[q];      // own rows only

        // ---------------------------------------------------------------- per 32-column half of X: dW0, dWs, dX
        if constexpr (FIRST) {
            for (int i = tid; i < TR * 3; i += 256) {                        // (the previous tile ended with a barrier)
                const int row = r0 + i / 3;
                Pts[i] = row < r1 ? a.pts[(size_t)row * a.dim + i % 3] : 0.f;
            }
        }
        __syncthreads();                                                     // Ds rows (and Pts) are complete
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            float x_cd[16];                                                  // X[row(q)][32 t + r], the C/D register layout
            if constexpr (FIRST) {
                // the tile's points are re-read from LDS at every use (broadcast reads): keeping them in registers across
                // the MFMA phases costs 48 VGPRs and spills
                asm volatile("" ::: "memory");
                const float *wp = Pts + TR * 3;
                const int n = 32 * t + r;
                const float w_0 = wp[n * 3 + 0], w_1 = wp[n * 3 + 1], w_2 = wp[n * 3 + 2], b_n = wp[192 + n];
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const float *pp = Pts + (wave * 32 + acc_row(q, lane)) * 3;
                    float acc = b_n;                                         // the forward's fc_pos arithmetic, bit for bit
                    acc = fmaf(pp[0], w_0, acc);
                    acc = fmaf(pp[1], w_1, acc);
                    acc = fmaf(pp[2], w_2, acc);
                    x_cd[q] = acc;                                           // (rows past the cloud: see hr_cd above)
                }
            } else {
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const float *xp = t == 0 ? a.x_left + (size_t)rb * a.ld_xl + r : a.x_right + (size_t)rb * a.ld_xr + r;
                    const int ldx = t == 0 ? a.ld_xl : a.ld_xr;
                    x_cd[q] = T2H_ROW_OK(q) ? xp[T2H_RO(q) * ldx] : 0.f;
                }
            }
            {
                float xr[16];
#pragma unroll
                for (int q = 0; q < 16; ++q) xr[q] = fmaxf(x_cd[q], 0.f);
                mfma_outer(d_cd, xr, acc_w0[t]);                             // dW0 += dhr^T relu(X)
                mfma_outer(g_cd, x_cd, acc_ws[t]);                           // dWs += g^T X
            }
            // dX = (dhr W0) * (X > 0) + g Ws
            f32x16 acc_x;
#pragma unroll
            for (int q = 0; q < 16; ++q) acc_x[q] = 0.f;
            mfma_rows<32, false>(Ds + (wave * 32 + r) * HS + 4 * h, W0t + (32 * t + r) * HS + 4 * h, acc_x);
#pragma unroll
            for (int q = 0; q < 16; ++q) acc_x[q] = FIRST ? keep_where_positive(x_cd[q], acc_x[q]) : (x_cd[q] > 0.f ? acc_x[q] : 0.f);
            mfma_rows<32, false>(Gs + (wave * 32 + r) * HS + 4 * h, Wst + (32 * t + r) * HS + 4 * h, acc_x);
            if constexpr (FIRST) {
                asm volatile("" ::: "memory");
#pragma unroll
                for (int q = 0; q < 16; ++q) {                               // fc_pos: dWpos += dX^T pts, dbpos += colsum dX
                    const float *pp = Pts + (wave * 32 + acc_row(q, lane)) * 3;
                    const float v = acc_x[q];                               // (0 for a row past the cloud: dhr = g = 0)
                    dbp[t] += v;
                    dwp[t][0] = fmaf(v, pp[0], dwp[t][0]);
                    dwp[t][1] = fmaf(v, pp[1], dwp[t][1]);
                    dwp[t][2] = fmaf(v, pp[2], dwp[t][2]);
                }
            } else {
#pragma unroll
                for (int q = 0; q < 16; ++q)
                    if (T2H_ROW_OK(q)) (a.dx + (size_t)rb * 64 + 32 * t + r)[T2H_RO(q) * 64] = acc_x[q];
            }
        }
        __syncthreads();                                                     // Gs / Ds free for the next tile
#undef T2H_RO
#undef T2H_ROW_OK
    }

    // ---- accumulators of the 4 waves -> one slab per workgroup, summed in wave order
    float *slab = a.slab + (size_t)blockIdx.x * a.slab_floats;
    float *scr = Gs;                                                         // Gs and Ds are adjacent: 2 * TR * HS floats
    static_assert(2 * TR * HS >= 4 * 1024, "scratch too small");
    auto flush = [&](const f32x16 &acc, float *dst, int ld, int col0) {
#pragma unroll
        for (int q = 0; q < 16; ++q) scr[wave * 1024 + acc_row(q, lane) * 32 + r] = acc[q];
        __syncthreads();
        const int e = tid * 4, n = e >> 5, k = e & 31;
        float4 s0 = *reinterpret_cast<const float4 *>(scr + e);
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            const float4 v = *reinterpret_cast<const float4 *>(scr + w * 1024 + e);
            s0.x += v.x; s0.y += v.y; s0.z += v.z; s0.w += v.w;
        }
        *reinterpret_cast<float4 *>(dst + n * ld + col0 + k) = s0;
        __syncthreads();
    };
    flush(acc_w0[0], slab + SL_W0, 64, 0);
    flush(acc_w0[1], slab + SL_W0, 64, 32);
    flush(acc_ws[0], slab + SL_WS, 64, 0);
    flush(acc_ws[1], slab + SL_WS, 64, 32);
    flush(acc_w1, slab + SL_W1, 32, 0);
    if (LAST) flush(acc_wc, slab + SL_X, 32, 0);
    // column sums: every lane holds the partial of column r over its rows; 8 partials (wave, h) per column
    auto flush_col = [&](float v, float *dst) {
        scr[(wave * 2 + h) * 32 + r] = v;
        __syncthreads();
        if (tid < 32) {
            float t = scr[tid];
#pragma unroll
            for (int j = 1; j < 8; ++j) t += scr[j * 32 + tid];
            dst[tid] = t;
        }
        __syncthreads();
    };
    flush_col(db0, slab + SL_B0);
    flush_col(db1, slab + SL_B1);
    if (LAST) flush_col(dbc, slab + SL_X + 1024);
    if constexpr (FIRST) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int d = 0; d < 3; ++d) {                                    // dWpos [64][3]: column n = 32 t + r
                scr[(wave * 2 + h) * 32 + r] = dwp[t][d];
                __syncthreads();
                if (tid < 32) {
                    float v = scr[tid];
#pragma unroll
                    for (int j = 1; j < 8; ++j) v += scr[j * 32 + tid];
                    slab[SL_X + (32 * t + tid) * 3 + d] = v;
                }
                __syncthreads();
            }
            flush_col(dbp[t], slab + SL_X + 192 + 32 * t);
        }
    }
}

// out[i] = [out[i] +] sum over slabs z (ascending) of src[z * stride + i], several independent segments per launch
struct ReduceSeg { const float *src; float *dst; int n, accumulate; };
constexpr int kMaxSegs = 48;
struct ReduceArgs { ReduceSeg seg[kMaxSegs]; int n_segs, n_slabs; long long stride; };

__global__ __launch_bounds__(256) void reduce_segments_kernel(ReduceArgs a) {
    // a block owns 32 consecutive outputs; its 8 slab lanes each add the slabs z = lane, lane + 8, ... (eight loads in
    // flight, four interleaved partial sums), the lanes' partials are added in lane order: fixed order, deterministic
    __shared__ float part[8][32];
    const ReduceSeg sg = a.seg[blockIdx.y];
    const int e = threadIdx.x & 31, sl = threadIdx.x >> 5;
    for (int i0 = blockIdx.x * 32; i0 < sg.n; i0 += gridDim.x * 32) {
        const int i = i0 + e;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        if (i < sg.n) {
            const float *src = sg.src + i;
            int z = sl;
            for (; z + 56 < a.n_slabs; z += 64) {
                const float v0 = src[(size_t)z * a.stride], v1 = src[(size_t)(z + 8) * a.stride];
                const float v2 = src[(size_t)(z + 16) * a.stride], v3 = src[(size_t)(z + 24) * a.stride];
                const float v4 = src[(size_t)(z + 32) * a.stride], v5 = src[(size_t)(z + 40) * a.stride];
                const float v6 = src[(size_t)(z + 48) * a.stride], v7 = src[(size_t)(z + 56) * a.stride];
                s0 += v0; s1 += v1; s2 += v2; s3 += v3; s0 += v4; s1 += v5; s2 += v6; s3 += v7;
            }
            for (; z < a.n_slabs; z += 8) s0 += src[(size_t)z * a.stride];
        }
        part[sl][e] = (s0 + s1) + (s2 + s3);
        __syncthreads();
        if (sl == 0 && i < sg.n) {
            float t = part[0][e];
#pragma unroll
            for (int j = 1; j < 8; ++j) t += part[j][e];
            sg.dst[i] = sg.accumulate ? sg.dst[i] + t : t;
        }
        __syncthreads();
    }
}

}  // namespace
}  // namespace t2h

using namespace t2h;

static bool al16(const void *p) { return ((uintptr_t)p & 15) == 0; }

T2H_API int t2h_trunk_block_fwd(const float *pts, int dim, const float *w_pos, const float *b_pos, const float *net_prev,
                                int ld_prev, const int32_t *cell, const int32_t *off0, const float *w0, const float *b0,
                                const float *w1, const float *b1, const float *ws, const float *wc, const float *bc, int64_t M,
                                float *x_full, float *pooled, float *hr, float *out, int ld_out, uint8_t *winner, float *c_out,
                                t2h_stream_t stream) {
    const bool first = pts != nullptr;
    const bool last = wc != nullptr;
    if (!w0 || !b0 || !w1 || !b1 || !ws || !hr || !out) return fail(T2H_ERR_ARG, "trunk_block_fwd: null pointer");
    if (first ? (!w_pos || !b_pos || dim < 3) : (!net_prev || !cell || !off0 || !winner || ld_prev < 32 || ld_prev % 4))
        return fail(T2H_ERR_ARG, "trunk_block_fwd: bad input description");
    if (first && last) return fail(T2H_ERR_ARG, "trunk_block_fwd: a block is either the first or the last");
    if (last && (!bc || !c_out)) return fail(T2H_ERR_ARG, "trunk_block_fwd: last block needs bc and c_out");
    if (M < 0 || M >= ((int64_t)1 << 31) - TR || ld_out < 32 || ld_out % 4) return fail(T2H_ERR_ARG, "trunk_block_fwd: bad shape");
    if (!al16(w0) || !al16(w1) || !al16(ws) || !al16(hr) || !al16(out) || (x_full && !al16(x_full)) || (pooled && !al16(pooled)) ||
        (net_prev && !al16(net_prev)) ||
        (wc && !al16(wc)) || (c_out && !al16(c_out)))
        return fail(T2H_ERR_ARG, "trunk_block_fwd: pointers must be 16-byte aligned");
    if (M == 0) return T2H_OK;
    TrunkFwdArgs a{};
    a.pts = pts; a.dim = dim; a.wpos = w_pos; a.bpos = b_pos;
    a.net_prev = net_prev; a.ld_prev = ld_prev; a.cell = cell; a.off0 = off0;
    a.w0 = w0; a.b0 = b0; a.w1 = w1; a.b1 = b1; a.ws = ws; a.wc = wc; a.bc = bc;
    a.M = (int)M; a.x_full = x_full; a.pooled = pooled; a.hr = hr; a.out = out; a.ld_out = ld_out; a.winner = winner; a.c_out = c_out;
    // r03 A/B (N = 131072, middle block): r02 loader 47.1 us, ballot / per-row loader 50.5 us -- the loader's dependent hops were
    // not what bounds the kernel (two workgroups of 80 KB LDS per CU run their phases nearly in sequence); the r02 form stays
    static const int loader = getenv("T2H_TRUNK_LOADER") ? atoi(getenv("T2H_TRUNK_LOADER")) : 1;
    a.loader = loader;
    const int64_t n_tiles = (M + TR - 1) / TR;
    const dim3 grid((unsigned)(n_tiles < 512 ? n_tiles : 512));      // two resident per CU, each walks its tiles
    hipStream_t s = as_stream(stream);
    if (first) hipLaunchKernelGGL((trunk_block_fwd_kernel<true, false>), grid, dim3(256), 0, s, a);
    else if (last) hipLaunchKernelGGL((trunk_block_fwd_kernel<false, true>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((trunk_block_fwd_kernel<false, false>), grid, dim3(256), 0, s, a);
    note_kernel(first ? "trunk_block_fwd_kernel<true,false>" : (last ? "trunk_block_fwd_kernel<false,true>" : "trunk_block_fwd_kernel<false,false>"));
    return check_launch("trunk_block_fwd");
}

// capacity (units) of a tile of M rows, and the int32 words t2h_trunk_units_build needs: [dense pairs 2 cap][count, 3 pad][raw pairs
// 2 cap][per-segment counts]
static int trunk_units_cap(int64_t M) { return M < 1 ? 0 : (int)((M + kUnitSeg - 1) / kUnitSeg) * kUnitSlots; }
T2H_API int64_t t2h_trunk_units_words(int64_t M) {
    const int64_t cap = trunk_units_cap(M);
    return cap == 0 ? 0 : 4 * cap + 4 + (M + kUnitSeg - 1) / kUnitSeg;
}

T2H_API int t2h_trunk_units_build(const int32_t *cell, const int32_t *off0, int64_t M, int32_t *units, t2h_stream_t stream) {
    if (!cell || !off0 || !units) return fail(T2H_ERR_ARG, "trunk_units_build: null pointer");
    if (M < 0 || M >= ((int64_t)1 << 31) - TR) return fail(T2H_ERR_ARG, "trunk_units_build: bad shape");
    if (((uintptr_t)units & 7) != 0) return fail(T2H_ERR_ARG, "trunk_units_build: units must be 8-byte aligned");
    if (M == 0) return T2H_OK;
    const int n_seg = (int)((M + kUnitSeg - 1) / kUnitSeg), cap = trunk_units_cap(M);
    int2 *dense = reinterpret_cast<int2 *>(units), *raw = reinterpret_cast<int2 *>(units + 2 * cap + 4);
    int *total = units + 2 * cap, *counts = units + 4 * cap + 4;
    hipLaunchKernelGGL(trunk_units_kernel, dim3((unsigned)((n_seg + 63) / 64)), dim3(64), 0, as_stream(stream), cell, off0, (int)M, n_seg,
                       raw, counts);
    hipLaunchKernelGGL(trunk_units_compact_kernel, dim3(1), dim3(1024), 0, as_stream(stream), raw, counts, n_seg, dense, total);
    return check_launch("trunk_units_build");
}

T2H_API int t2h_trunk_fused_fwd(const float *pts, int dim, const float *w_pos, const float *b_pos, const float *const *block_params,
                                int n_blocks, const float *wc, const float *bc, const int32_t *cell, const int32_t *off0, int64_t M,
                                float *const *hr, float *const *out, float *const *pooled, uint8_t *const *winner, float *c_out,
                                int stride, const int32_t *units, t2h_stream_t stream) {
    if (!pts || !w_pos || !b_pos || !block_params || !wc || !bc || !cell || !off0 || !hr || !out || !pooled || !winner || !c_out)
        return fail(T2H_ERR_ARG, "trunk_fused_fwd: null pointer");
    if (n_blocks < 2 || n_blocks > kMaxTrunkBlocks) return fail(T2H_ERR_ARG, "trunk_fused_fwd: 2 <= n_blocks <= %d", kMaxTrunkBlocks);
    if (dim < 3 || M < 0 || M >= ((int64_t)1 << 31) - TR) return fail(T2H_ERR_ARG, "trunk_fused_fwd: bad shape");
    int ablate = 0;
#ifdef T2H_TRUNK_ABLATE
    ablate = stride >> 8; stride &= 255;
#endif
    if (stride <= 0) stride = 96;
    if (stride > TR) return fail(T2H_ERR_ARG, "trunk_fused_fwd: stride must be <= %d rows", TR);
    if (M == 0) return T2H_OK;
    TrunkFusedArgs a{};
    a.pts = pts; a.dim = dim; a.wpos = w_pos; a.bpos = b_pos; a.wc = wc; a.bc = bc; a.cell = cell; a.off0 = off0;
    a.M = (int)M; a.nb = n_blocks; a.stride = stride; a.c_out = c_out; a.ablate = ablate; a.bounds = reinterpret_cast<const int2 *>(units);
    a.n_bounds = units ? units + 2 * trunk_units_cap(M) : nullptr;
    a.n_units = units ? trunk_units_cap(M) : (int)((M + stride - 1) / stride);       // (with a list: an upper bound for the grid; the kernel reads the count)
    for (int b = 0; b < n_blocks; ++b) {
        a.w0[b] = block_params[5 * b]; a.b0[b] = block_params[5 * b + 1]; a.w1[b] = block_params[5 * b + 2];
        a.b1[b] = block_params[5 * b + 3]; a.ws[b] = block_params[5 * b + 4];
        a.hr[b] = hr[b]; a.out[b] = out[b]; a.pooled[b] = pooled[b]; a.winner[b] = winner[b];
        if (!a.w0[b] || !a.b0[b] || !a.w1[b] || !a.b1[b] || !a.ws[b] || !a.hr[b] || !a.out[b] || (b > 0 && (!a.pooled[b] || !a.winner[b])))
            return fail(T2H_ERR_ARG, "trunk_fused_fwd: null pointer in block %d", b);
        if (!al16(a.w0[b]) || !al16(a.w1[b]) || !al16(a.ws[b]) || !al16(a.hr[b]) || !al16(a.out[b]) || (b > 0 && !al16(a.pooled[b])))
            return fail(T2H_ERR_ARG, "trunk_fused_fwd: pointers must be 16-byte aligned");
    }
    if (!al16(wc) || !al16(c_out)) return fail(T2H_ERR_ARG, "trunk_fused_fwd: pointers must be 16-byte aligned");
    const int64_t expect = units ? (M + 95) / 96 : a.n_units;
    const dim3 grid((unsigned)(expect < 512 ? (expect < 1 ? 1 : expect) : 512));  // two resident per CU, each walks its units
    hipLaunchKernelGGL(trunk_fused_fwd_kernel, grid, dim3(256), 0, as_stream(stream), a);
    note_kernel("trunk_fused_fwd_kernel");
    return check_launch("trunk_fused_fwd");
}

constexpr int kTrunkSlabFloats = 6240;      // 5184 shared + max(last: 1024 + 32, first: 192 + 64), multiple of 4
static int trunk_bwd_tiles_per_wg(int64_t M) {
    const int64_t n_tiles = (M + TR - 1) / TR;
    int64_t t = (n_tiles + 511) / 512;       // <= 512 workgroups: two resident per CU, each keeps its weight-gradient
    return (int)(t < 1 ? 1 : t);             // accumulators in registers over its tiles
}
static int trunk_bwd_slabs(int64_t M) {
    const int64_t n_tiles = (M + TR - 1) / TR;
    const int tpw = trunk_bwd_tiles_per_wg(M);
    return (int)((n_tiles + tpw - 1) / tpw);
}

T2H_API size_t t2h_trunk_block_bwd_workspace_bytes(int64_t M) {
    if (M < 1) return 0;
    return (size_t)trunk_bwd_slabs(M) * kTrunkSlabFloats * sizeof(float);
}

T2H_API int t2h_trunk_block_bwd(const float *g_net, int ld_gn, const float *g_pool, int ld_gp, const uint8_t *winner,
                                const int32_t *cell, const int32_t *off0, const float *gc, const float *wc,
                                const float *out_last, const float *hr, const float *x_left, int ld_xl, const float *x_right,
                                int ld_xr, const float *pts, int dim, const float *w_pos, const float *b_pos, const float *w0,
                                const float *w1, const float *ws, int64_t M, float *dx, void *workspace, size_t workspace_bytes,
                                t2h_stream_t stream) {
    const bool first = pts != nullptr, last = gc != nullptr;
    if (!hr || !w0 || !w1 || !ws || !workspace) return fail(T2H_ERR_ARG, "trunk_block_bwd: null pointer");
    if (first && last) return fail(T2H_ERR_ARG, "trunk_block_bwd: a block is either the first or the last");
    if (last ? (!wc || !out_last) : (!g_net || ld_gn < 32 || ld_gn % 4))
        return fail(T2H_ERR_ARG, "trunk_block_bwd: bad output-gradient description");
    if (g_pool && (last || !winner || !cell || !off0 || ld_gp < 32 || ld_gp % 4))
        return fail(T2H_ERR_ARG, "trunk_block_bwd: bad pooled-gradient description");
    if (first ? (!w_pos || !b_pos || dim < 3) : (!x_left || !x_right || !dx || ld_xl < 32 || ld_xr < 32))
        return fail(T2H_ERR_ARG, "trunk_block_bwd: bad input description");
    if (M < 0 || M >= ((int64_t)1 << 31) - TR) return fail(T2H_ERR_ARG, "trunk_block_bwd: bad shape");
    if (!al16(w0) || !al16(w1) || !al16(ws) || !al16(hr) || (g_net && !al16(g_net)) || (g_pool && !al16(g_pool)) ||
        (gc && (!al16(gc) || !al16(wc))) || !al16(workspace))
        return fail(T2H_ERR_ARG, "trunk_block_bwd: pointers must be 16-byte aligned");
    if (M == 0) return T2H_OK;
    if (workspace_bytes < t2h_trunk_block_bwd_workspace_bytes(M))
        return fail(T2H_ERR_WORKSPACE, "trunk_block_bwd: workspace %zu < %zu bytes", workspace_bytes, t2h_trunk_block_bwd_workspace_bytes(M));
    TrunkBwdArgs a{};
    a.g_net = g_net; a.ld_gn = ld_gn; a.g_pool = g_pool; a.ld_gp = ld_gp; a.winner = winner; a.cell = cell; a.off0 = off0;
    a.gc = gc; a.wc = wc; a.out_last = out_last; a.hr = hr; a.x_left = x_left; a.ld_xl = ld_xl; a.x_right = x_right; a.ld_xr = ld_xr;
    a.pts = pts; a.dim = dim; a.wpos = w_pos; a.bpos = b_pos; a.w0 = w0; a.w1 = w1; a.ws = ws;
    a.M = (int)M; a.tiles_per_wg = trunk_bwd_tiles_per_wg(M); a.dx = dx;
    a.slab = static_cast<float *>(workspace); a.slab_floats = kTrunkSlabFloats;
    const dim3 grid((unsigned)trunk_bwd_slabs(M));
    hipStream_t s = as_stream(stream);
    if (first) hipLaunchKernelGGL((trunk_block_bwd_kernel<true, false>), grid, dim3(256), 0, s, a);
    else if (last) hipLaunchKernelGGL((trunk_block_bwd_kernel<false, true>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((trunk_block_bwd_kernel<false, false>), grid, dim3(256), 0, s, a);
    note_kernel(first ? "trunk_block_bwd_kernel<true,false>" : (last ? "trunk_block_bwd_kernel<false,true>" : "trunk_block_bwd_kernel<false,false>"));
    return check_launch("trunk_block_bwd");
}

// Sum the per-workgroup slabs of t2h_trunk_block_bwd (ascending slab order) into the parameter gradients.
// dwx / dbx: fc_c's (last block) or fc_pos's (first block) weight / bias gradient, or NULL for a middle block.
T2H_API int t2h_trunk_block_reduce(const void *workspace, int64_t M, int first, int last, float *dw0, float *db0, float *dw1,
                                   float *db1, float *dws, float *dwx, float *dbx, int accumulate, t2h_stream_t stream) {
    if (!workspace || !dw0 || !db0 || !dw1 || !db1 || !dws) return fail(T2H_ERR_ARG, "trunk_block_reduce: null pointer");
    if ((first || last) && (!dwx || !dbx)) return fail(T2H_ERR_ARG, "trunk_block_reduce: missing fc_pos / fc_c gradient");
    if (M < 1) return fail(T2H_ERR_ARG, "trunk_block_reduce: bad shape");
    const float *base = static_cast<const float *>(workspace);
    ReduceArgs r{};
    int n = 0;
    auto add = [&](int off, float *dst, int len) { r.seg[n].src = base + off; r.seg[n].dst = dst; r.seg[n].n = len; r.seg[n].accumulate = accumulate; ++n; };
    add(SL_W0, dw0, 2048); add(SL_WS, dws, 2048); add(SL_W1, dw1, 1024); add(SL_B0, db0, 32); add(SL_B1, db1, 32);
    if (last) { add(SL_X, dwx, 1024); add(SL_X + 1024, dbx, 32); }
    if (first) { add(SL_X, dwx, 192); add(SL_X + 192, dbx, 64); }
    r.n_segs = n; r.n_slabs = trunk_bwd_slabs(M); r.stride = kTrunkSlabFloats;
    hipLaunchKernelGGL(reduce_segments_kernel, dim3(64, n), dim3(256), 0, as_stream(stream), r);
    return check_launch("trunk_block_reduce");
}
