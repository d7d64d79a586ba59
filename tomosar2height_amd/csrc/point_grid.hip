// Point <-> grid kernels on the cell-sorted point order (see include/t2h.h).
//
//   pool_max        scatter_max + gather          pointnet.py:92-99
//   segmean         scatter_mean into a plane     pointnet.py:101-111; alto.py:76-88,187-197
//   sample          grid_sample bilinear/border   alto.py:90-95,199-205
//
// All of them are HBM-bound row streaming: a "group" of G = 2^lg lanes owns one cell / point / pixel and
// each lane owns VEC consecutive channels of the row, so a row is one coalesced G*VEC*4-byte access and no
// cross-lane traffic is needed.  Segments are contiguous runs of rows thanks to the Morton sort, so the
// reductions are plain sequential loops over neighbouring rows: no atomics, deterministic sums.
#include <float.h>
#include <stdlib.h>

#include "t2h_common.h"

namespace t2h {

constexpr int kThreads = 256;

struct GroupCfg {
    int lg;        // log2 lanes per group
    int span;      // channels covered per pass = (1 << lg) * VEC
};

template <int VEC>
static GroupCfg group_cfg(int C) {
    GroupCfg g;
    g.lg = group_log2(C, VEC);
    g.span = (1 << g.lg) * VEC;
    return g;
}

static inline unsigned grid_for(int64_t groups, int lg) {
    int64_t threads = groups << lg;
    return (unsigned)((threads + kThreads - 1) / kThreads);
}

// ------------------------------------------------------------------------------------------------ pool_max
template <int VEC>
__global__ __launch_bounds__(kThreads) void pool_max_fwd_kernel(const float *__restrict__ feat,
                                                                const int32_t *__restrict__ off0, int64_t ncells, int C,
                                                                int ldf, int ldp, int lg, int wstride,
                                                                float *__restrict__ pooled,
                                                                uint8_t *__restrict__ winner) {
    int64_t t = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    int64_t cellid = t >> lg;
    if (cellid >= ncells) return;
    int s = off0[cellid], e = off0[cellid + 1];
    if (s == e) return;
    int span = VEC << lg;
    for (int c = ((int)t & ((1 << lg) - 1)) * VEC; c < C; c += span) {
        float best[VEC];
        int arg[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) { best[j] = -FLT_MAX; arg[j] = -1; }
        for (int n = s; n < e; n += 4) {
            // four row loads in flight; rows past the segment end re-read the last row (a duplicate never wins the
            // strict '>' so the first-occurrence arg-max is unaffected)
            const int n1 = min(n + 1, e - 1), n2 = min(n + 2, e - 1), n3 = min(n + 3, e - 1);
            Vec<VEC> v0 = Vec<VEC>::load(feat + (size_t)n * ldf + c);
            Vec<VEC> v1 = Vec<VEC>::load(feat + (size_t)n1 * ldf + c);
            Vec<VEC> v2 = Vec<VEC>::load(feat + (size_t)n2 * ldf + c);
            Vec<VEC> v3 = Vec<VEC>::load(feat + (size_t)n3 * ldf + c);
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                if (v0.v[j] > best[j]) { best[j] = v0.v[j]; arg[j] = n; }   // strict >: first point wins ties
                if (v1.v[j] > best[j]) { best[j] = v1.v[j]; arg[j] = n1; }
                if (v2.v[j] > best[j]) { best[j] = v2.v[j]; arg[j] = n2; }
                if (v3.v[j] > best[j]) { best[j] = v3.v[j]; arg[j] = n3; }
            }
        }
        Vec<VEC> o;
#pragma unroll
        for (int j = 0; j < VEC; ++j) o.v[j] = arg[j] < 0 ? 0.0f : best[j];
        for (int n = s; n < e; ++n) {
            o.store(pooled + (size_t)n * ldp + c);
            uint8_t bits = 0;
#pragma unroll
            for (int j = 0; j < VEC; ++j) bits |= (uint8_t)((arg[j] == n) << j);
            winner[(size_t)n * wstride + c / VEC] = bits;
        }
    }
}

// scatter_type='mean' (pointnet.py:55-56, 92-99): every row receives the mean of its cell's rows -- scatter_mean (a sum
// in point order, then one division by the count) + gather.  The operator is its own adjoint, so the backward is the
// same kernel on the gradient (accumulate: out += instead of out =).
template <int VEC>
__global__ __launch_bounds__(kThreads) void pool_mean_kernel(const float *__restrict__ feat, const int32_t *__restrict__ off0,
                                                             int64_t ncells, int C, int ldf, int ldo, int lg, int accumulate,
                                                             float *__restrict__ out) {
    int64_t t = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    int64_t cellid = t >> lg;
    if (cellid >= ncells) return;
    int s = off0[cellid], e = off0[cellid + 1];
    if (s == e) return;
    const float cnt = (float)(e - s);
    int span = VEC << lg;
    for (int c = ((int)t & ((1 << lg) - 1)) * VEC; c < C; c += span) {
        float sum[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) sum[j] = 0.0f;
        int n = s;
        for (; n + 3 < e; n += 4) {            // four row loads in flight, added in row order
            Vec<VEC> v0 = Vec<VEC>::load(feat + (size_t)n * ldf + c);
            Vec<VEC> v1 = Vec<VEC>::load(feat + (size_t)(n + 1) * ldf + c);
            Vec<VEC> v2 = Vec<VEC>::load(feat + (size_t)(n + 2) * ldf + c);
            Vec<VEC> v3 = Vec<VEC>::load(feat + (size_t)(n + 3) * ldf + c);
#pragma unroll
            for (int j = 0; j < VEC; ++j)
                sum[j] = __fadd_rn(__fadd_rn(__fadd_rn(__fadd_rn(sum[j], v0.v[j]), v1.v[j]), v2.v[j]), v3.v[j]);
        }
        for (; n < e; ++n) {
            Vec<VEC> v = Vec<VEC>::load(feat + (size_t)n * ldf + c);
#pragma unroll
            for (int j = 0; j < VEC; ++j) sum[j] = __fadd_rn(sum[j], v.v[j]);
        }
        Vec<VEC> o;
#pragma unroll
        for (int j = 0; j < VEC; ++j) o.v[j] = __fdiv_rn(sum[j], cnt);
        for (n = s; n < e; ++n) {
            if (accumulate) {
                Vec<VEC> a = Vec<VEC>::load(out + (size_t)n * ldo + c);
#pragma unroll
                for (int j = 0; j < VEC; ++j) a.v[j] = __fadd_rn(a.v[j], o.v[j]);
                a.store(out + (size_t)n * ldo + c);
            } else {
                o.store(out + (size_t)n * ldo + c);
            }
        }
    }
}

template <int VEC>
__global__ __launch_bounds__(kThreads) void pool_max_bwd_kernel(const float *__restrict__ gpooled,
                                                                const uint8_t *__restrict__ winner,
                                                                const int32_t *__restrict__ off0, int64_t ncells, int C,
                                                                int ldg, int ldo, int lg, int wstride, int accumulate,
                                                                float *__restrict__ gfeat) {
    int64_t t = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    int64_t cellid = t >> lg;
    if (cellid >= ncells) return;
    int s = off0[cellid], e = off0[cellid + 1];
    if (s == e) return;
    int span = VEC << lg;
    for (int c = ((int)t & ((1 << lg) - 1)) * VEC; c < C; c += span) {
        float sum[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) sum[j] = 0.0f;
        for (int n = s; n < e; n += 4) {   // four row loads in flight, rows past the end contribute 0
            const int n1 = min(n + 1, e - 1), n2 = min(n + 2, e - 1), n3 = min(n + 3, e - 1);
            Vec<VEC> g0 = Vec<VEC>::load(gpooled + (size_t)n * ldg + c);
            Vec<VEC> g1 = Vec<VEC>::load(gpooled + (size_t)n1 * ldg + c);
            Vec<VEC> g2 = Vec<VEC>::load(gpooled + (size_t)n2 * ldg + c);
            Vec<VEC> g3 = Vec<VEC>::load(gpooled + (size_t)n3 * ldg + c);
            const float k1 = n + 1 < e ? 1.0f : 0.0f, k2 = n + 2 < e ? 1.0f : 0.0f, k3 = n + 3 < e ? 1.0f : 0.0f;
#pragma unroll
            for (int j = 0; j < VEC; ++j)
                sum[j] = (((sum[j] + g0.v[j]) + k1 * g1.v[j]) + k2 * g2.v[j]) + k3 * g3.v[j];
        }
        for (int n = s; n < e; ++n) {
            uint8_t bits = winner[(size_t)n * wstride + c / VEC];
            Vec<VEC> o;
            if (accumulate) o = Vec<VEC>::load(gfeat + (size_t)n * ldo + c);
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                float r = ((bits >> j) & 1) ? sum[j] : 0.0f;
                o.v[j] = accumulate ? o.v[j] + r : r;
            }
            o.store(gfeat + (size_t)n * ldo + c);
        }
    }
}

// ---- row-parallel pooling (per-row cell ids available): balanced over rows instead of cells ------------------------
// A workgroup owns kPoolRows consecutive sorted rows (not cell aligned).  Rows go to LDS once (coalesced); the first row
// of every cell segment inside the chunk reduces its segment from LDS and leaves the result in that row's LDS slot;
// all rows then read their segment's slot.  A cell that crosses the chunk border (at most the first and the last one) has
// its outside rows reduced from global memory by the whole workgroup (strided over the lane groups, partials combined in
// a fixed order), so every chunk a cell touches computes the same whole-cell result.  Max: (value, first index) pairs
// -- larger value wins, equal values keep the smaller row index (= the first-occurrence arg-max of the cell-parallel
// kernel above); NaN and -FLT_MAX never win, like there.  Measured at the bench shape (N = 131072, C = 32, warm): backward
// 14 us against 26 us for the cell-parallel kernel (default in the trunk); forward 24 against 13 us -- the per-row
// cell -> offset lookups sit on its critical path -- so the forward stays on the cell-parallel kernel.
constexpr int kPoolRows = 128;

struct BestArg { float4 v; int4 a; };
__device__ inline float pool_clean(float x) { return x != x ? -FLT_MAX : x; }
__device__ inline void best_strict(BestArg &b, const float4 &v, int n) {      // rows visited in ascending order
    if (v.x > b.v.x) { b.v.x = v.x; b.a.x = n; }
    if (v.y > b.v.y) { b.v.y = v.y; b.a.y = n; }
    if (v.z > b.v.z) { b.v.z = v.z; b.a.z = n; }
    if (v.w > b.v.w) { b.v.w = v.w; b.a.w = n; }
}
__device__ inline void best_merge(float &bv, int &ba, float v, int a) {        // order-free: ties keep the smaller index
    if (a >= 0 && (v > bv || (v == bv && (ba < 0 || a < ba)))) { bv = v; ba = a; }
}

template <int LG>
__global__ __launch_bounds__(256) void pool_rows_fwd_kernel(const float *__restrict__ feat, int ldf,
                                                            const int32_t *__restrict__ cell,
                                                            const int32_t *__restrict__ off0, int nrows, int C,
                                                            float *__restrict__ pooled, int ldp,
                                                            uint8_t *__restrict__ winner, int wstride) {
    constexpr int G = 1 << LG, NG = 256 / G, R = NG > kPoolRows ? NG : kPoolRows, PASSES = R / NG;
    __shared__ float4 val[R * G];
    __shared__ short4 argv[R * G];       // winner row of the segment as chunk-local index; -1 none, -2 outside the chunk
    __shared__ float4 pval[NG * G];
    __shared__ int4 parg[NG * G];
    __shared__ float4 oval[2 * G];        // outside parts of the first / last cell of the chunk
    __shared__ int4 oarg[2 * G];
    __shared__ int bounds[4];
    const int tid = threadIdx.x, lane = tid & (G - 1), grp = tid >> LG;
    const int r0 = blockIdx.x * R, r1 = min(r0 + R, nrows);
    const bool cv = lane * 4 < C;
    int segs[PASSES], sege[PASSES];
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
        const int row = r0 + p * NG + grp;
        float4 v = make_float4(-FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX);
        segs[p] = sege[p] = -1;
        if (row < r1) {
            const int cid = cell[row];
            segs[p] = off0[cid]; sege[p] = off0[cid + 1];
            if (cv) {
                v = *reinterpret_cast<const float4 *>(feat + (size_t)row * ldf + lane * 4);
                v.x = pool_clean(v.x); v.y = pool_clean(v.y); v.z = pool_clean(v.z); v.w = pool_clean(v.w);
            }
        }
        val[(p * NG + grp) * G + lane] = v;
    }
    if (tid == 0) {
        const int ch = cell[r0], ct = cell[r1 - 1];
        bounds[0] = off0[ch]; bounds[1] = off0[ch + 1]; bounds[2] = off0[ct]; bounds[3] = off0[ct + 1];
    }
    __syncthreads();
    // outside rows of the border cells: [bounds[0], r0) before the chunk, [r1, bounds[3]) after it
#pragma unroll
    for (int side = 0; side < 2; ++side) {
        const int lo = side == 0 ? bounds[0] : r1, hi = side == 0 ? r0 : bounds[3];
        if (hi <= lo) continue;                      // uniform over the workgroup
        BestArg b;
        b.v = make_float4(-FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX); b.a = make_int4(-1, -1, -1, -1);
        if (cv)
            for (int n = lo + grp; n < hi; n += NG) {
                float4 v = *reinterpret_cast<const float4 *>(feat + (size_t)n * ldf + lane * 4);
                v.x = pool_clean(v.x); v.y = pool_clean(v.y); v.z = pool_clean(v.z); v.w = pool_clean(v.w);
                best_strict(b, v, n);
            }
        pval[grp * G + lane] = b.v; parg[grp * G + lane] = b.a;
        __syncthreads();
        if (grp == 0) {
            BestArg t; t.v = pval[lane]; t.a = parg[lane];
            for (int g = 1; g < NG; ++g) {
                const float4 v = pval[g * G + lane]; const int4 a = parg[g * G + lane];
                best_merge(t.v.x, t.a.x, v.x, a.x); best_merge(t.v.y, t.a.y, v.y, a.y);
                best_merge(t.v.z, t.a.z, v.z, a.z); best_merge(t.v.w, t.a.w, v.w, a.w);
            }
            oval[side * G + lane] = t.v; oarg[side * G + lane] = t.a;
        }
        __syncthreads();
    }
    // segment heads reduce their segment from LDS (rows in ascending order: strict '>' keeps the first maximum)
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
        const int row = r0 + p * NG + grp;
        if (row >= r1 || row != max(segs[p], r0)) continue;
        BestArg b;
        b.v = make_float4(-FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX); b.a = make_int4(-1, -1, -1, -1);
        if (segs[p] < r0) {                                                  // earlier rows: they win ties
            const int4 a = oarg[lane];
            b.v = oval[lane]; b.a = make_int4(a.x < 0 ? -1 : -2, a.y < 0 ? -1 : -2, a.z < 0 ? -1 : -2, a.w < 0 ? -1 : -2);
        }
        const int end = min(sege[p], r1);
        for (int n = row; n < end; ++n) best_strict(b, val[(n - r0) * G + lane], n - r0);
        if (sege[p] > r1) {                                                  // later rows: only a larger value wins
            const float4 v = oval[G + lane];
            if (v.x > b.v.x) { b.v.x = v.x; b.a.x = -2; }
            if (v.y > b.v.y) { b.v.y = v.y; b.a.y = -2; }
            if (v.z > b.v.z) { b.v.z = v.z; b.a.z = -2; }
            if (v.w > b.v.w) { b.v.w = v.w; b.a.w = -2; }
        }
        val[(row - r0) * G + lane] = b.v;
        argv[(row - r0) * G + lane] = make_short4((short)b.a.x, (short)b.a.y, (short)b.a.z, (short)b.a.w);
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
        const int row = r0 + p * NG + grp;
        if (row >= r1 || !cv) continue;
        const int slot = max(segs[p], r0) - r0;
        const float4 v = val[slot * G + lane]; const short4 a = argv[slot * G + lane];
        const int me = row - r0;
        float4 o = make_float4(a.x == -1 ? 0.f : v.x, a.y == -1 ? 0.f : v.y, a.z == -1 ? 0.f : v.z, a.w == -1 ? 0.f : v.w);
        *reinterpret_cast<float4 *>(pooled + (size_t)row * ldp + lane * 4) = o;
        winner[(size_t)row * wstride + lane] = (uint8_t)((a.x == me) | ((a.y == me) << 1) | ((a.z == me) << 2) | ((a.w == me) << 3));
    }
}

// gfeat[n] = [gfeat[n] +] (winner bits of n) ? sum over the cell of gpooled : 0; same structure, sums instead of maxima
template <int LG>
__global__ __launch_bounds__(256) void pool_rows_bwd_kernel(const float *__restrict__ gpooled, int ldg,
                                                            const uint8_t *__restrict__ winner, int wstride,
                                                            const int32_t *__restrict__ cell,
                                                            const int32_t *__restrict__ off0, int nrows, int C,
                                                            int accumulate, float *__restrict__ gfeat, int ldo) {
    constexpr int G = 1 << LG, NG = 256 / G, R = NG > kPoolRows ? NG : kPoolRows, PASSES = R / NG;
    __shared__ float4 val[R * G];
    __shared__ float4 pval[NG * G];
    __shared__ float4 oval[2 * G];
    __shared__ int bounds[4];
    const int tid = threadIdx.x, lane = tid & (G - 1), grp = tid >> LG;
    const int r0 = blockIdx.x * R, r1 = min(r0 + R, nrows);
    const bool cv = lane * 4 < C;
    int segs[PASSES], sege[PASSES];
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
        const int row = r0 + p * NG + grp;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        segs[p] = sege[p] = -1;
        if (row < r1) {
            const int cid = cell[row];
            segs[p] = off0[cid]; sege[p] = off0[cid + 1];
            if (cv) v = *reinterpret_cast<const float4 *>(gpooled + (size_t)row * ldg + lane * 4);
        }
        val[(p * NG + grp) * G + lane] = v;
    }
    if (tid == 0) {
        const int ch = cell[r0], ct = cell[r1 - 1];
        bounds[0] = off0[ch]; bounds[1] = off0[ch + 1]; bounds[2] = off0[ct]; bounds[3] = off0[ct + 1];
    }
    __syncthreads();
#pragma unroll
    for (int side = 0; side < 2; ++side) {
        const int lo = side == 0 ? bounds[0] : r1, hi = side == 0 ? r0 : bounds[3];
        if (hi <= lo) continue;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (cv)
            for (int n = lo + grp; n < hi; n += NG) {
                const float4 v = *reinterpret_cast<const float4 *>(gpooled + (size_t)n * ldg + lane * 4);
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
        pval[grp * G + lane] = acc;
        __syncthreads();
        if (grp == 0) {
            float4 t = pval[lane];
            for (int g = 1; g < NG; ++g) { const float4 v = pval[g * G + lane]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
            oval[side * G + lane] = t;
        }
        __syncthreads();
    }
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
        const int row = r0 + p * NG + grp;
        if (row >= r1 || row != max(segs[p], r0)) continue;
        float4 acc = segs[p] < r0 ? oval[lane] : make_float4(0.f, 0.f, 0.f, 0.f);
        const int end = min(sege[p], r1);
        for (int n = row; n < end; ++n) { const float4 v = val[(n - r0) * G + lane]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
        if (sege[p] > r1) { const float4 v = oval[G + lane]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
        val[(row - r0) * G + lane] = acc;
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
        const int row = r0 + p * NG + grp;
        if (row >= r1 || !cv) continue;
        const float4 sum = val[(max(segs[p], r0) - r0) * G + lane];
        const uint8_t bits = winner[(size_t)row * wstride + lane];
        float4 o = make_float4((bits & 1) ? sum.x : 0.f, (bits & 2) ? sum.y : 0.f, (bits & 4) ? sum.z : 0.f, (bits & 8) ? sum.w : 0.f);
        float4 *dst = reinterpret_cast<float4 *>(gfeat + (size_t)row * ldo + lane * 4);
        if (accumulate) { const float4 old = *dst; o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w; }
        *dst = o;
    }
}

// ------------------------------------------------------------------------------------------------- segmean
// One group per level-k cell, cells enumerated in Morton order (neighbouring groups read neighbouring rows).
// MEAN = false: the plain per-cell SUM (t2h_segsum_fwd)
template <int VEC, bool MEAN = true>
__global__ __launch_bounds__(kThreads) void segmean_fwd_kernel(const float *__restrict__ feat,
                                                               const int32_t *__restrict__ off0, int B, int nbits,
                                                               int level, int C, int lg, float *__restrict__ plane,
                                                               int ld_out) {
    int64_t t = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    int64_t gid = t >> lg;
    const int rbits = nbits - level;
    const int64_t cells_per_tile = (int64_t)1 << (2 * rbits);
    if (gid >= (int64_t)B * cells_per_tile) return;
    int b = (int)(gid >> (2 * rbits));
    uint32_t mk = (uint32_t)(gid & (cells_per_tile - 1));
    size_t obase = ((size_t)b << (2 * nbits)) + ((size_t)mk << (2 * level));
    int s = off0[obase], e = off0[obase + ((size_t)1 << (2 * level))];
    int cx = (int)compact1by1(mk), cy = (int)compact1by1(mk >> 1);
    int r = 1 << rbits;
    float *orow = plane + (((size_t)b * r + cy) * r + cx) * ld_out;     // ld_out >= C: a column block of a wider matrix
    float inv_den = (float)(e - s > 0 ? e - s : 1);
    int span = VEC << lg;
    for (int c = ((int)t & ((1 << lg) - 1)) * VEC; c < C; c += span) {
        float sum[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) sum[j] = 0.0f;
        int n = s;
        for (; n + 4 <= e; n += 4) {   // 4 independent row loads in flight
            Vec<VEC> a0 = Vec<VEC>::load(feat + (size_t)(n + 0) * C + c);
            Vec<VEC> a1 = Vec<VEC>::load(feat + (size_t)(n + 1) * C + c);
            Vec<VEC> a2 = Vec<VEC>::load(feat + (size_t)(n + 2) * C + c);
            Vec<VEC> a3 = Vec<VEC>::load(feat + (size_t)(n + 3) * C + c);
#pragma unroll
            for (int j = 0; j < VEC; ++j) sum[j] = (((sum[j] + a0.v[j]) + a1.v[j]) + a2.v[j]) + a3.v[j];
        }
        for (; n < e; ++n) {
            Vec<VEC> a = Vec<VEC>::load(feat + (size_t)n * C + c);
#pragma unroll
            for (int j = 0; j < VEC; ++j) sum[j] += a.v[j];
        }
        Vec<VEC> o;
#pragma unroll
        for (int j = 0; j < VEC; ++j) o.v[j] = MEAN ? __fdiv_rn(sum[j], inv_den) : sum[j];
        o.store(orow + c);
    }
}

// Gradient of per-cell sums taken at several (<= 8) resolutions of the same rows (t2h_segsum_bwd_multi): one group per point,
// gfeat[n] = (mask[n] > 0 ?) sum_l gplane_l[cell_l(n)] (+ addend[n]); planes in the order given (a fixed summation order).
constexpr int kMaxMultiPlanes = 8;
struct MultiPlanes { const float *g[kMaxMultiPlanes]; int level[kMaxMultiPlanes]; int ld[kMaxMultiPlanes]; int n; };
template <int VEC>
__global__ __launch_bounds__(kThreads) void segsum_bwd_multi_kernel(MultiPlanes mp, const int32_t *__restrict__ cell,
                                                                    int64_t npts, int nbits, int C, int lg,
                                                                    const float *__restrict__ mask,
                                                                    const float *__restrict__ addend,
                                                                    float *__restrict__ gfeat) {
    int64_t t = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    int64_t n = t >> lg;
    if (n >= npts) return;
    const uint32_t code = (uint32_t)cell[n];
    const uint32_t b = code >> (2 * nbits), m = code & ((1u << (2 * nbits)) - 1u);
    const uint32_t x0 = compact1by1(m), y0 = compact1by1(m >> 1);             // finest-level cell coordinates
    const float *rows[kMaxMultiPlanes];
#pragma unroll
    for (int q = 0; q < kMaxMultiPlanes; ++q) {
        rows[q] = nullptr;
        if (q < mp.n) {
            const int l = mp.level[q], r = 1 << (nbits - l);
            rows[q] = mp.g[q] + (((size_t)b * r + (y0 >> l)) * r + (x0 >> l)) * mp.ld[q];
        }
    }
    const int span = VEC << lg;
    for (int c = ((int)t & ((1 << lg) - 1)) * VEC; c < C; c += span) {
        Vec<VEC> acc = Vec<VEC>::load(rows[0] + c);
#pragma unroll
        for (int q = 1; q < kMaxMultiPlanes; ++q)
            if (q < mp.n) {
                Vec<VEC> g = Vec<VEC>::load(rows[q] + c);
#pragma unroll
                for (int j = 0; j < VEC; ++j) acc.v[j] = __fadd_rn(acc.v[j], g.v[j]);
            }
        if (mask) {
            Vec<VEC> h = Vec<VEC>::load(mask + (size_t)n * C + c);
#pragma unroll
            for (int j = 0; j < VEC; ++j) acc.v[j] = h.v[j] > 0.0f ? acc.v[j] : 0.0f;
        }
        if (addend) {
            Vec<VEC> o = Vec<VEC>::load(addend + (size_t)n * C + c);
#pragma unroll
            for (int j = 0; j < VEC; ++j) acc.v[j] = __fadd_rn(o.v[j], acc.v[j]);
        }
        acc.store(gfeat + (size_t)n * C + c);
    }
}

// coarse[b, y, x, :] = fine[b, 2y, 2x, :] + fine[b, 2y, 2x+1, :] + fine[b, 2y+1, 2x, :] + fine[b, 2y+1, 2x+1, :]  (NHWC, float4 lanes):
// the per-cell sums of resolution r / 2 from those of resolution r (a cell is the union of its four children)
__global__ __launch_bounds__(kThreads) void plane_sumpool2x2_kernel(const float *__restrict__ fine, int64_t total4, int rc,
                                                                   int C4, int ld_in, int ld_out, float *__restrict__ coarse) {
    int64_t t = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (t >= total4) return;
    const int c4 = (int)(t % C4);
    const int64_t pix = t / C4;
    const int x = (int)(pix % rc), y = (int)((pix / rc) % rc);
    const int64_t b = pix / ((int64_t)rc * rc);
    const int rf = 2 * rc;
    const float *f = fine + ((b * rf + 2 * y) * rf + 2 * x) * (int64_t)ld_in + 4 * c4;
    const float4 a = *reinterpret_cast<const float4 *>(f), bq = *reinterpret_cast<const float4 *>(f + ld_in);
    const float4 cq = *reinterpret_cast<const float4 *>(f + (int64_t)rf * ld_in);
    const float4 d = *reinterpret_cast<const float4 *>(f + (int64_t)rf * ld_in + ld_in);
    float4 o;
    o.x = (a.x + bq.x) + (cq.x + d.x); o.y = (a.y + bq.y) + (cq.y + d.y);
    o.z = (a.z + bq.z) + (cq.z + d.z); o.w = (a.w + bq.w) + (cq.w + d.w);
    *reinterpret_cast<float4 *>(coarse + pix * (int64_t)ld_out + 4 * c4) = o;
}

// One group per point: gfeat[n] = gplane[cell_k(n)] / count.
template <int VEC>
__global__ __launch_bounds__(kThreads) void segmean_bwd_kernel(const float *__restrict__ gplane,
                                                               const int32_t *__restrict__ cell,
                                                               const int32_t *__restrict__ off0, int64_t npts, int nbits,
                                                               int level, int C, int lg, const float *__restrict__ addend,
                                                               float *__restrict__ gfeat) {
    int64_t t = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    int64_t n = t >> lg;
    if (n >= npts) return;
    uint32_t code = (uint32_t)cell[n];
    uint32_t b = code >> (2 * nbits);
    uint32_t m = code & ((1u << (2 * nbits)) - 1u);
    uint32_t mk = m >> (2 * level);
    size_t obase = ((size_t)b << (2 * nbits)) + ((size_t)mk << (2 * level));
    int cnt = off0[obase + ((size_t)1 << (2 * level))] - off0[obase];
    int rbits = nbits - level, r = 1 << rbits;
    int cx = (int)compact1by1(mk), cy = (int)compact1by1(mk >> 1);
    const float *grow = gplane + (((size_t)b * r + cy) * r + cx) * C;
    float den = (float)(cnt > 0 ? cnt : 1);
    int span = VEC << lg;
    for (int c = ((int)t & ((1 << lg) - 1)) * VEC; c < C; c += span) {
        Vec<VEC> g = Vec<VEC>::load(grow + c);
#pragma unroll
        for (int j = 0; j < VEC; ++j) g.v[j] = __fdiv_rn(g.v[j], den);
        if (addend) {           // the other gradient of a point-feature tensor with two consumers: summed here, not by a pass of its own
            Vec<VEC> o = Vec<VEC>::load(addend + (size_t)n * C + c);
#pragma unroll
            for (int j = 0; j < VEC; ++j) g.v[j] = __fadd_rn(o.v[j], g.v[j]);
        }
        g.store(gfeat + (size_t)n * C + c);
    }
}

// -------------------------------------------------------------------------------------------------- sample
struct Taps {
    int x0, y0;
    float wx0, wx1, wy0, wy1;   // west/east, north/south factors
};
__device__ inline Taps make_taps(float x01, float y01, int r) {
    float ix = unnormalize_clip(x01, r), iy = unnormalize_clip(y01, r);
    float fx = floorf(ix), fy = floorf(iy);
    Taps t;
    t.x0 = (int)fx; t.y0 = (int)fy;
    t.wx0 = __fsub_rn(fx + 1.0f, ix); t.wx1 = __fsub_rn(ix, fx);
    t.wy0 = __fsub_rn(fy + 1.0f, iy); t.wy1 = __fsub_rn(iy, fy);
    return t;
}

// RELU: out = max(sample, 0) -- the hidden layer of fc_comm when its first Linear was applied on the grid
// (t2h_sample_fwd_relu)
// `bits` (RELU, VEC = 4, C % 256 == 0 only; may be null): the sign pattern of the result packed 1 bit per element -- for
// row n and 256-channel chunk q four 64-bit words at [q][n][4] (chunk-major: the rows of a chunk are consecutive 32-byte
// records, so the per-row stores of a wave fill whole cache lines and 64 rows load as one coalesced 2 KB run), bit l of word
// j <=> out[n][256 q + 4 l + j] > 0 -- which is all the backward needs of the hidden activations (32 B instead of 1 KB per
// row and chunk; t2h_sample_bwd_from_sums reads it)
template <int VEC, bool RELU = false>
__global__ __launch_bounds__(kThreads) void sample_fwd_kernel(const float *__restrict__ plane,
                                                              const float *__restrict__ pts, int dim, int64_t npts,
                                                              int N, int r, int C, int lg, float *__restrict__ out,
                                                              unsigned long long *__restrict__ bits = nullptr) {
    int64_t t = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    int64_t n = t >> lg;
    if (n >= npts) return;
    int b = N > 0 ? (int)(n / N) : __float_as_int(pts[n * dim + dim - 1]);    // ragged batch: tile id in the row's last float
    Taps tp = make_taps(pts[n * dim + 0], pts[n * dim + 1], r);
    // ATen: nw = (x1-ix)*(y1-iy), ne = (ix-x0)*(y1-iy), sw = (x1-ix)*(iy-y0), se = (ix-x0)*(iy-y0)
    float nw = __fmul_rn(tp.wx0, tp.wy0), ne = __fmul_rn(tp.wx1, tp.wy0);
    float sw = __fmul_rn(tp.wx0, tp.wy1), se = __fmul_rn(tp.wx1, tp.wy1);
    bool x1ok = tp.x0 + 1 < r, y1ok = tp.y0 + 1 < r;   // x0,y0 are in range after the border clip
    const float *base = plane + (size_t)b * r * r * C;
    const float *p00 = base + ((size_t)tp.y0 * r + tp.x0) * C;
    const float *p01 = p00 + C;
    const float *p10 = p00 + (size_t)r * C;
    const float *p11 = p10 + C;
    int span = VEC << lg;
    for (int c = ((int)t & ((1 << lg) - 1)) * VEC; c < C; c += span) {
        Vec<VEC> acc;
        Vec<VEC> v = Vec<VEC>::load(p00 + c);
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc.v[j] = __fmul_rn(v.v[j], nw);
        if (x1ok) {
            v = Vec<VEC>::load(p01 + c);
#pragma unroll
            for (int j = 0; j < VEC; ++j) acc.v[j] = __fadd_rn(acc.v[j], __fmul_rn(v.v[j], ne));
        }
        if (y1ok) {
            v = Vec<VEC>::load(p10 + c);
#pragma unroll
            for (int j = 0; j < VEC; ++j) acc.v[j] = __fadd_rn(acc.v[j], __fmul_rn(v.v[j], sw));
        }
        if (x1ok && y1ok) {
            v = Vec<VEC>::load(p11 + c);
#pragma unroll
            for (int j = 0; j < VEC; ++j) acc.v[j] = __fadd_rn(acc.v[j], __fmul_rn(v.v[j], se));
        }
        if (RELU) {
#pragma unroll
            for (int j = 0; j < VEC; ++j) acc.v[j] = fmaxf(acc.v[j], 0.0f);
            if (VEC == 4 && bits) {             // lg == 6 here: the wave is one (row, 256-channel chunk), lane l = channels 4 l ..
                unsigned long long w[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) w[j] = __ballot(acc.v[j] > 0.0f);
                if ((threadIdx.x & 63) == 0) {
                    unsigned long long *dst = bits + ((size_t)(c >> 8) * npts + n) * 4;
                    dst[0] = w[0]; dst[1] = w[1]; dst[2] = w[2]; dst[3] = w[3];
                }
            }
        }
        acc.store(out + (size_t)n * C + c);
    }
}

// wave-uniform lane reads (v_readlane: the index must be the same in every lane) and a lane-mask select
__device__ inline float keep_if(unsigned long long lanes, float v) {      // lane l: v if bit l of the wave-uniform mask, else 0
    float o;
    asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(o) : "v"(v), "s"(lanes));
    return o;
}
__device__ inline float readlane_f(float v, int i) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), i)); }
__device__ inline unsigned long long readlane_u64(unsigned long long v, int i) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, i);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), i);
    return ((unsigned long long)hi << 32) | lo;
}

// Hidden activations that never leave the chip (t2h_sample_relu_cellsums): for a coarse sampling level (many points per cell)
// one WAVE owns (cell of the sampling level, 256-channel chunk).  It stages the 3 x 3 pixel neighbourhood of the cell -- every
// tap of every point of the cell lies in it -- in LDS once (9 KB), then walks the cell's rows child by child (children = the
// cells of the finest resolution the sums are needed at, contiguous row runs in Morton order): per row the taps, four LDS
// reads, the same multiply-add chain as sample_fwd_kernel, ReLU, four ballots for the packed sign bits, and the running
// per-child sum in registers, written when the child ends (empty children write zeros, so nothing is memset).  Out go the
// per-cell sums (half the size of h at two points per finest cell) and 1 bit per element; h [N, C] itself is never written
// or re-read.  Same bits as t2h_sample_fwd_relu + t2h_segsum_fwd: identical operation order per element and per sum.
__global__ __launch_bounds__(256) void sample_relu_cellsums_kernel(const float *__restrict__ plane,
                                                                 const float *__restrict__ pts, int dim,
                                                                 const int32_t *__restrict__ off0, int nbits, int level,
                                                                 int sum_level, int C, float *__restrict__ sums, int ld_sums,
                                                                 unsigned long long *__restrict__ bits, int npts_m1,
                                                                 float *__restrict__ sums2, int ld_sums2) {
    extern __shared__ float nb_lds[];                                   // [waves][9][256]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int chunk = blockIdx.y * 4 + wave;
    if (chunk * 256 >= C) return;                                       // (no workgroup barrier below)
    float *T = nb_lds + wave * (9 * 256);
    const int rbits = nbits - level, r = 1 << rbits;
    const int64_t cellrow = blockIdx.x;
    const int b = (int)(cellrow >> (2 * rbits));
    const uint32_t mk = (uint32_t)(cellrow & (((int64_t)1 << (2 * rbits)) - 1));
    const int cx = (int)compact1by1(mk), cy = (int)compact1by1(mk >> 1);
    const int c0 = chunk * 256 + lane * 4;
#pragma unroll
    for (int sl = 0; sl < 9; ++sl) {
        const int py = cy - 1 + sl / 3, px = cx - 1 + sl % 3;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if ((unsigned)py < (unsigned)r && (unsigned)px < (unsigned)r)
            v = *reinterpret_cast<const float4 *>(plane + (((size_t)b * r + py) * r + px) * C + c0);
        *reinterpret_cast<float4 *>(T + sl * 256 + lane * 4) = v;
    }
    const int d = level - sum_level, nchild = 1 << (2 * d), rs = 1 << (nbits - sum_level);
    const size_t obase = ((size_t)b << (2 * nbits)) + ((size_t)mk << (2 * level));
    const int cpc = C >> 8;                                             // 256-channel chunks per row (bit words)
    // the taps of 64 consecutive rows are computed lane-parallel (lane l = row nb0 + l) and handed to the row loop by
    // readlane: ~40 instructions once per 64 rows instead of once per row and lane.  A tap outside the plane gets weight 0 and
    // reads the staged zero (the sample kernel skips it: the same value)
    int nb0 = 0, slot_l = 0;
    float nw_l = 0.f, ne_l = 0.f, sw_l = 0.f, se_l = 0.f;
    bool have = false;
    float4 pe = make_float4(0.f, 0.f, 0.f, 0.f), p01 = pe;             // pooled-output state of the current quad of children
    // blockIdx.z: this workgroup's share of the children (dense cells would otherwise set the kernel's duration: one wave
    // walking 500+ rows while most of the chip has finished); the children's sums are independent, so nothing is reduced
    const int per_group = nchild / (int)gridDim.z, child_lo = (int)blockIdx.z * per_group, child_hi = child_lo + per_group;
    for (int cb = child_lo; cb < child_hi; cb += 64) {                  // children in batches of 64: their row boundaries
        const int ci = cb + lane;
        const int bnd_lo = ci <= child_hi ? off0[obase + ((size_t)ci << (2 * sum_level))] : 0;
        const int bnd_hi = ci + 1 <= child_hi ? off0[obase + ((size_t)(ci + 1) << (2 * sum_level))] : 0;
        const int nc = min(64, child_hi - cb);
        for (int c = 0; c < nc; ++c) {
            const int s = __builtin_amdgcn_readlane(bnd_lo, c), e = __builtin_amdgcn_readlane(bnd_hi, c);
            float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
            int n = s;
            while (n < e) {
                if (!have || n >= nb0 + 64) {
                    nb0 = n; have = true;
                    const int nn = min(n + lane, npts_m1);                // rows past this cell belong to later cells: valid
                    const Taps tp = make_taps(pts[(size_t)nn * dim + 0], pts[(size_t)nn * dim + 1], r);
                    const float fx1 = (float)min(max(r - 1 - tp.x0, 0), 1), fy1 = (float)min(max(r - 1 - tp.y0, 0), 1);
                    nw_l = __fmul_rn(tp.wx0, tp.wy0);
                    ne_l = __fmul_rn(__fmul_rn(tp.wx1, tp.wy0), fx1);      // (no compare -> mask -> select here: see the v2 kernel)
                    sw_l = __fmul_rn(__fmul_rn(tp.wx0, tp.wy1), fy1);
                    se_l = __fmul_rn(__fmul_rn(tp.wx1, tp.wy1), __fmul_rn(fx1, fy1));
                    slot_l = min(max(tp.y0 - cy + 1, 0), 1) * 3 + min(max(tp.x0 - cx + 1, 0), 1);     // 0 or 1 each by construction
                }
                const int i0 = n - nb0, cnt = min(e - n, 64 - i0);        // rows of this child inside the current batch
                auto row = [&](int i, int nrow) -> float4 {
                    const float nw = readlane_f(nw_l, i), ne = readlane_f(ne_l, i), sw = readlane_f(sw_l, i), se = readlane_f(se_l, i);
                    const float *t00 = T + __builtin_amdgcn_readlane(slot_l, i) * 256 + lane * 4;
                    const float4 v00 = *reinterpret_cast<const float4 *>(t00), v01 = *reinterpret_cast<const float4 *>(t00 + 256);
                    const float4 v10 = *reinterpret_cast<const float4 *>(t00 + 3 * 256), v11 = *reinterpret_cast<const float4 *>(t00 + 4 * 256);
                    float4 a;
                    a.x = __fmul_rn(v00.x, nw); a.y = __fmul_rn(v00.y, nw); a.z = __fmul_rn(v00.z, nw); a.w = __fmul_rn(v00.w, nw);
                    a.x = __fadd_rn(a.x, __fmul_rn(v01.x, ne)); a.y = __fadd_rn(a.y, __fmul_rn(v01.y, ne));
                    a.z = __fadd_rn(a.z, __fmul_rn(v01.z, ne)); a.w = __fadd_rn(a.w, __fmul_rn(v01.w, ne));
                    a.x = __fadd_rn(a.x, __fmul_rn(v10.x, sw)); a.y = __fadd_rn(a.y, __fmul_rn(v10.y, sw));
                    a.z = __fadd_rn(a.z, __fmul_rn(v10.z, sw)); a.w = __fadd_rn(a.w, __fmul_rn(v10.w, sw));
                    a.x = __fadd_rn(a.x, __fmul_rn(v11.x, se)); a.y = __fadd_rn(a.y, __fmul_rn(v11.y, se));
                    a.z = __fadd_rn(a.z, __fmul_rn(v11.z, se)); a.w = __fadd_rn(a.w, __fmul_rn(v11.w, se));
                    a.x = fmaxf(a.x, 0.f); a.y = fmaxf(a.y, 0.f); a.z = fmaxf(a.z, 0.f); a.w = fmaxf(a.w, 0.f);
                    const unsigned long long w0 = __ballot(a.x > 0.f), w1 = __ballot(a.y > 0.f);
                    const unsigned long long w2 = __ballot(a.z > 0.f), w3 = __ballot(a.w > 0.f);
                    if (lane == 0 && bits) {                              // (null under no_grad: nothing will read the signs)
                        unsigned long long *dst = bits + ((size_t)chunk * ((size_t)npts_m1 + 1) + nrow) * 4;
                        dst[0] = w0; dst[1] = w1; dst[2] = w2; dst[3] = w3;
                    }
                    return a;
                };
                int j = 0;
                for (; j + 1 < cnt; j += 2) {                             // two rows in flight; summed in row order
                    const float4 a0 = row(i0 + j, n + j), a1 = row(i0 + j + 1, n + j + 1);
                    sum.x += a0.x; sum.y += a0.y; sum.z += a0.z; sum.w += a0.w;
                    sum.x += a1.x; sum.y += a1.y; sum.z += a1.z; sum.w += a1.w;
                }
                if (j < cnt) {
                    const float4 a0 = row(i0 + j, n + j);
                    sum.x += a0.x; sum.y += a0.y; sum.z += a0.z; sum.w += a0.w;
                }
                n += cnt;
            }
            const uint32_t cm = (uint32_t)(cb + c);                      // child's Morton code inside the cell
            const int fx = (cx << d) + (int)compact1by1(cm), fy = (cy << d) + (int)compact1by1(cm >> 1);
            *reinterpret_cast<float4 *>(sums + (((size_t)b * rs + fy) * rs + fx) * ld_sums + c0) = sum;
            if (sums2) {
                // the 2 x 2 pooled sums one level up, formed from the four children while they are in registers (children are
                // visited in Morton order = the order (c0 + c1) + (c2 + c3) of plane_sumpool2x2_kernel: same bits)
                const unsigned k4 = cm & 3u;
                if (k4 == 0u) pe = sum;
                else if (k4 == 1u) p01 = make_float4(pe.x + sum.x, pe.y + sum.y, pe.z + sum.z, pe.w + sum.w);
                else if (k4 == 2u) pe = sum;
                else {
                    const float4 o = make_float4(p01.x + (pe.x + sum.x), p01.y + (pe.y + sum.y), p01.z + (pe.z + sum.z),
                                                 p01.w + (pe.w + sum.w));
                    const int rs2 = rs >> 1;
                    *reinterpret_cast<float4 *>(sums2 + (((size_t)b * rs2 + (fy >> 1)) * rs2 + (fx >> 1)) * ld_sums2 + c0) = o;
                }
            }
        }
    }
}

// ---- r05: the same pass with the neighbourhood SHARED by the workgroup's four waves ----------------------------------------
// sample_relu_cellsums_kernel gives every wave its own 9 KB neighbourhood (one wave per 256-channel chunk): 36 KB of LDS per
// workgroup = 4 waves per SIMD, a 9 KB L2 -> LDS round trip in front of every wave's ~16 rows, ~75 vector instructions per row
// (the compiler pairs (v00.x nw, v01.x ne) for its packed multiplies and shuffles registers to do so).  Here a workgroup is
//   K = 0: ONE sampling cell x ONE 256-channel chunk: the 3 x 3 neighbourhood staged once (9 KB), each wave a quarter of the
//          workgroup's children (blockIdx.z splits them further);
//   K = 1: a 2 x 2 BLOCK of sampling cells, one per wave, sharing its 4 x 4 neighbourhood (16 KB) -- levels with few rows
//          per cell, where a cell has too few children to share out (r = 128: 4 children, 8 rows);
// so 8 workgroups = 32 waves fit a CU (<= 64 VGPRs) and the staging traffic drops 4x / 2.25x.  Per row: the channel pairs
// (x, y), (z, w) go through v_pk_mul_f32 / v_pk_add_f32 with the tap weight broadcast from an SGPR -- the SAME products and
// the same order of additions per element as sample_fwd_kernel (bit-identical), in 24 instead of ~55 vector instructions -- and
// the four ballots of a row are parked in lane (row mod 64) of eight registers (v_writelane) and leave as one coalesced 2 KB
// store per 64 rows instead of two lane-0 stores per row.  The next 64 rows' coordinates are requested while the current 64 run.
typedef float f32x2 __attribute__((ext_vector_type(2)));
// lane `i` (wave-uniform) of the eight registers = the four wave-uniform 64-bit words; the other lanes keep their values.
// (gfx9 encodings read one SGPR per vector instruction: the lane select goes through M0)
__device__ inline void park8(unsigned (&acc)[8], unsigned long long w0, unsigned long long w1, unsigned long long w2,
                             unsigned long long w3, int i) {
    asm("s_mov_b32 m0, %8\n\t"
        "v_writelane_b32 %0, %9, m0\n\tv_writelane_b32 %1, %10, m0\n\tv_writelane_b32 %2, %11, m0\n\tv_writelane_b32 %3, %12, m0\n\t"
        "v_writelane_b32 %4, %13, m0\n\tv_writelane_b32 %5, %14, m0\n\tv_writelane_b32 %6, %15, m0\n\tv_writelane_b32 %7, %16, m0"
        : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7])
        : "s"(__builtin_amdgcn_readfirstlane(i)), "s"((unsigned)w0), "s"((unsigned)(w0 >> 32)), "s"((unsigned)w1),
          "s"((unsigned)(w1 >> 32)), "s"((unsigned)w2), "s"((unsigned)(w2 >> 32)), "s"((unsigned)w3), "s"((unsigned)(w3 >> 32))
        : "m0");
}

template <int K, bool POOL>
__global__ __launch_bounds__(256, 2) void sample_relu_cellsums_v2_kernel(const float *__restrict__ plane,
                                                                        const float *__restrict__ pts, int dim,
                                                                        const int32_t *__restrict__ off0, int nbits, int level,
                                                                        int sum_level, int C, float *__restrict__ sums,
                                                                        int ld_sums, unsigned long long *__restrict__ bits,
                                                                        int npts_m1, float *__restrict__ sums2, int ld_sums2,
                                                                        const int32_t *__restrict__ order, int gz) {
    constexpr int SIDE = 3 + K, NS = SIDE * SIDE;
    __shared__ float4 T[NS][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // 1-D grid, the cell slowest: blockIdx.x = (cell rank * gz + z) * chunks + chunk -- workgroups start in blockIdx order, so
    // all parts of a cell start together and the cells in the order of their rank
    const int chunks = C >> 8;
    const int chunk = (int)(blockIdx.x % (unsigned)chunks), c0 = chunk * 256 + lane * 4;
    const int zpart = (int)((blockIdx.x / (unsigned)chunks) % (unsigned)gz);
    const unsigned rank = blockIdx.x / (unsigned)(chunks * gz);
    const int rbits = nbits - level, r = 1 << rbits, sbits = rbits - K;
    // Morton index (batch-major) of the cell at level + K; `order`: the cells by falling row count (t2h_cell_order_build), so
    // that the dispatcher starts the dense cells first and the light ones fill the gaps (longest-first scheduling)
    const int64_t wg = order ? (int64_t)order[rank] : (int64_t)rank;
    const int b = (int)(wg >> (2 * sbits));
    const uint32_t smk = (uint32_t)(wg & (((int64_t)1 << (2 * sbits)) - 1));
    const int d = level - sum_level, nchild = 1 << (2 * d), rs = 1 << (nbits - sum_level);
    const uint32_t mk = K == 0 ? smk : ((smk << 2) | (uint32_t)wave);   // this wave's sampling cell
    const int cx = (int)compact1by1(mk), cy = (int)compact1by1(mk >> 1);
    const int per_wave = K == 0 ? nchild / (gz * 4) : nchild;
    const int child_lo = K == 0 ? (zpart * 4 + wave) * per_wave : 0, child_hi = child_lo + per_wave;
    const size_t obase = ((size_t)b << (2 * nbits)) + ((size_t)mk << (2 * level));
    // the first 64 child boundaries and the rows' coordinates are on the wave's critical path: requested BEFORE the staging loads
    int bnd_lo, bnd_hi;
    {
        const int ci = child_lo + lane;
        bnd_lo = ci <= child_hi ? off0[obase + ((size_t)ci << (2 * sum_level))] : 0;
        bnd_hi = ci + 1 <= child_hi ? off0[obase + ((size_t)(ci + 1) << (2 * sum_level))] : 0;
    }
    const int wx0 = K == 0 ? cx - 1 : 2 * (int)compact1by1(smk) - 1, wy0 = K == 0 ? cy - 1 : 2 * (int)compact1by1(smk >> 1) - 1;
    for (int e = tid; e < NS * 64; e += 256) {
        const int sl = e >> 6, py = wy0 + sl / SIDE, px = wx0 + sl % SIDE;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if ((unsigned)py < (unsigned)r && (unsigned)px < (unsigned)r)
            v = *reinterpret_cast<const float4 *>(plane + (((size_t)b * r + py) * r + px) * C + chunk * 256 + (e & 63) * 4);
        T[sl][e & 63] = v;
    }
    const int row_first = __builtin_amdgcn_readfirstlane(bnd_lo);       // this wave's rows: [row_first, row_end), contiguous
    const int row_end = off0[obase + ((size_t)child_hi << (2 * sum_level))];
    float2 pxy = *reinterpret_cast<const float2 *>(pts + (size_t)min(row_first + lane, npts_m1) * dim);
    __syncthreads();
    const float4 *Tl = &T[0][lane];

    int nb0 = __builtin_amdgcn_readfirstlane(row_first - 64), slot_l = 0;  // taps of rows nb0 .. nb0 + 63, lane-parallel
    float nw_l = 0.f, ne_l = 0.f, sw_l = 0.f, se_l = 0.f;
    unsigned bl[8] = {0, 0, 0, 0, 0, 0, 0, 0};                           // the ballots of row nb0 + lane
    float4 pe = make_float4(0.f, 0.f, 0.f, 0.f), p01 = pe;              // pooled-output state of the current quad of children
    auto flush_bits = [&]() {
        if (bits && nb0 >= row_first && nb0 + lane < row_end) {
            uint4 *dst = reinterpret_cast<uint4 *>(bits + ((size_t)chunk * ((size_t)npts_m1 + 1) + (size_t)(nb0 + lane)) * 4);
            dst[0] = make_uint4(bl[0], bl[1], bl[2], bl[3]);
            dst[1] = make_uint4(bl[4], bl[5], bl[6], bl[7]);
        }
    };
    for (int cb = child_lo; cb < child_hi; cb += 64) {                  // children in batches of 64: their row boundaries
        if (cb != child_lo) {
            const int ci = cb + lane;
            bnd_lo = ci <= child_hi ? off0[obase + ((size_t)ci << (2 * sum_level))] : 0;
            bnd_hi = ci + 1 <= child_hi ? off0[obase + ((size_t)(ci + 1) << (2 * sum_level))] : 0;
        }
        const int nc = min(64, child_hi - cb);
        for (int c = 0; c < nc; ++c) {
            const int s = __builtin_amdgcn_readlane(bnd_lo, c), e = __builtin_amdgcn_readlane(bnd_hi, c);
            f32x2 sum01 = {0.f, 0.f}, sum23 = {0.f, 0.f};
            int n = s;
            while (n < e) {
                if (n >= nb0 + 64) {
                    flush_bits();
                    nb0 = __builtin_amdgcn_readfirstlane(nb0 + 64);
                    const Taps tp = make_taps(pxy.x, pxy.y, r);
                    if (nb0 + 64 < row_end)                               // the next 64 rows' coordinates, under this batch's rows
                        pxy = *reinterpret_cast<const float2 *>(pts + (size_t)min(nb0 + 64 + lane, npts_m1) * dim);
                    // 1.0 where the east / south tap lies inside the plane, else 0.0 -- by integer min / max and a product (p * 1 = p,
                    // p * 0 = +0 for these non-negative finite p: the same bits as a select).  NOT `x1ok ? p : 0`: that compiles to
                    // v_cmp -> SGPR pair -> s_and_b64 + v_cndmask, and with the split-convolution kernels (conv_bx3.hip) resident on
                    // the same CU the select of a wave's FIRST 64 rows saw the mask with its last quarter (lanes 48..63) cleared --
                    // 16 rows sampled without their east tap, a few cells' sums off by 1e-2 relative, 4-50 % of pipelined steps
                    // (r05: profiles/r05_coresidency.txt; found by replaying one backward's calls beside this kernel on a second
                    // stream; both this kernel and its r04 form, alone or beside any other kernel never).  The weights live for
                    // 64 rows, so one lost quarter is amplified; tests/test_coresidency.py keeps watch
#ifdef T2H_TAPS_BY_SELECT     // the r05 reproducer only (profiles/coresidency_lab.py builds it beside the product library): the pre-fix form
                    const bool x1ok = tp.x0 + 1 < r, y1ok = tp.y0 + 1 < r;
                    nw_l = __fmul_rn(tp.wx0, tp.wy0);
                    ne_l = x1ok ? __fmul_rn(tp.wx1, tp.wy0) : 0.f;
                    sw_l = y1ok ? __fmul_rn(tp.wx0, tp.wy1) : 0.f;
                    se_l = (x1ok && y1ok) ? __fmul_rn(tp.wx1, tp.wy1) : 0.f;
#else
                    const float fx1 = (float)min(max(r - 1 - tp.x0, 0), 1), fy1 = (float)min(max(r - 1 - tp.y0, 0), 1);
                    nw_l = __fmul_rn(tp.wx0, tp.wy0);
                    ne_l = __fmul_rn(__fmul_rn(tp.wx1, tp.wy0), fx1);
                    sw_l = __fmul_rn(__fmul_rn(tp.wx0, tp.wy1), fy1);
                    se_l = __fmul_rn(__fmul_rn(tp.wx1, tp.wy1), __fmul_rn(fx1, fy1));
#endif
                    // slot of the north-west tap in the staged window (a tap outside the plane has weight 0 and reads the
                    // staged zero -- the sample kernel skips it: the same value)
                    slot_l = (min(max(tp.y0 - cy + 1, 0), 1) + (K ? (wave >> 1) : 0)) * SIDE +
                             min(max(tp.x0 - cx + 1, 0), 1) + (K ? (wave & 1) : 0);
                }
                const int i0 = n - nb0, cnt = min(e - n, 64 - i0);        // rows of this child inside the current batch
                auto row = [&](int i) {
                    const float nw = readlane_f(nw_l, i), ne = readlane_f(ne_l, i), sw = readlane_f(sw_l, i), se = readlane_f(se_l, i);
                    const float4 *t00 = Tl + __builtin_amdgcn_readlane(slot_l, i) * 64;
                    const float4 v00 = t00[0], v01 = t00[64], v10 = t00[SIDE * 64], v11 = t00[(SIDE + 1) * 64];
                    f32x2 a01 = f32x2{v00.x, v00.y} * nw, a23 = f32x2{v00.z, v00.w} * nw;
                    a01 = a01 + f32x2{v01.x, v01.y} * ne; a23 = a23 + f32x2{v01.z, v01.w} * ne;
                    a01 = a01 + f32x2{v10.x, v10.y} * sw; a23 = a23 + f32x2{v10.z, v10.w} * sw;
                    a01 = a01 + f32x2{v11.x, v11.y} * se; a23 = a23 + f32x2{v11.z, v11.w} * se;
                    a01.x = fmaxf(a01.x, 0.f); a01.y = fmaxf(a01.y, 0.f); a23.x = fmaxf(a23.x, 0.f); a23.y = fmaxf(a23.y, 0.f);
                    const unsigned long long w0 = __ballot(a01.x > 0.f), w1 = __ballot(a01.y > 0.f);
                    const unsigned long long w2 = __ballot(a23.x > 0.f), w3 = __ballot(a23.y > 0.f);
                    park8(bl, w0, w1, w2, w3, i);
                    sum01 = sum01 + a01; sum23 = sum23 + a23;             // summed in row order
                };
                for (int j = 0; j < cnt; ++j) row(i0 + j);
                n += cnt;
            }
            const float4 sum = make_float4(sum01.x, sum01.y, sum23.x, sum23.y);
            const uint32_t cm = (uint32_t)(cb + c);                      // child's Morton code inside the cell
            const int fx = (cx << d) + (int)compact1by1(cm), fy = (cy << d) + (int)compact1by1(cm >> 1);
            *reinterpret_cast<float4 *>(sums + (((size_t)b * rs + fy) * rs + fx) * ld_sums + c0) = sum;
            if (POOL) {
                // the 2 x 2 pooled sums one level up, formed from the four children while they are in registers (children are
                // visited in Morton order = the order (c0 + c1) + (c2 + c3) of plane_sumpool2x2_kernel: same bits)
                const unsigned k4 = cm & 3u;
                if (k4 == 0u) pe = sum;
                else if (k4 == 1u) p01 = make_float4(pe.x + sum.x, pe.y + sum.y, pe.z + sum.z, pe.w + sum.w);
                else if (k4 == 2u) pe = sum;
                else {
                    const float4 o = make_float4(p01.x + (pe.x + sum.x), p01.y + (pe.y + sum.y), p01.z + (pe.z + sum.z),
                                                 p01.w + (pe.w + sum.w));
                    const int rs2 = rs >> 1;
                    *reinterpret_cast<float4 *>(sums2 + (((size_t)b * rs2 + (fy >> 1)) * rs2 + (fx >> 1)) * ld_sums2 + c0) = o;
                }
            }
        }
    }
    flush_bits();
}

// Deterministic backward: one group per pixel; a point of cell (cx,cy) only touches pixels
// {cx-1..cx+1} x {cy-1..cy+1} (px = x*(r-1) lies in (cx-1, cx+1)), so pixel (px,py) gathers from the
// 3x3 cells around it.  Cells are visited row-major, points in sorted order: a fixed summation order.
template <int VEC>
__global__ __launch_bounds__(kThreads) void sample_bwd_kernel(const float *__restrict__ gout,
                                                              const float *__restrict__ pts, int dim,
                                                              const int32_t *__restrict__ off0, int B, int nbits,
                                                              int level, int C, int lg, const float *__restrict__ addend,
                                                              float *__restrict__ gplane) {
    int64_t t = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    int64_t gid = t >> lg;
    const int rbits = nbits - level, r = 1 << rbits;
    if (gid >= (int64_t)B * r * r) return;
    int b = (int)(gid >> (2 * rbits));
    // pixels in Morton order: the pixels of a workgroup form a compact 2-D block, so the rows of the cells around them
    // are re-read from this CU's L1 instead of from L2 by ~9 different workgroups
    const uint32_t pm = (uint32_t)(gid & (((int64_t)1 << (2 * rbits)) - 1));
    int py = (int)compact1by1(pm >> 1), px = (int)compact1by1(pm);
    int span = VEC << lg;
    int c0 = ((int)t & ((1 << lg) - 1)) * VEC;
    // the row ranges of the (up to) nine cells, requested together
    int segs[9], sege[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) {
        const int cy = py - 1 + q / 3, cx = px - 1 + q % 3;
        segs[q] = sege[q] = 0;
        if (cx >= 0 && cx < r && cy >= 0 && cy < r) {
            size_t obase = ((size_t)b << (2 * nbits)) + ((size_t)morton2((uint32_t)cx, (uint32_t)cy) << (2 * level));
            segs[q] = off0[obase]; sege[q] = off0[obase + ((size_t)1 << (2 * level))];
        }
    }
    for (int c = c0; c < C; c += span) {
        float acc[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[j] = 0.0f;
#pragma unroll
        for (int q = 0; q < 9; ++q) {
                const int s = segs[q], e = sege[q];
                for (int n = s; n < e; n += 2) {   // two rows in flight; a row that misses this pixel gets weight 0
                    const int n1 = min(n + 1, e - 1);
                    Taps t0 = make_taps(pts[(size_t)n * dim + 0], pts[(size_t)n * dim + 1], r);
                    Taps t1 = make_taps(pts[(size_t)n1 * dim + 0], pts[(size_t)n1 * dim + 1], r);
                    Vec<VEC> g0 = Vec<VEC>::load(gout + (size_t)n * C + c);
                    Vec<VEC> g1 = Vec<VEC>::load(gout + (size_t)n1 * C + c);
                    float wx0 = (t0.x0 == px) ? t0.wx0 : ((t0.x0 + 1 == px) ? t0.wx1 : 0.0f);
                    float wy0 = (t0.y0 == py) ? t0.wy0 : ((t0.y0 + 1 == py) ? t0.wy1 : 0.0f);
                    float wx1 = (t1.x0 == px) ? t1.wx0 : ((t1.x0 + 1 == px) ? t1.wx1 : 0.0f);
                    float wy1 = (t1.y0 == py) ? t1.wy0 : ((t1.y0 + 1 == py) ? t1.wy1 : 0.0f);
                    float w0 = __fmul_rn(wx0, wy0);
                    float w1 = (n + 1 < e) ? __fmul_rn(wx1, wy1) : 0.0f;
#pragma unroll
                    for (int j = 0; j < VEC; ++j) {
                        acc[j] = __fadd_rn(acc[j], __fmul_rn(w0, g0.v[j]));
                        acc[j] = __fadd_rn(acc[j], __fmul_rn(w1, g1.v[j]));
                    }
                }
            }
        Vec<VEC> o;
#pragma unroll
        for (int j = 0; j < VEC; ++j) o.v[j] = acc[j];
        const size_t at = (((size_t)b * r + py) * r + px) * C + c;
        if (addend) {       // the plane's other gradient (it also feeds a convolution): summed here, not by a pass of its own
            Vec<VEC> a = Vec<VEC>::load(addend + at);
#pragma unroll
            for (int j = 0; j < VEC; ++j) o.v[j] = __fadd_rn(a.v[j], o.v[j]);
        }
        o.store(gplane + at);
    }
}


// Measured alternatives to this gather at the fine levels (r02, N = 131072 clustered, C = 64, r = 256: 78 us here):
//   * 8 x 8 pixel blocks accumulated in LDS, cells visited in nine colour phases so that no two lane groups touch one
//     pixel at a time (rows read once, deterministic): 99-260 us -- nine dependent global-load rounds per workgroup with
//     a handful of rows in flight each;
//   * a per-point tap table (pixel of the north-west tap + the four weights) so that a visited row costs one compare
//     chain instead of recomputing the taps: 103 us -- rocprofv3 shows ~2100 vector and ~90 load instructions per wave
//     here, and the two extra loads per visited row cost more than the arithmetic they save.
//   * a 2 x 2 pixel quad per lane group (a row visited by 4 quads instead of 9 pixels, taps computed once per quad,
//     bit-identical sums): 97 us -- a quarter of the threads with four accumulators each hides less latency.
//   * the first 1-3 rows of all nine cells requested before any is used (9-27 rows in flight per lane): 82-239 us --
//     the extra (dummy) loads for the many empty cells cost more than the latency they hide: the walk is bound by the
//     NUMBER of load instructions (~1.5 M wave-loads) plus ~35 M vector instructions, both proportional to the nine-fold
//     visit, not by latency.
//   * one wave per 4 x 4 pixel tile walking the rows of the 6 x 6 cells around it serially (row list and taps built 64
//     rows at a time in LDS, accumulators in LDS, lane group = tap, 4-16 gradient rows in flight; bit-identical sums, a
//     row visited by 1-4 tiles instead of 9 pixels): 138 us (r = 128, C = 128: 316 us against 74 for the per-cell
//     partials) -- the walk is a chain of dependent LDS read-modify-writes and its length is the tile's row count,
//     which the clustered clouds make very uneven (small buildings hold 20+ rows per cell); C = 32 took as long as 64.
// What did pay: x * 0.5f instead of the IEEE division in unnormalize_clip (bit-identical), mostly in sample_fwd.

// ---------------------------------------------------------------- sample backward through the transposed matrix
// The bilinear sample is a sparse matrix S [points x pixels] with four entries per row; its backward is S^T g.  The gather
// above re-derives S^T on every call (nine cells scanned per pixel, taps recomputed in every lane: ~34 M wave
// instructions at r = 256, C = 64).  The points of a tile do not move between the calls of one training step -- three
// sample backwards at r = 256, two at r = 128 -- so S^T is built once per (tile, level) as a CSR over pixels: entries
// (row, weight) of a pixel in exactly the order the gather visits them (cells row-major, rows sorted), which keeps the sums
// bit-identical.  A backward is then one pass over ~4 N entries: 16-byte row loads, one multiply-add per channel.
//   adjoint_scan_kernel<false>: entries per 16-pixel workgroup; adjoint_offsets_kernel: their exclusive prefix sum (one
//   workgroup); adjoint_scan_kernel<true>: offsets + fill; sample_bwd_adjoint_kernel: the product.
// 16 lanes per pixel, lane q < 9 scans the rows of neighbour cell q (a handful of rows each: short dependent-load chains,
// 16 x the waves of a lane-per-pixel walk); the entries of a pixel are laid out q-major, rows ascending.  Pixels are
// numbered in Morton order (gid); a workgroup owns 16 consecutive pixels.
//   FILL = false: block_sums[blockIdx + 1] = entries of the workgroup's pixels
//   FILL = true : offsets[gid] = first entry of pixel gid (block_sums now holds the exclusive block offsets), entries filled
template <bool FILL>
__global__ __launch_bounds__(kThreads) void adjoint_scan_kernel(const float *__restrict__ pts, int dim,
                                                                const int32_t *__restrict__ off0, int64_t npix, int nbits,
                                                                int level, int32_t *__restrict__ block_sums,
                                                                int32_t *__restrict__ offsets, int2 *__restrict__ entries) {
    __shared__ int pix_tot[kThreads / 16];
    const int tid = threadIdx.x, pl = tid >> 4, q = tid & 15;
    const int64_t gid = (int64_t)blockIdx.x * (kThreads / 16) + pl;
    const int rbits = nbits - level, r = 1 << rbits;
    const bool live = gid < npix;
    const int b = (int)(gid >> (2 * rbits));
    const uint32_t pm = (uint32_t)(gid & (((int64_t)1 << (2 * rbits)) - 1));
    const int py = (int)compact1by1(pm >> 1), px = (int)compact1by1(pm);
    int s = 0, e = 0;
    if (live && q < 9) {
        const int cy = py - 1 + q / 3, cx = px - 1 + q % 3;
        if (cx >= 0 && cx < r && cy >= 0 && cy < r) {
            const size_t obase = ((size_t)b << (2 * nbits)) + ((size_t)morton2((uint32_t)cx, (uint32_t)cy) << (2 * level));
            s = off0[obase]; e = off0[obase + ((size_t)1 << (2 * level))];
        }
    }
    int count = 0;
    for (int n = s; n < e; ++n) {
        const Taps t = make_taps(pts[(size_t)n * dim + 0], pts[(size_t)n * dim + 1], r);
        count += ((t.x0 == px || t.x0 + 1 == px) && (t.y0 == py || t.y0 + 1 == py)) ? 1 : 0;
    }
    int incl = count;                                          // prefix over the pixel's 16 lanes
#pragma unroll
    for (int d = 1; d < 16; d <<= 1) {
        const int up = __shfl_up(incl, d, 16);
        if (q >= d) incl += up;
    }
    if (q == 15) pix_tot[pl] = incl;
    __syncthreads();
    if (!FILL) {
        if (tid == 0) {
            int tot = 0;
#pragma unroll
            for (int i = 0; i < kThreads / 16; ++i) tot += pix_tot[i];
            block_sums[blockIdx.x + 1] = tot;
        }
        return;
    }
    int start = block_sums[blockIdx.x];
    for (int i = 0; i < pl; ++i) start += pix_tot[i];
    if (live && q == 15) {
        offsets[gid] = start;
        if (gid == npix - 1) offsets[npix] = start + incl;
    }
    int at = start + incl - count;
    for (int n = s; n < e; ++n) {
        const Taps t = make_taps(pts[(size_t)n * dim + 0], pts[(size_t)n * dim + 1], r);
        if ((t.x0 == px || t.x0 + 1 == px) && (t.y0 == py || t.y0 + 1 == py)) {
            const float wx = t.x0 == px ? t.wx0 : t.wx1, wy = t.y0 == py ? t.wy0 : t.wy1;
            entries[at++] = make_int2(n, __float_as_int(__fmul_rn(wx, wy)));
        }
    }
}

// in place: offsets[0] = 0, offsets[i + 1] = counts summed up to and including pixel i  (n = number of pixels)
__global__ __launch_bounds__(1024) void adjoint_offsets_kernel(int32_t *__restrict__ offsets, int64_t n) {
    __shared__ int wave_tot[16];
    __shared__ int carry_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) { carry_s = 0; offsets[0] = 0; }
    __syncthreads();
    for (int64_t base = 0; base < n; base += 1024) {
        const int64_t i = base + tid;
        const int v = i < n ? offsets[i + 1] : 0;
        int incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int up = __shfl_up(incl, d, 64);
            if (lane >= d) incl += up;
        }
        if (lane == 63) wave_tot[wave] = incl;
        __syncthreads();
        int before = carry_s;
        for (int w = 0; w < wave; ++w) before += wave_tot[w];
        if (i < n) offsets[i + 1] = before + incl;
        __syncthreads();
        if (tid == 1023) carry_s = before + incl;
        __syncthreads();
    }
}

template <int VEC>
__global__ __launch_bounds__(kThreads) void sample_bwd_adjoint_kernel(const float *__restrict__ gout,
                                                                      const int32_t *__restrict__ offsets,
                                                                      const int2 *__restrict__ entries, int64_t npix,
                                                                      int rbits, int C, int lg,
                                                                      const float *__restrict__ addend,
                                                                      float *__restrict__ gplane) {
    const int64_t t = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    const int64_t gid = t >> lg;
    if (gid >= npix) return;
    // pixels in Morton order inside a tile: the four pixels that read a row sit in the same or a neighbouring workgroup
    const int64_t b = gid >> (2 * rbits);
    const uint32_t pm = (uint32_t)(gid & (((int64_t)1 << (2 * rbits)) - 1));
    const int py = (int)compact1by1(pm >> 1), px = (int)compact1by1(pm);
    const int64_t pix = (b << (2 * rbits)) + ((int64_t)py << rbits) + px;
    const int s = offsets[gid], e = offsets[gid + 1];                // the CSR is indexed by the Morton pixel number
    const int span = VEC << lg;
    for (int c = ((int)t & ((1 << lg) - 1)) * VEC; c < C; c += span) {
        float acc[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[j] = 0.0f;
        int k = s;
        for (; k + 7 < e; k += 8) {                              // eight rows in flight, added in entry order (sixteen: slower)
            int2 en[8];
            Vec<VEC> g[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) en[u] = entries[k + u];
#pragma unroll
            for (int u = 0; u < 8; ++u) g[u] = Vec<VEC>::load(gout + (size_t)en[u].x * C + c);
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int j = 0; j < VEC; ++j) acc[j] = __fadd_rn(acc[j], __fmul_rn(__int_as_float(en[u].y), g[u].v[j]));
        }
        for (; k + 3 < e; k += 4) {                              // four rows in flight, added in entry order
            const int2 e0 = entries[k], e1 = entries[k + 1], e2 = entries[k + 2], e3 = entries[k + 3];
            const Vec<VEC> g0 = Vec<VEC>::load(gout + (size_t)e0.x * C + c);
            const Vec<VEC> g1 = Vec<VEC>::load(gout + (size_t)e1.x * C + c);
            const Vec<VEC> g2 = Vec<VEC>::load(gout + (size_t)e2.x * C + c);
            const Vec<VEC> g3 = Vec<VEC>::load(gout + (size_t)e3.x * C + c);
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                acc[j] = __fadd_rn(acc[j], __fmul_rn(__int_as_float(e0.y), g0.v[j]));
                acc[j] = __fadd_rn(acc[j], __fmul_rn(__int_as_float(e1.y), g1.v[j]));
                acc[j] = __fadd_rn(acc[j], __fmul_rn(__int_as_float(e2.y), g2.v[j]));
                acc[j] = __fadd_rn(acc[j], __fmul_rn(__int_as_float(e3.y), g3.v[j]));
            }
        }
        for (; k < e; ++k) {
            const int2 e0 = entries[k];
            const Vec<VEC> g0 = Vec<VEC>::load(gout + (size_t)e0.x * C + c);
#pragma unroll
            for (int j = 0; j < VEC; ++j) acc[j] = __fadd_rn(acc[j], __fmul_rn(__int_as_float(e0.y), g0.v[j]));
        }
        Vec<VEC> o;
#pragma unroll
        for (int j = 0; j < VEC; ++j) o.v[j] = acc[j];
        const size_t at = (size_t)pix * C + c;
        if (addend) {
            Vec<VEC> a = Vec<VEC>::load(addend + at);
#pragma unroll
            for (int j = 0; j < VEC; ++j) o.v[j] = __fadd_rn(a.v[j], o.v[j]);
        }
        o.store(gplane + at);
    }
}

// ------------------------------------------------------------------------------ coarse levels (many points / cell)
// At coarse ALTO levels a cell holds tens to hundreds of points, so "one lane-group walks one cell" leaves the
// chip idle and the 3x3 pixel gather re-reads every row ~9x.  Here one workgroup owns (cell, split): its rows
// are dealt round-robin to P = 256/G parallel row-slots (G lanes x float4 = the channel chunk), each slot keeps
// its partial sums in registers, and the slots are combined through LDS in slot order (fixed => deterministic).
constexpr int kCellThreads = 256;

__host__ __device__ inline int cell_chunk_channels(int C) { return C < 256 ? C : 256; }   // channels per workgroup (<= 64 lanes x float4)

// rows [lo, hi) of split `sp` out of S over segment [s, e)
__device__ inline void split_range(int s, int e, int sp, int S, int &lo, int &hi) {
    int len = e - s, per = (len + S - 1) / S;
    lo = min(e, s + sp * per);
    hi = min(e, lo + per);
}

// partial[((cellrow * S + sp)] [C]  -- per-(cell, split) sums of the rows; finalised by segmean_finalize_kernel
__global__ __launch_bounds__(kCellThreads) void segmean_cells_kernel(const float *__restrict__ feat,
                                                                     const int32_t *__restrict__ off0, int nbits,
                                                                     int level, int C, int lgG, int S,
                                                                     float *__restrict__ partial) {
    extern __shared__ float4 red[];                       // [P][G]
    const int G = 1 << lgG, P = kCellThreads >> lgG;
    int lane_c = threadIdx.x & (G - 1), slot = threadIdx.x >> lgG;
    int64_t cellrow = blockIdx.x;                          // b * cells_per_tile + mk   (Morton order)
    int sp = blockIdx.y % S, chunk = blockIdx.y / S;
    const int rbits = nbits - level;
    int b = (int)(cellrow >> (2 * rbits));
    uint32_t mk = (uint32_t)(cellrow & (((int64_t)1 << (2 * rbits)) - 1));
    size_t obase = ((size_t)b << (2 * nbits)) + ((size_t)mk << (2 * level));
    int s = off0[obase], e = off0[obase + ((size_t)1 << (2 * level))];
    int lo, hi;
    split_range(s, e, sp, S, lo, hi);
    int c = chunk * cell_chunk_channels(C) + lane_c * 4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < C) {
        int n = lo + slot;
        for (; n + P < hi; n += 2 * P) {                   // two independent rows in flight per slot
            float4 a0 = *reinterpret_cast<const float4 *>(feat + (size_t)n * C + c);
            float4 a1 = *reinterpret_cast<const float4 *>(feat + (size_t)(n + P) * C + c);
            acc.x = (acc.x + a0.x) + a1.x; acc.y = (acc.y + a0.y) + a1.y;
            acc.z = (acc.z + a0.z) + a1.z; acc.w = (acc.w + a0.w) + a1.w;
        }
        for (; n < hi; n += P) {
            float4 a0 = *reinterpret_cast<const float4 *>(feat + (size_t)n * C + c);
            acc.x += a0.x; acc.y += a0.y; acc.z += a0.z; acc.w += a0.w;
        }
    }
    red[slot * G + lane_c] = acc;
    __syncthreads();
    if (slot == 0 && c < C) {
        float4 t = red[lane_c];
        for (int q = 1; q < P; ++q) {
            float4 u = red[q * G + lane_c];
            t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
        }
        *reinterpret_cast<float4 *>(partial + ((size_t)cellrow * S + sp) * C + c) = t;
    }
}

// plane[cell] = (sum over splits, in split order) / max(count, 1)
__global__ __launch_bounds__(kThreads) void segmean_finalize_kernel(const float *__restrict__ partial,
                                                                   const int32_t *__restrict__ off0, int B, int nbits,
                                                                   int level, int C, int lg, int S,
                                                                   float *__restrict__ plane, int mean = 1, int ld_out = 0) {
    int64_t t = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    int64_t gid = t >> lg;
    const int rbits = nbits - level;
    const int64_t cells_per_tile = (int64_t)1 << (2 * rbits);
    if (gid >= (int64_t)B * cells_per_tile) return;
    int b = (int)(gid >> (2 * rbits));
    uint32_t mk = (uint32_t)(gid & (cells_per_tile - 1));
    size_t obase = ((size_t)b << (2 * nbits)) + ((size_t)mk << (2 * level));
    int cnt = off0[obase + ((size_t)1 << (2 * level))] - off0[obase];
    float den = (float)(cnt > 0 ? cnt : 1);
    int cx = (int)compact1by1(mk), cy = (int)compact1by1(mk >> 1), r = 1 << rbits;
    float *orow = plane + (((size_t)b * r + cy) * r + cx) * (ld_out > 0 ? ld_out : C);
    for (int c = ((int)t & ((1 << lg) - 1)) * 4; c < C; c += 4 << lg) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int sp = 0; sp < S; ++sp) {
            float4 u = *reinterpret_cast<const float4 *>(partial + ((size_t)gid * S + sp) * C + c);
            acc.x += u.x; acc.y += u.y; acc.z += u.z; acc.w += u.w;
        }
        if (mean) {
            acc.x = __fdiv_rn(acc.x, den); acc.y = __fdiv_rn(acc.y, den);
            acc.z = __fdiv_rn(acc.z, den); acc.w = __fdiv_rn(acc.w, den);
        }
        *reinterpret_cast<float4 *>(orow + c) = acc;
    }
}

// grid_sample backward, stage 1: per (cell, split) the contributions of its rows to the 3x3 pixels around the
// cell (slot = (py-cy+1)*3 + (px-cx+1)); every gradient row is read exactly once.
// FUSED: the gradient row is not read from `gout` but formed on the fly as (mask[n] > 0 ?) sum_q mp.g[q][cell_q(n)] -- the
// adjoint of the per-cell sums of the hidden activations (t2h_segsum_bwd_multi) folded into this kernel's row load, so the
// [N, C] hidden gradient is never written or re-read (t2h_sample_bwd_from_sums)
template <bool FUSED>
__global__ __launch_bounds__(kCellThreads) void sample_bwd_cells_kernel(const float *__restrict__ gout,
                                                                        const float *__restrict__ pts, int dim,
                                                                        const int32_t *__restrict__ off0, int nbits,
                                                                        int level, int C, int lgG, int S,
                                                                        float *__restrict__ partial, MultiPlanes mp,
                                                                        const int32_t *__restrict__ cell,
                                                                        const float *__restrict__ mask) {
    extern __shared__ float4 red[];                       // [P][G]
    const int G = 1 << lgG, P = kCellThreads >> lgG;
    int lane_c = threadIdx.x & (G - 1), slot = threadIdx.x >> lgG;
    int64_t cellrow = blockIdx.x;
    int sp = blockIdx.y % S, chunk = blockIdx.y / S;
    const int rbits = nbits - level, r = 1 << rbits;
    int b = (int)(cellrow >> (2 * rbits));
    uint32_t mk = (uint32_t)(cellrow & (((int64_t)1 << (2 * rbits)) - 1));
    int cx = (int)compact1by1(mk), cy = (int)compact1by1(mk >> 1);
    size_t obase = ((size_t)b << (2 * nbits)) + ((size_t)mk << (2 * level));
    int s = off0[obase], e = off0[obase + ((size_t)1 << (2 * level))];
    int lo, hi;
    split_range(s, e, sp, S, lo, hi);
    int c = chunk * cell_chunk_channels(C) + lane_c * 4;
    float4 acc[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) acc[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < C) {
        // a row's value: its gradient row, or (FUSED) the gathered + masked sum of the per-cell-sum gradients
        auto fetch = [&](int n, Taps &tp, float4 &g) {
            tp = make_taps(pts[(size_t)n * dim + 0], pts[(size_t)n * dim + 1], r);
            if (FUSED) {
                const uint32_t code = (uint32_t)cell[n];
                const uint32_t fb = code >> (2 * nbits), fm = code & ((1u << (2 * nbits)) - 1u);
                const uint32_t fx = compact1by1(fm), fy = compact1by1(fm >> 1);
                const float4 hm = *reinterpret_cast<const float4 *>(mask + (size_t)n * C + c);
                g = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int q = 0; q < kMaxMultiPlanes; ++q)
                    if (q < mp.n) {
                        const int l = mp.level[q], rq = 1 << (nbits - l);
                        const float4 u = *reinterpret_cast<const float4 *>(mp.g[q] + (((size_t)fb * rq + (fy >> l)) * rq + (fx >> l)) * mp.ld[q] + c);
                        if (q == 0) g = u;
                        else { g.x = __fadd_rn(g.x, u.x); g.y = __fadd_rn(g.y, u.y); g.z = __fadd_rn(g.z, u.z); g.w = __fadd_rn(g.w, u.w); }
                    }
                g.x = hm.x > 0.f ? g.x : 0.f; g.y = hm.y > 0.f ? g.y : 0.f;
                g.z = hm.z > 0.f ? g.z : 0.f; g.w = hm.w > 0.f ? g.w : 0.f;
            } else {
                g = *reinterpret_cast<const float4 *>(gout + (size_t)n * C + c);
            }
        };
        auto accumulate = [&](const Taps &tp, const float4 &g) {
            const int dx = tp.x0 - cx + 1, dy = tp.y0 - cy + 1;     // slot column/row of the north-west tap: 0 or 1
#pragma unroll
            for (int sy = 0; sy < 3; ++sy) {
                float wy = (sy == dy) ? tp.wy0 : ((sy == dy + 1) ? tp.wy1 : 0.0f);
#pragma unroll
                for (int sx = 0; sx < 3; ++sx) {
                    float wx = (sx == dx) ? tp.wx0 : ((sx == dx + 1) ? tp.wx1 : 0.0f);
                    float w = __fmul_rn(wx, wy);
                    float4 &a = acc[sy * 3 + sx];
                    // fused multiply-add: this kernel is VALU-bound (nine slots per row), and its partial sums are
                    // re-associated against the reference's order anyway (tolerance-checked, not bit-compared)
                    a.x = __fmaf_rn(w, g.x, a.x); a.y = __fmaf_rn(w, g.y, a.y);
                    a.z = __fmaf_rn(w, g.z, a.z); a.w = __fmaf_rn(w, g.w, a.w);
                }
            }
        };
        // two rows in flight per slot: the loads of the second row (points, cell code, planes, mask) are requested before
        // the first row's 36 multiply-adds; rows are accumulated in the same order as a one-row loop
        int n = lo + slot;
        for (; n + P < hi; n += 2 * P) {
            Taps t0, t1;
            float4 g0, g1;
            fetch(n, t0, g0);
            fetch(n + P, t1, g1);
            accumulate(t0, g0);
            accumulate(t1, g1);
        }
        if (n < hi) {
            Taps t0;
            float4 g0;
            fetch(n, t0, g0);
            accumulate(t0, g0);
        }
    }
    float *pbase = partial + ((size_t)cellrow * S + sp) * 9 * C;
#pragma unroll
    for (int q = 0; q < 9; ++q) {
        __syncthreads();
        red[slot * G + lane_c] = acc[q];
        __syncthreads();
        if (slot == 0 && c < C) {
            float4 t = red[lane_c];
            for (int k = 1; k < P; ++k) {
                float4 u = red[k * G + lane_c];
                t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
            }
            *reinterpret_cast<float4 *>(pbase + (size_t)q * C + c) = t;
        }
    }
}

// The same per-(cell, split) partials on the matrix cores (C % 64 == 0).  For one cell the contributions of its rows to the
// 3 x 3 pixel slots are a small dense product  partial[slot, :] = sum_rows W[slot, row] * g[row, :]  with W the bilinear tap
// weights -- v_mfma_f32_16x16x4_f32 with M = slots (9 of 16), K = rows (4 per instruction), N = channels.  Per block of 64
// rows 64 threads compute each row's taps ONCE (the VALU kernel above recomputes them in every lane: ~40 of its ~150
// instructions per row and lane, plus 27 for the zero-padded weights and 36 multiply-adds) and leave the [64][16] weight tile
// (FUSED: and every row's pixel index in each gradient plane) in LDS; a wave then owns 64 channels: lane (k = l >> 4,
// n = l & 15) loads the float4 of channels 4 n .. 4 n + 3 of row k -- 256 contiguous bytes per row -- and its four components
// are the B operands of four MFMAs (the N index of an MFMA is only a label: tile t holds channels 4 n + t), so the results of a
// slot come out as float4s again.  fp32 MFMA = an exact fma chain over the rows in order, as the VALU form.
using f32x4_t = __attribute__((ext_vector_type(4))) float;
constexpr int kMfmaRows = 64;
template <bool FUSED, bool BITS = false>
__global__ __launch_bounds__(kCellThreads) void sample_bwd_cells_mfma_kernel(const float *__restrict__ gout,
                                                                             const float *__restrict__ pts, int dim,
                                                                             const int32_t *__restrict__ off0, int nbits,
                                                                             int level, int C, int S,
                                                                             float *__restrict__ partial, MultiPlanes mp,
                                                                             const int32_t *__restrict__ cell,
                                                                             const float *__restrict__ mask, int tlevel,
                                                                             size_t npts) {
    __shared__ float Wl[kMfmaRows][16];
    __shared__ int Po[kMfmaRows][kMaxMultiPlanes];
    // FUSED: the planes at level >= tlevel (at most 16 cells of that level inside this workgroup's cell) are summed ONCE per
    // workgroup into this table -- entry t = child t (Morton order) of the cell at level tlevel -- so a row reads one LDS entry
    // for them instead of one global row per plane (r = 32, four planes: 16 KB of L2 reads per row -> 4 KB + this table)
    __shared__ float4 Tl[FUSED ? 16 : 1][FUSED ? 65 : 1];
    __shared__ int Ti[kMfmaRows];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t cellrow = blockIdx.x;
    const int sp = blockIdx.y % S, chunk = blockIdx.y / S;
    const int rbits = nbits - level, r = 1 << rbits;
    const int b = (int)(cellrow >> (2 * rbits));
    const uint32_t mk = (uint32_t)(cellrow & (((int64_t)1 << (2 * rbits)) - 1));
    const int cx = (int)compact1by1(mk), cy = (int)compact1by1(mk >> 1);
    const size_t obase = ((size_t)b << (2 * nbits)) + ((size_t)mk << (2 * level));
    const int s = off0[obase], e = off0[obase + ((size_t)1 << (2 * level))];
    int lo, hi;
    split_range(s, e, sp, S, lo, hi);
    const int ch = chunk * cell_chunk_channels(C) + wave * 64 + 4 * (lane & 15);
    const bool active = ch < C;                            // uniform over the wave (C % 64 == 0)
    const int kq = lane >> 4;                              // this lane's row inside a 4-row MFMA step
    if (FUSED && tlevel >= 0 && lo < hi) {
        const int tb = 2 * (level - tlevel), nT = 1 << tb;
        for (int e = tid; e < nT * 64; e += kCellThreads) {
            const int t = e >> 6, c4 = e & 63, che = chunk * cell_chunk_channels(C) + c4 * 4;
            if (che < C) {
                const uint32_t mt = (mk << tb) | (uint32_t)t;                // Morton code of the child at level tlevel
                const uint32_t tx = compact1by1(mt), ty = compact1by1(mt >> 1);
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int q = 0; q < kMaxMultiPlanes; ++q)
                    if (q < mp.n && mp.level[q] >= tlevel) {
                        const int l = mp.level[q], rq = 1 << (nbits - l), sh = l - tlevel;
                        const float4 w = *reinterpret_cast<const float4 *>(
                            mp.g[q] + (((size_t)b * rq + (ty >> sh)) * rq + (tx >> sh)) * mp.ld[q] + che);
                        v.x = __fadd_rn(v.x, w.x); v.y = __fadd_rn(v.y, w.y); v.z = __fadd_rn(v.z, w.z); v.w = __fadd_rn(v.w, w.w);
                    }
                Tl[t][c4] = v;
            }
        }
    }
    f32x4_t acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    for (int blk = lo; blk < hi; blk += kMfmaRows) {
        __syncthreads();                                   // the previous block's tiles are consumed
        if (tid < kMfmaRows) {
            const int n = blk + tid;
            float w[9];
#pragma unroll
            for (int q = 0; q < 9; ++q) w[q] = 0.0f;
            const int nn = min(n, hi - 1);
            if (n < hi) {
                const Taps tp = make_taps(pts[(size_t)n * dim + 0], pts[(size_t)n * dim + 1], r);
                const int dx = tp.x0 - cx + 1, dy = tp.y0 - cy + 1;       // slot column/row of the north-west tap: 0 or 1
#pragma unroll
                for (int sy = 0; sy < 3; ++sy) {
                    const float wy = (sy == dy) ? tp.wy0 : ((sy == dy + 1) ? tp.wy1 : 0.0f);
#pragma unroll
                    for (int sx = 0; sx < 3; ++sx) {
                        const float wx = (sx == dx) ? tp.wx0 : ((sx == dx + 1) ? tp.wx1 : 0.0f);
                        w[sy * 3 + sx] = __fmul_rn(wx, wy);
                    }
                }
            }
            float4 *wr = reinterpret_cast<float4 *>(&Wl[tid][0]);
            wr[0] = make_float4(w[0], w[1], w[2], w[3]);
            wr[1] = make_float4(w[4], w[5], w[6], w[7]);
            wr[2] = make_float4(w[8], 0.f, 0.f, 0.f);
            wr[3] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (FUSED) {
                const uint32_t code = (uint32_t)cell[nn];
                const uint32_t fb = code >> (2 * nbits), fm = code & ((1u << (2 * nbits)) - 1u);
                const uint32_t fx = compact1by1(fm), fy = compact1by1(fm >> 1);
#pragma unroll
                for (int q = 0; q < kMaxMultiPlanes; ++q)
                    if (q < mp.n) {
                        const int l = mp.level[q], rq = 1 << (nbits - l);
                        Po[tid][q] = (int)((fb * rq + (fy >> l)) * rq + (fx >> l));
                    }
                if (tlevel >= 0) Ti[tid] = (int)((fm >> (2 * tlevel)) & ((1u << (2 * (level - tlevel))) - 1u));
            }
        }
        __syncthreads();
        if (active) {
            const int steps = (min(kMfmaRows, hi - blk) + 3) >> 2;
            // two 4-row steps per iteration, both steps' loads requested before the first step's MFMAs; an odd last step runs
            // a padding step on clamped rows whose weights are 0
            for (int st0 = 0; st0 < steps; st0 += 2) {
                float a[2];
                float4 g[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int rr = 4 * (st0 + u) + kq;
                    const int n = min(blk + rr, hi - 1);   // rows past the end carry weight 0; keep their loads in bounds
                    a[u] = Wl[rr][lane & 15];
                    if (FUSED) {
                        float4 hm;
                        if (BITS) {            // packed sign bits (sample_fwd_kernel): 32 B per row and 256-channel chunk
                            const uint4 *bw = reinterpret_cast<const uint4 *>(
                                reinterpret_cast<const unsigned long long *>(mask) + ((size_t)(ch >> 8) * npts + n) * 4);
                            const uint4 b01 = bw[0], b23 = bw[1];
                            const int li = (ch & 255) >> 2;
                            const unsigned int s0 = li < 32 ? b01.x : b01.y, s1 = li < 32 ? b01.z : b01.w;
                            const unsigned int s2 = li < 32 ? b23.x : b23.y, s3 = li < 32 ? b23.z : b23.w;
                            const int sh = li & 31;
                            hm = make_float4((float)((s0 >> sh) & 1u), (float)((s1 >> sh) & 1u), (float)((s2 >> sh) & 1u),
                                             (float)((s3 >> sh) & 1u));
                        } else {
                            hm = *reinterpret_cast<const float4 *>(mask + (size_t)n * C + ch);
                        }
                        float4 v;
                        if (tlevel >= 0) {          // coarse planes from the table, the finer ones (level < tlevel) per row
                            v = Tl[Ti[rr]][wave * 16 + (lane & 15)];
#pragma unroll
                            for (int q = 0; q < kMaxMultiPlanes; ++q)
                                if (q < mp.n && mp.level[q] < tlevel) {
                                    const float4 w = *reinterpret_cast<const float4 *>(mp.g[q] + (size_t)Po[rr][q] * mp.ld[q] + ch);
                                    v.x = __fadd_rn(v.x, w.x); v.y = __fadd_rn(v.y, w.y); v.z = __fadd_rn(v.z, w.z); v.w = __fadd_rn(v.w, w.w);
                                }
                        } else {
                            v = *reinterpret_cast<const float4 *>(mp.g[0] + (size_t)Po[rr][0] * mp.ld[0] + ch);
#pragma unroll
                            for (int q = 1; q < kMaxMultiPlanes; ++q)
                                if (q < mp.n) {
                                    const float4 w = *reinterpret_cast<const float4 *>(mp.g[q] + (size_t)Po[rr][q] * mp.ld[q] + ch);
                                    v.x = __fadd_rn(v.x, w.x); v.y = __fadd_rn(v.y, w.y); v.z = __fadd_rn(v.z, w.z); v.w = __fadd_rn(v.w, w.w);
                                }
                        }
                        v.x = hm.x > 0.f ? v.x : 0.f; v.y = hm.y > 0.f ? v.y : 0.f;
                        v.z = hm.z > 0.f ? v.z : 0.f; v.w = hm.w > 0.f ? v.w : 0.f;
                        g[u] = v;
                    } else {
                        g[u] = *reinterpret_cast<const float4 *>(gout + (size_t)n * C + ch);
                    }
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], g[u].x, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], g[u].y, acc[1], 0, 0, 0);
                    acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], g[u].z, acc[2], 0, 0, 0);
                    acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], g[u].w, acc[3], 0, 0, 0);
                }
            }
        }
    }
    if (active) {
        // D layout: column = lane & 15 (-> channels 4 (lane & 15) + t of tile t), row = 4 (lane >> 4) + i = the pixel slot
        float *pbase = partial + ((size_t)cellrow * S + sp) * 9 * C;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = 4 * kq + i;
            if (m < 9) *reinterpret_cast<float4 *>(pbase + (size_t)m * C + ch) = make_float4(acc[0][i], acc[1][i], acc[2][i], acc[3][i]);
        }
    }
}

// ---- t2h_sample_bwd_from_sums with packed sign bits: the backward twin of sample_relu_cellsums_kernel -----------------------
// One WAVE walks a contiguous run of the finest cells ("children", level wl) of ONE sampling cell for a 256-channel chunk,
// lane l = channels 4 l .. 4 l + 3.  Per child the gathered gradient G = sum_q plane_q[parent_q(child)] is formed once (all
// rows of a child share it; the next child's plane rows are requested while this child's rows run); per row the four tap
// weights, the slot of the north-west tap and the four 64-bit sign words are computed / loaded lane-parallel for 64 rows at a
// time and handed to the row by readlane, the sign words ARE lane masks (v_cndmask with an SGPR pair: no bit arithmetic), and
// the row is added to the wave's 9 register accumulators (3 x 3 pixel neighbourhood of the cell) with its nine slot weights,
// five of them zero (a wave-uniform switch over the four tap positions instead makes the compiler copy the accumulator set
// between the cases: 256 VGPRs).  Nothing per row touches LDS or waits for global memory.
//   K = 0: the four waves of a workgroup share the sampling cell (a quarter of its children each -- a dense cell's 500 rows
//          would otherwise be one serial wave) and add their 9 slots in wave order through LDS;
//   K = 1: the workgroup is a 2 x 2 block of sampling cells, one per wave, added in wave order into the 4 x 4 neighbourhood
//          of the block -- 16 slots per 4 cells written and re-read instead of 36 (r = 128: 8 rows per cell on average).
// Fixed orders everywhere => deterministic.  Row values: fma(w, g, acc) with g = G or 0.
__device__ inline void fma4(float4 &a, float w, const float4 &g) {
    a.x = __fmaf_rn(w, g.x, a.x); a.y = __fmaf_rn(w, g.y, a.y); a.z = __fmaf_rn(w, g.z, a.z); a.w = __fmaf_rn(w, g.w, a.w);
}

constexpr int kWalkPlanes = 4;       // (more planes: the matrix-core kernel)
template <int K>
__global__ __launch_bounds__(256) void sample_bwd_walk_kernel(const float *__restrict__ pts, int dim,
                                                            const int32_t *__restrict__ off0, int nbits, int level, int wl,
                                                            int C, float *__restrict__ partial, MultiPlanes mp,
                                                            const unsigned long long *__restrict__ bits, int npts_m1,
                                                            const int32_t *__restrict__ order) {
    constexpr int SIDE = 3 + K, NS = SIDE * SIDE;
    __shared__ float4 NB[4 * 9][64];                                   // [wave][3 x 3 slot]: 36 KB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // 1-D grid, the cell slowest (blockIdx.x = rank * chunks + chunk): the chunks of a cell start together, the cells in the
    // order of their rank -- `order` (t2h_cell_order_build): by falling row count, longest first
    const int cpc = C >> 8;
    const int chunk = (int)(blockIdx.x % (unsigned)cpc), c0 = chunk * 256 + lane * 4;
    const unsigned rank = blockIdx.x / (unsigned)cpc;
    const int rbits = nbits - level, r = 1 << rbits, sbits = rbits - K;
    const int64_t wg = order ? (int64_t)order[rank] : (int64_t)rank;  // Morton index (batch-major) of the cell at level + K
    const int b = (int)(wg >> (2 * sbits));
    const uint32_t smk = (uint32_t)(wg & (((int64_t)1 << (2 * sbits)) - 1));
    const int d = level - wl, nchild = 1 << (2 * d);
    const uint32_t mk = K == 0 ? smk : ((smk << 2) | (uint32_t)wave);
    const int child_lo = K == 0 ? wave * (nchild >> 2) : 0, child_hi = K == 0 ? child_lo + (nchild >> 2) : nchild;
    const int cx = (int)compact1by1(mk), cy = (int)compact1by1(mk >> 1);
    const size_t obase = ((size_t)b << (2 * nbits)) + ((size_t)mk << (2 * level));
    f32x2 acc[9][2];                             // (x, y) and (z, w) halves: v_pk_fma_f32 with the slot weight broadcast
#pragma unroll
    for (int q = 0; q < 9; ++q) { acc[q][0] = f32x2{0.f, 0.f}; acc[q][1] = f32x2{0.f, 0.f}; }
    // the plane rows of a child, requested (raw) ahead of their use; a coarser plane's row is shared by 4, 16, .. consecutive
    // children and requested only when it changes.  Which row a child reads in each plane is computed for 64 children at once
    // (lane = child) and handed to the child loop by readlane: scalar compares and branches per child instead of ~30 vector
    // instructions of Morton arithmetic
    float4 raw[kWalkPlanes];
    int held[kWalkPlanes];
    int prow_l[kWalkPlanes];
#pragma unroll
    for (int q = 0; q < kWalkPlanes; ++q) { held[q] = -1; prow_l[q] = 0; raw[q] = make_float4(0.f, 0.f, 0.f, 0.f); }
    auto plane_rows = [&](int child) {                                  // lane-parallel: the rows of child `child` (this lane's)
        const uint32_t cm = (uint32_t)child;
        const int fx = (cx << d) + (int)compact1by1(cm), fy = (cy << d) + (int)compact1by1(cm >> 1);
#pragma unroll
        for (int q = 0; q < kWalkPlanes; ++q)
            if (q < mp.n) {
                const int sh = mp.level[q] - wl, rq = 1 << (nbits - mp.level[q]);
                prow_l[q] = (b * rq + (fy >> sh)) * rq + (fx >> sh);
            }
    };
    auto request = [&](int c) {                                         // c: the child's lane in the current batch (uniform)
#pragma unroll
        for (int q = 0; q < kWalkPlanes; ++q)
            if (q < mp.n) {
                const int row = __builtin_amdgcn_readlane(prow_l[q], c);
                if (row != held[q]) {
                    held[q] = row;
                    raw[q] = *reinterpret_cast<const float4 *>(mp.g[q] + (size_t)row * mp.ld[q] + c0);
                }
            }
    };
    int nb0 = 0;
    // lane i: the four tap weights of row nb0 + i; cm[k]: the rows of the batch whose north-west tap sits at position k = 2 dy + dx
    // of the cell's 3 x 3 neighbourhood.  A row's four products go to four accumulators that depend on k only, so the rows of a
    // child are taken class by class -- four straight-line loops over lane masks, 8 packed FMAs and 4 v_readlane per row -- where
    // r04 multiplied all nine accumulators by nine weights, five of them zero (18 packed FMAs, 9 v_readlane).  Per accumulator
    // the rows still arrive child by child in a fixed order: deterministic; the order inside a child is (class, row).
    float nw_l = 0.f, ne_l = 0.f, sw_l = 0.f, se_l = 0.f;
    unsigned long long cm0 = 0, cm1 = 0, cm2 = 0, cm3 = 0;
    unsigned long long w0_l = 0, w1_l = 0, w2_l = 0, w3_l = 0;
    bool have = false;
    for (int cb = child_lo; cb < child_hi; cb += 64) {
        const int ci = cb + lane;
        const int bnd_lo = ci < child_hi ? off0[obase + ((size_t)ci << (2 * wl))] : 0;
        const int bnd_hi = ci < child_hi ? off0[obase + ((size_t)(ci + 1) << (2 * wl))] : 0;
        unsigned long long todo = __ballot(bnd_hi > bnd_lo);             // the non-empty children of this batch
        plane_rows(ci);
        if (todo) request((int)__builtin_ctzll(todo));
        while (todo) {
            const int c = (int)__builtin_ctzll(todo);
            todo &= todo - 1;
            const int s = __builtin_amdgcn_readlane(bnd_lo, c), e = __builtin_amdgcn_readlane(bnd_hi, c);
            float4 G = raw[0];
#pragma unroll
            for (int q = 1; q < kWalkPlanes; ++q)
                if (q < mp.n) { G.x = __fadd_rn(G.x, raw[q].x); G.y = __fadd_rn(G.y, raw[q].y); G.z = __fadd_rn(G.z, raw[q].z); G.w = __fadd_rn(G.w, raw[q].w); }
            if (todo) request((int)__builtin_ctzll(todo));               // the next child's rows, under this child's work
            int n = s;
            while (n < e) {
                if (!have || n >= nb0 + 64) {
                    nb0 = n; have = true;
                    const int nn = min(n + lane, npts_m1);
                    const Taps tp = make_taps(pts[(size_t)nn * dim + 0], pts[(size_t)nn * dim + 1], r);
                    nw_l = __fmul_rn(tp.wx0, tp.wy0); ne_l = __fmul_rn(tp.wx1, tp.wy0);
                    sw_l = __fmul_rn(tp.wx0, tp.wy1); se_l = __fmul_rn(tp.wx1, tp.wy1);
                    const int cls = min(max(tp.y0 - cy + 1, 0), 1) * 2 + min(max(tp.x0 - cx + 1, 0), 1);
                    cm0 = __ballot(cls == 0); cm1 = __ballot(cls == 1); cm2 = __ballot(cls == 2); cm3 = __ballot(cls == 3);
                    const ulonglong2 *bw = reinterpret_cast<const ulonglong2 *>(bits + ((size_t)chunk * ((size_t)npts_m1 + 1) + nn) * 4);
                    const ulonglong2 b01 = bw[0], b23 = bw[1];
                    w0_l = b01.x; w1_l = b01.y; w2_l = b23.x; w3_l = b23.y;
                }
                const int i0 = n - nb0, cnt = min(e - n, 64 - i0);
                const unsigned long long range = (cnt >= 64 ? ~0ull : ((1ull << cnt) - 1ull)) << i0;
#define T2H_TAP(Q, W) acc[Q][0] = __builtin_elementwise_fma(f32x2{W, W}, g01, acc[Q][0]); \
                      acc[Q][1] = __builtin_elementwise_fma(f32x2{W, W}, g23, acc[Q][1])
#define T2H_CLASS(MASK, QA, QB, QC, QD)                                                                                      \
                for (unsigned long long mk = (MASK) & range; mk; mk &= mk - 1) {                                               \
                    const int i = (int)__builtin_ctzll(mk);                                                                    \
                    const f32x2 g01 = {keep_if(readlane_u64(w0_l, i), G.x), keep_if(readlane_u64(w1_l, i), G.y)};               \
                    const f32x2 g23 = {keep_if(readlane_u64(w2_l, i), G.z), keep_if(readlane_u64(w3_l, i), G.w)};               \
                    const float nw = readlane_f(nw_l, i), ne = readlane_f(ne_l, i), sw = readlane_f(sw_l, i), se = readlane_f(se_l, i); \
                    T2H_TAP(QA, nw); T2H_TAP(QB, ne); T2H_TAP(QC, sw); T2H_TAP(QD, se);                                           \
                }
                T2H_CLASS(cm0, 0, 1, 3, 4)
                T2H_CLASS(cm1, 1, 2, 4, 5)
                T2H_CLASS(cm2, 3, 4, 6, 7)
                T2H_CLASS(cm3, 4, 5, 7, 8)
#undef T2H_CLASS
#undef T2H_TAP
                n += cnt;
            }
        }
    }
    // every wave leaves its 3 x 3 neighbourhood in its own LDS slab; after ONE barrier the workgroup adds the four slabs per
    // element in wave order, starting from 0 -- the sums (and bits) of the r04 kernel, whose waves took turns behind a barrier each
#pragma unroll
    for (int sl = 0; sl < 9; ++sl) NB[wave * 9 + sl][lane] = make_float4(acc[sl][0].x, acc[sl][0].y, acc[sl][1].x, acc[sl][1].y);
    __syncthreads();
    float *pbase = partial + (size_t)wg * NS * C + chunk * 256;
    for (int e = tid; e < NS * 64; e += 256) {
        const int ws = e >> 6, l = e & 63, wy = ws / SIDE, wx = ws % SIDE;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const int ly = wy - (K ? (w >> 1) : 0), lx = wx - (K ? (w & 1) : 0);
            if ((unsigned)ly < 3u && (unsigned)lx < 3u) {
                const float4 u = NB[w * 9 + ly * 3 + lx][l];
                v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
            }
        }
        *reinterpret_cast<float4 *>(pbase + (size_t)ws * C + l * 4) = v;
    }
}

// stage 2 for K = 1: pixel (px, py) takes its slot from the (up to) four 2 x 2 cell blocks whose 4 x 4 windows hold it
__global__ __launch_bounds__(kThreads) void sample_bwd_gather4_kernel(const float *__restrict__ partial, int B, int rbits,
                                                                     int C, int lg, float *__restrict__ gplane) {
    int64_t t = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    int64_t gid = t >> lg;
    const int r = 1 << rbits, rs = r >> 1;
    if (gid >= (int64_t)B * r * r) return;
    const int b = (int)(gid >> (2 * rbits));
    const int py = (int)((gid >> rbits) & (r - 1)), px = (int)(gid & (r - 1));
    const int X = px >> 1, Y = py >> 1;
    const int X2 = (px & 1) ? X + 1 : X - 1, Y2 = (py & 1) ? Y + 1 : Y - 1;
    const int xs[2] = {min(X, X2), max(X, X2)}, ys[2] = {min(Y, Y2), max(Y, Y2)};
    for (int c = ((int)t & ((1 << lg) - 1)) * 4; c < C; c += 4 << lg) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int iy = 0; iy < 2; ++iy)
#pragma unroll
            for (int ix = 0; ix < 2; ++ix) {
                const int bx = xs[ix], by = ys[iy];
                if ((unsigned)bx < (unsigned)rs && (unsigned)by < (unsigned)rs) {
                    const size_t blk = ((size_t)b << (2 * (rbits - 1))) + morton2((uint32_t)bx, (uint32_t)by);
                    const int slot = (py - (2 * by - 1)) * 4 + (px - (2 * bx - 1));
                    const float4 u = *reinterpret_cast<const float4 *>(partial + (blk * 16 + slot) * C + c);
                    acc.x += u.x; acc.y += u.y; acc.z += u.z; acc.w += u.w;
                }
            }
        *reinterpret_cast<float4 *>(gplane + (((size_t)b * r + py) * r + px) * C + c) = acc;
    }
}

// stage 2: pixel (px,py) sums the matching slot of its (up to) 9 neighbouring cells, cells row-major, splits in order.
__global__ __launch_bounds__(kThreads) void sample_bwd_gather9_kernel(const float *__restrict__ partial, int B,
                                                                     int rbits, int C, int lg, int S,
                                                                     const float *__restrict__ addend,
                                                                     float *__restrict__ gplane) {
    int64_t t = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    int64_t gid = t >> lg;
    const int r = 1 << rbits;
    if (gid >= (int64_t)B * r * r) return;
    int b = (int)(gid >> (2 * rbits));
    int py = (int)((gid >> rbits) & (r - 1)), px = (int)(gid & (r - 1));
    for (int c = ((int)t & ((1 << lg) - 1)) * 4; c < C; c += 4 << lg) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int cy = max(py - 1, 0); cy <= min(py + 1, r - 1); ++cy)
            for (int cx = max(px - 1, 0); cx <= min(px + 1, r - 1); ++cx) {
                size_t cellrow = ((size_t)b << (2 * rbits)) + morton2((uint32_t)cx, (uint32_t)cy);
                int slot = (py - cy + 1) * 3 + (px - cx + 1);
                for (int sp = 0; sp < S; ++sp) {
                    float4 u = *reinterpret_cast<const float4 *>(partial + ((cellrow * S + sp) * 9 + slot) * C + c);
                    acc.x += u.x; acc.y += u.y; acc.z += u.z; acc.w += u.w;
                }
            }
        const size_t at = (((size_t)b * r + py) * r + px) * C + c;
        if (addend) {
            const float4 a = *reinterpret_cast<const float4 *>(addend + at);
            acc.x = __fadd_rn(a.x, acc.x); acc.y = __fadd_rn(a.y, acc.y); acc.z = __fadd_rn(a.z, acc.z); acc.w = __fadd_rn(a.w, acc.w);
        }
        *reinterpret_cast<float4 *>(gplane + at) = acc;
    }
}

// Generic backward (any r, any point order): float atomics, caller zeroes gplane.
__global__ __launch_bounds__(kThreads) void sample_bwd_atomic_kernel(const float *__restrict__ gout,
                                                                     const float *__restrict__ pts, int dim,
                                                                     int64_t npts, int N, int r, int C, int lg,
                                                                     float *__restrict__ gplane) {
    int64_t t = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    int64_t n = t >> lg;
    if (n >= npts) return;
    int b = N > 0 ? (int)(n / N) : __float_as_int(pts[n * dim + dim - 1]);    // ragged batch: tile id in the row's last float
    Taps tp = make_taps(pts[n * dim + 0], pts[n * dim + 1], r);
    float nw = tp.wx0 * tp.wy0, ne = tp.wx1 * tp.wy0, sw = tp.wx0 * tp.wy1, se = tp.wx1 * tp.wy1;
    bool x1ok = tp.x0 + 1 < r, y1ok = tp.y0 + 1 < r;
    float *base = gplane + (size_t)b * r * r * C;
    float *p00 = base + ((size_t)tp.y0 * r + tp.x0) * C;
    for (int c = (int)t & ((1 << lg) - 1); c < C; c += 1 << lg) {
        float g = gout[(size_t)n * C + c];
        atomicAdd(p00 + c, nw * g);
        if (x1ok) atomicAdd(p00 + C + c, ne * g);
        if (y1ok) atomicAdd(p00 + (size_t)r * C + c, sw * g);
        if (x1ok && y1ok) atomicAdd(p00 + (size_t)r * C + C + c, se * g);
    }
}

}  // namespace t2h

using namespace t2h;

#define T2H_DISPATCH_VEC(C, CALL4, CALL1) \
    do {                                  \
        if ((C) % 4 == 0) { CALL4; } else { CALL1; } \
    } while (0)

// ---- longest-first dispatch order of a level's cells (t2h_cell_order_build) -------------------------------------------------
// The on-chip walks give every sampling cell (or 2 x 2 block of cells) its own workgroup, whose duration grows with the cell's
// row count; workgroups start in blockIdx order, so where a dense cluster sits in the Morton order decides how long the chip
// idles at the end of the launch (Berlin-shaped tile, r = 32: 128 rows per cell on average, 951 in the densest -- 40 of 147 us).
// A counting sort of the cells by falling row count (keys: rows / quantum, capped at kOrderKeys - 1; ONE workgroup per list:
// LDS histogram, scan, scatter) lets the dispatcher start the dense cells first.  Which cell a workgroup takes never changes a result bit:
// every output element is still written by exactly one wave, with the same operations in the same order.  (Cells of equal
// count land in the order their atomics arrive: scheduling only.)
constexpr int kOrderKeys = 2048;
constexpr int kOrderRegs = 16;                  // cells per thread kept in registers between the two passes (16384 cells per list)
// blockIdx.x = 2 * (level - level_lo) + list: list 0 = the level's cells, list 1 = their 2 x 2 blocks (one level up).  `order`
// holds the lists of levels level_lo .. back to back: [cells_l][cells_l / 4] per level (t2h_cell_order_len each)
__global__ __launch_bounds__(1024) void cell_order_kernel(const int32_t *__restrict__ off0, int B, int nbits, int level_lo,
                                                         int32_t *__restrict__ order) {
    __shared__ int hist[kOrderKeys];
    __shared__ int wsum[16];
    const int tid = threadIdx.x;
    const int level = level_lo + ((int)blockIdx.x >> 1), list = (int)blockIdx.x & 1;
    int32_t *out = order;
    for (int l = level_lo; l < level; ++l) { const int64_t c = (int64_t)B << (2 * (nbits - l)); out += c + (c >> 2); }
    const int64_t cells = (int64_t)B << (2 * (nbits - level));
    const int shift = 2 * level + 2 * list;
    const int64_t n = cells >> (2 * list);
    if (list) out += cells;
    for (int k = tid; k < kOrderKeys; k += 1024) hist[k] = 0;
    // key = rows / quantum, capped: the quantum maps the list's AVERAGE row count to <= 64, so the 2048 keys reach 32 x the average
    const int quantum = (int)max((int64_t)1, ((int64_t)off0[n << shift] / max(n, (int64_t)1) + 63) / 64);
    int key[kOrderRegs];
#pragma unroll
    for (int k = 0; k < kOrderRegs; ++k) {
        const int64_t c = tid + (int64_t)k * 1024;
        key[k] = c < n ? kOrderKeys - 1 - min((off0[(c + 1) << shift] - off0[c << shift]) / quantum, kOrderKeys - 1) : -1;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kOrderRegs; ++k)
        if (key[k] >= 0) atomicAdd(&hist[key[k]], 1);                       // bucket 0 = the densest
    for (int64_t c = tid + (int64_t)kOrderRegs * 1024; c < n; c += 1024)   // (lists beyond 16384 cells: batches of tiles)
        atomicAdd(&hist[kOrderKeys - 1 - min((off0[(c + 1) << shift] - off0[c << shift]) / quantum, kOrderKeys - 1)], 1);
    __syncthreads();
    // exclusive scan of the 2048 buckets: two per thread, wave scan, then the 16 wave totals
    const int a = hist[2 * tid], bsum = a + hist[2 * tid + 1];
    int incl = bsum;
    for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(incl, o); if ((tid & 63) >= o) incl += v; }
    if ((tid & 63) == 63) wsum[tid >> 6] = incl;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < (tid >> 6); ++w) base += wsum[w];
    const int excl = base + incl - bsum;
    __syncthreads();
    hist[2 * tid] = excl; hist[2 * tid + 1] = excl + a;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kOrderRegs; ++k)
        if (key[k] >= 0) out[atomicAdd(&hist[key[k]], 1)] = (int32_t)(tid + k * 1024);
    for (int64_t c = tid + (int64_t)kOrderRegs * 1024; c < n; c += 1024)
        out[atomicAdd(&hist[kOrderKeys - 1 - min((off0[(c + 1) << shift] - off0[c << shift]) / quantum, kOrderKeys - 1)], 1)] = (int32_t)c;
}

static int check_level(const char *what, int B, int nbits, int level, int C) {
    if (B < 1 || nbits < 1 || nbits > T2H_MAX_NBITS || level < 0 || level > nbits || C < 1)
        return fail(T2H_ERR_ARG, "%s: unsupported shape (B=%d nbits=%d level=%d C=%d)", what, B, nbits, level, C);
    return T2H_OK;
}


// the matrix-core form of the sample adjoint's per-cell partials: 64-channel wave slices (T2H_CELLS_MFMA=0: the VALU form, A/B)
static bool cells_mfma(int C) {
    static const bool on = !(getenv("T2H_CELLS_MFMA") && getenv("T2H_CELLS_MFMA")[0] == '0');
    return on && C % 64 == 0;
}

// ---- coarse-level strategy: used when a cell holds >= 16 points on average and rows are float4-able ----------
struct CoarsePlan { bool use; int S; int lgG; int chunks; };
static CoarsePlan coarse_plan(int B, int N, int nbits, int level, int C, int min_pts_per_cell = 16) {
    CoarsePlan p{false, 1, 0, 1};
    if (C % 4 != 0 || N == 0) return p;
    int64_t cells = (int64_t)1 << (2 * (nbits - level));
    if ((int64_t)rows_per_tile(B, N) < (int64_t)min_pts_per_cell * cells) return p;      // (ragged batches: the average tile)
    p.use = true;
    int cc = cell_chunk_channels(C);
    p.chunks = (C + cc - 1) / cc;
    p.lgG = group_log2(cc, 4);
    // aim at >= 4096 workgroups; the channel chunks of a cell are workgroups of their own, so wide rows need fewer row
    // splits -- and every split costs a 9-slot partial row per cell to write and re-read (C = 512, r = 32: 4 -> 2 splits)
    const int64_t wgs = (int64_t)B * cells * p.chunks;
    static const int64_t min_wgs = getenv("T2H_CELLS_MIN_WGS") ? atoll(getenv("T2H_CELLS_MIN_WGS")) : 4096;
    int64_t want = (min_wgs + wgs - 1) / wgs;
    p.S = (int)(want < 1 ? 1 : (want > 8 ? 8 : want));
    return p;
}

T2H_API size_t t2h_segmean_workspace_bytes(int B, int N, int nbits, int level, int C) {
    if (B < 1 || nbits < 1 || nbits > T2H_MAX_NBITS || level < 0 || level > nbits || C < 1) return 0;
    CoarsePlan p = coarse_plan(B, N, nbits, level, C);
    if (!p.use) return 0;
    return ((size_t)B << (2 * (nbits - level))) * p.S * C * sizeof(float);
}

constexpr int kSampleBwdMinPts = 6;   // the 3x3 gather re-reads rows ~9x: switch to read-once partials early

T2H_API size_t t2h_sample_bwd_workspace_bytes(int B, int N, int nbits, int level, int C) {
    if (B < 1 || nbits < 1 || nbits > T2H_MAX_NBITS || level < 0 || level > nbits || C < 1) return 0;
    CoarsePlan p = coarse_plan(B, N, nbits, level, C, kSampleBwdMinPts);
    if (!p.use) return 0;
    return 9 * ((size_t)B << (2 * (nbits - level))) * p.S * C * sizeof(float);
}

T2H_API int t2h_pool_winner_stride(int C) { return C % 4 == 0 ? C / 4 : C; }

T2H_API int t2h_pool_max_fwd(const float *feat, int ldf, const int32_t *off0, int B, int nbits, int C, float *pooled,
                             int ldp, uint8_t *winner, t2h_stream_t stream) {
    if (!feat || !off0 || !pooled || !winner) return fail(T2H_ERR_ARG, "pool_max_fwd: null pointer");
    int rc = check_level("pool_max_fwd", B, nbits, 0, C);
    if (rc) return rc;
    if (ldf < C || ldp < C) return fail(T2H_ERR_ARG, "pool_max_fwd: row stride smaller than C");
    int64_t ncells = (int64_t)B << (2 * nbits);
    int ws = t2h_pool_winner_stride(C);
    bool v4 = C % 4 == 0 && ldf % 4 == 0 && ldp % 4 == 0 && (uintptr_t)feat % 16 == 0 && (uintptr_t)pooled % 16 == 0;
    if (C % 4 == 0 && !v4) return fail(T2H_ERR_ARG, "pool_max_fwd: rows must be 16-byte aligned when C %% 4 == 0");
    if (v4) {
        GroupCfg g = group_cfg<4>(C);
        hipLaunchKernelGGL(pool_max_fwd_kernel<4>, dim3(grid_for(ncells, g.lg)), dim3(kThreads), 0, as_stream(stream), feat,
                           off0, ncells, C, ldf, ldp, g.lg, ws, pooled, winner);
    } else {
        GroupCfg g = group_cfg<1>(C);
        hipLaunchKernelGGL(pool_max_fwd_kernel<1>, dim3(grid_for(ncells, g.lg)), dim3(kThreads), 0, as_stream(stream), feat,
                           off0, ncells, C, ldf, ldp, g.lg, ws, pooled, winner);
    }
    return check_launch("pool_max_fwd");
}

T2H_API int t2h_pool_max_bwd(const float *gpooled, int ldg, const uint8_t *winner, const int32_t *off0, int B, int nbits,
                             int C, int accumulate, float *gfeat, int ldo, t2h_stream_t stream) {
    if (!gpooled || !winner || !off0 || !gfeat) return fail(T2H_ERR_ARG, "pool_max_bwd: null pointer");
    int rc = check_level("pool_max_bwd", B, nbits, 0, C);
    if (rc) return rc;
    if (ldg < C || ldo < C) return fail(T2H_ERR_ARG, "pool_max_bwd: row stride smaller than C");
    int64_t ncells = (int64_t)B << (2 * nbits);
    int ws = t2h_pool_winner_stride(C);
    bool v4 = C % 4 == 0 && ldg % 4 == 0 && ldo % 4 == 0 && (uintptr_t)gpooled % 16 == 0 && (uintptr_t)gfeat % 16 == 0;
    if (C % 4 == 0 && !v4) return fail(T2H_ERR_ARG, "pool_max_bwd: rows must be 16-byte aligned when C %% 4 == 0");
    if (v4) {
        GroupCfg g = group_cfg<4>(C);
        hipLaunchKernelGGL(pool_max_bwd_kernel<4>, dim3(grid_for(ncells, g.lg)), dim3(kThreads), 0, as_stream(stream),
                           gpooled, winner, off0, ncells, C, ldg, ldo, g.lg, ws, accumulate, gfeat);
    } else {
        GroupCfg g = group_cfg<1>(C);
        hipLaunchKernelGGL(pool_max_bwd_kernel<1>, dim3(grid_for(ncells, g.lg)), dim3(kThreads), 0, as_stream(stream),
                           gpooled, winner, off0, ncells, C, ldg, ldo, g.lg, ws, accumulate, gfeat);
    }
    return check_launch("pool_max_bwd");
}

T2H_API int t2h_pool_mean(const float *feat, int ldf, const int32_t *off0, int B, int nbits, int C, int accumulate,
                          float *out, int ldo, t2h_stream_t stream) {
    if (!feat || !off0 || !out) return fail(T2H_ERR_ARG, "pool_mean: null pointer");
    if (feat == out) return fail(T2H_ERR_ARG, "pool_mean: in-place call (a cell's rows are re-read while they are written)");
    int rc = check_level("pool_mean", B, nbits, 0, C);
    if (rc) return rc;
    if (ldf < C || ldo < C) return fail(T2H_ERR_ARG, "pool_mean: row stride smaller than C");
    int64_t ncells = (int64_t)B << (2 * nbits);
    bool v4 = C % 4 == 0 && ldf % 4 == 0 && ldo % 4 == 0 && (uintptr_t)feat % 16 == 0 && (uintptr_t)out % 16 == 0;
    if (C % 4 == 0 && !v4) return fail(T2H_ERR_ARG, "pool_mean: rows must be 16-byte aligned when C %% 4 == 0");
    if (v4) {
        GroupCfg g = group_cfg<4>(C);
        hipLaunchKernelGGL(pool_mean_kernel<4>, dim3(grid_for(ncells, g.lg)), dim3(kThreads), 0, as_stream(stream), feat, off0,
                           ncells, C, ldf, ldo, g.lg, accumulate, out);
    } else {
        GroupCfg g = group_cfg<1>(C);
        hipLaunchKernelGGL(pool_mean_kernel<1>, dim3(grid_for(ncells, g.lg)), dim3(kThreads), 0, as_stream(stream), feat, off0,
                           ncells, C, ldf, ldo, g.lg, accumulate, out);
    }
    return check_launch("pool_mean");
}

T2H_API int t2h_pool_rows_fwd(const float *feat, int ldf, const int32_t *cell, const int32_t *off0, int64_t n_rows, int C,
                              float *pooled, int ldp, uint8_t *winner, t2h_stream_t stream) {
    if (!feat || !cell || !off0 || !pooled || !winner) return fail(T2H_ERR_ARG, "pool_rows_fwd: null pointer");
    if (n_rows < 0 || n_rows > (1LL << 30) || C < 4 || C % 4 || C > 64 || ldf < C || ldp < C || ldf % 4 || ldp % 4 ||
        (uintptr_t)feat % 16 || (uintptr_t)pooled % 16)
        return fail(T2H_ERR_ARG, "pool_rows_fwd: needs C in {4, 8, .., 64} and 16-byte aligned rows");
    if (n_rows == 0) return T2H_OK;
    const int ws = t2h_pool_winner_stride(C);
    const int lg = group_log2(C, 4);
    const int rows_per_wg = (256 >> lg) > kPoolRows ? (256 >> lg) : kPoolRows;
    const unsigned blocks = (unsigned)((n_rows + rows_per_wg - 1) / rows_per_wg);
#define T2H_POOL_FWD(LG) hipLaunchKernelGGL(pool_rows_fwd_kernel<LG>, dim3(blocks), dim3(256), 0, as_stream(stream), feat, ldf, \
                                            cell, off0, (int)n_rows, C, pooled, ldp, winner, ws)
    if (lg <= 0) T2H_POOL_FWD(0); else if (lg == 1) T2H_POOL_FWD(1); else if (lg == 2) T2H_POOL_FWD(2);
    else if (lg == 3) T2H_POOL_FWD(3); else T2H_POOL_FWD(4);
#undef T2H_POOL_FWD
    return check_launch("pool_rows_fwd");
}

T2H_API int t2h_pool_rows_bwd(const float *gpooled, int ldg, const uint8_t *winner, const int32_t *cell, const int32_t *off0,
                              int64_t n_rows, int C, int accumulate, float *gfeat, int ldo, t2h_stream_t stream) {
    if (!gpooled || !winner || !cell || !off0 || !gfeat) return fail(T2H_ERR_ARG, "pool_rows_bwd: null pointer");
    if (n_rows < 0 || n_rows > (1LL << 30) || C < 4 || C % 4 || C > 64 || ldg < C || ldo < C || ldg % 4 || ldo % 4 ||
        (uintptr_t)gpooled % 16 || (uintptr_t)gfeat % 16)
        return fail(T2H_ERR_ARG, "pool_rows_bwd: needs C in {4, 8, .., 64} and 16-byte aligned rows");
    if (n_rows == 0) return T2H_OK;
    const int ws = t2h_pool_winner_stride(C);
    const int lg = group_log2(C, 4);
    const int rows_per_wg = (256 >> lg) > kPoolRows ? (256 >> lg) : kPoolRows;
    const unsigned blocks = (unsigned)((n_rows + rows_per_wg - 1) / rows_per_wg);
#define T2H_POOL_BWD(LG) hipLaunchKernelGGL(pool_rows_bwd_kernel<LG>, dim3(blocks), dim3(256), 0, as_stream(stream), gpooled, \
                                            ldg, winner, ws, cell, off0, (int)n_rows, C, accumulate, gfeat, ldo)
    if (lg <= 0) T2H_POOL_BWD(0); else if (lg == 1) T2H_POOL_BWD(1); else if (lg == 2) T2H_POOL_BWD(2);
    else if (lg == 3) T2H_POOL_BWD(3); else T2H_POOL_BWD(4);
#undef T2H_POOL_BWD
    return check_launch("pool_rows_bwd");
}

static int segreduce_fwd(bool mean, const float *feat, const int32_t *off0, int B, int N, int nbits, int level, int C,
                         float *plane_nhwc, int ld_out, void *workspace, size_t workspace_bytes, t2h_stream_t stream) {
    if (!feat || !off0 || !plane_nhwc) return fail(T2H_ERR_ARG, "segmean_fwd: null pointer");
    if (ld_out < C || (C % 4 == 0 && ld_out % 4 != 0)) return fail(T2H_ERR_ARG, "segmean_fwd: plane row stride %d < C = %d", ld_out, C);
    int rc = check_level("segmean_fwd", B, nbits, level, C);
    if (rc) return rc;
    int64_t groups = (int64_t)B << (2 * (nbits - level));
    CoarsePlan cp = coarse_plan(B, N, nbits, level, C);
    if (cp.use) {
        size_t need = t2h_segmean_workspace_bytes(B, N, nbits, level, C);
        if (!workspace || workspace_bytes < need)
            return fail(T2H_ERR_WORKSPACE, "segmean_fwd: workspace %zu < %zu bytes", workspace_bytes, need);
        float *partial = static_cast<float *>(workspace);
        int G = 1 << cp.lgG, P = kCellThreads >> cp.lgG;
        hipLaunchKernelGGL(segmean_cells_kernel, dim3((unsigned)groups, cp.S * cp.chunks), dim3(kCellThreads),
                           (size_t)P * G * sizeof(float4), as_stream(stream), feat, off0, nbits, level, C, cp.lgG, cp.S,
                           partial);
        GroupCfg g = group_cfg<4>(C);
        hipLaunchKernelGGL(segmean_finalize_kernel, dim3(grid_for(groups, g.lg)), dim3(kThreads), 0, as_stream(stream),
                           partial, off0, B, nbits, level, C, g.lg, cp.S, plane_nhwc, mean ? 1 : 0, ld_out);
        return check_launch("segmean_fwd(coarse)");
    }
    if (mean) {
        T2H_DISPATCH_VEC(C,
            { GroupCfg g = group_cfg<4>(C);
              hipLaunchKernelGGL(segmean_fwd_kernel<4>, dim3(grid_for(groups, g.lg)), dim3(kThreads), 0, as_stream(stream),
                                 feat, off0, B, nbits, level, C, g.lg, plane_nhwc, ld_out); },
            { GroupCfg g = group_cfg<1>(C);
              hipLaunchKernelGGL(segmean_fwd_kernel<1>, dim3(grid_for(groups, g.lg)), dim3(kThreads), 0, as_stream(stream),
                                 feat, off0, B, nbits, level, C, g.lg, plane_nhwc, ld_out); });
    } else {
        T2H_DISPATCH_VEC(C,
            { GroupCfg g = group_cfg<4>(C);
              hipLaunchKernelGGL((segmean_fwd_kernel<4, false>), dim3(grid_for(groups, g.lg)), dim3(kThreads), 0, as_stream(stream),
                                 feat, off0, B, nbits, level, C, g.lg, plane_nhwc, ld_out); },
            { GroupCfg g = group_cfg<1>(C);
              hipLaunchKernelGGL((segmean_fwd_kernel<1, false>), dim3(grid_for(groups, g.lg)), dim3(kThreads), 0, as_stream(stream),
                                 feat, off0, B, nbits, level, C, g.lg, plane_nhwc, ld_out); });
    }
    return check_launch("segmean_fwd");
}

T2H_API int t2h_segmean_fwd(const float *feat, const int32_t *off0, int B, int N, int nbits, int level, int C,
                            float *plane_nhwc, void *workspace, size_t workspace_bytes, t2h_stream_t stream) {
    return segreduce_fwd(true, feat, off0, B, N, nbits, level, C, plane_nhwc, C, workspace, workspace_bytes, stream);
}

T2H_API int t2h_segsum_fwd(const float *feat, const int32_t *off0, int B, int N, int nbits, int level, int C,
                           float *plane_nhwc, int ld_plane, void *workspace, size_t workspace_bytes, t2h_stream_t stream) {
    return segreduce_fwd(false, feat, off0, B, N, nbits, level, C, plane_nhwc, ld_plane, workspace, workspace_bytes, stream);
}

T2H_API int t2h_plane_sumpool2x2(const float *fine_nhwc, int ld_fine, int B, int r_fine, int C, float *coarse_nhwc, int ld_coarse,
                                 t2h_stream_t stream) {
    if (!fine_nhwc || !coarse_nhwc) return fail(T2H_ERR_ARG, "plane_sumpool2x2: null pointer");
    if (B < 1 || r_fine < 2 || (r_fine & 1) || C < 4 || C % 4 != 0 || ((uintptr_t)fine_nhwc & 15) || ((uintptr_t)coarse_nhwc & 15) ||
        ld_fine < C || ld_coarse < C || ld_fine % 4 != 0 || ld_coarse % 4 != 0)
        return fail(T2H_ERR_ARG, "plane_sumpool2x2: needs an even resolution, C %% 4 == 0, row strides >= C and 16-byte aligned rows");
    const int rc = r_fine / 2;
    const int64_t total4 = (int64_t)B * rc * rc * (C / 4);
    hipLaunchKernelGGL(plane_sumpool2x2_kernel, dim3((unsigned)((total4 + kThreads - 1) / kThreads)), dim3(kThreads), 0,
                       as_stream(stream), fine_nhwc, total4, rc, C / 4, ld_fine, ld_coarse, coarse_nhwc);
    return check_launch("plane_sumpool2x2");
}

T2H_API int t2h_segsum_bwd_multi(const float *const *gplanes_nhwc, const int *levels, const int *lds, int n_planes,
                                 const int32_t *cell, int B, int N, int nbits, int C, const float *mask, const float *addend,
                                 float *gfeat, t2h_stream_t stream) {
    if (!gplanes_nhwc || !levels || !lds || !cell || !gfeat) return fail(T2H_ERR_ARG, "segsum_bwd_multi: null pointer");
    if (n_planes < 1 || n_planes > kMaxMultiPlanes)
        return fail(T2H_ERR_ARG, "segsum_bwd_multi: 1..%d planes, got %d", kMaxMultiPlanes, n_planes);
    if (B < 1 || nbits < 1 || nbits > T2H_MAX_NBITS || C < 1) return fail(T2H_ERR_ARG, "segsum_bwd_multi: unsupported shape");
    MultiPlanes mp{};
    mp.n = n_planes;
    for (int q = 0; q < n_planes; ++q) {
        if (!gplanes_nhwc[q] || levels[q] < 0 || levels[q] > nbits) return fail(T2H_ERR_ARG, "segsum_bwd_multi: bad plane %d", q);
        if (C % 4 == 0 && (((uintptr_t)gplanes_nhwc[q] & 15) || lds[q] % 4 != 0)) return fail(T2H_ERR_ARG, "segsum_bwd_multi: planes must be 16-byte aligned");
        if (lds[q] < C) return fail(T2H_ERR_ARG, "segsum_bwd_multi: plane %d row stride %d < C", q, lds[q]);
        mp.g[q] = gplanes_nhwc[q]; mp.level[q] = levels[q]; mp.ld[q] = lds[q];
    }
    if (C % 4 == 0 && ((mask && ((uintptr_t)mask & 15)) || (addend && ((uintptr_t)addend & 15)) || ((uintptr_t)gfeat & 15)))
        return fail(T2H_ERR_ARG, "segsum_bwd_multi: rows must be 16-byte aligned");
    const int64_t npts = rows_of(B, N);
    if (npts == 0) return T2H_OK;
    T2H_DISPATCH_VEC(C,
        { GroupCfg g = group_cfg<4>(C);
          hipLaunchKernelGGL(segsum_bwd_multi_kernel<4>, dim3(grid_for(npts, g.lg)), dim3(kThreads), 0, as_stream(stream),
                             mp, cell, npts, nbits, C, g.lg, mask, addend, gfeat); },
        { GroupCfg g = group_cfg<1>(C);
          hipLaunchKernelGGL(segsum_bwd_multi_kernel<1>, dim3(grid_for(npts, g.lg)), dim3(kThreads), 0, as_stream(stream),
                             mp, cell, npts, nbits, C, g.lg, mask, addend, gfeat); });
    return check_launch("segsum_bwd_multi");
}

T2H_API int t2h_segmean_bwd(const float *gplane_nhwc, const int32_t *cell, const int32_t *off0, int B, int N, int nbits,
                            int level, int C, float *gfeat, t2h_stream_t stream) {
    return t2h_segmean_bwd_add(gplane_nhwc, cell, off0, B, N, nbits, level, C, nullptr, gfeat, stream);
}

T2H_API int t2h_segmean_bwd_add(const float *gplane_nhwc, const int32_t *cell, const int32_t *off0, int B, int N, int nbits,
                                int level, int C, const float *addend, float *gfeat, t2h_stream_t stream) {
    if (!gplane_nhwc || !cell || !off0 || !gfeat) return fail(T2H_ERR_ARG, "segmean_bwd: null pointer");
    if (addend && C % 4 == 0 && ((uintptr_t)addend & 15)) return fail(T2H_ERR_ARG, "segmean_bwd: addend must be 16-byte aligned");
    int rc = check_level("segmean_bwd", B, nbits, level, C);
    if (rc) return rc;
    int64_t npts = rows_of(B, N);
    if (npts == 0) return T2H_OK;
    T2H_DISPATCH_VEC(C,
        { GroupCfg g = group_cfg<4>(C);
          hipLaunchKernelGGL(segmean_bwd_kernel<4>, dim3(grid_for(npts, g.lg)), dim3(kThreads), 0, as_stream(stream),
                             gplane_nhwc, cell, off0, npts, nbits, level, C, g.lg, addend, gfeat); },
        { GroupCfg g = group_cfg<1>(C);
          hipLaunchKernelGGL(segmean_bwd_kernel<1>, dim3(grid_for(npts, g.lg)), dim3(kThreads), 0, as_stream(stream),
                             gplane_nhwc, cell, off0, npts, nbits, level, C, g.lg, addend, gfeat); });
    return check_launch("segmean_bwd");
}

T2H_API int t2h_sample_fwd(const float *plane_nhwc, const float *pts, int dim, int B, int N, int r, int C, float *out,
                           t2h_stream_t stream) {
    if (!plane_nhwc || !pts || !out) return fail(T2H_ERR_ARG, "sample_fwd: null pointer");
    if (dim < 2 || B < 1 || r < 1 || C < 1 || (N < 0 && dim < 3)) return fail(T2H_ERR_ARG, "sample_fwd: unsupported shape");
    int64_t npts = rows_of(B, N);
    if (npts == 0) return T2H_OK;
    T2H_DISPATCH_VEC(C,
        { GroupCfg g = group_cfg<4>(C);
          hipLaunchKernelGGL(sample_fwd_kernel<4>, dim3(grid_for(npts, g.lg)), dim3(kThreads), 0, as_stream(stream),
                             plane_nhwc, pts, dim, npts, N, r, C, g.lg, out); },
        { GroupCfg g = group_cfg<1>(C);
          hipLaunchKernelGGL(sample_fwd_kernel<1>, dim3(grid_for(npts, g.lg)), dim3(kThreads), 0, as_stream(stream),
                             plane_nhwc, pts, dim, npts, N, r, C, g.lg, out); });
    return check_launch("sample_fwd");
}

T2H_API size_t t2h_cell_order_len(int B, int nbits, int level) {
    if (B < 1 || nbits < 1 || nbits > T2H_MAX_NBITS || level < 0 || level >= nbits) return 0;
    const size_t cells = (size_t)B << (2 * (nbits - level));
    return cells + (cells >> 2);
}

T2H_API int t2h_cell_order_build(const int32_t *off0, int B, int nbits, int level, int32_t *order, t2h_stream_t stream) {
    return t2h_cell_order_build_range(off0, B, nbits, level, level, order, stream);
}

T2H_API int t2h_cell_order_build_range(const int32_t *off0, int B, int nbits, int level_lo, int level_hi, int32_t *order,
                                       t2h_stream_t stream) {
    if (!off0 || !order) return fail(T2H_ERR_ARG, "cell_order_build: null pointer");
    if (level_lo > level_hi || t2h_cell_order_len(B, nbits, level_lo) == 0 || t2h_cell_order_len(B, nbits, level_hi) == 0)
        return fail(T2H_ERR_ARG, "cell_order_build: needs 0 <= level_lo <= level_hi < nbits");
    note_kernel("t2h::cell_order_kernel");
    hipLaunchKernelGGL(cell_order_kernel, dim3(2 * (level_hi - level_lo + 1)), dim3(1024), 0, as_stream(stream), off0, B, nbits,
                       level_lo, order);
    return check_launch("cell_order_build");
}

T2H_API int t2h_sample_relu_cellsums(const float *plane_nhwc, const float *pts, int dim, const int32_t *off0, int B, int N,
                                     int nbits, int level, int sum_level, int C, float *sums_nhwc, int ld_sums, void *sign_bits,
                                     t2h_stream_t stream) {
    return t2h_sample_relu_cellsums2(plane_nhwc, pts, dim, off0, B, N, nbits, level, sum_level, C, sums_nhwc, ld_sums, nullptr, 0,
                                     sign_bits, stream);
}

T2H_API int t2h_sample_relu_cellsums2(const float *plane_nhwc, const float *pts, int dim, const int32_t *off0, int B, int N,
                                      int nbits, int level, int sum_level, int C, float *sums_nhwc, int ld_sums,
                                      float *pooled_nhwc, int ld_pooled, void *sign_bits, t2h_stream_t stream) {
    return t2h_sample_relu_cellsums_ordered(plane_nhwc, pts, dim, off0, B, N, nbits, level, sum_level, C, sums_nhwc, ld_sums,
                                            pooled_nhwc, ld_pooled, sign_bits, nullptr, stream);
}

T2H_API int t2h_sample_relu_cellsums_ordered(const float *plane_nhwc, const float *pts, int dim, const int32_t *off0, int B, int N,
                                             int nbits, int level, int sum_level, int C, float *sums_nhwc, int ld_sums,
                                             float *pooled_nhwc, int ld_pooled, void *sign_bits, const int32_t *cell_order,
                                             t2h_stream_t stream) {
    if (!plane_nhwc || !pts || !off0 || !sums_nhwc) return fail(T2H_ERR_ARG, "sample_relu_cellsums: null pointer");
    if (pooled_nhwc && (sum_level >= level || ld_pooled < C || ld_pooled % 4 != 0 || ((uintptr_t)pooled_nhwc & 15)))
        return fail(T2H_ERR_ARG, "sample_relu_cellsums: the pooled sums need sum_level < level and 16-byte aligned rows of >= C floats");
    int rc = check_level("sample_relu_cellsums", B, nbits, level, C);
    if (rc) return rc;
    if (dim < 2 || sum_level < 0 || sum_level > level || C % 256 != 0 || ld_sums < C || ld_sums % 4 != 0 ||
        ((uintptr_t)plane_nhwc & 15) || ((uintptr_t)sums_nhwc & 15) || ((uintptr_t)sign_bits & 15))
        return fail(T2H_ERR_ARG, "sample_relu_cellsums: needs C %% 256 == 0, sum_level <= level, 16-byte aligned rows");
    if (rows_of(B, N) == 0) return fail(T2H_ERR_ARG, "sample_relu_cellsums: empty tile");
    const int64_t cells = (int64_t)B << (2 * (nbits - level));
    const int chunks = C / 256, waves = chunks < 4 ? chunks : 4;
    int groups = 1;                                                   // split the children until ~8192 workgroups exist (4096: +10 us at r = 64; 16384: +18 us at r = 32)
    const int nchild = 1 << (2 * (level - sum_level));
    static const int min_wgs = [] { const char* e = getenv("T2H_ON_CHIP_MIN_WGS"); return e ? atoi(e) : 8192; }();
    // r05: the workgroup shares one staged neighbourhood (sample_relu_cellsums_v2_kernel); T2H_CELLSUMS_V2=0: the r04 kernel (A/B)
    const char *e_v2 = getenv("T2H_CELLSUMS_V2"), *e_wgs = getenv("T2H_CELLSUMS_V2_WGS");        // (read per call: the probes flip them)
    const int v2 = e_v2 ? atoi(e_v2) : 1, v2_wgs = e_wgs ? atoi(e_wgs) : 8192;
    if (v2 && nbits - level >= 1) {
        const int npts_m1 = (int)(rows_of(B, N) - 1);
        unsigned long long *bw = static_cast<unsigned long long *>(sign_bits);
        const int quad = pooled_nhwc ? 4 : 1;                         // children a wave must hold (whole quads for the pooled sums)
        note_kernel("t2h::sample_relu_cellsums_v2_kernel");
        const bool k0 = v2 != 2 && nchild >= 4 * quad;              // K = 0: the four waves share out the cell's children
        // cell_order: [cells] the level's cells, then [cells / 4] its 2 x 2 blocks, each by falling row count
        const int32_t *order_k0 = cell_order, *order_k1 = cell_order ? cell_order + cells : nullptr;
        int gz = 1;
        if (k0) while (nchild / (4 * gz * 2) >= quad && cells * chunks * gz < v2_wgs) gz *= 2;
        const dim3 grid((unsigned)((k0 ? cells : (cells >> 2)) * chunks * gz));           // K = 1: a 2 x 2 block of cells per workgroup
#define T2H_V2_LAUNCH(KK, PP) hipLaunchKernelGGL((sample_relu_cellsums_v2_kernel<KK, PP>), grid, dim3(256), 0, as_stream(stream), \
            plane_nhwc, pts, dim, off0, nbits, level, sum_level, C, sums_nhwc, ld_sums, bw, npts_m1, pooled_nhwc, ld_pooled, k0 ? order_k0 : order_k1, gz)
        if (k0) { if (pooled_nhwc) T2H_V2_LAUNCH(0, true); else T2H_V2_LAUNCH(0, false); }
        else { if (pooled_nhwc) T2H_V2_LAUNCH(1, true); else T2H_V2_LAUNCH(1, false); }
#undef T2H_V2_LAUNCH
        return check_launch("sample_relu_cellsums(v2)");
    }
    // (with pooled sums a workgroup's share of the children must hold whole quads)
    while (groups < (pooled_nhwc ? nchild / 4 : nchild) && cells * ((chunks + 3) / 4) * groups < min_wgs) groups *= 2;
    hipLaunchKernelGGL(sample_relu_cellsums_kernel, dim3((unsigned)cells, (chunks + 3) / 4, groups), dim3(64 * waves),
                       (size_t)waves * 9 * 256 * sizeof(float), as_stream(stream), plane_nhwc, pts, dim, off0, nbits, level,
                       sum_level, C, sums_nhwc, ld_sums, static_cast<unsigned long long *>(sign_bits), (int)(rows_of(B, N) - 1),
                       pooled_nhwc, ld_pooled);
    return check_launch("sample_relu_cellsums");
}

T2H_API int t2h_sample_bwd_from_sums(const float *const *gplanes_nhwc, const int *levels, const int *lds, int n_planes,
                                     const int32_t *cell, const void *mask, int mask_is_bits, const float *pts, int dim, const int32_t *off0, int B, int N, int nbits,
                                     int level, int C, float *gplane_nhwc, void *workspace, size_t workspace_bytes,
                                     t2h_stream_t stream) {
    return t2h_sample_bwd_from_sums_ordered(gplanes_nhwc, levels, lds, n_planes, cell, mask, mask_is_bits, pts, dim, off0, B, N, nbits,
                                            level, C, gplane_nhwc, workspace, workspace_bytes, nullptr, stream);
}

T2H_API int t2h_sample_bwd_from_sums_ordered(const float *const *gplanes_nhwc, const int *levels, const int *lds, int n_planes,
                                             const int32_t *cell, const void *mask, int mask_is_bits, const float *pts, int dim,
                                             const int32_t *off0, int B, int N, int nbits, int level, int C, float *gplane_nhwc,
                                             void *workspace, size_t workspace_bytes, const int32_t *cell_order,
                                             t2h_stream_t stream) {
    if (!gplanes_nhwc || !levels || !lds || !cell || !mask || !pts || !off0 || !gplane_nhwc)
        return fail(T2H_ERR_ARG, "sample_bwd_from_sums: null pointer");
    if (n_planes < 1 || n_planes > kMaxMultiPlanes)
        return fail(T2H_ERR_ARG, "sample_bwd_from_sums: 1..%d planes, got %d", kMaxMultiPlanes, n_planes);
    int rc = check_level("sample_bwd_from_sums", B, nbits, level, C);
    if (rc) return rc;
    if (dim < 2 || C % 4 != 0 || ((uintptr_t)mask & 15)) return fail(T2H_ERR_ARG, "sample_bwd_from_sums: unsupported shape");
    if (mask_is_bits && (C % 256 != 0 || !cells_mfma(C)))
        return fail(T2H_ERR_ARG, "sample_bwd_from_sums: packed sign bits need C %% 256 == 0 (and the matrix-core partials)");
    const float *maskf = static_cast<const float *>(mask);
    CoarsePlan cp = coarse_plan(B, N, nbits, level, C, kSampleBwdMinPts);
    if (!cp.use)
        return fail(T2H_ERR_ARG, "sample_bwd_from_sums: level %d holds too few points per cell for the per-cell partials "
                                 "(t2h_sample_bwd_workspace_bytes == 0): use t2h_segsum_bwd_multi + t2h_sample_bwd", level);
    size_t need = t2h_sample_bwd_workspace_bytes(B, N, nbits, level, C);
    if (!workspace || workspace_bytes < need)
        return fail(T2H_ERR_WORKSPACE, "sample_bwd_from_sums: workspace %zu < %zu bytes", workspace_bytes, need);
    MultiPlanes mp{};
    mp.n = n_planes;
    for (int q = 0; q < n_planes; ++q) {
        if (!gplanes_nhwc[q] || levels[q] < 0 || levels[q] > nbits || ((uintptr_t)gplanes_nhwc[q] & 15) || lds[q] < C || lds[q] % 4)
            return fail(T2H_ERR_ARG, "sample_bwd_from_sums: bad plane %d", q);
        mp.g[q] = gplanes_nhwc[q]; mp.level[q] = levels[q]; mp.ld[q] = lds[q];
    }
    int64_t groups = (int64_t)B << (2 * (nbits - level));
    float *partial = static_cast<float *>(workspace);
    int G = 1 << cp.lgG, P = kCellThreads >> cp.lgG;
    // table level of the matrix-core kernel: two levels finer than the sampling level (16 entries), not finer than the
    // finest plane; -1 (no table) when no plane would go into it or the rows per workgroup would not repay building it
    static const int table_on = getenv("T2H_CELLS_TABLE") ? atoi(getenv("T2H_CELLS_TABLE")) : 1;
    int tlevel = -1;
    if (table_on && C % 256 == 0) {
        int finest = nbits;
        for (int q = 0; q < n_planes; ++q) finest = levels[q] < finest ? levels[q] : finest;
        int tl = level - 2 > finest ? level - 2 : finest;
        if (tl < 0) tl = 0;
        if (tl > level) tl = level;
        const int64_t entries = (int64_t)1 << (2 * (level - tl));
        int folded = 0;
        for (int q = 0; q < n_planes; ++q) folded += levels[q] >= tl;
        // rows per workgroup (average) against table entries: build when it replaces more row loads than it costs
        const int64_t rows_per_wg = rows_of(B, N) / (groups * cp.S > 0 ? groups * cp.S : 1);
        if (folded > 0 && rows_per_wg * table_on >= entries) tlevel = tl;
    }
    static const int walk_on = getenv("T2H_CELLS_WALK") ? atoi(getenv("T2H_CELLS_WALK")) : 1;
    if (mask_is_bits && walk_on && level >= 1 && nbits - level >= 1 && n_planes <= kWalkPlanes) {
        // walk level: the finest plane's cells, and at least one level below the sampling level (the waves split children)
        int wl = level - 1;
        for (int q = 0; q < n_planes; ++q) wl = levels[q] < wl ? levels[q] : wl;
        // few rows per sampling cell: one workgroup per 2 x 2 block of cells (walk_on == 2 / 3 force one of the forms)
        const bool blocks = walk_on == 3 || (walk_on == 1 && rows_of(B, N) < 64 * groups);
        const unsigned long long *bw = static_cast<const unsigned long long *>(mask);
        const int npts_m1 = (int)(rows_of(B, N) - 1);
        GroupCfg g = group_cfg<4>(C);
        const int32_t *order_k0 = cell_order, *order_k1 = cell_order ? cell_order + groups : nullptr;
        if (blocks) {
            hipLaunchKernelGGL(sample_bwd_walk_kernel<1>, dim3((unsigned)((groups >> 2) * (C / 256))), dim3(256), 0, as_stream(stream), pts,
                               dim, off0, nbits, level, wl, C, partial, mp, bw, npts_m1, order_k1);
            hipLaunchKernelGGL(sample_bwd_gather4_kernel, dim3(grid_for(groups, g.lg)), dim3(kThreads), 0, as_stream(stream), partial,
                               B, nbits - level, C, g.lg, gplane_nhwc);
        } else {
            hipLaunchKernelGGL(sample_bwd_walk_kernel<0>, dim3((unsigned)(groups * (C / 256))), dim3(256), 0, as_stream(stream), pts, dim,
                               off0, nbits, level, wl, C, partial, mp, bw, npts_m1, order_k0);
            hipLaunchKernelGGL(sample_bwd_gather9_kernel, dim3(grid_for(groups, g.lg)), dim3(kThreads), 0, as_stream(stream), partial,
                               B, nbits - level, C, g.lg, 1, nullptr, gplane_nhwc);
        }
        return check_launch("sample_bwd_from_sums(walk)");
    }
    if (mask_is_bits)
        hipLaunchKernelGGL((sample_bwd_cells_mfma_kernel<true, true>), dim3((unsigned)groups, cp.S * cp.chunks), dim3(kCellThreads), 0,
                           as_stream(stream), nullptr, pts, dim, off0, nbits, level, C, cp.S, partial, mp, cell, maskf, tlevel, (size_t)rows_of(B, N));
    else if (cells_mfma(C))
        hipLaunchKernelGGL(sample_bwd_cells_mfma_kernel<true>, dim3((unsigned)groups, cp.S * cp.chunks), dim3(kCellThreads), 0,
                           as_stream(stream), nullptr, pts, dim, off0, nbits, level, C, cp.S, partial, mp, cell, maskf, tlevel, (size_t)rows_of(B, N));
    else
        hipLaunchKernelGGL(sample_bwd_cells_kernel<true>, dim3((unsigned)groups, cp.S * cp.chunks), dim3(kCellThreads),
                           (size_t)P * G * sizeof(float4), as_stream(stream), nullptr, pts, dim, off0, nbits, level, C, cp.lgG, cp.S,
                           partial, mp, cell, maskf);
    GroupCfg g = group_cfg<4>(C);
    hipLaunchKernelGGL(sample_bwd_gather9_kernel, dim3(grid_for(groups, g.lg)), dim3(kThreads), 0, as_stream(stream),
                       partial, B, nbits - level, C, g.lg, cp.S, nullptr, gplane_nhwc);
    return check_launch("sample_bwd_from_sums");
}

T2H_API int t2h_sample_fwd_relu(const float *plane_nhwc, const float *pts, int dim, int B, int N, int r, int C, float *out,
                                void *sign_bits, t2h_stream_t stream) {
    if (!plane_nhwc || !pts || !out) return fail(T2H_ERR_ARG, "sample_fwd_relu: null pointer");
    if (dim < 2 || B < 1 || r < 1 || C < 1 || (N < 0 && dim < 3)) return fail(T2H_ERR_ARG, "sample_fwd_relu: unsupported shape");
    if (sign_bits && (C % 256 != 0 || ((uintptr_t)sign_bits & 15)))
        return fail(T2H_ERR_ARG, "sample_fwd_relu: the packed sign bits need C %% 256 == 0 and a 16-byte aligned buffer");
    int64_t npts = rows_of(B, N);
    if (npts == 0) return T2H_OK;
    T2H_DISPATCH_VEC(C,
        { GroupCfg g = group_cfg<4>(C);
          hipLaunchKernelGGL((sample_fwd_kernel<4, true>), dim3(grid_for(npts, g.lg)), dim3(kThreads), 0, as_stream(stream),
                             plane_nhwc, pts, dim, npts, N, r, C, g.lg, out, static_cast<unsigned long long *>(sign_bits)); },
        { GroupCfg g = group_cfg<1>(C);
          hipLaunchKernelGGL((sample_fwd_kernel<1, true>), dim3(grid_for(npts, g.lg)), dim3(kThreads), 0, as_stream(stream),
                             plane_nhwc, pts, dim, npts, N, r, C, g.lg, out); });
    return check_launch("sample_fwd_relu");
}

T2H_API int t2h_sample_bwd(const float *gout, const float *pts, int dim, const int32_t *off0, int B, int N, int nbits,
                           int level, int C, float *gplane_nhwc, void *workspace, size_t workspace_bytes,
                           t2h_stream_t stream) {
    return t2h_sample_bwd_add(gout, pts, dim, off0, B, N, nbits, level, C, nullptr, gplane_nhwc, workspace, workspace_bytes, stream);
}

T2H_API int t2h_sample_bwd_add(const float *gout, const float *pts, int dim, const int32_t *off0, int B, int N, int nbits,
                               int level, int C, const float *addend, float *gplane_nhwc, void *workspace,
                               size_t workspace_bytes, t2h_stream_t stream) {
    if (!gout || !pts || !off0 || !gplane_nhwc) return fail(T2H_ERR_ARG, "sample_bwd: null pointer");
    if (addend && ((uintptr_t)addend & 15)) return fail(T2H_ERR_ARG, "sample_bwd: addend must be 16-byte aligned");
    int rc = check_level("sample_bwd", B, nbits, level, C);
    if (rc) return rc;
    if (dim < 2) return fail(T2H_ERR_ARG, "sample_bwd: unsupported shape");
    int64_t groups = (int64_t)B << (2 * (nbits - level));
    CoarsePlan cp = coarse_plan(B, N, nbits, level, C, kSampleBwdMinPts);
    if (cp.use) {
        size_t need = t2h_sample_bwd_workspace_bytes(B, N, nbits, level, C);
        if (!workspace || workspace_bytes < need)
            return fail(T2H_ERR_WORKSPACE, "sample_bwd: workspace %zu < %zu bytes", workspace_bytes, need);
        float *partial = static_cast<float *>(workspace);
        int G = 1 << cp.lgG, P = kCellThreads >> cp.lgG;
        if (cells_mfma(C))
            hipLaunchKernelGGL(sample_bwd_cells_mfma_kernel<false>, dim3((unsigned)groups, cp.S * cp.chunks), dim3(kCellThreads), 0,
                               as_stream(stream), gout, pts, dim, off0, nbits, level, C, cp.S, partial, MultiPlanes{}, nullptr,
                               nullptr, -1, (size_t)rows_of(B, N));
        else
            hipLaunchKernelGGL(sample_bwd_cells_kernel<false>, dim3((unsigned)groups, cp.S * cp.chunks), dim3(kCellThreads),
                               (size_t)P * G * sizeof(float4), as_stream(stream), gout, pts, dim, off0, nbits, level, C,
                               cp.lgG, cp.S, partial, MultiPlanes{}, nullptr, nullptr);
        GroupCfg g = group_cfg<4>(C);
        hipLaunchKernelGGL(sample_bwd_gather9_kernel, dim3(grid_for(groups, g.lg)), dim3(kThreads), 0, as_stream(stream),
                           partial, B, nbits - level, C, g.lg, cp.S, addend, gplane_nhwc);
        return check_launch("sample_bwd(coarse)");
    }
    T2H_DISPATCH_VEC(C,
        { GroupCfg g = group_cfg<4>(C);
          hipLaunchKernelGGL(sample_bwd_kernel<4>, dim3(grid_for(groups, g.lg)), dim3(kThreads), 0, as_stream(stream),
                             gout, pts, dim, off0, B, nbits, level, C, g.lg, addend, gplane_nhwc); },
        { GroupCfg g = group_cfg<1>(C);
          hipLaunchKernelGGL(sample_bwd_kernel<1>, dim3(grid_for(groups, g.lg)), dim3(kThreads), 0, as_stream(stream),
                             gout, pts, dim, off0, B, nbits, level, C, g.lg, addend, gplane_nhwc); });
    return check_launch("sample_bwd");
}

T2H_API size_t t2h_sample_adjoint_offsets_len(int B, int nbits, int level) {
    if (B < 1 || nbits < 1 || nbits > T2H_MAX_NBITS || level < 0 || level > nbits) return 0;
    const int64_t npix = (int64_t)B << (2 * (nbits - level));
    return (size_t)(npix + 1 + (npix + 15) / 16 + 1);
}

T2H_API int t2h_sample_adjoint_build(const float *pts, int dim, const int32_t *off0, int B, int N, int nbits, int level,
                                     int32_t *offsets, void *entries, t2h_stream_t stream) {
    if (!pts || !off0 || !offsets || !entries) return fail(T2H_ERR_ARG, "sample_adjoint_build: null pointer");
    int rc = check_level("sample_adjoint_build", B, nbits, level, 1);
    if (rc) return rc;
    if (dim < 2 || rows_of(B, N) > ((int64_t)1 << 28)) return fail(T2H_ERR_ARG, "sample_adjoint_build: unsupported shape");
    if ((uintptr_t)entries & 7) return fail(T2H_ERR_ARG, "sample_adjoint_build: entries must be 8-byte aligned");
    const int64_t npix = (int64_t)B << (2 * (nbits - level));
    const int64_t nblocks = (npix + 15) / 16;
    int32_t *block_sums = offsets + npix + 1;                      // [nblocks + 1], behind the offsets proper
    hipStream_t s = as_stream(stream);
    hipLaunchKernelGGL(adjoint_scan_kernel<false>, dim3((unsigned)nblocks), dim3(kThreads), 0, s, pts, dim, off0, npix, nbits, level,
                       block_sums, offsets, static_cast<int2 *>(entries));
    hipLaunchKernelGGL(adjoint_offsets_kernel, dim3(1), dim3(1024), 0, s, block_sums, nblocks);
    hipLaunchKernelGGL(adjoint_scan_kernel<true>, dim3((unsigned)nblocks), dim3(kThreads), 0, s, pts, dim, off0, npix, nbits, level,
                       block_sums, offsets, static_cast<int2 *>(entries));
    return check_launch("sample_adjoint_build");
}

T2H_API int t2h_sample_bwd_adjoint(const float *gout, const int32_t *offsets, const void *entries, int B, int nbits, int level,
                                   int C, const float *addend, float *gplane_nhwc, t2h_stream_t stream) {
    if (!gout || !offsets || !entries || !gplane_nhwc) return fail(T2H_ERR_ARG, "sample_bwd_adjoint: null pointer");
    if (addend && ((uintptr_t)addend & 15)) return fail(T2H_ERR_ARG, "sample_bwd_adjoint: addend must be 16-byte aligned");
    int rc = check_level("sample_bwd_adjoint", B, nbits, level, C);
    if (rc) return rc;
    const int rbits = nbits - level;
    const int64_t npix = (int64_t)B << (2 * rbits);
    T2H_DISPATCH_VEC(C,
        { GroupCfg g = group_cfg<4>(C);
          hipLaunchKernelGGL(sample_bwd_adjoint_kernel<4>, dim3(grid_for(npix, g.lg)), dim3(kThreads), 0, as_stream(stream), gout,
                             offsets, static_cast<const int2 *>(entries), npix, rbits, C, g.lg, addend, gplane_nhwc); },
        { GroupCfg g = group_cfg<1>(C);
          hipLaunchKernelGGL(sample_bwd_adjoint_kernel<1>, dim3(grid_for(npix, g.lg)), dim3(kThreads), 0, as_stream(stream), gout,
                             offsets, static_cast<const int2 *>(entries), npix, rbits, C, g.lg, addend, gplane_nhwc); });
    return check_launch("sample_bwd_adjoint");
}

T2H_API int t2h_sample_bwd_atomic(const float *gout, const float *pts, int dim, int B, int N, int r, int C,
                                  float *gplane_nhwc, t2h_stream_t stream) {
    if (!gout || !pts || !gplane_nhwc) return fail(T2H_ERR_ARG, "sample_bwd_atomic: null pointer");
    if (dim < 2 || B < 1 || N < 0 || r < 1 || C < 1) return fail(T2H_ERR_ARG, "sample_bwd_atomic: unsupported shape (equal-N batches only)");
    int64_t npts = rows_of(B, N);
    if (npts == 0) return T2H_OK;
    int lg = group_log2(C, 1);
    hipLaunchKernelGGL(sample_bwd_atomic_kernel, dim3(grid_for(npts, lg)), dim3(kThreads), 0, as_stream(stream), gout, pts,
                       dim, npts, N, r, C, lg, gplane_nhwc);
    return check_launch("sample_bwd_atomic");
}
