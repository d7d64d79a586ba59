// Tile index: bin at the finest plane resolution, stable LSD radix sort by Morton cell code, CSR offsets.
//
// Replaces, once per tile, the reference's 9 coordinate2index calls and xy clone/index copies
// (utils/coordinate.py:12-28; pointnet.py:69-70; alto.py:79-80,189-190) and gives every later
// scatter/gather a contiguous-segment view of the points (see include/t2h.h).
//
// HBM-bound integer work: 8-bit digits, one 2048-key tile per 256-thread workgroup, per-wave digit
// counters in LDS, match-any via 8 ballots for a STABLE in-wave rank (no atomics in the scatter, so the
// sorted order -- and with it every fp32 sum downstream -- is identical from run to run).
#include "t2h_common.h"

namespace t2h {

char *err_buf() {
    static thread_local char buf[256] = "no error";
    return buf;
}
int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 256, fmt, ap);
    va_end(ap);
    return code;
}
static thread_local const char *g_noted_kernel = "";
void note_kernel(const char *name) { g_noted_kernel = name; }
const char *noted_kernel() { return g_noted_kernel; }
int check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(T2H_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
    return T2H_OK;
}

// rows of tile b of a batch: uniform (N > 0: rows [b N, (b + 1) N)) or ragged (N <= 0: rows [start[b], start[b + 1]), B <= kMaxRagged)
constexpr int kMaxRagged = T2H_MAX_RAGGED_TILES;
struct Spans { int start[kMaxRagged + 1]; };
__device__ inline void tile_span(const Spans &sp, int N, int b, size_t &base, int &n) {
    if (N > 0) { base = (size_t)b * N; n = N; }
    else { base = (size_t)sp.start[b]; n = sp.start[b + 1] - sp.start[b]; }
}

constexpr int kSortThreads = 256;
constexpr int kSortItems = 8;
constexpr int kSortTile = kSortThreads * kSortItems;  // 2048 keys per workgroup
constexpr int kSortWaves = kSortThreads / kWave;

// ---- coordinate2index, bit exact (operator-level drop-in) ------------------------------------------
__global__ void coordinate2index_kernel(const float *__restrict__ pts, int stride, int64_t total, int reso,
                                        int64_t *__restrict__ index) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    float fx = __fmul_rn(pts[i * stride + 0], (float)reso);
    float fy = __fmul_rn(pts[i * stride + 1], (float)reso);
    int64_t ix = (int64_t)fx, iy = (int64_t)fy;  // trunc toward zero == Tensor.long()
    index[i] = ix + (int64_t)reso * iy;
}

// ---- keys: Morton code of the finest-level cell; out-of-domain points are clamped and counted ---------
__global__ void tile_keys_kernel(const float *__restrict__ cloud, int dim, int N, Spans sp, int nbits,
                                 uint32_t *__restrict__ keys, int32_t *__restrict__ status) {
    int b = blockIdx.y;
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    size_t base; int n;
    tile_span(sp, N, b, base, n);
    if (i >= n) return;
    const float *p = cloud + (base + i) * dim;
    float x = p[0], y = p[1];
    int R = 1 << nbits;
    bool ok = (x >= 0.0f) && (x < 1.0f) && (y >= 0.0f) && (y < 1.0f);  // false for NaN too
    int ix = ok ? (int)__fmul_rn(x, (float)R) : (int)fminf(fmaxf(x * (float)R, 0.0f), (float)(R - 1));
    int iy = ok ? (int)__fmul_rn(y, (float)R) : (int)fminf(fmaxf(y * (float)R, 0.0f), (float)(R - 1));
    if (!ok) { atomicAdd(status, 1); atomicAdd(status + 1, 1); }
    keys[base + i] = morton2((uint32_t)ix, (uint32_t)iy);
}

// ---- radix pass 1/3: per-workgroup digit histogram -----------------------------------------------------
__global__ __launch_bounds__(kSortThreads) void sort_hist_kernel(const uint32_t *__restrict__ keys, int N, Spans sp, int shift,
                                                                 uint32_t *__restrict__ blockhist, int nblk) {
    __shared__ uint32_t h[256];
    int b = blockIdx.y, blk = blockIdx.x, tid = threadIdx.x;
    size_t tbase; int n;
    tile_span(sp, N, b, tbase, n);
    h[tid] = 0;
    __syncthreads();
    int base = blk * kSortTile;
#pragma unroll
    for (int i = 0; i < kSortItems; ++i) {
        int idx = base + i * kSortThreads + tid;
        if (idx < n) atomicAdd(&h[(keys[tbase + idx] >> shift) & 255u], 1u);
    }
    __syncthreads();
    blockhist[((size_t)b * nblk + blk) * 256 + tid] = h[tid];        // [tile][workgroup][digit]: coalesced here and in the scan
}

// ---- radix pass 2/3: exclusive scan over (digit, workgroup), one workgroup per tile of the batch ------------
__global__ __launch_bounds__(256) void sort_scan_kernel(uint32_t *__restrict__ blockhist, int nblk) {
    __shared__ uint32_t tot[256];
    int b = blockIdx.x, tid = threadIdx.x;
    uint32_t *p = blockhist + (size_t)b * nblk * 256 + tid;          // digit tid of workgroup j at p[j * 256]
    uint32_t s = 0;
    for (int j = 0; j < nblk; ++j) { uint32_t v = p[(size_t)j * 256]; p[(size_t)j * 256] = s; s += v; }
    tot[tid] = s;
    __syncthreads();
    // Hillis-Steele inclusive scan over the 256 digit totals
    for (int off = 1; off < 256; off <<= 1) {
        uint32_t add = tid >= off ? tot[tid - off] : 0u;
        __syncthreads();
        tot[tid] += add;
        __syncthreads();
    }
    uint32_t excl = tot[tid] - s;
    for (int j = 0; j < nblk; ++j) p[(size_t)j * 256] += excl;
}

// ---- radix pass 3/3: stable scatter --------------------------------------------------------------------
// Element order inside a workgroup tile is (wave, item, lane): wave w owns keys [w*512, (w+1)*512) of
// the tile, item i covers 64 consecutive keys.  rank = #equal digits before me in that order.
__global__ __launch_bounds__(kSortThreads) void sort_scatter_kernel(const uint32_t *__restrict__ keys_in,
                                                                    const uint32_t *__restrict__ vals_in, int N_, Spans sp,
                                                                    int shift, const uint32_t *__restrict__ blockhist,
                                                                    int nblk, uint32_t *__restrict__ keys_out,
                                                                    uint32_t *__restrict__ vals_out) {
    __shared__ uint32_t wc[kSortWaves][256];
    int b = blockIdx.y, blk = blockIdx.x, tid = threadIdx.x;
    int wave = tid >> 6, lane = tid & 63;
    size_t tbase; int N;
    tile_span(sp, N_, b, tbase, N);
#pragma unroll
    for (int w = 0; w < kSortWaves; ++w) wc[w][tid] = 0;
    __syncthreads();

    int base = blk * kSortTile + wave * (kWave * kSortItems);
    uint32_t key[kSortItems], val[kSortItems], rank[kSortItems];
    bool valid[kSortItems];
    const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
    for (int i = 0; i < kSortItems; ++i) {
        int idx = base + i * kWave + lane;
        valid[i] = idx < N;
        key[i] = valid[i] ? keys_in[tbase + idx] : 0u;
        val[i] = valid[i] ? (vals_in ? vals_in[tbase + idx] : (uint32_t)idx) : 0u;
        uint32_t d = (key[i] >> shift) & 255u;
        unsigned long long mask = __ballot(valid[i]);
#pragma unroll
        for (int bit = 0; bit < 8; ++bit) {
            bool set = (d >> bit) & 1u;
            unsigned long long bm = __ballot(valid[i] && set);
            mask &= set ? bm : ~bm;
        }
        uint32_t old = 0;
        if (valid[i]) {
            old = wc[wave][d];                        // same address for every lane of the match set
            int leader = __ffsll((long long)mask) - 1;
            if (lane == leader) wc[wave][d] = old + (uint32_t)__popcll(mask);
            rank[i] = old + (uint32_t)__popcll(mask & lt);
        } else {
            rank[i] = 0;
        }
    }
    __syncthreads();
    {   // thread tid == digit: turn per-wave counts into global destinations
        uint32_t g = blockhist[((size_t)b * nblk + blk) * 256 + tid];
#pragma unroll
        for (int w = 0; w < kSortWaves; ++w) { uint32_t c = wc[w][tid]; wc[w][tid] = g; g += c; }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < kSortItems; ++i) {
        if (valid[i]) {
            uint32_t d = (key[i] >> shift) & 255u;
            uint32_t dst = wc[wave][d] + rank[i];
            keys_out[tbase + dst] = key[i];
            vals_out[tbase + dst] = val[i];
        }
    }
}

// ---- finalize: gather points into sorted order, emit perm / cell codes / CSR offsets ---------------------
__global__ void tile_finalize_kernel(const float *__restrict__ cloud, int dim, int B, int N_, Spans sp, int nbits,
                                     const uint32_t *__restrict__ keys, const uint32_t *__restrict__ vals,
                                     float *__restrict__ pts_sorted, int out_dim, int32_t *__restrict__ perm,
                                     int32_t *__restrict__ cell, int32_t *__restrict__ off0) {
    int b = blockIdx.y;
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    size_t tbase; int N;
    tile_span(sp, N_, b, tbase, N);
    if (i >= N) return;
    const int M0 = 1 << (2 * nbits);
    size_t g = tbase + i;
    uint32_t key = keys[g], src = vals[g];
    const float *p = cloud + (tbase + src) * dim;
    float *q = pts_sorted + g * out_dim;
    for (int d = 0; d < dim; ++d) q[d] = p[d];
    if (out_dim > dim) q[out_dim - 1] = __int_as_float(b);               // ragged batches: the row's tile (t2h_common.h rows_of)
    perm[g] = (int32_t)src;
    cell[g] = b * M0 + (int32_t)key;
    int prev = (i == 0) ? -1 : (int)keys[g - 1];
    for (int m = prev + 1; m <= (int)key; ++m) off0[(size_t)b * M0 + m] = (int32_t)g;
    if (i == N - 1) {
        for (int m = (int)key + 1; m < M0; ++m) off0[(size_t)b * M0 + m] = (int32_t)(g + 1);
        if (b == B - 1) off0[(size_t)B * M0] = (int32_t)(g + 1);
    }
}

static size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

// t2h_debug_poison_lds: the whole LDS of a CU written with quiet-NaN patterns, the workgroup held for a few microseconds so that a
// round of them lands on distinct CUs (each takes a CU's whole LDS)
__global__ __launch_bounds__(256) void poison_lds_kernel(int words, unsigned *sink) {
    extern __shared__ unsigned lds_words[];
    for (int i = threadIdx.x; i < words; i += 256) lds_words[i] = 0x7fc00000u | (unsigned)(i & 0xffff);
    __syncthreads();
    for (int k = 0; k < 40; ++k) __builtin_amdgcn_s_sleep(127);
    if (lds_words[(threadIdx.x * 97) % words] == 1u) sink[0] = 1u;       // (keeps the stores)
}

}  // namespace t2h

using namespace t2h;

T2H_API int t2h_abi_version(void) { return T2H_ABI_VERSION; }
T2H_API const char *t2h_last_error_string(void) { return err_buf(); }
T2H_API const char *t2h_last_kernel_name(void) { return noted_kernel(); }
T2H_API void t2h_clear_kernel_name(void) { note_kernel(""); }

T2H_API int t2h_debug_poison_lds(t2h_stream_t stream) {
    static unsigned *sink = nullptr;
    static int bytes = 0;
    if (!sink) {
        if (hipMalloc(&sink, 256) != hipSuccess) return fail(T2H_ERR_LAUNCH, "debug_poison_lds: hipMalloc failed");
        bytes = 160 * 1024;
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(poison_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) {
            (void)hipGetLastError();
            bytes = 64 * 1024;
        }
    }
    hipLaunchKernelGGL(poison_lds_kernel, dim3(512), dim3(256), (size_t)bytes, as_stream(stream), bytes / 4, sink);
    return check_launch("debug_poison_lds");
}

T2H_API int t2h_coordinate2index(const float *pts, int stride, int64_t total, int reso, int64_t *index,
                                 t2h_stream_t stream) {
    if (!pts || !index || stride < 2 || reso < 1 || total < 0) return fail(T2H_ERR_ARG, "coordinate2index: bad argument");
    if (total == 0) return T2H_OK;
    int64_t blocks = (total + 255) / 256;
    hipLaunchKernelGGL(coordinate2index_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), pts, stride,
                       total, reso, index);
    return check_launch("coordinate2index");
}

static size_t tile_ws_bytes(int B, size_t rows, int max_n, int nbits) {
    int nblk = (max_n + kSortTile - 1) / kSortTile;
    if (nblk < 1) nblk = 1;
    return 4 * align256((rows > 0 ? rows : 1) * sizeof(uint32_t)) + align256((size_t)B * 256 * nblk * sizeof(uint32_t));
}

T2H_API size_t t2h_tile_workspace_bytes(int B, int N, int nbits) {
    if (B < 1 || N < 0 || nbits < 1 || nbits > T2H_MAX_NBITS) return 0;
    return tile_ws_bytes(B, (size_t)B * (size_t)N, N, nbits);
}

T2H_API size_t t2h_tile_ragged_workspace_bytes(int B, int64_t total_rows, int max_rows, int nbits) {
    if (B < 1 || B > kMaxRagged || total_rows < 0 || max_rows < 0 || nbits < 1 || nbits > T2H_MAX_NBITS) return 0;
    return tile_ws_bytes(B, (size_t)total_rows, max_rows, nbits);
}

// shared body: N > 0 uniform tiles, else the spans
static int tile_build_body(const float *cloud, int dim, int B, int N, const Spans &sp, int max_n, size_t bn, int nbits,
                           float *pts_sorted, int out_dim, int32_t *perm, int32_t *cell, int32_t *off0, int32_t *status,
                           void *workspace, hipStream_t s) {
    int nblk = (max_n + kSortTile - 1) / kSortTile;
    char *w = static_cast<char *>(workspace);
    uint32_t *keys_a = reinterpret_cast<uint32_t *>(w); w += align256(bn * 4);
    uint32_t *keys_b = reinterpret_cast<uint32_t *>(w); w += align256(bn * 4);
    uint32_t *vals_a = reinterpret_cast<uint32_t *>(w); w += align256(bn * 4);
    uint32_t *vals_b = reinterpret_cast<uint32_t *>(w); w += align256(bn * 4);
    uint32_t *blockhist = reinterpret_cast<uint32_t *>(w);

    hipLaunchKernelGGL(tile_keys_kernel, dim3((max_n + 255) / 256, B), dim3(256), 0, s, cloud, dim, N, sp, nbits, keys_a, status);
    int rc = check_launch("tile_keys");
    if (rc) return rc;

    const int passes = (2 * nbits + 7) / 8;
    uint32_t *kin = keys_a, *kout = keys_b, *vin = nullptr, *vout = vals_a;
    for (int p = 0; p < passes; ++p) {
        int shift = 8 * p;
        hipLaunchKernelGGL(sort_hist_kernel, dim3(nblk, B), dim3(kSortThreads), 0, s, kin, N, sp, shift, blockhist, nblk);
        hipLaunchKernelGGL(sort_scan_kernel, dim3(B), dim3(256), 0, s, blockhist, nblk);
        hipLaunchKernelGGL(sort_scatter_kernel, dim3(nblk, B), dim3(kSortThreads), 0, s, kin, vin, N, sp, shift, blockhist,
                           nblk, kout, vout);
        rc = check_launch("tile_sort");
        if (rc) return rc;
        uint32_t *t = kin; kin = kout; kout = t;
        vin = vout;
        vout = (vout == vals_a) ? vals_b : vals_a;
    }
    hipLaunchKernelGGL(tile_finalize_kernel, dim3((max_n + 255) / 256, B), dim3(256), 0, s, cloud, dim, B, N, sp, nbits, kin, vin,
                       pts_sorted, out_dim, perm, cell, off0);
    return check_launch("tile_finalize");
}

T2H_API int t2h_tile_build(const float *cloud, int dim, int B, int N, int nbits, float *pts_sorted, int32_t *perm,
                           int32_t *cell, int32_t *off0, int32_t *status, void *workspace, size_t workspace_bytes,
                           t2h_stream_t stream) {
    if (!cloud || !pts_sorted || !perm || !cell || !off0 || !status || !workspace)
        return fail(T2H_ERR_ARG, "tile_build: null pointer");
    if (dim < 2 || B < 1 || N < 0 || nbits < 1 || nbits > T2H_MAX_NBITS)
        return fail(T2H_ERR_ARG, "tile_build: unsupported shape (dim=%d B=%d N=%d nbits=%d)", dim, B, N, nbits);
    if ((int64_t)B * N >= (int64_t)1 << 31 || ((int64_t)B << (2 * nbits)) >= (int64_t)1 << 31)
        return fail(T2H_ERR_ARG, "tile_build: B*N or B*4^nbits exceeds int32");
    if (workspace_bytes < t2h_tile_workspace_bytes(B, N, nbits))
        return fail(T2H_ERR_WORKSPACE, "tile_build: workspace %zu < %zu bytes", workspace_bytes,
                    t2h_tile_workspace_bytes(B, N, nbits));
    hipStream_t s = as_stream(stream);
    const size_t M0 = (size_t)1 << (2 * nbits);
    if (hipMemsetAsync(status, 0, sizeof(int32_t), s) != hipSuccess) return check_launch("tile_build/memset status");
    if (N == 0) {
        if (hipMemsetAsync(off0, 0, (B * M0 + 1) * sizeof(int32_t), s) != hipSuccess)
            return check_launch("tile_build/memset off0");
        return T2H_OK;
    }
    Spans sp{};
    return tile_build_body(cloud, dim, B, N, sp, N, (size_t)B * N, nbits, pts_sorted, dim, perm, cell, off0, status, workspace, s);
}

// A batch of B tiles with DIFFERENT point counts (the tiles of the reference's accumulation window, trainer.py:72-89, which it
// runs one by one because N varies: tomosar2height.yaml:40): `cloud` [total, dim] holds the tiles back to back, tile b =
// rows [starts[b], starts[b + 1]) (HOST array of B + 1 ints, starts[0] = 0, every tile non-empty).  Same sort per tile and same
// outputs as t2h_tile_build -- rows of tile b stay in [starts[b], starts[b + 1]), `cell` = b 4^nbits + Morton code, `perm` = the
// row's index inside its tile -- plus the tile index as int bits in float `out_dim - 1` of every pts row (out_dim = dim + 1).
// Every other entry point takes such a batch as (B, N = -total).
T2H_API int t2h_tile_build_ragged(const float *cloud, int dim, int B, const int32_t *starts, int nbits, float *pts_sorted,
                                  int out_dim, int32_t *perm, int32_t *cell, int32_t *off0, int32_t *status, void *workspace,
                                  size_t workspace_bytes, t2h_stream_t stream) {
    if (!cloud || !starts || !pts_sorted || !perm || !cell || !off0 || !status || !workspace)
        return fail(T2H_ERR_ARG, "tile_build_ragged: null pointer");
    if (dim < 2 || B < 1 || B > kMaxRagged || nbits < 1 || nbits > T2H_MAX_NBITS || out_dim != dim + 1)
        return fail(T2H_ERR_ARG, "tile_build_ragged: unsupported shape (dim=%d out_dim=%d B=%d (max %d) nbits=%d)", dim, out_dim, B,
                    kMaxRagged, nbits);
    Spans sp{};
    int max_n = 0;
    if (starts[0] != 0) return fail(T2H_ERR_ARG, "tile_build_ragged: starts[0] must be 0");
    for (int b = 0; b < B; ++b) {
        const int n = starts[b + 1] - starts[b];
        if (n < 1) return fail(T2H_ERR_ARG, "tile_build_ragged: tile %d has %d points (every tile must be non-empty)", b, n);
        if (n > max_n) max_n = n;
        sp.start[b] = starts[b];
    }
    sp.start[B] = starts[B];
    const int64_t total = starts[B];
    if (total >= (int64_t)1 << 31 || ((int64_t)B << (2 * nbits)) >= (int64_t)1 << 31)
        return fail(T2H_ERR_ARG, "tile_build_ragged: rows or B*4^nbits exceed int32");
    const size_t need = t2h_tile_ragged_workspace_bytes(B, total, max_n, nbits);
    if (workspace_bytes < need) return fail(T2H_ERR_WORKSPACE, "tile_build_ragged: workspace %zu < %zu bytes", workspace_bytes, need);
    hipStream_t s = as_stream(stream);
    if (hipMemsetAsync(status, 0, sizeof(int32_t), s) != hipSuccess) return check_launch("tile_build_ragged/memset status");
    return tile_build_body(cloud, dim, B, 0, sp, max_n, (size_t)total, nbits, pts_sorted, out_dim, perm, cell, off0, status, workspace, s);
}
