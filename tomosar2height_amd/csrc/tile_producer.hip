// Tile producer: the point-cloud half of TomoSARDataset.__getitem__ (reference dataset.py:233-278,
// utils/crop_cloud.py:8-29) on a chunk cloud resident in HBM:
//   1. strict 2-D crop of the float64 world points to the 512 m window          (x > min && x < max, same for y)
//   2. z shift = min z of the cropped points ('local_min', conf/dataset/base.yaml)
//   3. float64 normalise to [0,1]: xn = (x - cx)/sx + 0.5, yn likewise, zn = (z - zmin)/sz; cast to float32
//   4. re-crop on the float32 result (strictly inside (0,1)), dataset.py:278
// as ONE stable compaction (point order preserved, like torch.where): count pass (+ atomic min on an order-preserving
// integer image of z: min is order independent, so this is deterministic), block-offset scan, scatter pass.
#include "t2h_common.h"

namespace t2h {

constexpr int kCropThreads = 256;
constexpr int kCropItems = 8;
constexpr int kCropTile = kCropThreads * kCropItems;

struct CropArgs {
    double minx, maxx, miny, maxy;   // window (world)
    double cx, cy;                   // window centre
    double sx, sy, sz;               // scale_mat diagonal: patch size / normalised range, z_bound span
    int rot, flip;                   // augmentation (dataset.py:253-270): rot x 90 deg clockwise about z, flip -1 / 0 (x) / 1 (y)
};

__device__ inline unsigned long long ordered_bits(double v) {
    unsigned long long b = (unsigned long long)__double_as_longlong(v);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ inline double from_ordered_bits(unsigned long long k) {
    unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)b);
}

__device__ inline bool crop_test(const double *__restrict__ p, const CropArgs &a, bool &first, float &xn, float &yn) {
    double x = p[0], y = p[1];
    first = (x > a.minx) && (x < a.maxx) && (y > a.miny) && (y < a.maxy);
    // centred coordinates in [-0.5, 0.5], then flip_mat @ rot_mat (dataset.py:268: rotation by -90 deg per step about z:
    // (u, v) -> (v, -u); flips negate one axis), then the shift to [0, 1]; exact sign / swap operations in float64
    double u = (x - a.cx) / a.sx, v = (y - a.cy) / a.sy;
    for (int k = 0; k < a.rot; ++k) { const double t = u; u = v; v = -t; }
    if (a.flip == 0) u = -u;
    if (a.flip == 1) v = -v;
    xn = (float)(u + 0.5);
    yn = (float)(v + 0.5);
    return first && xn > 0.0f && xn < 1.0f && yn > 0.0f && yn < 1.0f;
}

__global__ __launch_bounds__(kCropThreads) void crop_count_kernel(const double *__restrict__ pts, long long P, CropArgs a,
                                                                  unsigned int *__restrict__ block_counts,
                                                                  unsigned long long *__restrict__ zmin_key) {
    __shared__ unsigned int cnt;
    __shared__ unsigned long long zmin_s;
    if (threadIdx.x == 0) { cnt = 0; zmin_s = ~0ull; }
    __syncthreads();
    long long base = (long long)blockIdx.x * kCropTile;
    unsigned int mine = 0;
    unsigned long long zk = ~0ull;
#pragma unroll
    for (int i = 0; i < kCropItems; ++i) {
        long long idx = base + (long long)i * kCropThreads + threadIdx.x;
        if (idx < P) {
            bool first; float xn, yn;
            bool keep = crop_test(pts + idx * 3, a, first, xn, yn);
            mine += keep ? 1u : 0u;
            if (first) zk = min(zk, ordered_bits(pts[idx * 3 + 2]));
        }
    }
    atomicAdd(&cnt, mine);
    atomicMin(&zmin_s, zk);
    __syncthreads();
    if (threadIdx.x == 0) {
        block_counts[blockIdx.x] = cnt;
        if (zmin_s != ~0ull) atomicMin(zmin_key, zmin_s);
    }
}

// exclusive scan of the block counts in place; total -> *count_out  (single workgroup)
__global__ __launch_bounds__(256) void crop_scan_kernel(unsigned int *__restrict__ block_counts, int nblocks,
                                                        int *__restrict__ count_out) {
    __shared__ unsigned int tot[256];
    int per = (nblocks + 255) / 256;
    int lo = threadIdx.x * per, hi = min(nblocks, lo + per);
    unsigned int s = 0;
    for (int j = lo; j < hi; ++j) { unsigned int v = block_counts[j]; block_counts[j] = s; s += v; }
    tot[threadIdx.x] = s;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        unsigned int add = threadIdx.x >= off ? tot[threadIdx.x - off] : 0u;
        __syncthreads();
        tot[threadIdx.x] += add;
        __syncthreads();
    }
    unsigned int excl = tot[threadIdx.x] - s;
    for (int j = lo; j < hi; ++j) block_counts[j] += excl;
    if (threadIdx.x == 255) *count_out = (int)tot[255];
}

// element order inside a workgroup tile is (wave, item, lane): wave w owns points [w*512, (w+1)*512) of the tile
__global__ __launch_bounds__(kCropThreads) void crop_scatter_kernel(const double *__restrict__ pts, long long P, CropArgs a,
                                                                    const unsigned int *__restrict__ block_offsets,
                                                                    const unsigned long long *__restrict__ zmin_key,
                                                                    float *__restrict__ out, int *__restrict__ src_index) {
    __shared__ unsigned int wave_cnt[kCropThreads / kWave];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const unsigned long long lt = (1ull << lane) - 1ull;
    long long base = (long long)blockIdx.x * kCropTile + (long long)wave * (kWave * kCropItems);
    const double zmin = from_ordered_bits(*zmin_key);
    bool keep[kCropItems];
    float xn[kCropItems], yn[kCropItems];
    unsigned int rank[kCropItems];
    unsigned int running = 0;
#pragma unroll
    for (int i = 0; i < kCropItems; ++i) {
        long long idx = base + (long long)i * kWave + lane;
        bool first = false;
        keep[i] = idx < P && crop_test(pts + idx * 3, a, first, xn[i], yn[i]);
        unsigned long long m = __ballot(keep[i]);
        rank[i] = running + (unsigned int)__popcll(m & lt);
        running += (unsigned int)__popcll(m);
    }
    if (lane == 0) wave_cnt[wave] = running;
    __syncthreads();
    unsigned int wbase = block_offsets[blockIdx.x];
    for (int w = 0; w < wave; ++w) wbase += wave_cnt[w];
#pragma unroll
    for (int i = 0; i < kCropItems; ++i) {
        if (keep[i]) {
            long long idx = base + (long long)i * kWave + lane;
            size_t o = (size_t)(wbase + rank[i]);
            out[o * 3 + 0] = xn[i];
            out[o * 3 + 1] = yn[i];
            out[o * 3 + 2] = (float)((pts[idx * 3 + 2] - zmin) / a.sz);
            if (src_index) src_index[o] = (int)idx;
        }
    }
}

}  // namespace t2h

using namespace t2h;

T2H_API size_t t2h_tile_crop_workspace_bytes(int64_t P) {
    if (P < 0) return 0;
    size_t nblocks = (size_t)((P + kCropTile - 1) / kCropTile);
    return (nblocks + 1) * sizeof(unsigned int) + 16;
}

T2H_API int t2h_tile_crop_normalise(const double *chunk, int64_t P, double min_x, double min_y, double max_x, double max_y,
                                    double scale_x, double scale_y, double scale_z, float *out, int32_t *src_index,
                                    int32_t *count, double *z_shift, void *workspace, size_t workspace_bytes,
                                    t2h_stream_t stream) {
    return t2h_tile_crop_normalise_aug(chunk, P, min_x, min_y, max_x, max_y, scale_x, scale_y, scale_z, 0, -1, out, src_index,
                                       count, z_shift, workspace, workspace_bytes, stream);
}

T2H_API int t2h_tile_crop_normalise_aug(const double *chunk, int64_t P, double min_x, double min_y, double max_x, double max_y,
                                        double scale_x, double scale_y, double scale_z, int rot_times, int flip_dim, float *out,
                                        int32_t *src_index, int32_t *count, double *z_shift, void *workspace,
                                        size_t workspace_bytes, t2h_stream_t stream) {
    if (!chunk || !out || !count || !z_shift || !workspace) return fail(T2H_ERR_ARG, "tile_crop_normalise: null pointer");
    if (rot_times < 0 || rot_times > 3 || flip_dim < -1 || flip_dim > 1) return fail(T2H_ERR_ARG, "tile_crop_normalise: bad augmentation");
    if (P < 0 || P >= ((int64_t)1 << 31) || !(scale_x > 0) || !(scale_y > 0) || !(scale_z > 0))
        return fail(T2H_ERR_ARG, "tile_crop_normalise: bad argument");
    if (workspace_bytes < t2h_tile_crop_workspace_bytes(P)) return fail(T2H_ERR_WORKSPACE, "tile_crop_normalise: workspace too small");
    hipStream_t s = as_stream(stream);
    // z_shift doubles as the 64-bit min key while the kernels run; converted back by the caller-visible copy below
    unsigned long long *zkey = reinterpret_cast<unsigned long long *>(z_shift);
    if (hipMemsetAsync(zkey, 0xff, sizeof(unsigned long long), s) != hipSuccess) return check_launch("tile_crop/memset");
    if (hipMemsetAsync(count, 0, sizeof(int32_t), s) != hipSuccess) return check_launch("tile_crop/memset");
    if (P == 0) return T2H_OK;
    int nblocks = (int)((P + kCropTile - 1) / kCropTile);
    unsigned int *block_counts = static_cast<unsigned int *>(workspace);
    CropArgs a{min_x, max_x, min_y, max_y, (min_x + max_x) / 2.0, (min_y + max_y) / 2.0, scale_x, scale_y, scale_z, rot_times, flip_dim};
    hipLaunchKernelGGL(crop_count_kernel, dim3(nblocks), dim3(kCropThreads), 0, s, chunk, (long long)P, a, block_counts, zkey);
    hipLaunchKernelGGL(crop_scan_kernel, dim3(1), dim3(256), 0, s, block_counts, nblocks, count);
    hipLaunchKernelGGL(crop_scatter_kernel, dim3(nblocks), dim3(kCropThreads), 0, s, chunk, (long long)P, a, block_counts, zkey,
                       out, src_index);
    return check_launch("tile_crop_normalise");
}

// z_shift holds the order-preserving key after t2h_tile_crop_normalise; this turns it into the double (NaN if no point)
__global__ void crop_zkey_to_double_kernel(double *z) {
    unsigned long long k = *reinterpret_cast<unsigned long long *>(z);
    *z = (k == ~0ull) ? __longlong_as_double(0x7ff8000000000000ll) : from_ordered_bits(k);
}

T2H_API int t2h_tile_crop_finish(double *z_shift, t2h_stream_t stream) {
    if (!z_shift) return fail(T2H_ERR_ARG, "tile_crop_finish: null pointer");
    hipLaunchKernelGGL(crop_zkey_to_double_kernel, dim3(1), dim3(1), 0, as_stream(stream), z_shift);
    return check_launch("tile_crop_finish");
}

// ---- raster half of the tile: DSM / image patch (dataset.py:291-328) ---------------------------------------------------
//   patch = raster[:, r0 : r0 + ph, c0 : c0 + pw];  rot90(rot, [-1, -2]);  flip(-1) if flip == 0, flip(-2) if flip == 1;
//   .float().flip(-2)                                 (row 0 of the result = south, like the points' y axis)
// as one gather: every output element reads its source pixel through the composed index map.
namespace t2h {
template <typename T>
__global__ __launch_bounds__(256) void raster_patch_kernel(const T *__restrict__ src, int C, int H, int W, int r0, int c0,
                                                           int ph, int pw, int rot, int flip, float *__restrict__ out) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)C * ph * pw) return;
    const int w = (int)(i % pw), h = (int)((i / pw) % ph), c = (int)(i / ((long long)pw * ph));
    int h1 = ph - 1 - h, w1 = w;                    // undo the final .flip(-2)
    if (flip == 0) w1 = pw - 1 - w1;                // augmentation flips
    if (flip == 1) h1 = ph - 1 - h1;
    int hs = h1, ws = w1;                           // undo rot90(rot, dims = [W, H]) (square patches when rot > 0)
    if (rot == 1) { hs = ph - 1 - w1; ws = h1; }
    if (rot == 2) { hs = ph - 1 - h1; ws = pw - 1 - w1; }
    if (rot == 3) { hs = w1; ws = pw - 1 - h1; }
    out[i] = (float)src[((size_t)c * H + r0 + hs) * W + c0 + ws];
}
}  // namespace t2h

T2H_API int t2h_raster_patch(const void *raster, int is_f64, int C, int H, int W, int row0, int col0, int ph, int pw,
                             int rot_times, int flip_dim, float *out, t2h_stream_t stream) {
    if (!raster || !out) return fail(T2H_ERR_ARG, "raster_patch: null pointer");
    if (C < 1 || H < 1 || W < 1 || ph < 1 || pw < 1 || rot_times < 0 || rot_times > 3 || flip_dim < -1 || flip_dim > 1)
        return fail(T2H_ERR_ARG, "raster_patch: bad argument");
    if (rot_times % 2 == 1 && ph != pw) return fail(T2H_ERR_ARG, "raster_patch: a quarter-turn needs a square patch");
    if (row0 < 0 || col0 < 0 || row0 + ph > H || col0 + pw > W)
        return fail(T2H_ERR_ARG, "raster_patch: rows [%d, %d) x cols [%d, %d) leave the %d x %d raster", row0, row0 + ph, col0,
                    col0 + pw, H, W);
    const long long n = (long long)C * ph * pw;
    const dim3 grid((unsigned)((n + 255) / 256));
    if (is_f64)
        hipLaunchKernelGGL(raster_patch_kernel<double>, grid, dim3(256), 0, as_stream(stream), static_cast<const double *>(raster), C,
                           H, W, row0, col0, ph, pw, rot_times, flip_dim, out);
    else
        hipLaunchKernelGGL(raster_patch_kernel<float>, grid, dim3(256), 0, as_stream(stream), static_cast<const float *>(raster), C, H,
                           W, row0, col0, ph, pw, rot_times, flip_dim, out);
    return check_launch("raster_patch");
}
