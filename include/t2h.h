/* t2h.h -- C ABI of libt2h_hip.so: the MI355X (gfx950) implementation of the dual-topology
 * point-cloud hot path of zhu-xlab/tomosar2height.
 *
 * The reference has no FFI of its own: the seam is a handful of Python operator calls
 * (SURVEY.md section 8b).  Each entry point below names the reference call(s) it replaces
 * (paths relative to the reference root).  Conventions:
 *
 *   - every pointer is a DEVICE pointer owned by the caller (PyTorch's caching allocator in
 *     the shipped binding); the library allocates nothing and keeps no state, so every call
 *     is stream-ordered, re-entrant and hipGraph-capturable;
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream);
 *   - return value: 0 on success, a negative T2H_ERR_* code otherwise (never throws, never
 *     exits); t2h_last_error_string() describes the last failure on the calling thread;
 *   - all feature arrays are fp32; point features are POINT-MAJOR [B*N, C] rows in the
 *     CELL-SORTED point order produced by t2h_tile_build; planes exchanged with the point side
 *     are PIXEL-MAJOR [B, r, r, C] (NHWC), the layout in which a cell / pixel is one contiguous
 *     row -- t2h_nchw_to_nhwc / t2h_nhwc_to_nchw convert from/to the conv side's NCHW.
 *
 * Cell-sorted order.  t2h_tile_build bins the points at the finest plane resolution
 * R = 2^nbits exactly as coordinate2index does (utils/coordinate.py:12-28: trunc(x*R),
 * ix + R*iy), then stably sorts them by the Morton code of (ix, iy).  A cell of ANY coarser
 * level k (resolution R >> k, the resolutions the ALTO U-Net visits) is then one contiguous
 * run of points: points of level-k cell (cx, cy) are [off0[M << 2k], off0[(M+1) << 2k]) with
 * M = morton(cx, cy) and off0 the CSR offsets of the finest level.  One sort per tile serves
 * the 9 coordinate2index calls, 4 scatter_max, 9 scatter_mean and 8 grid_sample backward
 * passes of one forward/backward.
 */
#ifndef T2H_H_
#define T2H_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define T2H_OK 0
#define T2H_ERR_ARG (-1)      /* bad argument (null pointer, unsupported size) */
#define T2H_ERR_LAUNCH (-2)   /* HIP reported an error at launch */
#define T2H_ERR_WORKSPACE (-3) /* workspace too small */

#define T2H_ABI_VERSION 19
#define T2H_MAX_NBITS 10      /* finest plane resolution up to 1024 */
#define T2H_MAX_RAGGED_TILES 64 /* tiles per ragged batch (t2h_tile_build_ragged) */

typedef void *t2h_stream_t;

int t2h_abi_version(void);
const char *t2h_last_error_string(void);
/* Profiling aid (no reference counterpart): the name of the main device kernel the last entry point on this
 * thread launched ("" if it does not record one), so that a host-side timeline can aggregate per kernel symbol
 * exactly like `rocprofv3 --kernel-trace --stats` does.  t2h_clear_kernel_name() resets it. */
const char *t2h_last_kernel_name(void);
void t2h_clear_kernel_name(void);
/* Test instrumentation (no reference counterpart): fills the LDS of every CU that is free on `stream`'s turn with NaN
 * patterns -- one 160 KB workgroup per CU, two rounds.  A kernel that reads LDS before writing it (or before the barrier
 * behind the write) normally finds the previous workgroup's values there, which are the right ones when that workgroup was
 * the same kernel's; behind this call it finds NaNs.  T2H_POISON_LDS=1 makes the Python layer issue it before every entry
 * point.  It catches LDS that is never written; a race that is usually won (the writer usually first) stays invisible. */
int t2h_debug_poison_lds(t2h_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * coordinate2index(x, reso)                                     utils/coordinate.py:12-28
 * pts: [total, stride] floats (x, y first); index: [total] int64 = trunc(x*reso) + reso*trunc(y*reso).
 * Any reso >= 1 (the operator-level drop-in; the fused path uses t2h_tile_build instead). */
int t2h_coordinate2index(const float *pts, int stride, int64_t total, int reso, int64_t *index,
                         t2h_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Tile index: replaces the 9 coordinate2index calls + clone()[..., [0, 1]] copies per forward
 * (pointnet.py:69-70; alto.py:79-80, 189-190) and provides the segment structure used instead
 * of torch_scatter's atomics.
 *
 *   cloud       [B, N, dim] input points, x/y in [0,1) (dataset.py:278 guarantees (0,1))
 *   pts_sorted  [B*N, dim]  the same points in cell-sorted order
 *   perm        [B*N]       index (within its tile, 0..N-1) of the original point at each sorted slot
 *   cell        [B*N]       b * 4^nbits + morton(ix, iy) of each sorted point (finest level)
 *   off0        [B * 4^nbits + 1]  CSR offsets into the sorted order (global, i.e. including b*N)
 *   status      [2]         [0]: number of points of THIS call with x or y outside [0,1) or NaN (clamped into
 *                           the border cell; the reference would index out of range) -- zeroed by every call;
 *                           [1]: the same count, ADDED to whatever the caller left there (never zeroed by the
 *                           library), so that a caller can keep one running total across many tiles and read
 *                           it once, without a synchronisation per tile
 * The sort is stable (original order inside a cell), so results are run-to-run deterministic. */
size_t t2h_tile_workspace_bytes(int B, int N, int nbits);
int t2h_tile_build(const float *cloud, int dim, int B, int N, int nbits, float *pts_sorted, int32_t *perm,
                   int32_t *cell, int32_t *off0, int32_t *status, void *workspace, size_t workspace_bytes,
                   t2h_stream_t stream);
/* RAGGED batches: B tiles with different point counts in one index -- the tiles of the reference's 64-tile accumulation
 * window (trainer.py:72-89), which it runs one at a time only because N varies per tile (tomosar2height.yaml:40,
 * generator.py:44).  `cloud` [total, dim] holds the tiles back to back; tile b = rows [starts[b], starts[b+1]) (`starts`: HOST
 * array of B + 1 ints, starts[0] = 0, every tile non-empty, B <= T2H_MAX_RAGGED_TILES).  Per tile the same stable sort and
 * the same outputs as t2h_tile_build (rows of tile b stay in its row range; cell = b 4^nbits + Morton code; perm = index
 * inside the tile), and pts_sorted rows have out_dim = dim + 1 floats: the last one holds the tile index as int bits.
 * EVERY point-side entry point below takes such a batch as (B, N = -total): N < 0 means "-N rows in all, tile boundaries
 * from off0 / cell / the pts rows" (sums inside a tile are those of the single-tile call: bit-identical results per tile). */
size_t t2h_tile_ragged_workspace_bytes(int B, int64_t total_rows, int max_rows, int nbits);
int t2h_tile_build_ragged(const float *cloud, int dim, int B, const int32_t *starts, int nbits, float *pts_sorted, int out_dim,
                          int32_t *perm, int32_t *cell, int32_t *off0, int32_t *status, void *workspace,
                          size_t workspace_bytes, t2h_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * pool_local (scatter_max + gather)                              pointnet.py:92-99
 * feat/pooled [B*N, C] sorted rows with row strides ldf/ldp (>= C: the pooled half of the reference's
 * torch.cat([net, pooled]) at pointnet.py:78 is written straight into its column slice of the next
 * block's input); every point receives the per-channel max over the points of
 * its finest-level cell.  Ties: the first point in ORIGINAL order wins (pytorch-scatter CPU rule;
 * the stable sort keeps original order inside a cell).
 * winner: [B*N, t2h_pool_winner_stride(C)] bytes; bit j of byte (n, g) is set iff point n is the
 * arg-max of its cell for channel g*vec + j (vec = 4 if C % 4 == 0 else 1).
 * Backward (gather-backward = per-cell sum of gpooled, routed to the arg-max point):
 * gfeat = (accumulate ? gfeat : 0) + routed gradient. */
int t2h_pool_winner_stride(int C);
int t2h_pool_max_fwd(const float *feat, int ldf, const int32_t *off0, int B, int nbits, int C, float *pooled,
                     int ldp, uint8_t *winner, t2h_stream_t stream);
int t2h_pool_max_bwd(const float *gpooled, int ldg, const uint8_t *winner, const int32_t *off0, int B, int nbits,
                     int C, int accumulate, float *gfeat, int ldo, t2h_stream_t stream);
/* The OPERATOR itself, in the reference's own layout, for callers of the operators rather than of the modules
 * (SURVEY 8b "operator-level seam"):
 *     out, arg = torch_scatter.scatter_max(src [B, C, N], index [B, 1, N], dim_size = R * R)              pointnet.py:95
 * feat: the points' features POINT-major in the CALLER's order, [B * N, ld] (src is a permuted view of such a tensor at
 * the reference's call site); perm / off0: from t2h_tile_build on the cell centres of `index` (equal-N batches).
 * val [B, C, 4^nbits] fp32, arg [B, C, 4^nbits] int64, cell p = ix + R iy as coordinate2index numbers them.  Cells no
 * point falls into: value 0, arg = N.  Ties: the first point in the caller's order (pytorch-scatter's CPU rule; NaN and
 * -inf never win).  Backward (pytorch-scatter: the gradient of `out` goes to `arg` only): gsrc [B, N, C] point-major is
 * zeroed and gsrc[b, arg[b, c, p], c] = gval[b, c, p] for every touched cell -- no atomics, a point lies in one cell. */
int t2h_scatter_max_fwd(const float *feat, int ld, const int32_t *perm, const int32_t *off0, int B, int N, int nbits, int C,
                        float *val, int64_t *arg, t2h_stream_t stream);
int t2h_scatter_max_bwd(const float *gval, const int64_t *arg, int B, int C, int N, int64_t cells, float *gsrc,
                        t2h_stream_t stream);
/* scatter_type='mean' (pointnet.py:55-56: self.scatter = scatter_mean; no shipped config selects it): out[n] = mean of
 * feat over the rows of n's finest-level cell -- torch_scatter.scatter_mean (sum in point order, one division by the
 * count) followed by the gather of pointnet.py:98.  The operator is its own adjoint: the backward is the same call on the
 * gradient, accumulate != 0 adding into out (the pooled half's gradient folded into the left half of the block input's
 * gradient, as t2h_pool_max_bwd does).  feat != out. */
int t2h_pool_mean(const float *feat, int ldf, const int32_t *off0, int B, int nbits, int C, int accumulate, float *out,
                  int ldo, t2h_stream_t stream);
/* The same two operators balanced over ROWS instead of cells, for callers that hold the per-row cell ids of
 * t2h_tile_build (cell [n_rows], values index off0): a workgroup owns 128 consecutive sorted rows, reduces the cell
 * segments inside it from LDS and the outside rows of the (at most two) cells crossing its border cooperatively, so dense
 * cells no longer serialise on one lane group.  Same results (values, first-occurrence winners); C in {4, 8, .., 64}.
 * At N = 131072, C = 32: backward 14 us against 26 us (the trunk uses it), forward 24 against 13 us (it does not). */
int t2h_pool_rows_fwd(const float *feat, int ldf, const int32_t *cell, const int32_t *off0, int64_t n_rows, int C,
                      float *pooled, int ldp, uint8_t *winner, t2h_stream_t stream);
int t2h_pool_rows_bwd(const float *gpooled, int ldg, const uint8_t *winner, const int32_t *cell, const int32_t *off0,
                      int64_t n_rows, int C, int accumulate, float *gfeat, int ldo, t2h_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * generate_plane_features (scatter_mean into a zero plane)       pointnet.py:101-111; alto.py:76-88,187-197
 * level k: resolution r = 2^(nbits-k).  plane [B, r, r, C] (NHWC); empty cells are written 0
 * (no separate memset).  Sum order inside a cell = sorted order (deterministic).
 * Backward: gfeat[n] = gplane[cell_k(n)] / count(cell_k(n)).
 * Coarse levels (>= 16 points per cell on average) run as per-(cell, split) partial sums in `workspace`
 * (t2h_segmean_workspace_bytes; 0 = not needed, NULL accepted) + a finalise pass. */
size_t t2h_segmean_workspace_bytes(int B, int N, int nbits, int level, int C);
int t2h_segmean_fwd(const float *feat, const int32_t *off0, int B, int N, int nbits, int level, int C,
                    float *plane_nhwc, void *workspace, size_t workspace_bytes, t2h_stream_t stream);
int t2h_segmean_bwd(const float *gplane_nhwc, const int32_t *cell, const int32_t *off0, int B, int N,
                    int nbits, int level, int C, float *gfeat, t2h_stream_t stream);
/* Per-cell SUMS instead of means (same kernels, no division), their 2x2 pooling to the next coarser resolution (a cell is
 * the union of its four children: plane [B, r, r, C] -> [B, r/2, r/2, C]) and the joint backward of sums taken at up to eight
 * resolutions of the same rows: gfeat[n] = (mask[n] > 0 ?) sum_q gplane_q[cell_{level_q}(n)] (+ addend[n]).
 * They serve the "deferred point features" form of the ALTO point update (alto.py:121-130, 245-255): the per-point features
 * c_k = fc_comm.2(h_k) + fc_c(c_{k-1}) are linear in the hidden activations h_j (j <= k), and so is their scatter_mean, hence
 *   scatter_mean(c_k) = ( sum_j cellsum_r(h_j) A_{k,j}^T ) / count + const_k        (A_{k,j} = Wc_k .. Wc_{j+1} W1_j)
 * is formed on the r^2 pixels from per-cell sums of the hidden activations and c_k itself is never computed on the N points. */
int t2h_segsum_fwd(const float *feat, const int32_t *off0, int B, int N, int nbits, int level, int C,
                   float *plane_nhwc, int ld_plane, void *workspace, size_t workspace_bytes, t2h_stream_t stream);
int t2h_plane_sumpool2x2(const float *fine_nhwc, int ld_fine, int B, int r_fine, int C, float *coarse_nhwc, int ld_coarse,
                         t2h_stream_t stream);
int t2h_segsum_bwd_multi(const float *const *gplanes_nhwc, const int *levels, const int *lds, int n_planes, const int32_t *cell,
                         int B, int N, int nbits, int C, const float *mask, const float *addend, float *gfeat,
                         t2h_stream_t stream);
/* Row strides (`ld_*`, in floats, >= C): a plane may be a COLUMN BLOCK of a wider [B r r, K] matrix -- the per-cell sums of all
 * sources of the deferred form sit side by side per resolution, so that scatter_mean(c_k) is ONE product over K = sum K_j. */
/* Grid-side glue of the deferred form (csrc/deferred.hip): the points-per-cell plane [B, r, r] (float, row-major) of an ALTO
 * level from the tile CSR; scatter_mean's division / empty-cell rule / composed bias applied to a product of per-cell sums,
 *   raster[p, :] = acc[p, :] / max(cnt[p], 1) + [cnt[p] > 0] * cvec[:]                       (alto.py:76-88: count clamped to 1),
 * and its backward (dacc = g / max(cnt, 1); dcvec = sum over non-empty cells of g, two-stage fixed-order column sum;
 * either output may be NULL). */
int t2h_cell_counts(const int32_t *off0, int B, int nbits, int level, float *cnt, t2h_stream_t stream);
int t2h_mean_bias_fwd(const float *acc, const float *cnt, const float *cvec, int64_t P, int C, float *out, t2h_stream_t stream);
size_t t2h_mean_bias_bwd_workspace_bytes(int64_t P, int C);
int t2h_mean_bias_bwd(const float *g, const float *cnt, int64_t P, int C, float *dacc, float *dcvec, void *workspace,
                      size_t workspace_bytes, t2h_stream_t stream);
/* The forward of a deferred level at a coarse sampling resolution in ONE pass, with the hidden activations never leaving the
 * chip: per (cell of the sampling level, 256-channel chunk) one wave stages the cell's 3 x 3 pixel neighbourhood in LDS, walks the
 * cell's rows and emits the per-cell SUMS of relu(sample(plane)) at the finer resolution `sum_level` (into a column block with row
 * stride ld_sums) and the packed sign bits (layout of t2h_sample_fwd_relu).  Bit-identical to t2h_sample_fwd_relu +
 * t2h_segsum_fwd; for levels with many points per cell (the walk is sequential inside a cell).  C % 256 == 0.
 * `sign_bits` may be NULL (forward-only callers, generator.py:142-147: nothing reads the signs, nothing is written). */
int t2h_sample_relu_cellsums(const float *plane_nhwc, const float *pts, int dim, const int32_t *off0, int B, int N, int nbits,
                             int level, int sum_level, int C, float *sums_nhwc, int ld_sums, void *sign_bits, t2h_stream_t stream);
/* ... and, from the same registers, the sums one level coarser (`pooled_nhwc`, [B, R_sum/2, R_sum/2, C] with row stride ld_pooled;
 * may be NULL; needs sum_level < level): what t2h_plane_sumpool2x2 would form from `sums_nhwc` with the same bits, without
 * re-reading it. */
int t2h_sample_relu_cellsums2(const float *plane_nhwc, const float *pts, int dim, const int32_t *off0, int B, int N, int nbits,
                              int level, int sum_level, int C, float *sums_nhwc, int ld_sums, float *pooled_nhwc, int ld_pooled,
                              void *sign_bits, t2h_stream_t stream);
/* ... with a dispatch order: `cell_order` (may be NULL) from t2h_cell_order_build for the same tile and level -- the level's
 * cells, then its 2 x 2 blocks of cells, each list by FALLING row count ([cells + cells / 4] int32, t2h_cell_order_len).  The
 * on-chip walks give each cell (or block) its own workgroup whose duration grows with the cell's rows; started longest-first the
 * dense cells no longer decide how long the chip idles at the end of the launch.  Scheduling only: results are bit-identical
 * with and without an order.  (No reference counterpart: torch_scatter / grid_sample have no such structure to balance.) */
size_t t2h_cell_order_len(int B, int nbits, int level);
int t2h_cell_order_build(const int32_t *off0, int B, int nbits, int level, int32_t *order, t2h_stream_t stream);
/* ... for levels level_lo .. level_hi in ONE launch; `order` holds their lists back to back (t2h_cell_order_len each). */
int t2h_cell_order_build_range(const int32_t *off0, int B, int nbits, int level_lo, int level_hi, int32_t *order,
                               t2h_stream_t stream);
int t2h_sample_relu_cellsums_ordered(const float *plane_nhwc, const float *pts, int dim, const int32_t *off0, int B, int N,
                                     int nbits, int level, int sum_level, int C, float *sums_nhwc, int ld_sums, float *pooled_nhwc,
                                     int ld_pooled, void *sign_bits, const int32_t *cell_order, t2h_stream_t stream);
/* t2h_segsum_bwd_multi folded into the sample adjoint's per-cell partial kernel: gplane [B, r, r, C] =
 * S^T ( (mask > 0) * sum_q gplanes_q[cell_q(.)] ) without the [N, C] hidden gradient ever being written.  Only where the level
 * takes the per-cell partials (t2h_sample_bwd_workspace_bytes > 0); same workspace.  `mask`: the hidden activations [N, C]
 * (mask_is_bits = 0) or their packed sign bits (mask_is_bits = 1, layout of t2h_sample_fwd_relu; C % 256 == 0).  With sign bits
 * and <= 4 planes of distinct levels the kernel is a row walk -- the gathered gradient formed once per finest cell, the sign
 * words used as lane masks, 3 x 3 register accumulators per wave, per-workgroup partials for one cell or a 2 x 2 block of cells
 * -- otherwise slots x rows x channels products on the matrix cores with the gather in the row load.  Deterministic either way. */
int t2h_sample_bwd_from_sums(const float *const *gplanes_nhwc, const int *levels, const int *lds, int n_planes,
                             const int32_t *cell, const void *mask, int mask_is_bits, const float *pts, int dim, const int32_t *off0, int B, int N,
                             int nbits, int level, int C, float *gplane_nhwc, void *workspace, size_t workspace_bytes,
                             t2h_stream_t stream);
/* ... with the dispatch order of t2h_cell_order_build (`cell_order`, may be NULL; used by the row-walk form): see
 * t2h_sample_relu_cellsums_ordered.  Bit-identical results with and without. */
int t2h_sample_bwd_from_sums_ordered(const float *const *gplanes_nhwc, const int *levels, const int *lds, int n_planes,
                                     const int32_t *cell, const void *mask, int mask_is_bits, const float *pts, int dim,
                                     const int32_t *off0, int B, int N, int nbits, int level, int C, float *gplane_nhwc,
                                     void *workspace, size_t workspace_bytes, const int32_t *cell_order, t2h_stream_t stream);
/* The same with `addend` [B*N, C] (may be NULL) added to the result: the point features of a level feed both the
 * rasterisation and the next level's fc_c (alto.py:123-130), so their gradient is a sum of two -- formed here instead of
 * by an extra elementwise pass (gfeat may alias addend). */
int t2h_segmean_bwd_add(const float *gplane_nhwc, const int32_t *cell, const int32_t *off0, int B, int N, int nbits,
                        int level, int C, const float *addend, float *gfeat, t2h_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * sample_plane_feature: F.grid_sample(c, 2*xy-1, bilinear, border, align_corners=True)
 *                                                                alto.py:90-95, 199-205
 * plane [B, r, r, C] NHWC, pts [B*N, dim] -> out [B*N, C]  (== the reference's [B,C,N] result
 * after its transpose at alto.py:122).  Any r >= 1.
 * Backward w.r.t. the plane (points carry no gradient):
 *   t2h_sample_bwd        deterministic gather over the 3x3 neighbouring cells of each pixel using
 *                         the tile CSR; requires pts in cell-sorted order and r == 2^(nbits-level);
 *                         at coarse levels each row is instead read once into per-cell 3x3 pixel partials
 *                         (`workspace`, t2h_sample_bwd_workspace_bytes) that a second pass gathers;
 *   t2h_sample_bwd_atomic any r / any point order, float atomics (order-dependent rounding);
 *                         gplane must be zeroed by the caller. */
int t2h_sample_fwd(const float *plane_nhwc, const float *pts, int dim, int B, int N, int r, int C, float *out,
                   t2h_stream_t stream);
/* max(sample, 0): the hidden activations relu(fc_comm.0(sampled)) of alto.py:121-123 / 245-248 when fc_comm.0 was applied
 * to the plane's PIXELS first (nn.Linear commutes with the bilinear interpolation, whose tap weights sum to 1 under
 * padding_mode='border', align_corners=True):  relu(W0 sample(P) + b0) == relu(sample(P W0^T + b0)).  `plane_nhwc` is then the
 * [B, r, r, 2C] plane P W0^T + b0; where N >> r^2 this replaces an [N, C] x [C, 2C] product by an [r^2, C] x [C, 2C] one. */
int t2h_sample_fwd_relu(const float *plane_nhwc, const float *pts, int dim, int B, int N, int r, int C, float *out,
                        void *sign_bits, t2h_stream_t stream);
/* `sign_bits` (may be NULL; C % 256 == 0): the pattern out > 0 packed 1 bit per element, [C / 256][B*N][4] 64-bit words (chunk-
 * major: the rows of a 256-channel chunk q are consecutive 32-byte records), bit l of word j <=> channel 256 q + 4 l + j -- all
 * the backward needs of the hidden activations (t2h_sample_bwd_from_sums with
 * mask_is_bits = 1 reads 32 B instead of 1 KB per row and chunk, and the activations need not be kept for the backward). */
size_t t2h_sample_bwd_workspace_bytes(int B, int N, int nbits, int level, int C);
int t2h_sample_bwd(const float *gout, const float *pts, int dim, const int32_t *off0, int B, int N, int nbits,
                   int level, int C, float *gplane_nhwc, void *workspace, size_t workspace_bytes,
                   t2h_stream_t stream);
/* The same with `addend` [B, r, r, C] (may be NULL) added to the result: a conv output that is sampled also feeds the next
 * level's residual convolution (alto.py:104-114, 233-236), so its gradient is a sum of two -- formed in this kernel's final
 * store instead of by an extra elementwise pass. */
int t2h_sample_bwd_add(const float *gout, const float *pts, int dim, const int32_t *off0, int B, int N, int nbits,
                       int level, int C, const float *addend, float *gplane_nhwc, void *workspace,
                       size_t workspace_bytes, t2h_stream_t stream);
/* The same backward through the transposed sampling matrix, for callers that sample one tile at one level more than once
 * (the training step: three backwards at r = 256, two at r = 128; the points do not move in between).  build: a CSR over
 * the pixels of level `level`, numbered in Morton order inside a tile -- offsets [t2h_sample_adjoint_offsets_len()] int32
 * (B * r * r + 1 offsets, then scratch), entries [4 * B * N] pairs (int32 row, float weight), a pixel's entries in the
 * order t2h_sample_bwd visits them -- from the points and the tile index.  bwd_adjoint: gplane =
 * [addend +] S^T gout with those entries; the sums are bit-identical to t2h_sample_bwd's per-pixel gather (same order,
 * same roundings).  Measured at N = 131072, r = 256, C = 64: see DESIGN.md section 4. */
size_t t2h_sample_adjoint_offsets_len(int B, int nbits, int level);
int t2h_sample_adjoint_build(const float *pts, int dim, const int32_t *off0, int B, int N, int nbits, int level,
                             int32_t *offsets, void *entries, t2h_stream_t stream);
int t2h_sample_bwd_adjoint(const float *gout, const int32_t *offsets, const void *entries, int B, int nbits, int level, int C,
                           const float *addend, float *gplane_nhwc, t2h_stream_t stream);

int t2h_sample_bwd_atomic(const float *gout, const float *pts, int dim, int B, int N, int r, int C,
                          float *gplane_nhwc, t2h_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Per-point nn.Linear layers in exact fp32 on the matrix cores (v_mfma_f32_32x32x2_f32):
 * pointnet.py:36-40 (fc_pos, fc_c), block/resnet.py:26-31 (fc_0, fc_1, shortcut), alto.py:63-69,164-170
 * (fc_comm.0, fc_comm.2, fc_c) and their autograd backward.  Rows are points (row stride ld*, so a layer can
 * read / write a column slice of a wider buffer and the reference's torch.cat at pointnet.py:78 never
 * materialises); W is torch's [N_out, K_in] row-major weight.  The elementwise neighbours are fused:
 *
 *   t2h_linear_fwd    y = [y +] act_out( act_in(x) W^T + bias )        flags: T2H_RELU_IN | T2H_RELU_OUT | T2H_ACCUM
 *   t2h_linear_dgrad  dx = [dx +] (dy W) * (mask > 0)                   flags: T2H_ACCUM;  mask [M, ldmask] or NULL
 *                     (mask = the tensor whose ReLU fed this layer / whose ReLU output dy belongs to)
 *   t2h_linear_wgrad  dw = [dw +] dy^T act_in(x);  db = [db +] colsum(dy)  flags: T2H_RELU_IN | T2H_ACCUM; db may be NULL
 *                     reduction over the M points is split across workgroups into slabs in `workspace`
 *                     (t2h_linear_wgrad_workspace_bytes) and summed in split order: deterministic.
 * K_in that is not a multiple of 4 (fc_pos: K = 3) takes a VALU path (K <= 64 forward, K <= 8 wgrad), always fp32.
 * T2H_BF16 (all three): v_mfma_f32_32x32x16_bf16 on bf16-rounded operands, fp32 accumulate and fp32 I/O. */
#define T2H_RELU_IN 1
#define T2H_RELU_OUT 2
#define T2H_ACCUM 4
#define T2H_BF16 8   /* operands rounded to bf16 (RNE) while staging, fp32 accumulate: BASELINE.json configs[2] */
#define T2H_DEFER_REDUCE 32 /* weight-gradient entry points: see t2h_reduce_capture_begin */
#define T2H_F16X2 64 /* the t2h_*_bx3* entry points: fp16 two-way split with per-block power-of-two scales, 3 MFMAs per product
                        (weights prepared by the *_f16x2_prepare entry points); see "fp16 two-way split" below */
#define T2H_BF16X3 16 /* fp32-grade products from bf16 MFMAs: exact 3-way bf16 split of both operands, 6 piece products
                         (error <= ~2^-23 relative per product, fp32 accumulate); opt-in, never the default */
int t2h_linear_fwd(const float *x, int ldx, const float *w, const float *bias, float *y, int ldy, int M, int K,
                   int N, int flags, t2h_stream_t stream);
/* t2h_linear_fwd + addend [M, ldadd] added to the finished result (fp32 kernels only; NULL = t2h_linear_fwd): the 1x1-conv
 * residuals `x + conv1x1(prev)` of alto.py:110,114,236 without a separate elementwise add */
int t2h_linear_fwd_add(const float *x, int ldx, const float *w, const float *bias, const float *addend, int ldadd, float *y,
                       int ldy, int M, int K, int N, int flags, t2h_stream_t stream);
int t2h_linear_dgrad(const float *dy, int lddy, const float *w, float *dx, int lddx, int M, int K, int N,
                     const float *mask, int ldmask, int flags, t2h_stream_t stream);
size_t t2h_linear_wgrad_workspace_bytes(int M, int K, int N);
int t2h_linear_wgrad(const float *dy, int lddy, const float *x, int ldx, int M, int K, int N, int flags,
                     float *dw, float *db, void *workspace, size_t workspace_bytes, t2h_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * F.interpolate(size=(H, W), mode='bilinear', align_corners=True) pixel.py:107,110
 * NCHW in / NCHW out (the decoder convs consume it).  `addend` (may be NULL) is added to the
 * result: the image-plane sum of pixel.py:110 when the image plane is already H x W.
 * Backward: deterministic gather (no atomics). */
int t2h_upsample_bilinear_fwd(const float *in, const float *addend, int B, int C, int h, int w, int H, int W,
                              float *out, t2h_stream_t stream);
int t2h_upsample_bilinear_bwd(const float *gout, int B, int C, int h, int w, int H, int W, float *gin,
                              t2h_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Grid-side fusions around the (MIOpen) convolutions -- SURVEY 8f-1, first step.  NHWC tensors ([B,H,W,C],
 * channels_last), P = B*H*W pixels, C % 4 == 0.
 *   t2h_bias_relu_fwd   y = act(y + bias[c]) in place           conv bias + F.relu (alto.py:98-99,226-227; pixel.py:27-30)
 *   t2h_bias_relu_bwd   g_masked = g * (y > 0) (if relu; else untouched, may be NULL); dbias = [dbias +] sum_p g_masked
 *   t2h_head1x1_fwd/bwd out[p] = bias + sum_i <w_i, x_i[p,:]> over up to 4 inputs: torch.cat([x,x1,x2,x3]) + the 1x1
 *                       conv4 of ConvDecoder (pixel.py:31) without materialising the 288-channel concat; backward gives
 *                       dx_i = [dx_i +] g[p] w_i, dw (Ctot floats, concatenated order) and dbias.  out_channels = 1.
 *                       dx_flags: bit 0 = accumulate into dx_i; bit 1 = add to what dw / dbias hold (the trainer's gradient
 *                       bucket) instead of overwriting; bit 8+i = multiply dx_i by (x_i > 0), i.e. hand the producer of a ReLU
 *                       output x_i a gradient that already passed its ReLU backward.
 *   t2h_upsample_bilinear_nhwc_fwd/bwd   F.interpolate(bilinear, align_corners=True) on NHWC planes (pixel.py:107,110) */
int t2h_bias_relu_fwd(float *y, const float *bias, int64_t P, int C, int relu, t2h_stream_t stream);
size_t t2h_bias_relu_bwd_workspace_bytes(int64_t P, int C);
int t2h_bias_relu_bwd(const float *g, const float *y, float *g_masked, int64_t P, int C, int relu, int accumulate,
                      float *dbias, void *workspace, size_t workspace_bytes, t2h_stream_t stream);
int t2h_head1x1_fwd(const float *const *x, const int *C, int n_in, const float *w, const float *bias, int64_t P,
                    float *out, t2h_stream_t stream);
size_t t2h_head1x1_bwd_workspace_bytes(int64_t P, int Ctot);
int t2h_head1x1_bwd(const float *const *x, float *const *dx, const int *C, int n_in, const float *w, const float *g,
                    int64_t P, int dx_flags, float *dw, float *dbias, void *workspace, size_t workspace_bytes,
                    t2h_stream_t stream);
int t2h_upsample_bilinear_nhwc_fwd(const float *in, const float *addend, int B, int C, int h, int w, int H, int W,
                                   float *out, t2h_stream_t stream);
/* nn.MaxPool2d(kernel_size=2, stride=2) on NHWC planes (DownConv.pool, alto.py:61,110,136): in [B,H,W,C] (H, W even),
 * out [B,H/2,W/2,C]; which [B,H/2,W/2,C] bytes = 2*dy+dx of the first maximum in scan order (ATen's tie-break); the
 * backward writes every element of gin [B,H,W,C] exactly once. */
int t2h_maxpool2x2_nhwc_fwd(const float *in, int B, int H, int W, int C, float *out, uint8_t *which, t2h_stream_t stream);
int t2h_maxpool2x2_nhwc_bwd(const float *gout, const uint8_t *which, int B, int H, int W, int C, float *gin,
                            t2h_stream_t stream);
/* ... with `addend` [B, H, W, C] (may be NULL) added: the pooled plane is also a U-Net skip connection (alto.py:135-138).
 * ld_addend: its pixel stride in floats (>= C): the skip's gradient arrives as one half of a concatenation's gradient
 * (alto.py:227: torch.cat((from_up, from_down), 1)) and is read in place. */
int t2h_maxpool2x2_nhwc_bwd_add(const float *gout, const uint8_t *which, int B, int H, int W, int C, const float *addend,
                                int ld_addend, float *gin, t2h_stream_t stream);
int t2h_upsample_bilinear_nhwc_bwd(const float *gout, int B, int C, int h, int w, int H, int W, float *gin,
                                   t2h_stream_t stream);
/* nn.Upsample(mode='bilinear', scale_factor=2) -- align_corners=False, ATen's half-pixel source index -- on NHWC planes:
 * the non-parametric up path upconv2x2(mode='upsample') puts in front of a 1x1 convolution (alto.py:23-35, unet.py).
 * in [B, h, w, C] -> out [B, 2h, 2w, C]; the backward gathers, per input pixel, the output pixels that read it in raster
 * order (no atomics). */
int t2h_upsample2x_nhwc_fwd(const float *in, int B, int C, int h, int w, float *out, t2h_stream_t stream);
int t2h_upsample2x_nhwc_bwd(const float *gout, int B, int C, int h, int w, float *gin, t2h_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * 3x3 grid convolutions as implicit GEMMs on the matrix cores (exact fp32) -- SURVEY 8f-1, second step.
 * Replaces nn.Conv2d(kernel_size=3, stride=1, padding=1) forward and its autograd: conv3x3 of alto.py:59-61 used at
 * alto.py:98-99,226-227 (with F.relu) and ConvDecoder.conv1..3 of pixel.py:20-30.
 *   x  [B,H,W,Cin]  NHWC (torch channels_last), y / dy [B,H,W,Cout]; H, W powers of two
 *   w  [Cout][3][3][Cin] = the channels_last memory of torch's [Cout,Cin,3,3] weight; dw has the same layout
 *   fwd    y  = act(conv(x, w) + bias)             flags: T2H_RELU_OUT, T2H_ACCUM;   Cin % 16 == 0, Cout % 4 == 0
 *   dgrad  dx = [dx +] conv_transpose(dy, w) * (mask > 0)   mask [B,H,W,Cin] or NULL; flags: T2H_ACCUM; Cout % 16 == 0
 *   wgrad  dw = [dw +] sum_p dy[p] (x) x[p+tap];  db = [db +] sum_p dy[p]  (db may be NULL); flags: T2H_ACCUM
 * t2h_relu_mask is the F.relu backward in front of dgrad / wgrad (the bias gradient comes out of wgrad).
 * Plane sizes that give fewer workgroups than the chip has CUs split the reduction into slabs in the caller's
 * workspace (sizes from *_workspace_bytes; 0 = none needed); slabs are summed in a fixed order: deterministic. */
int t2h_relu_mask(const float *g, const float *y, float *g_masked, int64_t n, t2h_stream_t stream); /* g * (y > 0) */
size_t t2h_conv3x3_fwd_workspace_bytes(int B, int H, int W, int Cin, int Cout);
int t2h_conv3x3_fwd(const float *x, const float *w, const float *bias, float *y, int B, int H, int W, int Cin, int Cout,
                    int flags, void *workspace, size_t workspace_bytes, t2h_stream_t stream);
size_t t2h_conv3x3_dgrad_workspace_bytes(int B, int H, int W, int Cin, int Cout);
int t2h_conv3x3_dgrad(const float *dy, const float *w, float *dx, const float *mask, int B, int H, int W, int Cin,
                      int Cout, int flags, void *workspace, size_t workspace_bytes, t2h_stream_t stream);
size_t t2h_conv3x3_wgrad_workspace_bytes(int B, int H, int W, int Cin, int Cout);
int t2h_conv3x3_wgrad(const float *dy, const float *x, float *dw, float *db, int B, int H, int W, int Cin, int Cout,
                      int flags, void *workspace, size_t workspace_bytes, t2h_stream_t stream);

/* The same three convolutions on the bf16 matrix cores with an EXACT 3-way bf16 split of both operands (six bf16 MFMAs per
 * product, fp32 accumulate: fp32-grade results, same tolerance as the kernels above; csrc/conv_bx3.hip), for planes with
 * W >= 32, H >= 4 and Cin, Cout multiples of 32 (t2h_conv3x3_bx3_supported).  Same reference calls (alto.py:59-61,98-99,
 * 157-182,226-227; pixel.py:20-32), same layouts; planes with fewer tiles than the chip has workgroups split the reduction into
 * slabs in the caller's workspace, summed in a fixed order (deterministic).  The weights are split once per optimizer step into the byte order of the
 * MFMA operand: t2h_conv3x3_bx3_prepare(w [Cout][3][3][Cin], transposed = 0 for fwd / 1 for dgrad) -> wf of
 * t2h_conv3x3_bx3_weights_bytes; activations are split inside the kernels, once per staged element.
 *   fwd    y  = [y +] act(conv(x, w) + bias)          flags: T2H_RELU_OUT, T2H_ACCUM
 *   dgrad  dx = [dx +] conv_transpose(dy, w) * (mask > 0)     flags: T2H_ACCUM
 *   wgrad  dw = [dw +] sum_p dy[p] (x) x[p+tap]; db = [db +] sum_p dy[p] (db may be NULL); flags: T2H_ACCUM; deterministic slabs
 * T2H_BF16 (all three): only the leading bf16 part of both operands, ONE MFMA per product -- the arithmetic of BASELINE
 * configs[2] ("bf16 ... GEMMs on MFMA": operands rounded to bf16, fp32 accumulate, fp32 tensors), ~2e-3 relative. */
int t2h_conv3x3_bx3_supported(int B, int H, int W, int Cin, int Cout);
size_t t2h_conv3x3_bx3_weights_bytes(int Cin, int Cout);
int t2h_conv3x3_bx3_prepare(const float *w, int Cin, int Cout, int transposed, void *wf, t2h_stream_t stream);
size_t t2h_conv3x3_bx3_fwd_workspace_bytes(int B, int H, int W, int Cin, int Cout);     /* 0 unless the reduction is split */
int t2h_conv3x3_bx3_fwd(const float *x, const void *wf, const float *bias, float *y, int B, int H, int W, int Cin, int Cout,
                        int flags, void *workspace, size_t workspace_bytes, t2h_stream_t stream);
size_t t2h_conv3x3_bx3_dgrad_workspace_bytes(int B, int H, int W, int Cin, int Cout);
int t2h_conv3x3_bx3_dgrad(const float *dy, const void *wf_transposed, float *dx, const float *mask, int B, int H, int W, int Cin,
                          int Cout, int flags, void *workspace, size_t workspace_bytes, t2h_stream_t stream);

/* r06: the same data gradient with the 1 x 1 head's share of the gradient formed in its epilogue: dx = mask(dgrad) + mask(g[pixel] *
 * w1[ci]) (mask: (mask > 0), or none), WRITTEN -- for the decoder activations that feed both the next 3 x 3 convolution and the head
 * (pixel.py:28-31: out = conv4(cat[x, x1, x2, x3])): the head's backward then writes no gradient for them and this kernel reads no
 * old values.  Bit-identical to t2h_head1x1_bwd's dx followed by t2h_conv3x3_bx3_dgrad(T2H_ACCUM).  g [B H W], w1 [Cin].
 * Only for shapes whose reduction is not split (t2h_conv3x3_bx3_dgrad_rank1_supported). */
int t2h_conv3x3_bx3_dgrad_rank1_supported(int B, int H, int W, int Cin, int Cout);
int t2h_conv3x3_bx3_dgrad_rank1(const float *dy, const void *wf_t, float *dx, const float *mask, const float *g, const float *w1,
                                int B, int H, int W, int Cin, int Cout, int flags, t2h_stream_t stream);

size_t t2h_conv3x3_bx3_wgrad_workspace_bytes(int B, int H, int W, int Cin, int Cout);
int t2h_conv3x3_bx3_wgrad(const float *dy, const float *x, float *dw, float *db, int B, int H, int W, int Cin, int Cout,
                          int flags, void *workspace, size_t workspace_bytes, t2h_stream_t stream);

/* The 1-tap form of the same kernels: a GEMM on rows, y[M, N] = [y +] act( x[M, K] W^T + bias ) * (mask > 0), fp32 in / out, products
 * by the exact 3-way bf16 split (T2H_BF16: leading part only).  For nn.Linear on pixel rows and the grid-side products of the
 * deferred ALTO point update (alto.py:123-130 re-associated, tomosar2height_amd/deferred.py), whose long reductions (K up to 2752
 * stacked columns) ran at 0.4-0.6 of the fp32 matrix peak.  W: [N][K] with row stride ldw (w_is_kn = 0, nn.Linear layout) or
 * [K][N] (w_is_kn = 1), split once by t2h_gemm_bx3_prepare.  M % 128 == 0, K % 64 == 0, N % 32 == 0 (t2h_gemm_bx3_supported);
 * ldx / ldy / ldm: row strides in floats (column slices of wider matrices are fine).  Few-row products split K into slabs. */
int t2h_gemm_bx3_supported(int64_t M, int K, int N);
size_t t2h_gemm_bx3_weights_bytes(int K, int N);
int t2h_gemm_bx3_prepare(const float *w, int ldw, int K, int N, int w_is_kn, void *wf, t2h_stream_t stream);
size_t t2h_gemm_bx3_workspace_bytes(int64_t M, int K, int N);
int t2h_gemm_bx3(const float *x, int ldx, const void *wf, const float *bias, const float *mask, int ldm, float *y, int ldy,
                 int64_t M, int K, int N, int flags, void *workspace, size_t workspace_bytes, t2h_stream_t stream);

/* r06: the weight gradient of the same product on the split kernels: dw [N, K] = [dw +] dy^T x, db [N] = [db +] colsum(dy) (db may be
 * NULL) -- the transposed product over the M rows (alto.py:123-130 re-associated: the Linear layers on pixel rows / per-cell sums;
 * autograd of nn.Linear).  Every fp32 product from three fp16 MFMAs (flag T2H_F16X2: the two-way split with one power-of-two scale
 * per 32-row unit and operand; the only arithmetic built for this form), deterministic split slabs + reduction.
 * M % 32 == 0, K in {64, 128, 256}, N % 64 == 0; row strides lddy >= N, ldx >= K in floats.  T2H_ACCUM, T2H_DEFER_REDUCE as
 * t2h_conv3x3_bx3_wgrad. */
int t2h_gemm_bx3_wgrad_supported(int64_t M, int K, int N);
size_t t2h_gemm_bx3_wgrad_workspace_bytes(int64_t M, int K, int N);
int t2h_gemm_bx3_wgrad(const float *dy, int lddy, const float *x, int ldx, int64_t M, int K, int N, float *dw, float *db, int flags,
                       void *workspace, size_t workspace_bytes, t2h_stream_t stream);


/* fp16 two-way split (flag T2H_F16X2 on t2h_conv3x3_bx3_fwd / _dgrad / _wgrad, t2h_gemm_bx3, t2h_upconv2x2_bx3_*).  Each operand
 * element is x 2^e = h1 + h2, two fp16 numbers (11 + 11 significant bits), with ONE power of two 2^e per staged block -- activations:
 * the halo tile x channel chunk a workgroup stages (3x3 / 1-tap forms) or the 32-pixel unit (weight gradient); weights: the tensor --
 * chosen so that the block's largest magnitude lands in [2^14, 2^15).  Three of the four partial products are formed (3 MFMAs
 * instead of the 6 of the bf16 three-way split), accumulated in fp32; the accumulators are rescaled exactly when a block with a
 * larger scale arrives.  Error per product: <= 3 * 2^-22 |a b| + 2^-40 (block max |a|) |b| (+ the same with a, b swapped) -- fp32
 * grade unless an element lies more than 2^18 below the largest element of its own block; measured against float64 at the level
 * of the fp32 kernels (tests/test_hip_conv.py).  The weights are prepared by the entry points below into a buffer of
 * t2h_*_f16x2_weights_bytes (two f16 planes + a 256-byte trailer holding the tensor's scale) and passed where the bf16 buffers go. */
size_t t2h_conv3x3_f16x2_weights_bytes(int Cin, int Cout);
int t2h_conv3x3_f16x2_prepare(const float *w, int Cin, int Cout, int transposed, void *wf, t2h_stream_t stream);
size_t t2h_gemm_f16x2_weights_bytes(int K, int N);
int t2h_gemm_f16x2_prepare(const float *w, int ldw, int K, int N, int w_is_kn, void *wf, t2h_stream_t stream);

/* Every split weight of a model in a few launches (24 buffers per launch; the weights change once per optimizer step, trainer.py:78-84,
 * and preparing ~110 buffers one by one cost 0.85 ms of host time after each step).  One descriptor per prepared buffer:
 *   kind 0 / 1: the 3x3 weight [Cout][3][3][Cin] forward / transposed (a = Cin, b = Cout) -- t2h_conv3x3_{bx3,f16x2}_prepare
 *   kind 2 / 3: a matrix stored [K][N] / [N][K] with row stride ldw (a = K, b = N)        -- t2h_gemm_{bx3,f16x2}_prepare
 *   h2: 1 = fp16 two-plane buffer (T2H_F16X2), 0 = bf16 three-plane buffer;  trailer_word = planes bytes / 4 (h2 only);
 *   maxslot (h2 only): 0 or 3, ALTERNATING between consecutive runs on the same buffer, which must start zeroed -- each run clears
 *   the word the next one accumulates max |w| into, so no memset runs in between. */
typedef struct t2h_prep_desc {
    const float *w;
    void *wf;
    int kind, h2, a, b, ldw, maxslot;
    unsigned trailer_word, reserved;
} t2h_prep_desc;
int t2h_split_weights_batch(const t2h_prep_desc *descs, int n, t2h_stream_t stream);

/* ConvTranspose2d(kernel_size = 2, stride = 2) (upconv2x2, alto.py:175,215-218,236) on the same 1-tap kernels.  The forward is the
 * GEMM [B H W, Cin] x [Cin, (tap, co)] with a scattering epilogue: column tile -> tap = 2 dy + dx -> output pixel (2y + dy, 2x + dx),
 * bias and the residual `addend` (output-shaped, or null) added there; y[B, 2H, 2W, Cout] NHWC.  The data gradient gathers its A
 * rows instead: K = (tap, co) reads dy at the four output pixels of each input pixel.  Weights: the module's [Cin, Cout, 2, 2]
 * tensor in [Cin][2][2][Cout] memory order IS the matrix [Cin][4 Cout]: split it with t2h_gemm_bx3_prepare(w, 4 Cout, K = Cin,
 * N = 4 Cout, w_is_kn = 1) for the forward and (w, 4 Cout, K = 4 Cout, N = Cin, w_is_kn = 0) for the data gradient.
 * H, W powers of two, B H W % 128 == 0, Cin % 64 == 0, Cout % 64 == 0 (t2h_upconv2x2_bx3_supported).  Replaces
 * t2h_upconv2x2_fwd_add / t2h_upconv2x2_dgrad where supported.  The weight gradient dW[ci][tap][co] = sum_p x[p][ci] dy[up(p, tap)][co]
 * (and db = column sums of dy) is bx3_wgrad_kernel with the roles swapped and 4 taps (W >= 32 in addition; same flags, workspace and
 * slab reduction as t2h_conv3x3_bx3_wgrad); replaces t2h_upconv2x2_wgrad_bias. */
int t2h_upconv2x2_bx3_supported(int B, int H, int W, int Cin, int Cout);
int t2h_upconv2x2_bx3_fwd(const float *x, const void *wf, const float *bias, const float *addend, float *y, int B, int H, int W,
                          int Cin, int Cout, int flags, t2h_stream_t stream);
size_t t2h_upconv2x2_bx3_dgrad_workspace_bytes(int B, int H, int W, int Cin, int Cout);
/* lddy: pixel stride of dy in floats (>= Cout, a multiple of 4): the output's gradient is usually the `from_up` half of a
 * concatenation's gradient (alto.py:227) and is read in place */
int t2h_upconv2x2_bx3_dgrad(const float *dy, int lddy, const void *wf_t, float *dx, int B, int H, int W, int Cin, int Cout, int flags,
                            void *workspace, size_t workspace_bytes, t2h_stream_t stream);
size_t t2h_upconv2x2_bx3_wgrad_workspace_bytes(int B, int H, int W, int Cin, int Cout);
int t2h_upconv2x2_bx3_wgrad(const float *dy, int lddy, const float *x, float *dw, float *db, int B, int H, int W, int Cin, int Cout,
                            int flags, void *workspace, size_t workspace_bytes, t2h_stream_t stream);

/* Batched slab reductions.  Every weight-gradient entry point (t2h_linear_wgrad, t2h_conv3x3_wgrad, t2h_conv3x3_bx3_wgrad,
 * t2h_upconv2x2_wgrad[_bias]) ends with a launch that sums its split slabs (fixed order: deterministic) into dw / db.  Within one
 * backward pass (trainer.py:70 `loss.backward()`) nothing reads those gradients before the pass ends, so a caller may bracket the
 * pass with t2h_reduce_capture_begin() / t2h_reduce_capture_end(stream): while the capture is active, calls that carry
 * T2H_DEFER_REDUCE record their reduction instead of launching it, and _end runs all recorded ones in one launch per 24 (same
 * summation tree per output: bit-identical results; ~50 launches per tile-step become 2-3).  The caller must keep the workspaces of
 * the recorded calls alive until _end and must not read dw / db before it.  The capture is process-wide state (the backward may run
 * on another thread than the caller's); two reductions into the same output are never batched together (the earlier ones are
 * flushed first).  t2h_reduce_capture_pending: recorded reductions, -1 when no capture is active. */
int t2h_reduce_capture_begin(void);
int t2h_reduce_capture_pending(void);
int t2h_reduce_capture_end(t2h_stream_t stream);

/* nn.ConvTranspose2d(kernel_size=2, stride=2) of the ALTO up path (upconv2x2 in alto.py, used at alto.py:175,215-218,236):
 * output pixel (2y+dy, 2x+dx) = bias + sum_ci x[y, x, ci] w[ci, dy, dx, co].  H, W = INPUT plane dims (powers of two),
 * x [B,H,W,Cin], y / dy [B,2H,2W,Cout] NHWC; w [Cin][2][2][Cout] = channels_last memory of torch's [Cin,Cout,2,2]
 * weight (dw likewise); Cin, Cout multiples of 16.  The bias gradient is sum_p dy[p] = t2h_bias_relu_bwd with relu = 0. */
int t2h_upconv2x2_fwd(const float *x, const float *w, const float *bias, float *y, int B, int H, int W, int Cin, int Cout,
                      int flags, t2h_stream_t stream);
/* ... + addend [B,2H,2W,Cout] (may be NULL): the residual `x + upconv(prev)` of alto.py:236 without a separate add */
int t2h_upconv2x2_fwd_add(const float *x, const float *w, const float *bias, const float *addend, float *y, int B, int H,
                          int W, int Cin, int Cout, int flags, t2h_stream_t stream);
size_t t2h_upconv2x2_dgrad_workspace_bytes(int B, int H, int W, int Cin, int Cout);
int t2h_upconv2x2_dgrad(const float *dy, const float *w, float *dx, int B, int H, int W, int Cin, int Cout, int flags,
                        void *workspace, size_t workspace_bytes, t2h_stream_t stream);
size_t t2h_upconv2x2_wgrad_workspace_bytes(int B, int H, int W, int Cin, int Cout);
int t2h_upconv2x2_wgrad(const float *dy, const float *x, float *dw, int B, int H, int W, int Cin, int Cout, int flags,
                        void *workspace, size_t workspace_bytes, t2h_stream_t stream);
/* The same, also producing the bias gradient db[Cout] (= the column sums of dy, formed from the operand tiles the kernel
 * stages anyway; db may be NULL).  Workspace: t2h_upconv2x2_wgrad_workspace_bytes. */
int t2h_upconv2x2_wgrad_bias(const float *dy, const float *x, float *dw, float *db, int B, int H, int W, int Cin, int Cout,
                             int flags, void *workspace, size_t workspace_bytes, t2h_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * DSM mosaic of the inference path                     generator.py:147-157
 *   per tile: dsm[t:t+H, l:l+W] += flip_rows(height) * patch_weight;  weight[...] += patch_weight   (float64)
 *   finally : dsm = maximum(dsm / weight, 0), NaN where no tile contributed
 * height [H, W] fp32 = model(...)[0].squeeze(); flip_rows = 1 applies the reference's .flip(1). Region rows/cols that
 * fall outside the [rows, cols] mosaic are skipped.  Launch one tile at a time on one stream (tiles overlap). */
int t2h_mosaic_accumulate(const float *height, int H, int W, const double *patch_weight, double *dsm,
                          double *weight, int rows, int cols, int t_row, int l_col, int flip_rows,
                          t2h_stream_t stream);
int t2h_mosaic_finalize(double *dsm, const double *weight, int64_t n, t2h_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Tile producer: the point half of TomoSARDataset.__getitem__   dataset.py:233-278, utils/crop_cloud.py:8-29
 * chunk [P,3] float64 world points (resident in HBM) -> out [count,3] float32 normalised tile, original order kept:
 *   strict crop to (min, max) in x/y; z_shift = min z of the cropped points ('local_min'); float64 normalise
 *   (x - cx)/scale_x + 0.5, (y - cy)/scale_y + 0.5, (z - z_shift)/scale_z; cast to float32; strict re-crop to (0,1).
 * out must hold P rows (upper bound); *count = rows written; src_index (may be NULL) = chunk row of each output row.
 * t2h_tile_crop_finish converts the internal min key in *z_shift to the double value (NaN when the crop is empty);
 * call it after t2h_tile_crop_normalise on the same stream. */
size_t t2h_tile_crop_workspace_bytes(int64_t P);
int t2h_tile_crop_normalise(const double *chunk, int64_t P, double min_x, double min_y, double max_x, double max_y,
                            double scale_x, double scale_y, double scale_z, float *out, int32_t *src_index,
                            int32_t *count, double *z_shift, void *workspace, size_t workspace_bytes,
                            t2h_stream_t stream);
int t2h_tile_crop_finish(double *z_shift, t2h_stream_t stream);
/* The same with the training augmentation of dataset.py:253-270: rot_times in 0..3 quarter turns (clockwise about z, in
 * the centred tile frame), then flip_dim -1 (none) / 0 (x := -x) / 1 (y := -y); (0, -1) is t2h_tile_crop_normalise. */
int t2h_tile_crop_normalise_aug(const double *chunk, int64_t P, double min_x, double min_y, double max_x, double max_y,
                                double scale_x, double scale_y, double scale_z, int rot_times, int flip_dim, float *out,
                                int32_t *src_index, int32_t *count, double *z_shift, void *workspace,
                                size_t workspace_bytes, t2h_stream_t stream);

/* Raster half of the tile (dataset.py:291-328): the DSM target / satellite image patch of a raster resident in HBM.
 * raster [C, H, W] (float32, or float64 when is_f64: the mean/std-normalised image is held in double by the reference);
 * out [C, ph, pw] float32 = raster[:, row0:row0+ph, col0:col0+pw] .rot90(rot_times, [-1, -2]) .flip(-1 if flip_dim == 0,
 * -2 if flip_dim == 1) .float() .flip(-2)  -- one gather through the composed index map. */
int t2h_raster_patch(const void *raster, int is_f64, int C, int H, int W, int row0, int col0, int ph, int pw,
                     int rot_times, int flip_dim, float *out, t2h_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Layout glue between the conv side (NCHW) and the point side (NHWC): [B, C, P] <-> [B, P, C]. */
int t2h_nchw_to_nhwc(const float *in, int B, int C, int P, float *out, t2h_stream_t stream);
int t2h_nhwc_to_nchw(const float *in, int B, int C, int P, float *out, t2h_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * sample_mode = 'bicubic' (r06; no shipped config selects it: tomosar2height.yaml:27)
 *   F.interpolate(c, size, mode='bicubic', align_corners=True)                            decoder/pixel.py:107,110
 *   F.grid_sample(c, vgrid, padding_mode='border', align_corners=True, mode='bicubic')    encoder/alto.py:95,204
 * ATen's cubic convolution (A = -0.75), taps outside the plane clamped to the border pixel.  Planes [B, C, h, w] in NCHW or
 * (channels_last != 0) NHWC memory.  t2h_upsample_bicubic_bwd is a gather (deterministic); t2h_sample_bicubic_bwd zeroes gplane
 * and adds with atomics (like t2h_sample_bwd_atomic: the order of its additions is not reproducible run to run).
 *   pts   [B * N, dim] rows with x, y in the plane's [0, 1] coordinates (vgrid = 2 xy - 1);  out / gout [B * N, C] point-major */
int t2h_upsample_bicubic_fwd(const float *in, const float *addend, int B, int C, int h, int w, int H, int W, int channels_last,
                             float *out, t2h_stream_t stream);
int t2h_upsample_bicubic_bwd(const float *gout, int B, int C, int h, int w, int H, int W, int channels_last, float *gin,
                             t2h_stream_t stream);
int t2h_sample_bicubic_fwd(const float *plane, const float *pts, int dim, int B, int64_t N, int r, int C, int channels_last,
                           float *out, t2h_stream_t stream);
int t2h_sample_bicubic_bwd(const float *gout, const float *pts, int dim, int B, int64_t N, int r, int C, int channels_last,
                           float *gplane, t2h_stream_t stream);
/* mode = 'nearest' of the same call (grid_sampler_2d: the coordinate clipped to the plane, then rounded half to even) */
int t2h_sample_nearest_fwd(const float *plane, const float *pts, int dim, int B, int64_t N, int r, int C, int channels_last,
                           float *out, t2h_stream_t stream);
int t2h_sample_nearest_bwd(const float *gout, const float *pts, int dim, int B, int64_t N, int r, int C, int channels_last,
                           float *gplane, t2h_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * nn.Conv2d(Cin <= 8, Cout, 3, padding=1): the image U-Net's first layer              encoder/unet.py:112-187
 * (Conv2d(3, 32, ...) at 512 x 512: the one convolution of the image configs that t2h_conv3x3_* does not take --
 * their reduction runs in 16-channel slabs).  NHWC activations, weight memory [Cout][3][3][Cin], Cout % 4 == 0, <= 64.
 * flags: T2H_RELU_OUT (forward), T2H_ACCUM (dgrad / wgrad add to the destination).  Deterministic (slabs, no atomics). */
int t2h_conv3x3_smallcin_fwd(const float *x, const float *w, const float *bias, float *y, int B, int H, int W, int Cin,
                             int Cout, int flags, t2h_stream_t stream);
int t2h_conv3x3_smallcin_dgrad(const float *dy, const float *w, float *dx, int B, int H, int W, int Cin, int Cout,
                               int flags, t2h_stream_t stream);
size_t t2h_conv3x3_smallcin_wgrad_workspace_bytes(int Cin, int Cout);
int t2h_conv3x3_smallcin_wgrad(const float *dy, const float *x, float *dw, float *db, int B, int H, int W, int Cin,
                               int Cout, int flags, void *workspace, size_t workspace_bytes, t2h_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Fused PointNet trunk block, hidden_dim = 32               pointnet.py:72-82, 92-99; resnet.py:36-54
 * One launch per ResnetBlockFC(64 -> 32) over the cell-sorted rows:
 *     X   = [net_prev | pool_local(net_prev)]       (pts != NULL: X = fc_pos(pts) instead, pointnet.py:72)
 *     hr  = relu(fc_0(relu(X)))                     [M, 32]  (kept for the backward)
 *     out = shortcut(X) + (fc_1(hr) + b1)           [M, ld_out >= 32]
 *     c   = fc_c(relu(out))                         [M, 32]  (wc != NULL: the last block, pointnet.py:81-82)
 * scatter_max + gather + torch.cat (pointnet.py:76-78) run inside the block's loader: the pooled half of X is never
 * written to memory.  winner [M, 8] receives the arg-max bits of net_prev's pooling (bit j of byte l: the row holds the
 * first maximum of its cell for channel 4l + j -- torch_scatter's CPU tie-break), as t2h_pool_max_fwd would.
 * pooled (optional, [M, 32]): the pooled half of X, which the backward reads; x_full (optional, [M, 64]): all of X (tests).
 * w0, ws [32, 64]; w1, wc [32, 32]; w_pos [64, 3]: nn.Linear layouts.  cell / off0: from t2h_tile_build. */
int t2h_trunk_block_fwd(const float *pts, int dim, const float *w_pos, const float *b_pos, const float *net_prev,
                        int ld_prev, const int32_t *cell, const int32_t *off0, const float *w0, const float *b0,
                        const float *w1, const float *b1, const float *ws, const float *wc, const float *bc, int64_t M,
                        float *x_full, float *pooled, float *hr, float *out, int ld_out, uint8_t *winner, float *c_out,
                        t2h_stream_t stream);

/* r06 (ABI 15): the whole trunk forward of pointnet.py:72-82 -- fc_pos, n_blocks x ResnetBlockFC(64 -> 32) with the
 * pool_local (scatter_max + gather + cat, pointnet.py:76-78, 92-99) between them, fc_c -- in ONE launch, for hidden_dim = 32.
 * All poolings use the same finest-level cells (pointnet.py:70), and a cell is a run of consecutive sorted rows, so a workgroup
 * that owns whole cells needs nothing from its neighbours between the blocks: the activations of its rows stay in LDS from the
 * points to c.  What is written is what the backward (t2h_trunk_block_bwd) reads: per block b the hidden activations hr[b] and
 * the output out[b] ([M, 32] each), for b >= 1 the pooled half pooled[b] [M, 32] and the arg-max bits winner[b] [M, 8] of the
 * pooling that feeds block b (pooled[0] / winner[0] are not read), and c_out [M, 32].  Results are bit-identical to
 * n_blocks t2h_trunk_block_fwd launches.
 *   block_params  HOST array of 5 * n_blocks DEVICE pointers: w0, b0, w1, b1, ws of block 0, then of block 1, ...
 *   hr, out, pooled, winner   HOST arrays of n_blocks device pointers
 *   units         device buffer of t2h_trunk_units_words(M) int32 words (8-byte aligned) filled by t2h_trunk_units_build: the work
 *                 units as a dense list of (first row, end row) pairs + their number -- whole cells packed greedily into at most
 *                 128 rows (a cell of more than 128 rows is a unit of its own and runs block by block through memory inside the
 *                 launch).  They depend on the tile index only: build once per tile, reuse for every forward.  NULL: fixed windows of `stride` rows (0: 96, at most 128) snapped to cell
 *                 boundaries, looked up by every workgroup itself; a window's unit of more than 128 rows takes the slow path
 * 2 <= n_blocks <= 8. */
int64_t t2h_trunk_units_words(int64_t M);
int t2h_trunk_units_build(const int32_t *cell, const int32_t *off0, int64_t M, int32_t *units, t2h_stream_t stream);
int t2h_trunk_fused_fwd(const float *pts, int dim, const float *w_pos, const float *b_pos, const float *const *block_params,
                        int n_blocks, const float *wc, const float *bc, const int32_t *cell, const int32_t *off0, int64_t M,
                        float *const *hr, float *const *out, float *const *pooled, uint8_t *const *winner, float *c_out,
                        int stride, const int32_t *units, t2h_stream_t stream);

/* Backward of t2h_trunk_block_fwd, one launch per block (+ one small reduction):
 *     g   = g_net + route(sum over each row's cell of g_pool)     (g_pool != NULL: the backward of the pooling that
 *                                                                  consumed this block's output: gather -> scatter-add,
 *                                                                  scatter_max -> the arg-max row, pointnet.py:95-98)
 *           (gc != NULL, last block: g = (gc wc) * (out_last > 0), and dWc / dbc of fc_c, pointnet.py:81-82)
 *     dhr = (g w1) * (hr > 0);   dX = g ws + (dhr w0) * (X > 0)  -> dx [M, 64] = [d net_prev | d pooled_prev]
 *     slabs of dW0, dWs, dW1, db0, db1 (and dWc, dbc / dWpos, dbpos) per workgroup in `workspace`
 * X = [x_left | x_right] (row strides ld_xl, ld_xr): the previous block's output and the pooled half the forward kept.
 * pts != NULL (first block): X is recomputed from the points, dx is not written, fc_pos's gradients are produced.
 * t2h_trunk_block_reduce adds the slabs in a fixed order into the gradients ([dst +=] when accumulate != 0). */
size_t t2h_trunk_block_bwd_workspace_bytes(int64_t M);
int t2h_trunk_block_bwd(const float *g_net, int ld_gn, const float *g_pool, int ld_gp, const uint8_t *winner,
                        const int32_t *cell, const int32_t *off0, const float *gc, const float *wc,
                        const float *out_last, const float *hr, const float *x_left, int ld_xl, const float *x_right,
                        int ld_xr, const float *pts, int dim, const float *w_pos, const float *b_pos, const float *w0,
                        const float *w1, const float *ws, int64_t M, float *dx, void *workspace, size_t workspace_bytes,
                        t2h_stream_t stream);
int t2h_trunk_block_reduce(const void *workspace, int64_t M, int first, int last, float *dw0, float *db0, float *dw1,
                           float *db1, float *dws, float *dwx, float *dbx, int accumulate, t2h_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * AdamW step over all parameters in one launch                       train.py:97, trainer.py:78-79
 * (torch.optim.AdamW arithmetic: decoupled weight decay, bias corrections in double on the host, amsgrad off).
 *   table   [n_tensors] device records {float *p; const float *g; float *m; float *v; int64 n}  (5 x 8 bytes each):
 *           parameter, its gradient (a view of the trainer's flat bucket), first / second moment, element count;
 *           p, g, m, v of one tensor share one dense memory layout, so the update runs in storage order
 *   chunks  [n_chunks] device records {int32 tensor, int32 first_element}, t2h_adamw_chunk_elems() elements each
 *   step    1-based step count (the bias corrections 1 - beta^step)
 *   zero_grad != 0: the gradient is set to zero in the same pass (trainer.py:80 optimizer.zero_grad()). */
int t2h_adamw_chunk_elems(void);
int t2h_adamw_flat_step(const void *table, const int32_t *chunks, int n_chunks, double lr, double beta1, double beta2,
                        double eps, double weight_decay, int64_t step, int zero_grad, t2h_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* T2H_H_ */
