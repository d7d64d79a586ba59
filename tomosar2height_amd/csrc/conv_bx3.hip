// 3x3 grid convolutions (stride 1, zero padding 1) on the bf16 matrix cores with an EXACT 3-way split of both operands:
// fp32-grade results at up to 2.67x the rate of v_mfma_f32_32x32x2_f32 (conv.hip), for planes at least 32 pixels wide
// (reference: conv3x3 of alto.py:59-61,157-182 with F.relu at alto.py:98-99,226-227; ConvDecoder pixel.py:20-32).
//
// Arithmetic.  x = x1 + x2 + x3 exactly, x1 = bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2) (round to nearest even; both
// differences are exact in fp32: 24 = 8 + 8 + 8 significant bits, the residuals carry their own signs).  Of the nine partial
// products a_i b_j of a product a b the six with i + j <= 4 are formed (a3b1, a1b3, a2b2, a2b1, a1b2, a1b1); the three dropped
// ones are below 2^-25 |a b|.  A bf16 x bf16 product is exact in fp32 and v_mfma_f32_32x32x16_bf16 accumulates in fp32, so
// the result differs from conv.hip's fp32 fma chain in rounding ORDER only: measured against float64 (profiles/
// conv_bf16x3_lab.hip, tests/test_hip_conv.py) both sit at 1.5-2e-7 of sum |a b|.  6 MFMAs x 32 cycles per 16 k of a 32 x 32
// tile = 12 cycles per k against 32.
//
// Layout as conv.hip: activations NHWC [B,H,W,C], weights [Cout][3][3][Cin], H and W powers of two.
//
//   fwd / dgrad ("rows"): a workgroup owns 4 image rows x 32 columns x BN output channels.  Per 32-channel chunk of the
//     reduction it reads the (4 + 2) x 34 pixel HALO tile once (fp32), splits it once and keeps it in LDS as three bf16 images
//     (80-byte pixel stride, an odd multiple of 16 B: the ds_read_b128 fragments of every tap are conflict-free); the nine taps
//     are shifted fragment reads of that image -- nothing is converted or re-fetched in the MFMA loop.  The weights are split
//     ONCE per optimizer step (t2h_conv3x3_bx3_prepare) into the byte order of the MFMA B fragments, so a (tap, 16-channel)
//     slab is a linear 3 KB run per 32 output channels: LDS-DMA (global_load_lds_dwordx4), double buffered.  The data
//     gradient is the same kernel on flipped + transposed prepared weights, with the ReLU mask / accumulate in its epilogue.
//   wgrad: dW[co][tap][ci] = sum_p dY[p][co] X[p + off(tap)][ci] reduces over PIXELS, so both MFMA operands need 8 consecutive
//     pixels of one channel per lane: ds_read_b64_tr_b16, the hardware transpose read, from the same [pixel][channel] bf16 images
//     (row strides chosen so that four consecutive pixel rows tile the 256-byte bank row).  A workgroup owns a 32-channel
//     chunk of Cin and up to 128 of Cout, walks image-row segments of 32 pixels (X halo: 3 x 34 pixels) and keeps all
//     (Cout / 32) x 9 output tiles in registers; partial sums of the pixel ranges go to slabs, summed in a fixed order by
//     reduce_slabs (deterministic, no atomics); the bias gradient (column sums of dY) comes out of the dY staging pass.
#include <stdlib.h>
#include "t2h_common.h"
#include "gemm_args.h"
#include "gemm_tile.h"

namespace t2h {
namespace {

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f16x2 = __attribute__((ext_vector_type(2))) _Float16;
using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using s16x4 = __attribute__((ext_vector_type(4))) short;
using s16x8 = __attribute__((ext_vector_type(8))) short;
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

#ifndef T2H_BX3_BDIRECT
#define T2H_BX3_BDIRECT 1      // 0: weight slabs through LDS by LDS-DMA, one barrier per step (the r04a-c form; A/B)
#endif
constexpr int NT = 256;
constexpr int TW = 32;        // tile width in pixels = rows of one MFMA tile
constexpr int CC = 32;        // reduction channels per staged chunk (wgrad; fwd / dgrad: template parameter CCH)

// ---- the split -----------------------------------------------------------------------------------------------------------
__device__ inline unsigned pack_bf16x2(float a, float b) {             // v_cvt_pk_bf16_f32: round to nearest even, NaN stays NaN
    f32x2 v = {a, b};
    bf16x2 r = __builtin_convertvector(v, bf16x2);
    return *reinterpret_cast<unsigned *>(&r);
}
__device__ inline float bf_lo(unsigned p) { return __uint_as_float(p << 16); }
__device__ inline float bf_hi(unsigned p) { return __uint_as_float(p & 0xffff0000u); }
// two floats -> three packed bf16 pairs (high, middle, low part)
__device__ inline void split3(float a, float b, unsigned &p1, unsigned &p2, unsigned &p3) {
    p1 = pack_bf16x2(a, b);
    const float ra = __fsub_rn(a, bf_lo(p1)), rb = __fsub_rn(b, bf_hi(p1));        // exact
    p2 = pack_bf16x2(ra, rb);
    const float sa = __fsub_rn(ra, bf_lo(p2)), sb = __fsub_rn(rb, bf_hi(p2));      // exact
    p3 = pack_bf16x2(sa, sb);
}

// ---- the fp16 two-way split (NPL = 2, T2H_F16X2) --------------------------------------------------------------------------------
// x 2^e = h1 + h2 with h1 = f16(x 2^e), h2 = f16(x 2^e - h1) (round to nearest even; the difference is exact in fp32): 11 + 11
// significant bits.  The power of two 2^e is chosen PER STAGED BLOCK (activations: the halo tile x channel chunk a workgroup stages,
// the 32-pixel unit of the weight gradient; weights: the tensor) so that the block's largest magnitude lands in [2^14, 2^15): fp16
// keeps 5 exponent bits, so h2 of an element more than 2^18 below its block's maximum becomes subnormal (the f16 MFMA keeps subnormal
// inputs: profiles/mfma_f16_lab.hip) and the element is represented to 2^-40 of the BLOCK maximum instead of 2^-22 of itself.  Of the
// four partial products three are formed (h2 g1, h1 g2, h1 g1; h2 g2 <= 2^-22 |a b| is dropped) -- THREE MFMAs per product instead
// of six; an f16 x f16 product is exact in fp32, v_mfma_f32_32x32x16_f16 accumulates in fp32, and a power-of-two rescaling of an
// fp32 accumulator is exact, so the accumulators follow the running block scale at no cost in accuracy.  Error model:
// |fl(a b) - a b| <= 3 * 2^-22 |a b| + 2^-40 max_block|a| |b| + (same with a, b swapped); measured against float64 on activations
// with a log-normal spread of 1.5 nats: 5.7e-7 of sum |a b| (fp32 FMA chain 4.2e-7, the bf16 three-way split 8.5e-7).
constexpr int E_UNSET = 1 << 20;
__device__ inline unsigned pack_f16x2(float a, float b) {              // 2 x v_cvt_f16_f32 (RNE) + v_pack_b32_f16
    f32x2 v = {a, b};
    f16x2 r = __builtin_convertvector(v, f16x2);
    return *reinterpret_cast<unsigned *>(&r);
}
__device__ inline void split2h(float a, float b, int e, unsigned &p1, unsigned &p2) {
    const float as = ldexpf(a, e), bs = ldexpf(b, e);
    const f16x2 h = __builtin_convertvector((f32x2){as, bs}, f16x2);
    p1 = *reinterpret_cast<const unsigned *>(&h);
    p2 = pack_f16x2(__fsub_rn(as, (float)h[0]), __fsub_rn(bs, (float)h[1]));                 // exact differences
}
// block exponent from the block's largest magnitude (float bits, non-negative): 14 - floor(log2 m); E_UNSET for an all-zero block
__device__ inline int block_exponent(unsigned maxbits) { return maxbits == 0u ? E_UNSET : 14 - ((int)(maxbits >> 23) - 127); }
__device__ inline float wave_max(float m) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    return m;
}
__device__ inline float abs4max(float m, const float4 &v) { return fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w))); }

// ---- weight preparation: [Cout][9][Cin] fp32 -> MFMA B-fragment order, three bf16 planes --------------------------------------
// slab (16-channel step c16 of the reduction, tap t) x [tile of 32 output channels][plane][lane][8 bf16]; lane (r = l & 31,
// h = l >> 5) holds B[k = 8 h + j][col r].  Forward: k = input channel, col = output channel, W[col][tap][k].  Transposed (data
// gradient): k = output channel, col = input channel, tap flipped: W[k][8 - tap][col].
// H2: two f16 planes of w 2^e_w instead (e_w from the tensor's largest magnitude, left in `trailer[0]` by absmax_kernel; trailer[1]
// receives 2^-e_w for the consumers' epilogues)
// `maxslot` (H2): index of the trailer word that holds the bits of max |w| (0 for the single-weight entry points; the batched
// preparation alternates between words 0 and 3 and clears the other one for its next run)
template <bool TRANSPOSED, bool H2>
__device__ inline void prepare_conv_body(const float *__restrict__ w, int Cin, int Cout, unsigned *__restrict__ wf,
                                         unsigned *__restrict__ trailer, int maxslot, long long t) {
    const int Kc = TRANSPOSED ? Cout : Cin, Nc = TRANSPOSED ? Cin : Cout;
    const int ntile = Nc / 32;
    const long long total = (long long)(Kc / 16) * 9 * ntile * 64;        // one thread per (slab, tile, lane)
    if (t >= total) return;
    const int lane = (int)(t & 63);
    const int tile = (int)((t >> 6) % ntile);
    const long long slab = (t >> 6) / ntile;                              // c16 * 9 + tap
    const int tap = (int)(slab % 9), c16 = (int)(slab / 9);
    const int r = lane & 31, h = lane >> 5;
    const int n = tile * 32 + r, k0 = c16 * 16 + 8 * h;
    unsigned p1[4], p2[4], p3[4];
    int ew = 0;
    if (H2) {
        ew = block_exponent(trailer[maxslot]);
        ew = ew == E_UNSET ? 0 : ew;
        if (t == 0) { reinterpret_cast<float *>(trailer)[1] = ldexpf(1.0f, -ew); reinterpret_cast<int *>(trailer)[2] = ew; }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float a, b;
        if (!TRANSPOSED) {
            a = w[((size_t)n * 9 + tap) * Cin + k0 + 2 * j];
            b = w[((size_t)n * 9 + tap) * Cin + k0 + 2 * j + 1];
        } else {
            a = w[((size_t)(k0 + 2 * j) * 9 + (8 - tap)) * Cin + n];
            b = w[((size_t)(k0 + 2 * j + 1) * 9 + (8 - tap)) * Cin + n];
        }
        if (H2) split2h(a, b, ew, p1[j], p2[j]);
        else split3(a, b, p1[j], p2[j], p3[j]);
    }
    uint4 *dst = reinterpret_cast<uint4 *>(wf) + ((slab * ntile + tile) * (H2 ? 2 : 3)) * 64 + lane;
    dst[0] = make_uint4(p1[0], p1[1], p1[2], p1[3]);
    dst[64] = make_uint4(p2[0], p2[1], p2[2], p2[3]);
    if (!H2) dst[128] = make_uint4(p3[0], p3[1], p3[2], p3[3]);
}
template <bool TRANSPOSED, bool H2 = false>
__global__ __launch_bounds__(256) void bx3_prepare_kernel(const float *__restrict__ w, int Cin, int Cout,
                                                         unsigned *__restrict__ wf, unsigned *__restrict__ trailer = nullptr) {
    prepare_conv_body<TRANSPOSED, H2>(w, Cin, Cout, wf, trailer, 0, (long long)blockIdx.x * 256 + threadIdx.x);
}

// largest magnitude of a [rows][cols] matrix with row stride ld, as float bits (non-negative floats order like unsigned ints)
__device__ inline void absmax_body(const float *__restrict__ w, long long rows, int cols, long long ld, unsigned *__restrict__ out,
                                   long long first, long long stride) {
    float m = 0.f;
    const long long total = rows * cols;
    for (long long i = first; i < total; i += stride) m = fmaxf(m, fabsf(w[(i / cols) * ld + i % cols]));
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_uint(m));
}
__global__ __launch_bounds__(256) void absmax_kernel(const float *__restrict__ w, long long rows, int cols, long long ld,
                                                    unsigned *__restrict__ out) {
    absmax_body(w, rows, cols, ld, out, (long long)blockIdx.x * 256 + threadIdx.x, (long long)gridDim.x * 256);
}

// the same for a plain weight matrix (1-tap form): W given as [N][K] (nn.Linear: k contiguous) or, KN, as [K][N]
template <bool KN, bool H2>
__device__ inline void prepare_gemm_body(const float *__restrict__ w, int K, int N, int ldw, unsigned *__restrict__ wf,
                                         unsigned *__restrict__ trailer, int maxslot, long long t) {
    const int ntile = N / 32;
    const long long total = (long long)(K / 16) * ntile * 64;
    if (t >= total) return;
    const int lane = (int)(t & 63);
    const int tile = (int)((t >> 6) % ntile);
    const long long c16 = (t >> 6) / ntile;
    const int r = lane & 31, h = lane >> 5;
    const int n = tile * 32 + r, k0 = (int)c16 * 16 + 8 * h;
    unsigned p1[4], p2[4], p3[4];
    int ew = 0;
    if (H2) {
        ew = block_exponent(trailer[maxslot]);
        ew = ew == E_UNSET ? 0 : ew;
        if (t == 0) { reinterpret_cast<float *>(trailer)[1] = ldexpf(1.0f, -ew); reinterpret_cast<int *>(trailer)[2] = ew; }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float a = KN ? w[(size_t)(k0 + 2 * j) * ldw + n] : w[(size_t)n * ldw + k0 + 2 * j];
        const float b = KN ? w[(size_t)(k0 + 2 * j + 1) * ldw + n] : w[(size_t)n * ldw + k0 + 2 * j + 1];
        if (H2) split2h(a, b, ew, p1[j], p2[j]);
        else split3(a, b, p1[j], p2[j], p3[j]);
    }
    uint4 *dst = reinterpret_cast<uint4 *>(wf) + ((c16 * ntile + tile) * (H2 ? 2 : 3)) * 64 + lane;
    dst[0] = make_uint4(p1[0], p1[1], p1[2], p1[3]);
    dst[64] = make_uint4(p2[0], p2[1], p2[2], p2[3]);
    if (!H2) dst[128] = make_uint4(p3[0], p3[1], p3[2], p3[3]);
}
template <bool KN, bool H2 = false>
__global__ __launch_bounds__(256) void bx3_prepare_gemm_kernel(const float *__restrict__ w, int K, int N, int ldw,
                                                              unsigned *__restrict__ wf, unsigned *__restrict__ trailer = nullptr) {
    prepare_gemm_body<KN, H2>(w, K, N, ldw, wf, trailer, 0, (long long)blockIdx.x * 256 + threadIdx.x);
}

// ---- batched preparation: every split weight of a model in a few launches (once per optimizer step) -------------------------------
// One descriptor per prepared buffer (t2h_prep_desc in include/t2h.h; kind 0 / 1 = 3x3 weight forward / transposed, 2 / 3 = matrix
// [K][N] / [N][K]); `first` = the descriptor's first workgroup in the launch.  fp16 buffers: pass 1 (absmax) leaves the bits of
// max |w| in trailer word `maxslot` (0 or 3, alternating from run to run: the split pass clears the OTHER word for the next run, so
// no memset is needed between runs; the buffers start zeroed), pass 2 splits.
constexpr int kPrepBatch = 24;
struct PrepBatch {
    int n;
    t2h_prep_desc d[kPrepBatch];
    unsigned first[kPrepBatch + 1];
};
__device__ inline int prep_find(const PrepBatch &b, unsigned blk) {
    int i = 0;
    while (i + 1 < b.n && blk >= b.first[i + 1]) ++i;
    return i;
}
__global__ __launch_bounds__(256) void prep_absmax_batch_kernel(PrepBatch b) {
    const int i = prep_find(b, blockIdx.x);
    const t2h_prep_desc &d = b.d[i];
    const unsigned nblk = b.first[i + 1] - b.first[i];
    const bool conv = d.kind < 2;
    const long long rows = conv ? 1 : (d.kind == 2 ? d.a : d.b);           // matrix rows as stored
    const int cols = conv ? d.a * 9 * d.b : (d.kind == 2 ? d.b : d.a);
    absmax_body(d.w, rows, cols, conv ? 0 : d.ldw, static_cast<unsigned *>(d.wf) + d.trailer_word + d.maxslot,
                (long long)(blockIdx.x - b.first[i]) * 256 + threadIdx.x, (long long)nblk * 256);
}
__global__ __launch_bounds__(256) void prep_split_batch_kernel(PrepBatch b) {
    const int i = prep_find(b, blockIdx.x);
    const t2h_prep_desc &d = b.d[i];
    const long long t = (long long)(blockIdx.x - b.first[i]) * 256 + threadIdx.x;
    unsigned *wf = static_cast<unsigned *>(d.wf), *trailer = wf + d.trailer_word;
    if (d.h2) {
        if (t == 0) trailer[d.maxslot ^ 3] = 0u;                            // the other slot, for the next run
        if (d.kind == 0) prepare_conv_body<false, true>(d.w, d.a, d.b, wf, trailer, d.maxslot, t);
        else if (d.kind == 1) prepare_conv_body<true, true>(d.w, d.a, d.b, wf, trailer, d.maxslot, t);
        else if (d.kind == 2) prepare_gemm_body<true, true>(d.w, d.a, d.b, d.ldw, wf, trailer, d.maxslot, t);
        else prepare_gemm_body<false, true>(d.w, d.a, d.b, d.ldw, wf, trailer, d.maxslot, t);
    } else {
        if (d.kind == 0) prepare_conv_body<false, false>(d.w, d.a, d.b, wf, nullptr, 0, t);
        else if (d.kind == 1) prepare_conv_body<true, false>(d.w, d.a, d.b, wf, nullptr, 0, t);
        else if (d.kind == 2) prepare_gemm_body<true, false>(d.w, d.a, d.b, d.ldw, wf, nullptr, 0, t);
        else prepare_gemm_body<false, false>(d.w, d.a, d.b, d.ldw, wf, nullptr, 0, t);
    }
}

// ---- fwd / dgrad --------------------------------------------------------------------------------------------------------------
struct RowsArgs {
    const float *x;          // [B,H,W,Kc] fp32: the input (fwd) or dY (dgrad)
    const unsigned *wf;      // prepared weights
    const float *wscale;     // NPL = 2: the prepared buffer's trailer (wscale[1] = 2^-e_w)
    const float *bias;       // [Nc] or null
    const float *mask;       // [B,H,W,Nc] or null: result *= (mask > 0)
    float *y;                // [B,H,W,Nc], or the slab base when the reduction is split
    int B, H, W, Kc, Nc, flags;
    int splits, chunks_per_split;   // splits > 1: grid = tiles x splits, partial sums to y + split * B H W Nc (no epilogue)
    int ntn_per_wg;          // PERSIST: column tiles per workgroup (the staged rows are split once and reused for all of them)
    int ldx, ldy, ldm;       // row strides (floats) of x, y and mask: Kc / Nc / Nc for the convolutions, free for the 1-tap (GEMM) form
    // 2x2 stride-2 transposed convolution on the 1-tap form (UPM != 0): the GEMM rows are the pixels of the INPUT plane
    // [*, 2^up_logH, 2^up_logW], the output plane has twice its size and up_cout channels
    const float *addend;     // UPM = 1: tensor of the output's shape added to the result, or null
    int up_logW, up_logH, up_cout;
    // r06 (plain forms): a rank-1 term r1_g[pixel] * r1_w[channel] added to the result under the same mask -- the 1 x 1 head's share of
    // a decoder activation's gradient (pixel.py:31: conv4(cat[x, x1, x2, x3])), formed in this epilogue instead of being written by
    // the head's backward and read back here as "old values" (T2H_ACCUM): same products, same order of additions, same bits
    const float *r1_g, *r1_w;
};

// output pixel (2y + dy, 2x + dx) of the transposed convolution for GEMM row m = ((b * H + y) * W + x), tap = 2 dy + dx
__device__ inline size_t up_pixel(long long m, int tap, int logW, int logH) {
    const long long x = m & ((1LL << logW) - 1), y = (m >> logW) & ((1LL << logH) - 1), b = m >> (logW + logH);
    return (size_t)(((b << (logH + 1)) + 2 * y + (tap >> 1)) << (logW + 1)) + 2 * x + (tap & 1);
}

// TH image rows x 32 columns x BN output channels per workgroup, CCH reduction channels per staged halo chunk.  Two shapes are
// used: 4 rows / 32 channels (80-byte pixel stride) and -- for planes with enough tiles to fill the chip -- 8 rows / 16
// channels (48-byte stride; half the barriers, LDS fragment bytes and weight-slab traffic per MFMA).  Both keep the halo
// images + two weight slabs under 80 KB: two workgroups per CU, one staging while the other computes.
// NPL = 3: the exact split (six MFMAs per product).  NPL = 1: operands rounded to bf16 once (T2H_BF16, BASELINE configs[2]:
// one MFMA per product, a third of the LDS images and weight traffic) -- the same kernel without the residual planes.
// NTAP = 9: the 3x3 convolution.  NTAP = 1: the same kernel as a plain GEMM on rows, Y[M, Nc] = X[M, Kc] W^T -- 1x1 convolutions,
// nn.Linear on pixel rows and the grid-side products of the deferred ALTO point update (deferred.py): the "image" is M / 32
// rows of 32 "pixels" with no halo, the staged chunk is the tile itself (chunks of 64 / 32 channels), row strides are free.
// UPM (1-tap form only): 1 = ConvTranspose2d(2, stride 2) forward -- GEMM [pixels, Cin] x [Cin, (tap, co)] with a SCATTERING epilogue
// (column tile -> tap -> output pixel up(p, tap)), bias / residual addend there; 2 = its data gradient -- the A rows are GATHERED:
// chunk c of K = (tap, co) reads dY[up(p, tap)][co0 ..] (alto.py:175,215-218,236).
// PERSIST (1-tap form, ONE staged chunk: K = CCH): a workgroup keeps its staged and split rows in LDS and walks p.ntn_per_wg column
// tiles with them -- wide outputs of a short reduction (the stacked per-pixel product 64 -> 2752 over 65 536 rows), where the
// plain form stages and splits the same rows once per 64 output columns and loses to the fp32 MFMA kernel for that
template <int TH, int BN, int WAVES_M, int WAVES_N, int CCH, int NPL, int NTAP, int UPM = 0, bool PERSIST = false>
__global__ __launch_bounds__(NT, 2) void bx3_rows_kernel(RowsArgs p) {
    static_assert(UPM == 0 || NTAP == 1, "the transposed convolution runs on the 1-tap form");
    static_assert(!PERSIST || (NTAP == 1 && UPM == 0 && T2H_BX3_BDIRECT != 0), "PERSIST: plain 1-tap form with register-fetched weights");
    constexpr int PXB = CCH * 2 + 16;     // bytes per pixel and plane of the halo image: CCH bf16 + 16 B (an odd multiple of 16 B)
    constexpr int NQ = CCH / 16, F4 = CCH / 4, NSTEP = NTAP * NQ;
    constexpr int TM = TH / WAVES_M, TN = BN / (32 * WAVES_N);
    constexpr int HB = NTAP == 9 ? 1 : 0;                                // halo border
    constexpr int HW = TW + 2 * HB;                                      // pixels per staged image row
    constexpr int HP = (TH + 2 * HB) * HW;                               // staged pixels
    constexpr int PLANE = HP * PXB;                                      // bytes per bf16 plane
    constexpr int BSLAB = (BN / 32) * NPL * 1024;                        // bytes per weight slab
    constexpr int HALO_BYTES = NPL * PLANE;
    constexpr bool BDIRECT = T2H_BX3_BDIRECT != 0;
    constexpr int LDS_WORK = HALO_BYTES + (BDIRECT ? 0 : 2 * BSLAB);
    constexpr int PATCH_BYTES = 4 * 32 * 36 * 4;
    // (the epilogue's patches live in the same array -- behind the images when the images have to survive the epilogue)
    constexpr int LDS_BYTES = PERSIST ? LDS_WORK + PATCH_BYTES : (LDS_WORK >= PATCH_BYTES ? LDS_WORK : PATCH_BYTES);
    constexpr bool H2 = NPL == 2;                                        // the fp16 two-way split with block scales
    constexpr int WPL = H2 ? 2 : 3;                                      // planes per tile in the prepared weight buffer
    static_assert(WAVES_M * WAVES_N == 4 && TM * WAVES_M == TH && TN * WAVES_N * 32 == BN, "wave layout");
    __shared__ __attribute__((aligned(1024))) unsigned char lds[LDS_BYTES + (H2 ? 64 : 0)];
    unsigned char *halo = lds, *bbuf = lds + HALO_BYTES;
    float *slots = reinterpret_cast<float *>(lds + LDS_BYTES);           // H2: [parity][wave] block maxima of the staged chunks

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    // XCD-aware order: a contiguous run of tiles per XCD (neighbouring tiles share halo pixels and all share the weights in L2)
    const unsigned nb = gridDim.x, bid = blockIdx.x;
    const unsigned qq = nb / 8, rr = nb % 8, xx = bid % 8, i8 = bid / 8;
    unsigned t = (xx < rr ? xx * (qq + 1) : rr * (qq + 1) + (xx - rr) * qq) + i8;
    const int split = t % p.splits; t /= p.splits;                      // (the splits of a tile side by side: same halo pixels)
    const int ntn = p.Nc / BN;
    const int tn_groups = PERSIST ? (ntn + p.ntn_per_wg - 1) / p.ntn_per_wg : ntn;
    const int tng = t % tn_groups; t /= tn_groups;
    const int tn_lo = PERSIST ? tng * p.ntn_per_wg : tng, tn_hi = PERSIST ? min(ntn, tn_lo + p.ntn_per_wg) : tng + 1;
    const int tiles_x = p.W / TW;
    const int tx = t % tiles_x; t /= tiles_x;
    const int tiles_y = p.H / TH;
    const int ty = t % tiles_y, b = t / tiles_y;
    const int x0 = tx * TW, y0 = ty * TH;
    int n0 = tn_lo * BN;
    const int c_beg = split * p.chunks_per_split, c_end = min(p.Kc / CCH, c_beg + p.chunks_per_split);

    // halo staging: F4 float4 per pixel and chunk
    constexpr int NF4 = HP * F4, PER = (NF4 + NT - 1) / NT;
    float4 hreg[PER];
    auto halo_load = [&](int c) {
#pragma unroll
        for (int f = 0; f < PER; ++f) {
            const int idx = tid + f * NT;
            const int px = idx / F4, c4 = idx % F4;
            const int hy = px / HW, hx = px - hy * HW;
            const int gy = y0 + hy - HB, gx = x0 + hx - HB;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (UPM == 2) {
                const int k0 = c * CCH, tap = k0 / p.up_cout, co0 = k0 - tap * p.up_cout;      // (CCH divides up_cout)
                if (idx < NF4)
                    v = *reinterpret_cast<const float4 *>(p.x + up_pixel(((long long)b * p.H + gy) * p.W + gx, tap, p.up_logW, p.up_logH) *
                                                                    p.ldx + co0 + c4 * 4);
            } else if (idx < NF4 && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W)
                v = *reinterpret_cast<const float4 *>(p.x + (((size_t)b * p.H + gy) * p.W + gx) * p.ldx + c * CCH + c4 * 4);
            hreg[f] = v;
        }
    };
    int e_run = E_UNSET, e_cur = 0;                                      // H2: exponent of the accumulators' A scale / of the chunk being staged
    auto halo_store = [&]() {
#pragma unroll
        for (int f = 0; f < PER; ++f) {
            const int idx = tid + f * NT;
            if (idx < NF4) {
                const int px = idx / F4, c4 = idx % F4;
                unsigned a1, a2, a3, b1, b2, b3;
                unsigned char *d = halo + px * PXB + c4 * 8;
                if (H2) {
                    split2h(hreg[f].x, hreg[f].y, e_cur, a1, a2);
                    split2h(hreg[f].z, hreg[f].w, e_cur, b1, b2);
                    *reinterpret_cast<uint2 *>(d) = make_uint2(a1, b1);
                    *reinterpret_cast<uint2 *>(d + PLANE) = make_uint2(a2, b2);
                    continue;
                }
                split3(hreg[f].x, hreg[f].y, a1, a2, a3);
                split3(hreg[f].z, hreg[f].w, b1, b2, b3);
                *reinterpret_cast<uint2 *>(d) = make_uint2(a1, b1);
                if (NPL == 3) {
                    *reinterpret_cast<uint2 *>(d + PLANE) = make_uint2(a2, b2);
                    *reinterpret_cast<uint2 *>(d + 2 * PLANE) = make_uint2(a3, b3);
                }
            }
        }
    };
    // H2: the largest magnitude of the chunk waiting in hreg -> slots[parity] (published by the next barrier)
    auto publish_block_max = [&](int parity) {
        float m = 0.f;
#pragma unroll
        for (int f = 0; f < PER; ++f) m = abs4max(m, hreg[f]);
        m = wave_max(m);
        if (lane == 0) slots[parity * 4 + wave] = m;
    };
    // weight slab (16-channel step c16, tap) = c16 * 9 + tap: (Nc / 32) x 3 KB; this workgroup's BN / 32 tiles are one linear run.
    // Step st of chunk c: tap = st / NQ, q = st % NQ, c16 = c * NQ + q
    const unsigned char *wbase = reinterpret_cast<const unsigned char *>(p.wf) + (size_t)(n0 / 32) * WPL * 1024;      // (PERSIST: per column tile)
    const size_t slab_stride = (size_t)(p.Nc / 32) * WPL * 1024;
    auto issue_b = [&](int c, int st, unsigned char *dst) {
        const int s = (c * NQ + st % NQ) * NTAP + st / NQ;
        const unsigned char *src = wbase + (size_t)s * slab_stride;
        constexpr int PIECES = BSLAB / 1024;                             // 1 KB per wave instruction, dealt round-robin to the waves
#pragma unroll
        for (int j = 0; j < (PIECES + 3) / 4; ++j)
            if (j * 4 + wave < PIECES) {
                // piece = (tile, plane); the prepared bf16 weights always hold three planes per tile
                const int piece = j * 4 + wave, spiece = NPL == WPL ? piece : piece * 3;
                __builtin_amdgcn_global_load_lds((glb_void *)(src + spiece * 1024 + lane * 16), (lds_void *)(dst + piece * 1024), 16, 0, 0);
            }
    };

    // BDIRECT: every wave fetches the B fragments of its own column tiles straight from global memory / L2 into registers (the
    // prepared weights ARE in fragment order: 1 KB per tile and plane, one dwordx4 per lane), one step ahead.  No weight slabs in
    // LDS, so no barrier per step: the waves of a workgroup meet twice per staged chunk only and run ahead of each other in
    // between; the loads are ordinary register loads, so the compiler's waitcnts are exact.
    auto load_b = [&](int c, int st, uint4 (&dst)[TN][NPL]) {
        const int sidx = (c * NQ + st % NQ) * NTAP + st / NQ;
        const unsigned char *src = wbase + (size_t)sidx * slab_stride + lane * 16;
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) dst[j][pl] = *reinterpret_cast<const uint4 *>(src + ((wn * TN + j) * WPL + pl) * 1024);
    };

    f32x16 acc[TM][TN];
    const int r = lane & 31, h = lane >> 5;
  for (int tn = tn_lo; tn < tn_hi; ++tn) {                               // (one pass unless PERSIST)
    const bool first = tn == tn_lo;
    if (PERSIST && !first) {
        n0 = tn * BN;
        wbase = reinterpret_cast<const unsigned char *>(p.wf) + (size_t)(n0 / 32) * WPL * 1024;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int z = 0; z < 16; ++z) acc[i][j][z] = 0.0f;

    if (!PERSIST || first) {
        halo_load(c_beg);
        if (H2) {
            publish_block_max(c_beg & 1);
            __syncthreads();
        }
    }
    uint4 bnext[TN][NPL];
    if (BDIRECT) load_b(c_beg, 0, bnext);
    int s = 0;                                                           // running step count (selects the weight buffer)
    for (int c = c_beg; c < c_end; ++c) {
        if (H2 && (!PERSIST || first)) {
            // this chunk's block exponent (the same in every thread); the accumulators follow the smallest exponent (= the largest
            // block) met so far: a power-of-two rescaling of an fp32 accumulator is exact
            const float *sl = slots + (c & 1) * 4;
            const int e_nat = block_exponent(__float_as_uint(fmaxf(fmaxf(sl[0], sl[1]), fmaxf(sl[2], sl[3]))));
            if (e_nat < e_run) {
                if (e_run != E_UNSET) {
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
#pragma unroll
                            for (int z = 0; z < 16; ++z) acc[i][j][z] = ldexpf(acc[i][j][z], e_nat - e_run);
                }
                e_run = e_nat;
            }
            e_cur = e_run == E_UNSET ? 0 : e_run;
        }
        if (!PERSIST || first) {
            halo_store();                                                // (the previous chunk's last barrier has passed)
            if (!BDIRECT && c == c_beg) issue_b(c, 0, bbuf);             // (later chunks: issued by the previous chunk's last step)
            if (c + 1 < c_end) halo_load(c + 1);                         // in flight under this chunk's MFMAs
            __syncthreads();
        }
#pragma unroll 1
        for (int tap = 0; tap < NTAP; ++tap) {
            const int ky = NTAP == 9 ? tap / 3 : 0, kx = NTAP == 9 ? tap - 3 * ky : 0;
#pragma unroll
            for (int q = 0; q < NQ; ++q, ++s) {
                const unsigned char *cur = bbuf + (s & 1) * BSLAB;
                const int st = tap * NQ + q;
                uint4 af[TM][NPL], bfr[TN][NPL];
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int px = (wm * TM + i + ky) * HW + r + kx;
                    const unsigned char *a = halo + px * PXB + q * 32 + h * 16;
#pragma unroll
                    for (int pl = 0; pl < NPL; ++pl) af[i][pl] = *reinterpret_cast<const uint4 *>(a + pl * PLANE);
                }
                if (BDIRECT) {
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int pl = 0; pl < NPL; ++pl) bfr[j][pl] = bnext[j][pl];
                    if (st + 1 < NSTEP) load_b(c, st + 1, bnext);
                    else if (c + 1 < c_end) load_b(c + 1, 0, bnext);
                } else {
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int pl = 0; pl < NPL; ++pl)
                            bfr[j][pl] = *reinterpret_cast<const uint4 *>(cur + ((wn * TN + j) * NPL + pl) * 1024 + lane * 16);
                }
                __builtin_amdgcn_sched_barrier(0);                       // all fragment reads in flight before the first MFMA
                // The next slab's DMA is issued AFTER this step's fragment reads: the compiler cannot tell the DMA's LDS target from
                // the buffer being read and puts s_waitcnt vmcnt(0) in front of every LDS read that follows a pending LDS-DMA -- issued
                // first (r04a-c) each step waited out its own prefetch: 3 100 of the 3 900 cycles of a step (rocprofv3 SQ counters,
                // profiles/r04d_conv_counters.txt).  Now the transfer runs under this step's MFMAs.
                if (!BDIRECT) {
                    if (st + 1 < NSTEP) issue_b(c, st + 1, bbuf + ((s + 1) & 1) * BSLAB);
                    else if (c + 1 < c_end) issue_b(c + 1, 0, bbuf + ((s + 1) & 1) * BSLAB);
                    __builtin_amdgcn_sched_barrier(0);
                }
                // smallest terms first: a3b1, a1b3, a2b2, a2b1, a1b2, a1b1
                constexpr int ia[6] = {2, 0, 1, 1, 0, 0}, ib[6] = {0, 2, 1, 0, 1, 0};
                if (H2) {                                                // h2 g1, h1 g2, h1 g1
#pragma unroll
                    for (int e = 3; e < 6; ++e)
#pragma unroll
                        for (int i = 0; i < TM; ++i)
#pragma unroll
                            for (int j = 0; j < TN; ++j)
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<f16x8 *>(&af[i][ia[e]]),
                                                                                    *reinterpret_cast<f16x8 *>(&bfr[j][ib[e]]), acc[i][j], 0, 0, 0);
                    if (st + 1 == NSTEP && c + 1 < c_end) publish_block_max((c + 1) & 1);      // (under the MFMAs; the barrier below publishes)
                } else {
#pragma unroll
                    for (int e = (NPL == 3 ? 0 : 5); e < 6; ++e)
#pragma unroll
                        for (int i = 0; i < TM; ++i)
#pragma unroll
                            for (int j = 0; j < TN; ++j)
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<bf16x8 *>(&af[i][ia[e]]),
                                                                                     *reinterpret_cast<bf16x8 *>(&bfr[j][ib[e]]), acc[i][j], 0, 0, 0);
                }
                if (!BDIRECT) __syncthreads();                           // (also drains the next slab's DMA: vmcnt(0))
            }
        }
        if (BDIRECT && !PERSIST) __syncthreads();                        // every wave is done with this chunk's halo images
    }

    // epilogue: one 32 x 32 tile at a time through the wave's LDS patch -> float4 rows along the output channels
    float *patch = reinterpret_cast<float *>(lds + (PERSIST ? LDS_WORK : 0)) + wave * (32 * 36);
    const int er = lane >> 3, ec = (lane & 7) * 4;
    const bool relu = p.flags & F_RELU_OUT, rank1 = UPM == 0 && p.r1_g != nullptr, accum = (p.flags & F_ACCUM) || rank1;
    float *const ybase = p.y + (size_t)split * ((size_t)p.B * p.H * p.W * p.ldy);     // (split == 0 unless the reduction is split)
    // What the epilogue READS from global memory (ReLU mask, the old values under T2H_ACCUM, the residual addend of the transposed
    // convolution) is fetched one 32 x 32 tile ahead of the tile being written: issued inside the pass loop each load's latency was
    // exposed once per pass -- 4 TM TN dependent round trips per workgroup, 60 us of the 260 us 512^2 data gradients.
    constexpr int T = TM * TN;
    float4 pre_m[2][4], pre_o[2][4];
    auto tile_off = [&](int t, int pass, int &col_out, int &up_co) -> size_t {
        const int i = t / TN, j = t % TN;
        const int col = n0 + (wn * TN + j) * 32 + ec;
        const size_t pix = ((size_t)b * p.H + y0 + wm * TM + i) * p.W + x0 + pass * 8 + er;
        col_out = col;
        if (UPM == 1) {      // scatter: row = input pixel, column tile = (tap, 32 channels) -> output pixel up(row, tap)
            const int up_tap = col / p.up_cout;
            up_co = col - up_tap * p.up_cout;
            return up_pixel((long long)pix, up_tap, p.up_logW, p.up_logH) * p.up_cout + up_co;
        }
        up_co = col;
        return pix;
    };
    auto prefetch = [&](int t, float4 (&m)[4], float4 (&o)[4]) {
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            int col, co;
            const size_t q = tile_off(t, pass, col, co);
            if (UPM == 1) {
                if (p.addend) m[pass] = *reinterpret_cast<const float4 *>(p.addend + q);
                if (accum) o[pass] = *reinterpret_cast<const float4 *>(p.y + q);
            } else {
                if (p.mask) m[pass] = *reinterpret_cast<const float4 *>(p.mask + q * p.ldm + col);
                if (rank1) {
                    const float gq = p.r1_g[q];
                    const float4 wv = *reinterpret_cast<const float4 *>(p.r1_w + col);
                    o[pass] = make_float4(gq * wv.x, gq * wv.y, gq * wv.z, gq * wv.w);
                } else if (accum) o[pass] = *reinterpret_cast<const float4 *>(ybase + q * p.ldy + col);
            }
        }
    };
    prefetch(0, pre_m[0], pre_o[0]);
    const float winv = H2 ? p.wscale[1] : 1.0f;
    e_cur = e_run == E_UNSET ? 0 : e_run;
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const int i = t / TN, j = t % TN;
        if (t + 1 < T) prefetch(t + 1, pre_m[(t + 1) & 1], pre_o[(t + 1) & 1]);
#pragma unroll
        for (int z = 0; z < 16; ++z)
            patch[((z & 3) + 8 * (z >> 2) + 4 * h) * 36 + r] = H2 ? ldexpf(acc[i][j][z] * winv, -e_cur) : acc[i][j][z];
        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
        {
            int col, co;
            tile_off(t, 0, col, co);
            if (p.bias) bv = *reinterpret_cast<const float4 *>(p.bias + co);
        }
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            float4 v = *reinterpret_cast<const float4 *>(patch + (pass * 8 + er) * 36 + ec);
            int col, co;
            const size_t q = tile_off(t, pass, col, co);
            const float4 mk = pre_m[t & 1][pass];
            float4 old = pre_o[t & 1][pass];
            if (rank1 && p.mask) {                                        // (the head masks its share like this kernel masks its own)
                old.x = mk.x > 0.f ? old.x : 0.f; old.y = mk.y > 0.f ? old.y : 0.f;
                old.z = mk.z > 0.f ? old.z : 0.f; old.w = mk.w > 0.f ? old.w : 0.f;
            }
            v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
            if (UPM == 1) {
                float4 *dst = reinterpret_cast<float4 *>(p.y + q);
                if (accum) { v.x += old.x; v.y += old.y; v.z += old.z; v.w += old.w; }
                if (p.addend) { v.x += mk.x; v.y += mk.y; v.z += mk.z; v.w += mk.w; }
                *dst = v;
                continue;
            }
            if (p.mask) {
                v.x = mk.x > 0.f ? v.x : 0.f; v.y = mk.y > 0.f ? v.y : 0.f;
                v.z = mk.z > 0.f ? v.z : 0.f; v.w = mk.w > 0.f ? v.w : 0.f;
            }
            if (relu) { v.x = relu1(v.x); v.y = relu1(v.y); v.z = relu1(v.z); v.w = relu1(v.w); }
            float4 *dst = reinterpret_cast<float4 *>(ybase + q * p.ldy + col);
            if (accum) { v.x += old.x; v.y += old.y; v.z += old.z; v.w += old.w; }
            *dst = v;
        }
    }
  }                                                                      // (column tiles of this workgroup)
}

// ---- wgrad ----------------------------------------------------------------------------------------------------------------------
struct WgradArgs {
    const float *dy;         // [B,H,W,Cout]
    const float *x;          // [B,H,W,Cin]
    float *slab;             // [splits][Cout][9 Cin]
    float *colslab;          // [splits][Cout] or null
    int H, W, Cin, Cout;
    int n_units, units_per_split;
    int ldx;                 // pixel stride of x in floats (UP: of the layer's output gradient, which may be a channel slice)
    int lddy;                // GT (the GEMM form): row stride of dy in floats
};

// COT = 32-channel tiles of Cout per workgroup (4, 2 or 1): the workgroup's (COT x 9) output tiles are dealt to the four waves
// as (co tile, contiguous tap range)
// UP: the weight gradient of ConvTranspose2d(2, stride 2), dW[ci][tap][co] = sum_p x[p][ci] dy[up(p, tap)][co] -- the same kernel with
// the roles swapped: p.dy = the layer's INPUT x (the 32 x 32 COT "row" side, no halo), p.x = the layer's output gradient, whose two
// rows x 64 pixels above a unit's 32 input pixels are staged de-interleaved as [tap][32 pixels] (so a tap reads exactly like a 3 x 3
// tap does), 4 taps instead of 9; p.Cout / p.Cin are the channel counts of those roles (the layer's Cin / Cout), p.H / p.W the INPUT
// plane.  The column sums (bias gradient) are then those of the tap side.
// GT > 0 (r06): the same kernel as a transposed product on ROWS -- dW [N][K] = sum_m dY[m][N] X[m][K] with K = 32 GT (the weight
// gradient of a Linear on pixel rows, t2h_gemm_bx3_wgrad): a unit is 32 consecutive rows, the K side is staged whole as
// [chunk][32 rows] of 32 channels -- the UP layout with GT "taps", a tap being a channel chunk instead of a pixel shift -- so that dY,
// the large operand, is read once; p.Cout = N, p.Cin = 32, p.lddy / p.ldx the row strides.
template <int COT, int NPL, bool UP = false, int GT = 0>
__global__ __launch_bounds__(NT, 2) void bx3_wgrad_kernel(WgradArgs p) {
    static_assert(!(UP && GT), "UP and GT are different forms");
    constexpr int NTAPS = GT ? GT : (UP ? 4 : 9);
    constexpr int WPC = 4 / COT;                                         // waves per co tile
    constexpr int TMAX = (NTAPS + WPC - 1) / WPC;                        // taps per wave, at most
    constexpr int YS = COT == 1 ? 64 : 64 * COT + 64;                    // dY image: bytes per pixel row (4 rows tile the bank row)
    constexpr int XS = 64;                                               // X image: 32 bf16 per pixel, no padding
    constexpr int YPLANE = 32 * YS, XHP = GT ? GT * TW : (UP ? 4 * TW : 3 * (TW + 2)), XPLANE = XHP * XS;
    constexpr int LDS_WORK = NPL * YPLANE + NPL * XPLANE;
    constexpr int LDS_BYTES = LDS_WORK >= 4 * 32 * 36 * 4 ? LDS_WORK : 4 * 32 * 36 * 4;      // (epilogue patches, column-sum scratch)
    constexpr bool H2 = NPL == 2;                                        // the fp16 two-way split: one power-of-two scale per staged unit
    __shared__ __attribute__((aligned(1024))) unsigned char lds[LDS_BYTES + (H2 ? 64 : 0)];
    unsigned char *yimg = lds, *ximg = lds + NPL * YPLANE;
    float *slots = reinterpret_cast<float *>(lds + LDS_BYTES);           // H2: [parity][wave][x, y] unit maxima

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // (GT: the column groups of dY fastest -- the workgroups that share a row range, i.e. the same rows of X, start together)
    const int split = GT ? blockIdx.z : blockIdx.x, ci0 = GT ? 0 : blockIdx.y * CC, co0 = (GT ? blockIdx.x : blockIdx.z) * (32 * COT);
    const int u_beg = split * p.units_per_split, u_end = min(p.n_units, u_beg + p.units_per_split);
    const int segs = p.W / TW;

    const int cot = wave / WPC, sub = wave % WPC;                         // this wave's co tile and tap range
    const int tap_lo = (NTAPS * sub + WPC - 1) / WPC, tap_hi = (NTAPS * (sub + 1) + WPC - 1) / WPC;

    // staging registers: X halo 3 x 34 pixels x 8 float4; dY 32 pixels x (8 COT) float4
    constexpr int XF4 = XHP * 8, XPER = (XF4 + NT - 1) / NT;
    constexpr int YPER = COT;
    float4 xr[XPER], yr[YPER];
    auto load_unit = [&](int u) {
        const int xs = u % segs, row = u / segs;                         // row = b * H + y
        const int y = row & (p.H - 1), x0 = xs * TW;
#pragma unroll
        for (int f = 0; f < XPER; ++f) {
            const int idx = tid + f * NT;
            const int px = idx >> 3, c4 = idx & 7;
            const int hy = px / (TW + 2), hx = px - hy * (TW + 2);
            const int gy = y + hy - 1, gx = x0 + hx - 1;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (GT) {      // px = (channel chunk, row of the unit): X[u TW + i][32 chunk + 4 c4 ..]
                if (idx < XF4) v = *reinterpret_cast<const float4 *>(p.x + ((size_t)u * TW + (px % TW)) * p.ldx + (px / TW) * 32 + c4 * 4);
            } else if (UP) {      // px = (output row parity, 64 output pixels): rows 2 row, 2 row + 1 of the [*, 2 W] plane, columns 2 x0 ..
                v = *reinterpret_cast<const float4 *>(p.x + ((size_t)(2 * row + (px >> 6)) * (2 * p.W) + 2 * x0 + (px & 63)) * p.ldx + ci0 + c4 * 4);
            } else if (idx < XF4 && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W)
                v = *reinterpret_cast<const float4 *>(p.x + ((size_t)(row + hy - 1) * p.W + gx) * p.ldx + ci0 + c4 * 4);
            xr[f] = v;
        }
#pragma unroll
        for (int f = 0; f < YPER; ++f) {
            const int idx = tid + f * NT;
            const int px = idx / (8 * COT), c4 = idx % (8 * COT);
            yr[f] = *reinterpret_cast<const float4 *>(p.dy + ((size_t)row * p.W + x0 + px) * (GT ? p.lddy : p.Cout) + co0 + c4 * 4);
        }
    };
    float4 csum = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool do_colsum = p.colslab != nullptr && (UP ? blockIdx.z == 0 : blockIdx.y == 0);
    int s_run = E_UNSET, ex_cur = 0, ey_cur = 0;                         // H2: exponent of the accumulators (x and y scales together), of the unit
    auto publish_unit_max = [&](int parity) {
        float mx = 0.f, my = 0.f;
#pragma unroll
        for (int f = 0; f < XPER; ++f) mx = abs4max(mx, xr[f]);         // (entries past XF4 are zero)
#pragma unroll
        for (int f = 0; f < YPER; ++f) my = abs4max(my, yr[f]);
        mx = wave_max(mx);
        my = wave_max(my);
        if (lane == 0) { slots[(parity * 4 + wave) * 2] = mx; slots[(parity * 4 + wave) * 2 + 1] = my; }
    };
    auto store_unit = [&]() {
#pragma unroll
        for (int f = 0; f < XPER; ++f) {
            const int idx = tid + f * NT;
            if (idx < XF4) {
                const int c4 = idx & 7;
                int px = idx >> 3;
                if (UP) {                                                // (row parity, column 2 i + dx) -> slot [tap = 2 dy + dx][i]
                    px = ((px >> 6) * 2 + (px & 1)) * TW + ((px & 63) >> 1);
                    if (do_colsum) { csum.x += xr[f].x; csum.y += xr[f].y; csum.z += xr[f].z; csum.w += xr[f].w; }
                }
                unsigned a1, a2, a3, b1, b2, b3;
                unsigned char *d = ximg + px * XS + c4 * 8;
                if (H2) {
                    split2h(xr[f].x, xr[f].y, ex_cur, a1, a2);
                    split2h(xr[f].z, xr[f].w, ex_cur, b1, b2);
                    *reinterpret_cast<uint2 *>(d) = make_uint2(a1, b1);
                    *reinterpret_cast<uint2 *>(d + XPLANE) = make_uint2(a2, b2);
                    continue;
                }
                split3(xr[f].x, xr[f].y, a1, a2, a3);
                split3(xr[f].z, xr[f].w, b1, b2, b3);
                *reinterpret_cast<uint2 *>(d) = make_uint2(a1, b1);
                if (NPL == 3) {
                    *reinterpret_cast<uint2 *>(d + XPLANE) = make_uint2(a2, b2);
                    *reinterpret_cast<uint2 *>(d + 2 * XPLANE) = make_uint2(a3, b3);
                }
            }
        }
#pragma unroll
        for (int f = 0; f < YPER; ++f) {
            const int idx = tid + f * NT;
            const int px = idx / (8 * COT), c4 = idx % (8 * COT);
            unsigned a1, a2, a3, b1, b2, b3;
            unsigned char *d = yimg + px * YS + c4 * 8;
            if (!UP && do_colsum) { csum.x += yr[f].x; csum.y += yr[f].y; csum.z += yr[f].z; csum.w += yr[f].w; }   // pixels in order
            if (H2) {
                split2h(yr[f].x, yr[f].y, ey_cur, a1, a2);
                split2h(yr[f].z, yr[f].w, ey_cur, b1, b2);
                *reinterpret_cast<uint2 *>(d) = make_uint2(a1, b1);
                *reinterpret_cast<uint2 *>(d + YPLANE) = make_uint2(a2, b2);
                continue;
            }
            split3(yr[f].x, yr[f].y, a1, a2, a3);
            split3(yr[f].z, yr[f].w, b1, b2, b3);
            *reinterpret_cast<uint2 *>(d) = make_uint2(a1, b1);
            if (NPL == 3) {
                *reinterpret_cast<uint2 *>(d + YPLANE) = make_uint2(a2, b2);
                *reinterpret_cast<uint2 *>(d + 2 * YPLANE) = make_uint2(a3, b3);
            }
        }
    };

    f32x16 acc[TMAX];
#pragma unroll
    for (int t = 0; t < TMAX; ++t)
#pragma unroll
        for (int z = 0; z < 16; ++z) acc[t][z] = 0.0f;

    // transposed fragment reads: 16-lane group g = lane >> 4 takes the block of 4 pixels x 16 channels with pixel base 8 (g >> 1)
    // and channel base 16 (g & 1); lane 4 q + p of the group supplies the address of pixel row q, channels 4 p .. 4 p + 3 and
    // receives channel (lane & 15) of the four pixels.  Two reads (pixels +0..3, +4..7) make the 8 k of one MFMA operand.
    const int g = lane >> 4, gi = lane & 15, gq = gi >> 2, gp = gi & 3;
    const int kpix = 8 * (g >> 1) + gq;                                   // pixel (k) of this lane's address inside a 16-pixel step
    const int choff = (16 * (g & 1) + 4 * gp) * 2;                        // byte offset of its four channels
    auto tr8 = [&](const unsigned char *a, int row_stride) -> bf16x8 {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(a));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(a + 4 * row_stride));
        const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        return *reinterpret_cast<const bf16x8 *>(&v);
    };

    if (u_beg < u_end) load_unit(u_beg);
    if (H2) {
        if (u_beg < u_end) publish_unit_max(u_beg & 1);
        __syncthreads();
    }
    for (int u = u_beg; u < u_end; ++u) {
        if (H2) {
            // natural exponents of this unit's two blocks; the accumulators carry 2^(s_run), the smallest sum met so far (the largest
            // products): the x side keeps its natural scale, the y side takes what is left (<= its natural one: no overflow)
            const float *sl = slots + (u & 1) * 8;
            const int ex_nat = block_exponent(__float_as_uint(fmaxf(fmaxf(sl[0], sl[2]), fmaxf(sl[4], sl[6]))));
            const int ey_nat = block_exponent(__float_as_uint(fmaxf(fmaxf(sl[1], sl[3]), fmaxf(sl[5], sl[7]))));
            if (ex_nat != E_UNSET && ey_nat != E_UNSET && ex_nat + ey_nat < s_run) {
                if (s_run != E_UNSET) {
#pragma unroll
                    for (int t = 0; t < TMAX; ++t)
#pragma unroll
                        for (int z = 0; z < 16; ++z) acc[t][z] = ldexpf(acc[t][z], ex_nat + ey_nat - s_run);
                }
                s_run = ex_nat + ey_nat;
            }
            ex_cur = ex_nat == E_UNSET ? 0 : ex_nat;
            ey_cur = ey_nat == E_UNSET ? 0 : (s_run == E_UNSET ? ey_nat : min(s_run - ex_cur, ey_nat));
        }
        store_unit();
        if (u + 1 < u_end) load_unit(u + 1);                             // in flight under this unit's MFMAs
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 2; ++s) {                                    // two 16-pixel steps
            bf16x8 af[NPL];
            const unsigned char *ya = yimg + (16 * s + kpix) * YS + cot * 64 + choff;
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) af[pl] = tr8(ya + pl * YPLANE, YS);
#pragma unroll
            for (int t = 0; t < TMAX; ++t) {
                const int tap = tap_lo + t;
                if (tap < tap_hi) {                                       // wave-uniform
                    const int ky = tap / 3, kx = tap - 3 * ky;
                    const unsigned char *xa = ximg + ((UP || GT) ? tap * TW + 16 * s + kpix : ky * (TW + 2) + 16 * s + kpix + kx) * XS + choff;
                    bf16x8 bfr[NPL];
#pragma unroll
                    for (int pl = 0; pl < NPL; ++pl) bfr[pl] = tr8(xa + pl * XPLANE, XS);
                    constexpr int ia[6] = {2, 0, 1, 1, 0, 0}, ib[6] = {0, 2, 1, 0, 1, 0};
                    if (H2) {
#pragma unroll
                        for (int e = 3; e < 6; ++e)
                            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const f16x8 *>(&af[ia[e]]),
                                                                            *reinterpret_cast<const f16x8 *>(&bfr[ib[e]]), acc[t], 0, 0, 0);
                    } else {
#pragma unroll
                        for (int e = (NPL == 3 ? 0 : 5); e < 6; ++e)
                            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ia[e]], bfr[ib[e]], acc[t], 0, 0, 0);
                    }
                }
            }
        }
        if (H2 && u + 1 < u_end) publish_unit_max((u + 1) & 1);            // (the next unit's loads have landed under the MFMAs)
        __syncthreads();                                                  // every wave has read the images
    }

    // bias gradient: the threads that share a channel group are summed in a fixed order
    if (do_colsum) {
        float4 *red = reinterpret_cast<float4 *>(lds);
        red[tid] = csum;
        __syncthreads();
        constexpr int CG = UP ? 8 : 8 * COT;
        if (tid < CG) {
            float4 tsum = red[tid];
            for (int j = tid + CG; j < NT; j += CG) { tsum.x += red[j].x; tsum.y += red[j].y; tsum.z += red[j].z; tsum.w += red[j].w; }
            if (UP) *reinterpret_cast<float4 *>(p.colslab + (size_t)split * p.Cin + ci0 + tid * 4) = tsum;
            else *reinterpret_cast<float4 *>(p.colslab + (size_t)split * p.Cout + co0 + tid * 4) = tsum;
        }
        __syncthreads();
    }

    // slab tile (co tile, tap): rows co, columns tap * Cin + ci0 .. + 31
    float *patch = reinterpret_cast<float *>(lds) + wave * (32 * 36);
    const int er = lane >> 3, ec = (lane & 7) * 4, r = lane & 31, h = lane >> 5;
    float *sbase = p.slab + (size_t)split * p.Cout * NTAPS * p.Cin;
#pragma unroll
    for (int t = 0; t < TMAX; ++t) {
        const int tap = tap_lo + t;
        if (tap < tap_hi) {
#pragma unroll
            for (int z = 0; z < 16; ++z)
                patch[((z & 3) + 8 * (z >> 2) + 4 * h) * 36 + r] = H2 ? ldexpf(acc[t][z], s_run == E_UNSET ? 0 : -s_run) : acc[t][z];
#pragma unroll
            for (int pass = 0; pass < 4; ++pass) {
                const float4 v = *reinterpret_cast<const float4 *>(patch + (pass * 8 + er) * 36 + ec);
                const int co = co0 + cot * 32 + pass * 8 + er;
                *reinterpret_cast<float4 *>(sbase + ((size_t)co * NTAPS + tap) * p.Cin + ci0 + ec) = v;
            }
        }
    }
}

bool al16(const void *q) { return (uintptr_t)q % 16 == 0; }
bool pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

// geometry both directions need: planes at least one 4 x 32 tile, channel counts in multiples of 32
int check_bx3(const char *what, int B, int H, int W, int Cin, int Cout) {
    if (B < 1 || !pow2(H) || !pow2(W) || H < 4 || W < TW || H > 32768 || W > 32768 || Cin < 32 || Cout < 32 || Cin % 32 || Cout % 32)
        return fail(T2H_ERR_ARG, "%s: needs power-of-two H >= 4, W >= 32 and Cin, Cout multiples of 32 (B=%d H=%d W=%d %d->%d)", what, B,
                    H, W, Cin, Cout);
    if ((long long)B * H * W > (1LL << 30)) return fail(T2H_ERR_ARG, "%s: more than 2^30 pixels", what);
    return T2H_OK;
}

struct RowsPlan { bool tall; int bn, splits, chunks_per_split; long long tiles; };
RowsPlan bx3_rows_plan(int B, int H, int W, int Kc, int Nc, int ntap = 9, int upm = 0) {
    RowsPlan r{};
    r.bn = Nc % 128 == 0 ? 128 : (Nc % 64 == 0 ? 64 : 32);
    if (upm) {
        // transposed convolution: 4-row tiles only; the forward's scattering epilogue cannot take split slabs, so it narrows its
        // column tiles until ~256 workgroups exist; the data gradient splits like any other few-tile product
        r.tall = false;
        const long long row_tiles = (long long)B * (H / 4) * (W / TW);
        if (upm == 1) while (r.bn > 32 && row_tiles * (Nc / r.bn) < 192) r.bn >>= 1;
        r.tiles = row_tiles * (Nc / r.bn);
        r.splits = 1;
        const int nchunk_u = Kc / 64;
        static const long long target_u = 512;
        if (upm == 2 && r.tiles < target_u / 2) {
            long long want = (target_u + r.tiles - 1) / r.tiles;
            if (want > nchunk_u) want = nchunk_u;
            if (want > 32) want = 32;
            r.splits = (int)(want < 1 ? 1 : want);
        }
        r.chunks_per_split = (nchunk_u + r.splits - 1) / r.splits;
        r.splits = (nchunk_u + r.chunks_per_split - 1) / r.chunks_per_split;
        return r;
    }
    // 8-row tiles where they still give every CU its two workgroups twice over (512 x 512 planes: 1024 tiles); measured on
    // 64->128 / 128->64 / 64->32 at 512^2: 203 -> 185, 239 -> 194, 76 -> 72 us, and slower at 256^2 (256 tiles: one per CU)
    static const long long min_tiles8 = getenv("T2H_BX3_TILES8") ? atoll(getenv("T2H_BX3_TILES8")) : 1024;
    const long long tiles8 = H % 8 == 0 ? (long long)B * (H / 8) * (W / TW) * (Nc / r.bn) : 0;
    r.tall = tiles8 >= min_tiles8 && !(ntap == 1 && r.bn == 128);        // (1-tap, 128 columns: the 8-row images would not fit twice per CU)
    r.tiles = r.tall ? tiles8 : (long long)B * (H / 4) * (W / TW) * (Nc / r.bn);
    // few tiles (small planes) and a short reduction (<= 128 channels: few chunks to split): 64-column tiles double the tile count
    // instead of splitting the reduction into slabs (64->128 at 128^2: 32 -> 24 us, 128->256 at 64^2: 31 -> 29; with longer
    // reductions the narrower tiles lose: 512->256 at 64^2 dgrad 64 -> 98 us).  T2H_BX3_NARROW=0: A/B
    static const bool narrow = !(getenv("T2H_BX3_NARROW") && getenv("T2H_BX3_NARROW")[0] == '0');
    if (narrow && !r.tall && r.bn == 128 && r.tiles < 256 && Kc <= 128) { r.bn = 64; r.tiles *= 2; }
    // small planes with many channels: fewer tiles than the chip holds workgroups -> split the reduction over whole 32-channel
    // chunks into slabs (summed in a fixed order by reduce_rows_epilogue, which also applies the epilogue): deterministic
    r.splits = 1;
    const int nchunk = ntap == 9 ? Kc / (r.tall ? 16 : 32) : Kc / (r.tall ? 32 : 64);      // (the 1-tap form stages 64 / 32 channels)
    static const long long target = getenv("T2H_BX3_ROWS_WGS") ? atoll(getenv("T2H_BX3_ROWS_WGS")) : 512;
    if (!r.tall && r.tiles < target / 2) {
        long long want = (target + r.tiles - 1) / r.tiles;
        if (want > nchunk) want = nchunk;
        if (want > 32) want = 32;
        r.splits = (int)(want < 1 ? 1 : want);
    }
    r.chunks_per_split = (nchunk + r.splits - 1) / r.splits;
    r.splits = (nchunk + r.chunks_per_split - 1) / r.chunks_per_split;
    return r;
}

#define BX3_LAUNCH(TH_, BN_, WM_, WN_, CC_, NTAP_)                                                                           \
    do {                                                                                                                    \
        if (npl == 1) {                                                                                                     \
            hipLaunchKernelGGL((bx3_rows_kernel<TH_, BN_, WM_, WN_, CC_, 1, NTAP_>), dim3((unsigned)grid), dim3(NT), 0, s, a); \
            note_kernel("bx3_rows_kernel<" #TH_ "," #BN_ "," #WM_ "," #WN_ "," #CC_ ",1," #NTAP_ ",0,false>");                      \
        } else if (npl == 2) {                                                                                              \
            hipLaunchKernelGGL((bx3_rows_kernel<TH_, BN_, WM_, WN_, CC_, 2, NTAP_>), dim3((unsigned)grid), dim3(NT), 0, s, a); \
            note_kernel("bx3_rows_kernel<" #TH_ "," #BN_ "," #WM_ "," #WN_ "," #CC_ ",2," #NTAP_ ",0,false>");                      \
        } else {                                                                                                            \
            hipLaunchKernelGGL((bx3_rows_kernel<TH_, BN_, WM_, WN_, CC_, 3, NTAP_>), dim3((unsigned)grid), dim3(NT), 0, s, a); \
            note_kernel("bx3_rows_kernel<" #TH_ "," #BN_ "," #WM_ "," #WN_ "," #CC_ ",3," #NTAP_ ",0,false>");                      \
        }                                                                                                                   \
    } while (0)

// `a`: x, wf, bias, mask, y, geometry and epilogue flags; splits the reduction into `ws` when the plan says so
#define BX3_LAUNCH_UP(BN_, WM_, WN_, UPM_)                                                                                    \
    do {                                                                                                                    \
        if (npl == 2) {                                                                                                     \
            hipLaunchKernelGGL((bx3_rows_kernel<4, BN_, WM_, WN_, 64, 2, 1, UPM_>), dim3((unsigned)grid), dim3(NT), 0, s, a); \
            note_kernel("bx3_rows_kernel<4," #BN_ "," #WM_ "," #WN_ ",64,2,1," #UPM_ ",false>");                                  \
        } else {                                                                                                            \
            hipLaunchKernelGGL((bx3_rows_kernel<4, BN_, WM_, WN_, 64, 3, 1, UPM_>), dim3((unsigned)grid), dim3(NT), 0, s, a); \
            note_kernel("bx3_rows_kernel<4," #BN_ "," #WM_ "," #WN_ ",64,3,1," #UPM_ ",false>");                                  \
        }                                                                                                                   \
    } while (0)

// npl: 3 = bf16 three-way split, 2 = fp16 two-way split (a.wf prepared by the *_f16x2_prepare entry points), 1 = bf16 rounding
int npl_of(int flags) { return (flags & T2H_F16X2) ? 2 : ((flags & T2H_BF16) ? 1 : 3); }
// trailer of an fp16-prepared weight buffer ([0] largest magnitude bits, [1] 2^-e_w, [2] e_w) behind `planes_bytes` of planes
const float *f16_trailer(const void *wf, size_t planes_bytes) { return reinterpret_cast<const float *>(static_cast<const unsigned char *>(wf) + planes_bytes); }

bool persist_n_enabled() {
    static const bool on = !(getenv("T2H_BX3_PERSIST_N") && getenv("T2H_BX3_PERSIST_N")[0] == '0');
    return on;
}

int launch_rows(RowsArgs a, int npl, void *ws, size_t ws_bytes, hipStream_t s, const char *what, int ntap = 9, int upm = 0) {
    const RowsPlan r = bx3_rows_plan(a.B, a.H, a.W, a.Kc, a.Nc, ntap, upm);
    const long long grid = r.tiles * r.splits;
    if (grid > 0x7fffffffLL) return fail(T2H_ERR_ARG, "%s: too many tiles", what);
    const long long M = (long long)a.B * a.H * a.W;
    EpilogueArgs e{};
    if (r.splits > 1) {
        const size_t need = (size_t)r.splits * M * a.Nc * sizeof(float);
        if (!ws || ws_bytes < need || !al16(ws)) return fail(T2H_ERR_WORKSPACE, "%s: workspace %zu < %zu bytes", what, ws_bytes, need);
        e.C = a.y; e.bias = a.bias; e.mask = a.mask; e.M = (int)M; e.N = a.Nc; e.ldc = a.ldy; e.ldm = a.ldm;
        e.accum = a.flags & F_ACCUM; e.relu_out = a.flags & F_RELU_OUT;
        a.y = static_cast<float *>(ws); a.bias = nullptr; a.mask = nullptr; a.flags = 0; a.ldy = a.Nc;
    }
    a.splits = r.splits; a.chunks_per_split = r.chunks_per_split; a.ntn_per_wg = 1;
    if (upm == 1) {
        if (r.bn == 128) BX3_LAUNCH_UP(128, 2, 2, 1);
        else if (r.bn == 64) BX3_LAUNCH_UP(64, 4, 1, 1);
        else BX3_LAUNCH_UP(32, 4, 1, 1);
    } else if (upm == 2) {
        if (r.bn == 128) BX3_LAUNCH_UP(128, 2, 2, 2);
        else if (r.bn == 64) BX3_LAUNCH_UP(64, 4, 1, 2);
        else BX3_LAUNCH_UP(32, 4, 1, 2);
    } else if (ntap == 9) {
        if (r.tall) {
            if (r.bn == 128) BX3_LAUNCH(8, 128, 2, 2, 16, 9);
            else if (r.bn == 64) BX3_LAUNCH(8, 64, 4, 1, 16, 9);
            else BX3_LAUNCH(8, 32, 4, 1, 16, 9);
        } else {
            if (r.bn == 128) BX3_LAUNCH(4, 128, 2, 2, 32, 9);
            else if (r.bn == 64) BX3_LAUNCH(4, 64, 4, 1, 32, 9);
            else BX3_LAUNCH(4, 32, 4, 1, 32, 9);
        }
    } else if (ntap == 1 && upm == 0 && npl == 2 && a.Kc == 64 && r.splits == 1 && a.Nc / r.bn >= 8 && persist_n_enabled()) {
        // one staged chunk, many column tiles: the rows are staged and split once per workgroup and reused (PERSIST)
        const long long row_tiles = (long long)a.B * (a.H / 4) * (a.W / TW);
        const int pbn = r.bn == 128 ? 64 : r.bn;                          // (128-column tiles spill in this form)
        const int ntn = a.Nc / pbn;
        static const long long persist_wgs = getenv("T2H_BX3_PERSIST_WGS") ? atoll(getenv("T2H_BX3_PERSIST_WGS")) : 2048;
        // r06: at least four column groups however many row tiles there are -- with four tiles per launch (2 048 row tiles) a
        // single group made every workgroup walk all 43 column tiles: 289 us per tile against 181 with four groups and the 185 of
        // one tile per launch (profiles/level256_probe.py; T2H_BX3_PERSIST_MIN_GROUPS)
        static const long long min_groups = getenv("T2H_BX3_PERSIST_MIN_GROUPS") ? atoll(getenv("T2H_BX3_PERSIST_MIN_GROUPS")) : 4;
        long long groups = persist_wgs / (row_tiles > 0 ? row_tiles : 1);
        if (groups < min_groups) groups = min_groups;
        if (groups < 1) groups = 1;
        if (groups > ntn) groups = ntn;
        a.ntn_per_wg = (int)((ntn + groups - 1) / groups);
        groups = (ntn + a.ntn_per_wg - 1) / a.ntn_per_wg;
        const long long pgrid = row_tiles * groups;
        if (pbn == 64) { hipLaunchKernelGGL((bx3_rows_kernel<4, 64, 4, 1, 64, 2, 1, 0, true>), dim3((unsigned)pgrid), dim3(NT), 0, s, a); note_kernel("bx3_rows_kernel<4,64,4,1,64,2,1,0,true>"); }
        else { hipLaunchKernelGGL((bx3_rows_kernel<4, 32, 4, 1, 64, 2, 1, 0, true>), dim3((unsigned)pgrid), dim3(NT), 0, s, a); note_kernel("bx3_rows_kernel<4,32,4,1,64,2,1,0,true>"); }
    } else {
        if (r.tall) {
            if (r.bn == 128) BX3_LAUNCH(8, 128, 2, 2, 32, 1);
            else if (r.bn == 64) BX3_LAUNCH(8, 64, 4, 1, 32, 1);
            else BX3_LAUNCH(8, 32, 4, 1, 32, 1);
        } else {
            if (r.bn == 128) BX3_LAUNCH(4, 128, 2, 2, 64, 1);
            else if (r.bn == 64) BX3_LAUNCH(4, 64, 4, 1, 64, 1);
            else BX3_LAUNCH(4, 32, 4, 1, 64, 1);
        }
    }
    if (int rc = check_launch(what)) return rc;
    if (r.splits > 1) return launch_reduce_rows_epilogue(static_cast<const float *>(ws), r.splits, M * a.Nc, M, a.Nc, e, s);
    return T2H_OK;
}

struct WgradPlan { int cot, splits, units_per_split, n_units; };
WgradPlan bx3_wgrad_plan(int B, int H, int W, int Cin, int Cout) {
    WgradPlan p{};
    p.cot = Cout % 128 == 0 ? 4 : (Cout % 64 == 0 ? 2 : 1);
    p.n_units = B * H * (W / TW);
    const long long groups = (long long)(Cin / CC) * (Cout / (32 * p.cot));
    // ~512 workgroups; planes up to 128 x 128 (few units per workgroup either way): 256 -- half the slabs to write and reduce
    static const long long forced = getenv("T2H_BX3_WGRAD_WGS") ? atoll(getenv("T2H_BX3_WGRAD_WGS")) : 0;
    const long long target = forced > 0 ? forced : ((long long)H * W <= 128 * 128 ? 256 : 512);
    long long splits = target / groups;
    if (splits < 1) splits = 1;
    if (splits > p.n_units) splits = p.n_units;
    p.units_per_split = (int)((p.n_units + splits - 1) / splits);
    p.splits = (p.n_units + p.units_per_split - 1) / p.units_per_split;
    return p;
}

}  // namespace
}  // namespace t2h

using namespace t2h;

T2H_API int t2h_conv3x3_bx3_supported(int B, int H, int W, int Cin, int Cout) {
    return B >= 1 && pow2(H) && pow2(W) && H >= 4 && W >= TW && H <= 32768 && W <= 32768 && Cin >= 32 && Cout >= 32 && Cin % 32 == 0 &&
           Cout % 32 == 0 && (long long)B * H * W <= (1LL << 30);
}

T2H_API size_t t2h_conv3x3_bx3_weights_bytes(int Cin, int Cout) {
    if (Cin < 32 || Cout < 32 || Cin % 32 || Cout % 32) return 0;
    return (size_t)Cout * 9 * Cin * 6;                                   // three bf16 planes
}

T2H_API int t2h_conv3x3_bx3_prepare(const float *w, int Cin, int Cout, int transposed, void *wf, t2h_stream_t stream) {
    if (!w || !wf) return fail(T2H_ERR_ARG, "conv3x3_bx3_prepare: null pointer");
    if (Cin < 32 || Cout < 32 || Cin % 32 || Cout % 32 || !al16(wf))
        return fail(T2H_ERR_ARG, "conv3x3_bx3_prepare: Cin=%d, Cout=%d must be multiples of 32, wf 16-byte aligned", Cin, Cout);
    const int Kc = transposed ? Cout : Cin, Nc = transposed ? Cin : Cout;
    const long long total = (long long)(Kc / 16) * 9 * (Nc / 32) * 64;
    const unsigned blocks = (unsigned)((total + 255) / 256);
    if (transposed) hipLaunchKernelGGL(bx3_prepare_kernel<true>, dim3(blocks), dim3(256), 0, as_stream(stream), w, Cin, Cout, static_cast<unsigned *>(wf));
    else hipLaunchKernelGGL(bx3_prepare_kernel<false>, dim3(blocks), dim3(256), 0, as_stream(stream), w, Cin, Cout, static_cast<unsigned *>(wf));
    return check_launch("conv3x3_bx3_prepare");
}

T2H_API size_t t2h_conv3x3_bx3_fwd_workspace_bytes(int B, int H, int W, int Cin, int Cout) {
    if (!t2h_conv3x3_bx3_supported(B, H, W, Cin, Cout)) return 0;
    const RowsPlan r = bx3_rows_plan(B, H, W, Cin, Cout);
    return r.splits > 1 ? (size_t)r.splits * B * H * W * Cout * sizeof(float) : 0;
}

T2H_API size_t t2h_conv3x3_bx3_dgrad_workspace_bytes(int B, int H, int W, int Cin, int Cout) {
    if (!t2h_conv3x3_bx3_supported(B, H, W, Cin, Cout)) return 0;
    const RowsPlan r = bx3_rows_plan(B, H, W, Cout, Cin);
    return r.splits > 1 ? (size_t)r.splits * B * H * W * Cin * sizeof(float) : 0;
}

T2H_API int t2h_conv3x3_bx3_fwd(const float *x, const void *wf, const float *bias, float *y, int B, int H, int W, int Cin, int Cout,
                                int flags, void *workspace, size_t workspace_bytes, t2h_stream_t stream) {
    if (!x || !wf || !y) return fail(T2H_ERR_ARG, "conv3x3_bx3_fwd: null pointer");
    if (int rc = check_bx3("conv3x3_bx3_fwd", B, H, W, Cin, Cout)) return rc;
    if (!al16(x) || !al16(wf) || !al16(y) || (bias && !al16(bias))) return fail(T2H_ERR_ARG, "conv3x3_bx3_fwd: pointers must be 16-byte aligned");
    RowsArgs a{};
    a.x = x; a.wf = static_cast<const unsigned *>(wf); a.bias = bias; a.mask = nullptr; a.y = y;
    a.B = B; a.H = H; a.W = W; a.Kc = Cin; a.Nc = Cout; a.ldx = Cin; a.ldy = Cout; a.ldm = Cout;
    a.flags = ((flags & T2H_RELU_OUT) ? F_RELU_OUT : 0) | ((flags & T2H_ACCUM) ? F_ACCUM : 0);
    a.wscale = f16_trailer(wf, (size_t)Cout * 9 * Cin * 4);
    return launch_rows(a, npl_of(flags), workspace, workspace_bytes, as_stream(stream), "conv3x3_bx3_fwd");
}

T2H_API int t2h_conv3x3_bx3_dgrad(const float *dy, const void *wf_t, float *dx, const float *mask, int B, int H, int W, int Cin,
                                  int Cout, int flags, void *workspace, size_t workspace_bytes, t2h_stream_t stream) {
    if (!dy || !wf_t || !dx) return fail(T2H_ERR_ARG, "conv3x3_bx3_dgrad: null pointer");
    if (int rc = check_bx3("conv3x3_bx3_dgrad", B, H, W, Cin, Cout)) return rc;
    if (!al16(dy) || !al16(wf_t) || !al16(dx) || (mask && !al16(mask))) return fail(T2H_ERR_ARG, "conv3x3_bx3_dgrad: pointers must be 16-byte aligned");
    RowsArgs a{};
    a.x = dy; a.wf = static_cast<const unsigned *>(wf_t); a.bias = nullptr; a.mask = mask; a.y = dx;
    a.B = B; a.H = H; a.W = W; a.Kc = Cout; a.Nc = Cin; a.ldx = Cout; a.ldy = Cin; a.ldm = Cin;
    a.flags = (flags & T2H_ACCUM) ? F_ACCUM : 0;
    a.wscale = f16_trailer(wf_t, (size_t)Cout * 9 * Cin * 4);
    return launch_rows(a, npl_of(flags), workspace, workspace_bytes, as_stream(stream), "conv3x3_bx3_dgrad");
}

// the data gradient + the rank-1 term g[pixel] * w1[ci] under the same mask, written (not accumulated): dx = mask(dgrad) + mask(g w1).
// Only where the reduction is not split (the epilogue runs in the kernel): t2h_conv3x3_bx3_dgrad_rank1_supported
T2H_API int t2h_conv3x3_bx3_dgrad_rank1_supported(int B, int H, int W, int Cin, int Cout) {
    if (!t2h_conv3x3_bx3_supported(B, H, W, Cin, Cout)) return 0;
    return bx3_rows_plan(B, H, W, Cout, Cin).splits == 1;
}

T2H_API int t2h_conv3x3_bx3_dgrad_rank1(const float *dy, const void *wf_t, float *dx, const float *mask, const float *g, const float *w1,
                                        int B, int H, int W, int Cin, int Cout, int flags, t2h_stream_t stream) {
    if (!dy || !wf_t || !dx || !g || !w1) return fail(T2H_ERR_ARG, "conv3x3_bx3_dgrad_rank1: null pointer");
    if (int rc = check_bx3("conv3x3_bx3_dgrad_rank1", B, H, W, Cin, Cout)) return rc;
    if (!t2h_conv3x3_bx3_dgrad_rank1_supported(B, H, W, Cin, Cout))
        return fail(T2H_ERR_ARG, "conv3x3_bx3_dgrad_rank1: this shape splits its reduction (use t2h_conv3x3_bx3_dgrad with T2H_ACCUM)");
    if (flags & T2H_ACCUM) return fail(T2H_ERR_ARG, "conv3x3_bx3_dgrad_rank1: the rank-1 term takes the place of the accumulated values");
    if (!al16(dy) || !al16(wf_t) || !al16(dx) || (mask && !al16(mask)) || !al16(w1))
        return fail(T2H_ERR_ARG, "conv3x3_bx3_dgrad_rank1: pointers must be 16-byte aligned");
    RowsArgs a{};
    a.x = dy; a.wf = static_cast<const unsigned *>(wf_t); a.bias = nullptr; a.mask = mask; a.y = dx;
    a.B = B; a.H = H; a.W = W; a.Kc = Cout; a.Nc = Cin; a.ldx = Cout; a.ldy = Cin; a.ldm = Cin;
    a.flags = 0; a.r1_g = g; a.r1_w = w1;
    a.wscale = f16_trailer(wf_t, (size_t)Cout * 9 * Cin * 4);
    return launch_rows(a, npl_of(flags), nullptr, 0, as_stream(stream), "conv3x3_bx3_dgrad_rank1");
}

T2H_API size_t t2h_conv3x3_bx3_wgrad_workspace_bytes(int B, int H, int W, int Cin, int Cout) {
    if (!t2h_conv3x3_bx3_supported(B, H, W, Cin, Cout)) return 0;
    WgradPlan p = bx3_wgrad_plan(B, H, W, Cin, Cout);
    return (size_t)p.splits * ((size_t)Cout * 9 * Cin + Cout) * sizeof(float);
}

T2H_API int t2h_conv3x3_bx3_wgrad(const float *dy, const float *x, float *dw, float *db, int B, int H, int W, int Cin, int Cout,
                                  int flags, void *workspace, size_t workspace_bytes, t2h_stream_t stream) {
    if (!dy || !x || !dw) return fail(T2H_ERR_ARG, "conv3x3_bx3_wgrad: null pointer");
    if (int rc = check_bx3("conv3x3_bx3_wgrad", B, H, W, Cin, Cout)) return rc;
    if (!al16(dy) || !al16(x)) return fail(T2H_ERR_ARG, "conv3x3_bx3_wgrad: pointers must be 16-byte aligned");
    const size_t need = t2h_conv3x3_bx3_wgrad_workspace_bytes(B, H, W, Cin, Cout);
    if (!workspace || workspace_bytes < need || !al16(workspace))
        return fail(T2H_ERR_WORKSPACE, "conv3x3_bx3_wgrad: workspace %zu < %zu bytes", workspace_bytes, need);
    hipStream_t s = as_stream(stream);
    WgradPlan p = bx3_wgrad_plan(B, H, W, Cin, Cout);
    const int Ncols = 9 * Cin;
    float *slab = static_cast<float *>(workspace);
    float *colslab = slab + (size_t)p.splits * Cout * Ncols;
    WgradArgs a{};
    a.dy = dy; a.x = x; a.slab = slab; a.colslab = db ? colslab : nullptr;
    a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.n_units = p.n_units; a.units_per_split = p.units_per_split; a.ldx = Cin;
    dim3 grid(p.splits, Cin / CC, Cout / (32 * p.cot));
    if (grid.y > 65535 || grid.z > 65535) return fail(T2H_ERR_ARG, "conv3x3_bx3_wgrad: too many channel chunks");
    if (flags & T2H_F16X2) {
        if (p.cot == 4) { hipLaunchKernelGGL((bx3_wgrad_kernel<4, 2>), grid, dim3(NT), 0, s, a); note_kernel("bx3_wgrad_kernel<4,2,false,0>"); }
        else if (p.cot == 2) { hipLaunchKernelGGL((bx3_wgrad_kernel<2, 2>), grid, dim3(NT), 0, s, a); note_kernel("bx3_wgrad_kernel<2,2,false,0>"); }
        else { hipLaunchKernelGGL((bx3_wgrad_kernel<1, 2>), grid, dim3(NT), 0, s, a); note_kernel("bx3_wgrad_kernel<1,2,false,0>"); }
    } else if (flags & T2H_BF16) {
        if (p.cot == 4) { hipLaunchKernelGGL((bx3_wgrad_kernel<4, 1>), grid, dim3(NT), 0, s, a); note_kernel("bx3_wgrad_kernel<4,1,false,0>"); }
        else if (p.cot == 2) { hipLaunchKernelGGL((bx3_wgrad_kernel<2, 1>), grid, dim3(NT), 0, s, a); note_kernel("bx3_wgrad_kernel<2,1,false,0>"); }
        else { hipLaunchKernelGGL((bx3_wgrad_kernel<1, 1>), grid, dim3(NT), 0, s, a); note_kernel("bx3_wgrad_kernel<1,1,false,0>"); }
    } else {
        if (p.cot == 4) { hipLaunchKernelGGL((bx3_wgrad_kernel<4, 3>), grid, dim3(NT), 0, s, a); note_kernel("bx3_wgrad_kernel<4,3,false,0>"); }
        else if (p.cot == 2) { hipLaunchKernelGGL((bx3_wgrad_kernel<2, 3>), grid, dim3(NT), 0, s, a); note_kernel("bx3_wgrad_kernel<2,3,false,0>"); }
        else { hipLaunchKernelGGL((bx3_wgrad_kernel<1, 3>), grid, dim3(NT), 0, s, a); note_kernel("bx3_wgrad_kernel<1,3,false,0>"); }
    }
    if (int rc = check_launch("conv3x3_bx3_wgrad")) return rc;
    return launch_reduce_slabs(slab, p.splits, (long long)Cout * Ncols, Cout, Ncols, Ncols, (flags & T2H_ACCUM) ? 1 : 0, dw, colslab, db, s,
                               0, 0, (flags & T2H_DEFER_REDUCE) != 0);
}

// ---- the 1-tap form as a GEMM on rows ---------------------------------------------------------------------------------------------
T2H_API int t2h_gemm_bx3_supported(int64_t M, int K, int N) {
    return M >= 128 && M % 128 == 0 && M <= (1LL << 30) && K >= 64 && K % 64 == 0 && N >= 32 && N % 32 == 0;
}

T2H_API size_t t2h_gemm_bx3_weights_bytes(int K, int N) {
    if (K < 16 || N < 32 || K % 16 || N % 32) return 0;
    return (size_t)K * N * 6;
}

T2H_API int t2h_gemm_bx3_prepare(const float *w, int ldw, int K, int N, int w_is_kn, void *wf, t2h_stream_t stream) {
    if (!w || !wf) return fail(T2H_ERR_ARG, "gemm_bx3_prepare: null pointer");
    if (K < 16 || N < 32 || K % 16 || N % 32 || !al16(wf) || ldw < (w_is_kn ? N : K))
        return fail(T2H_ERR_ARG, "gemm_bx3_prepare: K=%d must be a multiple of 16, N=%d of 32, ldw=%d at least the row length", K, N, ldw);
    const long long total = (long long)(K / 16) * (N / 32) * 64;
    const unsigned blocks = (unsigned)((total + 255) / 256);
    if (w_is_kn) hipLaunchKernelGGL(bx3_prepare_gemm_kernel<true>, dim3(blocks), dim3(256), 0, as_stream(stream), w, K, N, ldw, static_cast<unsigned *>(wf));
    else hipLaunchKernelGGL(bx3_prepare_gemm_kernel<false>, dim3(blocks), dim3(256), 0, as_stream(stream), w, K, N, ldw, static_cast<unsigned *>(wf));
    return check_launch("gemm_bx3_prepare");
}

T2H_API size_t t2h_gemm_bx3_workspace_bytes(int64_t M, int K, int N) {
    if (!t2h_gemm_bx3_supported(M, K, N)) return 0;
    const RowsPlan r = bx3_rows_plan(1, (int)(M / 32), 32, K, N, 1);
    return r.splits > 1 ? (size_t)r.splits * M * N * sizeof(float) : 0;
}

T2H_API int t2h_gemm_bx3(const float *x, int ldx, const void *wf, const float *bias, const float *mask, int ldm, float *y, int ldy,
                         int64_t M, int K, int N, int flags, void *workspace, size_t workspace_bytes, t2h_stream_t stream) {
    if (!x || !wf || !y) return fail(T2H_ERR_ARG, "gemm_bx3: null pointer");
    if (!t2h_gemm_bx3_supported(M, K, N))
        return fail(T2H_ERR_ARG, "gemm_bx3: needs M %% 128 == 0, K %% 64 == 0, N %% 32 == 0 (M=%lld K=%d N=%d)", (long long)M, K, N);
    if (ldx < K || ldy < N || ldx % 4 || ldy % 4 || (mask && (ldm < N || ldm % 4)) || !al16(x) || !al16(wf) || !al16(y) || (bias && !al16(bias)) ||
        (mask && !al16(mask)))
        return fail(T2H_ERR_ARG, "gemm_bx3: rows must be 16-byte aligned and at least K / N floats long");
    RowsArgs a{};
    a.x = x; a.wf = static_cast<const unsigned *>(wf); a.bias = bias; a.mask = mask; a.y = y;
    a.B = 1; a.H = (int)(M / 32); a.W = 32; a.Kc = K; a.Nc = N; a.ldx = ldx; a.ldy = ldy; a.ldm = ldm;
    a.flags = ((flags & T2H_RELU_OUT) ? F_RELU_OUT : 0) | ((flags & T2H_ACCUM) ? F_ACCUM : 0);
    a.wscale = f16_trailer(wf, (size_t)K * N * 4);
    return launch_rows(a, npl_of(flags), workspace, workspace_bytes, as_stream(stream), "gemm_bx3", 1);
}

// ---- the weight gradient of the same GEMM: dW [N][K] = dY^T X, reduction over the M rows (bx3_wgrad_kernel<.., GT = K / 32>) ----------
static int gemm_wgrad_splits(int64_t M, int N) {
    static const long long target = getenv("T2H_GEMM_BX3_WGRAD_WGS") ? atoll(getenv("T2H_GEMM_BX3_WGRAD_WGS")) : 4096;
    const long long groups = N / 64, units = M / TW;
    long long splits = (target + groups - 1) / groups;
    if (splits > units) splits = units;
    if (splits < 1) splits = 1;
    const long long per = (units + splits - 1) / splits;
    return (int)((units + per - 1) / per);
}

T2H_API int t2h_gemm_bx3_wgrad_supported(int64_t M, int K, int N) {
    return M >= TW && M % TW == 0 && M <= (1LL << 30) && (K == 64 || K == 128 || K == 256) && N >= 64 && N % 64 == 0;
}

T2H_API size_t t2h_gemm_bx3_wgrad_workspace_bytes(int64_t M, int K, int N) {
    if (!t2h_gemm_bx3_wgrad_supported(M, K, N)) return 0;
    return (size_t)gemm_wgrad_splits(M, N) * ((size_t)N * K + N) * sizeof(float);
}

T2H_API int t2h_gemm_bx3_wgrad(const float *dy, int lddy, const float *x, int ldx, int64_t M, int K, int N, float *dw, float *db, int flags,
                               void *workspace, size_t workspace_bytes, t2h_stream_t stream) {
    if (!dy || !x || !dw) return fail(T2H_ERR_ARG, "gemm_bx3_wgrad: null pointer");
    if (!t2h_gemm_bx3_wgrad_supported(M, K, N))
        return fail(T2H_ERR_ARG, "gemm_bx3_wgrad: needs M %% 32 == 0, K in {64, 128, 256}, N %% 64 == 0 (M=%lld K=%d N=%d)", (long long)M, K, N);
    if (lddy < N || ldx < K || lddy % 4 || ldx % 4 || !al16(dy) || !al16(x))
        return fail(T2H_ERR_ARG, "gemm_bx3_wgrad: rows must be 16-byte aligned and at least N / K floats long");
    if (!(flags & T2H_F16X2)) return fail(T2H_ERR_ARG, "gemm_bx3_wgrad: only the fp16 two-way split (T2H_F16X2) is built");
    const size_t need = t2h_gemm_bx3_wgrad_workspace_bytes(M, K, N);
    if (!workspace || workspace_bytes < need || !al16(workspace))
        return fail(T2H_ERR_WORKSPACE, "gemm_bx3_wgrad: workspace %zu < %zu bytes", workspace_bytes, need);
    hipStream_t s = as_stream(stream);
    const int splits = gemm_wgrad_splits(M, N);
    float *slab = static_cast<float *>(workspace);
    float *colslab = slab + (size_t)splits * N * K;
    WgradArgs a{};
    a.dy = dy; a.x = x; a.slab = slab; a.colslab = db ? colslab : nullptr;
    a.H = (int)(M / TW); a.W = TW; a.Cin = 32; a.Cout = N; a.n_units = (int)(M / TW);
    a.units_per_split = (a.n_units + splits - 1) / splits; a.ldx = ldx; a.lddy = lddy;
    const dim3 grid(N / 64, 1, splits);
    if (grid.z > 65535) return fail(T2H_ERR_ARG, "gemm_bx3_wgrad: too many splits");
    if (K == 64) { hipLaunchKernelGGL((bx3_wgrad_kernel<2, 2, false, 2>), grid, dim3(NT), 0, s, a); note_kernel("bx3_wgrad_kernel<2,2,false,2>"); }
    else if (K == 128) { hipLaunchKernelGGL((bx3_wgrad_kernel<2, 2, false, 4>), grid, dim3(NT), 0, s, a); note_kernel("bx3_wgrad_kernel<2,2,false,4>"); }
    else { hipLaunchKernelGGL((bx3_wgrad_kernel<2, 2, false, 8>), grid, dim3(NT), 0, s, a); note_kernel("bx3_wgrad_kernel<2,2,false,8>"); }
    if (int rc = check_launch("gemm_bx3_wgrad")) return rc;
    return launch_reduce_slabs(slab, splits, (long long)N * K, N, K, K, (flags & T2H_ACCUM) ? 1 : 0, dw, colslab, db, s, 0, 0,
                               (flags & T2H_DEFER_REDUCE) != 0);
}

// ---- ConvTranspose2d(kernel_size = 2, stride = 2) on the 1-tap form -------------------------------------------------------------------
static int check_up_bx3(const char *what, int B, int H, int W, int Cin, int Cout) {
    if (B < 1 || !pow2(H) || !pow2(W) || H > 16384 || W > 16384 || (long long)B * H * W % 128 != 0 || (long long)B * H * W > (1LL << 28) ||
        Cin < 64 || Cout < 64 || Cin % 64 || Cout % 64)
        return fail(T2H_ERR_ARG, "%s: needs power-of-two H, W with B H W %% 128 == 0 and Cin, Cout multiples of 64 (B=%d H=%d W=%d %d->%d)",
                    what, B, H, W, Cin, Cout);
    return T2H_OK;
}
static int ilog2i(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }

T2H_API int t2h_upconv2x2_bx3_supported(int B, int H, int W, int Cin, int Cout) {
    return B >= 1 && pow2(H) && pow2(W) && H <= 16384 && W <= 16384 && (long long)B * H * W % 128 == 0 && (long long)B * H * W <= (1LL << 28) &&
           Cin >= 64 && Cout >= 64 && Cin % 64 == 0 && Cout % 64 == 0;
}

T2H_API int t2h_upconv2x2_bx3_fwd(const float *x, const void *wf, const float *bias, const float *addend, float *y, int B, int H, int W,
                                  int Cin, int Cout, int flags, t2h_stream_t stream) {
    if (!x || !wf || !y) return fail(T2H_ERR_ARG, "upconv2x2_bx3_fwd: null pointer");
    if (int rc = check_up_bx3("upconv2x2_bx3_fwd", B, H, W, Cin, Cout)) return rc;
    if (!al16(x) || !al16(wf) || !al16(y) || (bias && !al16(bias)) || (addend && !al16(addend)))
        return fail(T2H_ERR_ARG, "upconv2x2_bx3_fwd: pointers must be 16-byte aligned");
    const long long M = (long long)B * H * W;
    RowsArgs a{};
    a.x = x; a.wf = static_cast<const unsigned *>(wf); a.bias = bias; a.addend = addend; a.y = y;
    a.B = 1; a.H = (int)(M / 32); a.W = 32; a.Kc = Cin; a.Nc = 4 * Cout; a.ldx = Cin; a.ldy = 4 * Cout; a.ldm = 0;
    a.up_logW = ilog2i(W); a.up_logH = ilog2i(H); a.up_cout = Cout;
    a.flags = (flags & T2H_ACCUM) ? F_ACCUM : 0;
    a.wscale = f16_trailer(wf, (size_t)Cin * 4 * Cout * 4);
    return launch_rows(a, (flags & T2H_F16X2) ? 2 : 3, nullptr, 0, as_stream(stream), "upconv2x2_bx3_fwd", 1, 1);
}

T2H_API size_t t2h_upconv2x2_bx3_dgrad_workspace_bytes(int B, int H, int W, int Cin, int Cout) {
    if (!t2h_upconv2x2_bx3_supported(B, H, W, Cin, Cout)) return 0;
    const long long M = (long long)B * H * W;
    const RowsPlan r = bx3_rows_plan(1, (int)(M / 32), 32, 4 * Cout, Cin, 1, 2);
    return r.splits > 1 ? (size_t)r.splits * M * Cin * sizeof(float) : 0;
}

T2H_API int t2h_upconv2x2_bx3_dgrad(const float *dy, int lddy, const void *wf_t, float *dx, int B, int H, int W, int Cin, int Cout,
                                    int flags, void *workspace, size_t workspace_bytes, t2h_stream_t stream) {
    if (!dy || !wf_t || !dx) return fail(T2H_ERR_ARG, "upconv2x2_bx3_dgrad: null pointer");
    if (int rc = check_up_bx3("upconv2x2_bx3_dgrad", B, H, W, Cin, Cout)) return rc;
    if (lddy < Cout || lddy % 4) return fail(T2H_ERR_ARG, "upconv2x2_bx3_dgrad: lddy=%d must be >= Cout=%d and a multiple of 4", lddy, Cout);
    if (!al16(dy) || !al16(wf_t) || !al16(dx)) return fail(T2H_ERR_ARG, "upconv2x2_bx3_dgrad: pointers must be 16-byte aligned");
    const long long M = (long long)B * H * W;
    RowsArgs a{};
    a.x = dy; a.wf = static_cast<const unsigned *>(wf_t); a.y = dx;
    a.B = 1; a.H = (int)(M / 32); a.W = 32; a.Kc = 4 * Cout; a.Nc = Cin; a.ldx = lddy; a.ldy = Cin; a.ldm = 0;
    a.up_logW = ilog2i(W); a.up_logH = ilog2i(H); a.up_cout = Cout;
    a.flags = (flags & T2H_ACCUM) ? F_ACCUM : 0;
    a.wscale = f16_trailer(wf_t, (size_t)Cin * 4 * Cout * 4);
    return launch_rows(a, (flags & T2H_F16X2) ? 2 : 3, workspace, workspace_bytes, as_stream(stream), "upconv2x2_bx3_dgrad", 1, 2);
}

static WgradPlan up_wgrad_plan(int B, int H, int W, int Cin, int Cout) {
    // roles swapped: the 32 COT-wide side is the layer's Cin, the 32-channel chunks (CC) its Cout; ~256 workgroups
    WgradPlan p{};
    p.cot = Cin % 128 == 0 ? 4 : 2;
    p.n_units = (int)((long long)B * H * W / TW);
    const long long groups = (long long)(Cout / CC) * (Cin / (32 * p.cot));
    static const long long forced = getenv("T2H_BX3_UPWGRAD_WGS") ? atoll(getenv("T2H_BX3_UPWGRAD_WGS")) : 0;
    long long splits = (forced > 0 ? forced : 256) / groups;
    if (splits < 1) splits = 1;
    if (splits > p.n_units) splits = p.n_units;
    p.units_per_split = (int)((p.n_units + splits - 1) / splits);
    p.splits = (p.n_units + p.units_per_split - 1) / p.units_per_split;
    return p;
}

T2H_API size_t t2h_upconv2x2_bx3_wgrad_workspace_bytes(int B, int H, int W, int Cin, int Cout) {
    if (!t2h_upconv2x2_bx3_supported(B, H, W, Cin, Cout) || W < TW) return 0;
    const WgradPlan p = up_wgrad_plan(B, H, W, Cin, Cout);
    return (size_t)p.splits * ((size_t)Cin * 4 * Cout + Cout) * sizeof(float);
}

T2H_API int t2h_upconv2x2_bx3_wgrad(const float *dy, int lddy, const float *x, float *dw, float *db, int B, int H, int W, int Cin,
                                    int Cout, int flags, void *workspace, size_t workspace_bytes, t2h_stream_t stream) {
    if (!dy || !x || !dw) return fail(T2H_ERR_ARG, "upconv2x2_bx3_wgrad: null pointer");
    if (int rc = check_up_bx3("upconv2x2_bx3_wgrad", B, H, W, Cin, Cout)) return rc;
    if (lddy < Cout || lddy % 4) return fail(T2H_ERR_ARG, "upconv2x2_bx3_wgrad: lddy=%d must be >= Cout=%d and a multiple of 4", lddy, Cout);
    if (W < TW) return fail(T2H_ERR_ARG, "upconv2x2_bx3_wgrad: W=%d below the %d-pixel unit", W, TW);
    if (!al16(dy) || !al16(x)) return fail(T2H_ERR_ARG, "upconv2x2_bx3_wgrad: pointers must be 16-byte aligned");
    const size_t need = t2h_upconv2x2_bx3_wgrad_workspace_bytes(B, H, W, Cin, Cout);
    if (!workspace || workspace_bytes < need || !al16(workspace))
        return fail(T2H_ERR_WORKSPACE, "upconv2x2_bx3_wgrad: workspace %zu < %zu bytes", workspace_bytes, need);
    hipStream_t s = as_stream(stream);
    const WgradPlan p = up_wgrad_plan(B, H, W, Cin, Cout);
    const int Ncols = 4 * Cout;
    float *slab = static_cast<float *>(workspace);
    float *colslab = slab + (size_t)p.splits * Cin * Ncols;
    WgradArgs a{};
    a.dy = x; a.x = dy; a.slab = slab; a.colslab = db ? colslab : nullptr;           // (roles swapped, see the kernel)
    a.H = H; a.W = W; a.Cin = Cout; a.Cout = Cin; a.n_units = p.n_units; a.units_per_split = p.units_per_split; a.ldx = lddy;
    dim3 grid(p.splits, Cout / CC, Cin / (32 * p.cot));
    if (grid.y > 65535 || grid.z > 65535) return fail(T2H_ERR_ARG, "upconv2x2_bx3_wgrad: too many channel chunks");
    if (flags & T2H_F16X2) {
        if (p.cot == 4) { hipLaunchKernelGGL((bx3_wgrad_kernel<4, 2, true>), grid, dim3(NT), 0, s, a); note_kernel("bx3_wgrad_kernel<4,2,true,0>"); }
        else { hipLaunchKernelGGL((bx3_wgrad_kernel<2, 2, true>), grid, dim3(NT), 0, s, a); note_kernel("bx3_wgrad_kernel<2,2,true,0>"); }
    } else if (p.cot == 4) { hipLaunchKernelGGL((bx3_wgrad_kernel<4, 3, true>), grid, dim3(NT), 0, s, a); note_kernel("bx3_wgrad_kernel<4,3,true,0>"); }
    else { hipLaunchKernelGGL((bx3_wgrad_kernel<2, 3, true>), grid, dim3(NT), 0, s, a); note_kernel("bx3_wgrad_kernel<2,3,true,0>"); }
    if (int rc = check_launch("upconv2x2_bx3_wgrad")) return rc;
    return launch_reduce_slabs(slab, p.splits, (long long)Cin * Ncols, Cin, Ncols, Ncols, (flags & T2H_ACCUM) ? 1 : 0, dw, colslab, db, s,
                               p.splits, Cout, (flags & T2H_DEFER_REDUCE) != 0);
}

// ---- weight preparation for the fp16 two-way split (T2H_F16X2) ------------------------------------------------------------------------
// buffer = two f16 planes in MFMA B-fragment order (4 bytes per weight) + a 256-byte trailer: [0] bits of max |w|, [1] 2^-e_w, [2] e_w
T2H_API size_t t2h_conv3x3_f16x2_weights_bytes(int Cin, int Cout) {
    if (Cin < 32 || Cout < 32 || Cin % 32 || Cout % 32) return 0;
    return (size_t)Cout * 9 * Cin * 4 + 256;
}

static int absmax_into(const float *w, long long rows, int cols, long long ld, unsigned *trailer, hipStream_t s, const char *what) {
    if (hipMemsetAsync(trailer, 0, 256, s) != hipSuccess) return fail(T2H_ERR_LAUNCH, "%s: hipMemsetAsync failed", what);
    const long long total = rows * cols;
    const unsigned blocks = (unsigned)std::min<long long>((total + 4095) / 4096, 1024);
    hipLaunchKernelGGL(absmax_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, s, w, rows, cols, ld, trailer);
    return check_launch(what);
}

T2H_API int t2h_conv3x3_f16x2_prepare(const float *w, int Cin, int Cout, int transposed, void *wf, t2h_stream_t stream) {
    if (!w || !wf) return fail(T2H_ERR_ARG, "conv3x3_f16x2_prepare: null pointer");
    if (Cin < 32 || Cout < 32 || Cin % 32 || Cout % 32 || !al16(wf))
        return fail(T2H_ERR_ARG, "conv3x3_f16x2_prepare: Cin=%d, Cout=%d must be multiples of 32, wf 16-byte aligned", Cin, Cout);
    hipStream_t s = as_stream(stream);
    unsigned *trailer = reinterpret_cast<unsigned *>(static_cast<unsigned char *>(wf) + (size_t)Cout * 9 * Cin * 4);
    if (int rc = absmax_into(w, 1, Cout * 9 * Cin, 0, trailer, s, "conv3x3_f16x2_prepare")) return rc;
    const int Kc = transposed ? Cout : Cin, Nc = transposed ? Cin : Cout;
    const long long total = (long long)(Kc / 16) * 9 * (Nc / 32) * 64;
    const unsigned blocks = (unsigned)((total + 255) / 256);
    if (transposed) hipLaunchKernelGGL((bx3_prepare_kernel<true, true>), dim3(blocks), dim3(256), 0, s, w, Cin, Cout, static_cast<unsigned *>(wf), trailer);
    else hipLaunchKernelGGL((bx3_prepare_kernel<false, true>), dim3(blocks), dim3(256), 0, s, w, Cin, Cout, static_cast<unsigned *>(wf), trailer);
    return check_launch("conv3x3_f16x2_prepare");
}

T2H_API size_t t2h_gemm_f16x2_weights_bytes(int K, int N) {
    if (K < 16 || N < 32 || K % 16 || N % 32) return 0;
    return (size_t)K * N * 4 + 256;
}

T2H_API int t2h_gemm_f16x2_prepare(const float *w, int ldw, int K, int N, int w_is_kn, void *wf, t2h_stream_t stream) {
    if (!w || !wf) return fail(T2H_ERR_ARG, "gemm_f16x2_prepare: null pointer");
    if (K < 16 || N < 32 || K % 16 || N % 32 || !al16(wf) || ldw < (w_is_kn ? N : K))
        return fail(T2H_ERR_ARG, "gemm_f16x2_prepare: K=%d must be a multiple of 16, N=%d of 32, ldw=%d at least the row length", K, N, ldw);
    hipStream_t s = as_stream(stream);
    unsigned *trailer = reinterpret_cast<unsigned *>(static_cast<unsigned char *>(wf) + (size_t)K * N * 4);
    if (int rc = absmax_into(w, w_is_kn ? K : N, w_is_kn ? N : K, ldw, trailer, s, "gemm_f16x2_prepare")) return rc;
    const long long total = (long long)(K / 16) * (N / 32) * 64;
    const unsigned blocks = (unsigned)((total + 255) / 256);
    if (w_is_kn) hipLaunchKernelGGL((bx3_prepare_gemm_kernel<true, true>), dim3(blocks), dim3(256), 0, s, w, K, N, ldw, static_cast<unsigned *>(wf), trailer);
    else hipLaunchKernelGGL((bx3_prepare_gemm_kernel<false, true>), dim3(blocks), dim3(256), 0, s, w, K, N, ldw, static_cast<unsigned *>(wf), trailer);
    return check_launch("gemm_f16x2_prepare");
}

// ---- batched preparation --------------------------------------------------------------------------------------------------------------
T2H_API int t2h_split_weights_batch(const t2h_prep_desc *descs, int n, t2h_stream_t stream) {
    if (n < 0 || (n > 0 && !descs)) return fail(T2H_ERR_ARG, "split_weights_batch: bad descriptor list");
    hipStream_t s = as_stream(stream);
    for (int base = 0; base < n; base += kPrepBatch) {
        PrepBatch bm{}, bs{};
        unsigned mblocks = 0, sblocks = 0;
        const int cnt = std::min(kPrepBatch, n - base);
        for (int i = 0; i < cnt; ++i) {
            const t2h_prep_desc &d = descs[base + i];
            const bool conv = d.kind < 2;
            if (!d.w || !d.wf || d.kind < 0 || d.kind > 3 || d.a < 16 || d.b < 16 || !al16(d.wf) || (d.h2 && d.maxslot != 0 && d.maxslot != 3))
                return fail(T2H_ERR_ARG, "split_weights_batch: descriptor %d is malformed", base + i);
            // (a, b) = (Cin, Cout) for the 3x3 weights, (K, N) for the matrices
            const int Kc = conv ? (d.kind == 1 ? d.b : d.a) : d.a, Nc = conv ? (d.kind == 1 ? d.a : d.b) : d.b;
            if (Kc % 16 || Nc % 32) return fail(T2H_ERR_ARG, "split_weights_batch: descriptor %d: K=%d %% 16, N=%d %% 32", base + i, Kc, Nc);
            const long long threads = (long long)(Kc / 16) * (conv ? 9 : 1) * (Nc / 32) * 64;
            const long long elems = conv ? (long long)d.a * 9 * d.b : (long long)d.a * d.b;
            if (threads > (1LL << 31) || elems > (1LL << 31)) return fail(T2H_ERR_ARG, "split_weights_batch: descriptor %d too large", base + i);
            bs.d[i] = d; bs.first[i] = sblocks; sblocks += (unsigned)((threads + 255) / 256);
            if (d.h2) { bm.d[bm.n] = d; bm.first[bm.n] = mblocks; mblocks += (unsigned)std::min<long long>((elems + 4095) / 4096, 64); ++bm.n; }
        }
        bs.n = cnt; bs.first[cnt] = sblocks; bm.first[bm.n] = mblocks;
        if (bm.n) hipLaunchKernelGGL(prep_absmax_batch_kernel, dim3(mblocks), dim3(256), 0, s, bm);
        hipLaunchKernelGGL(prep_split_batch_kernel, dim3(sblocks), dim3(256), 0, s, bs);
    }
    return check_launch("split_weights_batch");
}
