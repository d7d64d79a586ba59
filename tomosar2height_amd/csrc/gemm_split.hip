// bf16-MFMA variants of the per-point GEMMs with the fp32 -> bf16 conversion done ONCE per element while staging:
//
//   PIECES = 1  T2H_BF16    operands rounded to bf16 (RNE), fp32 accumulate                      (BASELINE configs[2])
//   PIECES = 3  T2H_BF16X3  every fp32 operand split exactly into three bf16 pieces v1 + v2 + v3; the six piece
//                           products with i + j <= 4 are summed (small terms first): fp32-grade products (dropped
//                           terms <= ~2^-23 relative), fp32 accumulation.  Opt-in; the default path is gemm.hip.
//
// Same NT / NN / TN roles, flags and epilogue as gemm.hip (see there).  128 x 128 x 16 tiles, 4 waves x (2 x 2) MFMA
// 32x32x16 tiles.  LDS image per operand and piece: [row][16 k] bf16 = 32-byte rows, k contiguous, the two 16-byte
// halves of a row XOR-swizzled with bit 3 of the row so that the ds_read_b128 fragment reads (lane = row, half = k / 8)
// are conflict free without padding.  K-contiguous sources (x rows, weight rows) are converted float4 -> 4 bf16 and
// stored with one ds_write_b64 per piece; row-contiguous sources (dY^T, x in the weight gradient, W in the data
// gradient) are loaded as two float4 of adjacent k and transposed on the way in as packed bf16 pairs (ds_write_b32).
// The inner loop is then 12 ds_read_b128 + 24 MFMAs (PIECES = 3) with no conversion work left in it.
#include "t2h_common.h"
#include "gemm_args.h"

namespace t2h {

using f32x16s = __attribute__((ext_vector_type(16))) float;
using bf16x8s = __attribute__((ext_vector_type(8))) __bf16;
using bf16x4s = __attribute__((ext_vector_type(4))) __bf16;
using bf16x2s = __attribute__((ext_vector_type(2))) __bf16;

constexpr int SBK = 16;              // reduction slab
constexpr int SROW = 32;             // bytes per LDS row (16 bf16)

using SplitArgs = GemmArgs;
constexpr int SF_RELU_A = F_RELU_A, SF_RELU_B = F_RELU_B, SF_RELU_OUT = F_RELU_OUT, SF_ACCUM = F_ACCUM;

template <int PIECES>
__device__ inline void split4(float4 v, bf16x4s out[PIECES]) {
    float f[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        __bf16 p1 = (__bf16)f[q];
        out[0][q] = p1;
        if (PIECES == 3) {
            float r1 = f[q] - (float)p1;
            __bf16 p2 = (__bf16)r1;
            float r2 = r1 - (float)p2;
            out[1][q] = p2;
            out[2][q] = (__bf16)r2;
        }
    }
}

__device__ inline int swz(int row, int half) { return half ^ ((row >> 3) & 1); }

// one operand tile (128 rows x 16 k) -> PIECES LDS images of 4 KiB each
template <bool KC, int PIECES>
struct SplitLoader {
    float4 r[2];

    __device__ inline void load(const float *__restrict__ src, int ld, int row0, int rows, int k0, int kend, int tid,
                                bool relu) {
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (KC) {               // element (i, k) at src[(row0+i)*ld + k0+k]; float4 along k
                int idx = tid + f * 256, i = idx >> 2, kc = idx & 3;
                int m = row0 + i, k = k0 + kc * 4;
                if (m < rows && k < kend) v = *reinterpret_cast<const float4 *>(src + (size_t)m * ld + k);
            } else {                // element (i, k) at src[(k0+k)*ld + row0+i]; float4 along i, rows k = 2*kpair + f
                int ic = tid >> 3, kpair = tid & 7;
                int m = row0 + ic * 4, k = k0 + kpair * 2 + f;
                if (m < rows && k < kend) v = *reinterpret_cast<const float4 *>(src + (size_t)k * ld + m);
            }
            if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            r[f] = v;
        }
    }

    __device__ inline void store(char *__restrict__ lds /* PIECES images, 128*32 B apart */, int tid) const {
        if (KC) {
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                int idx = tid + f * 256, i = idx >> 2, kc = idx & 3;
                bf16x4s p[PIECES];
                split4<PIECES>(r[f], p);
                int off = i * SROW + swz(i, kc >> 1) * 16 + (kc & 1) * 8;
#pragma unroll
                for (int q = 0; q < PIECES; ++q) *reinterpret_cast<bf16x4s *>(lds + q * (128 * SROW) + off) = p[q];
            }
        } else {
            int ic = tid >> 3, kpair = tid & 7;
            bf16x4s p0[PIECES], p1[PIECES];
            split4<PIECES>(r[0], p0);      // k = 2*kpair
            split4<PIECES>(r[1], p1);      // k = 2*kpair + 1
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                int i = ic * 4 + e;
                int off = i * SROW + swz(i, kpair >> 2) * 16 + (kpair & 3) * 4;
#pragma unroll
                for (int q = 0; q < PIECES; ++q) {
                    bf16x2s pr;
                    pr[0] = p0[q][e];
                    pr[1] = p1[q][e];
                    *reinterpret_cast<bf16x2s *>(lds + q * (128 * SROW) + off) = pr;
                }
            }
        }
    }
};

template <bool A_KC, bool B_KC, int PIECES, int MINW>
__global__ __launch_bounds__(256, MINW) void gemm_split_kernel(SplitArgs p) {
    constexpr int IMG = 128 * SROW;                 // bytes per piece image
    constexpr int OPER = PIECES * IMG;              // bytes per operand
    constexpr int BUF = 2 * OPER;                   // A then B
    constexpr int EPI = 4 * 32 * 36 * 4;            // epilogue patches
    constexpr int LDS_BYTES = 2 * BUF > EPI ? 2 * BUF : EPI;
    __shared__ __attribute__((aligned(16))) char lds[LDS_BYTES];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    int tile_m, tile_n, split;
    {   // XCD-aware work order (see gemm.hip)
        const unsigned nb = gridDim.x * gridDim.y * gridDim.z;
        const unsigned b = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        const unsigned q = nb / 8, r = nb % 8, x = b % 8, i = b / 8;
        const unsigned t = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
        tile_n = t % gridDim.x;
        tile_m = (t / gridDim.x) % gridDim.y;
        split = t / (gridDim.x * gridDim.y);
    }
    const int m0 = tile_m * 128, n0 = tile_n * 128;
    const int kbeg = split * p.k_chunk, kend = min(p.K, kbeg + p.k_chunk);
    const bool relu_a = p.flags & SF_RELU_A, relu_b = p.flags & SF_RELU_B;

    SplitLoader<A_KC, PIECES> la;
    SplitLoader<B_KC, PIECES> lb;
    f32x16s acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.0f;

    float4 csum = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool do_colsum = !A_KC && p.colsum != nullptr && tile_n == 0;

    const int nk = (kend - kbeg + SBK - 1) / SBK;
    if (nk > 0) {
        la.load(p.A, p.lda, m0, p.M, kbeg, kend, tid, relu_a);
        lb.load(p.B, p.ldb, n0, p.N, kbeg, kend, tid, relu_b);
        la.store(lds, tid);
        lb.store(lds + OPER, tid);
    }
    __syncthreads();
    const int r = lane & 31, h = lane >> 5;
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (do_colsum) {      // direct layout: this thread's two float4 are rows k, k+1 of columns 4*ic .. 4*ic+3
            csum.x += la.r[0].x + la.r[1].x; csum.y += la.r[0].y + la.r[1].y;
            csum.z += la.r[0].z + la.r[1].z; csum.w += la.r[0].w + la.r[1].w;
        }
        if (kt + 1 < nk) {
            la.load(p.A, p.lda, m0, p.M, kbeg + (kt + 1) * SBK, kend, tid, relu_a);
            lb.load(p.B, p.ldb, n0, p.N, kbeg + (kt + 1) * SBK, kend, tid, relu_b);
        }
        const char *abase = lds + cur * BUF, *bbase = lds + cur * BUF + OPER;
        bf16x8s a[2][PIECES], b[2][PIECES];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int row = wm * 64 + i * 32 + r;
            int off = row * SROW + swz(row, h) * 16;
#pragma unroll
            for (int q = 0; q < PIECES; ++q) a[i][q] = *reinterpret_cast<const bf16x8s *>(abase + q * IMG + off);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            int row = wn * 64 + j * 32 + r;
            int off = row * SROW + swz(row, h) * 16;
#pragma unroll
            for (int q = 0; q < PIECES; ++q) b[j][q] = *reinterpret_cast<const bf16x8s *>(bbase + q * IMG + off);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                f32x16s c = acc[i][j];
                if (PIECES == 3) {
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], b[j][0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][2], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][1], c, 0, 0, 0);
                }
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][0], c, 0, 0, 0);
                acc[i][j] = c;
            }
        if (kt + 1 < nk) {
            la.store(lds + (cur ^ 1) * BUF, tid);
            lb.store(lds + (cur ^ 1) * BUF + OPER, tid);
        }
        __syncthreads();
    }

    if (do_colsum) {      // threads with equal ic (= tid >> 3) cover the same 4 columns: fixed-order reduction
        float4 *red = reinterpret_cast<float4 *>(lds);
        red[tid] = csum;
        __syncthreads();
        if (tid < 32) {
            float4 t = red[tid * 8];
            for (int j = 1; j < 8; ++j) { float4 u = red[tid * 8 + j]; t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w; }
            float *dst = p.colsum + (size_t)split * p.M + m0 + tid * 4;
            float tv[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (m0 + tid * 4 + q < p.M) dst[q] = tv[q];
        }
        __syncthreads();
    }

    float *C = p.C + (size_t)split * p.slab_stride;
    const bool accum = p.flags & SF_ACCUM, relu_out = p.flags & SF_RELU_OUT;
    constexpr int EP = 36;
    float *patch = reinterpret_cast<float *>(lds) + wave * (32 * EP);
    const int er = lane >> 3, ec = (lane & 7) * 4;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row0 = m0 + wm * 64 + i * 32, col0 = n0 + wn * 64 + j * 32;
#pragma unroll
            for (int q = 0; q < 16; ++q)
                patch[((q & 3) + 8 * (q >> 2) + 4 * (lane >> 5)) * EP + (lane & 31)] = acc[i][j][q];
            const int col = col0 + ec;
            float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p.bias && col < p.N) bv = *reinterpret_cast<const float4 *>(p.bias + col);
#pragma unroll
            for (int pass = 0; pass < 4; ++pass) {
                const int row = row0 + pass * 8 + er;
                float4 v = *reinterpret_cast<const float4 *>(patch + (pass * 8 + er) * EP + ec);
                if (row < p.M && col < p.N) {
                    v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
                    if (p.mask) {
                        float4 mk = *reinterpret_cast<const float4 *>(p.mask + (size_t)row * p.ldm + col);
                        v.x = mk.x > 0.f ? v.x : 0.f; v.y = mk.y > 0.f ? v.y : 0.f;
                        v.z = mk.z > 0.f ? v.z : 0.f; v.w = mk.w > 0.f ? v.w : 0.f;
                    }
                    if (relu_out) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                    float4 *dst = reinterpret_cast<float4 *>(C + (size_t)row * p.ldc + col);
                    if (accum) { float4 o = *dst; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
                    *dst = v;
                }
            }
        }
}

// mode: 1 = bf16, 2 = bf16x3.  Returns T2H_ERR_* or T2H_OK.
int launch_gemm_split(int mode, bool a_kc, bool b_kc, const SplitArgs &a, int splits, hipStream_t s, const char *what) {
    dim3 grid((a.N + 127) / 128, (a.M + 127) / 128, splits);
    if (grid.y > 65535 || grid.z > 65535) return fail(T2H_ERR_ARG, "%s: grid too large", what);
#define T2H_SPLIT_LAUNCH(AKC, BKC, P, MW) hipLaunchKernelGGL((gemm_split_kernel<AKC, BKC, P, MW>), grid, dim3(256), 0, s, a)
    if (mode == 2) {
        if (a_kc && b_kc) T2H_SPLIT_LAUNCH(true, true, 3, 2);
        else if (a_kc) T2H_SPLIT_LAUNCH(true, false, 3, 2);
        else T2H_SPLIT_LAUNCH(false, false, 3, 2);
    } else {
        if (a_kc && b_kc) T2H_SPLIT_LAUNCH(true, true, 1, 3);
        else if (a_kc) T2H_SPLIT_LAUNCH(true, false, 1, 3);
        else T2H_SPLIT_LAUNCH(false, false, 1, 3);
    }
#undef T2H_SPLIT_LAUNCH
    note_kernel(mode == 2 ? "gemm_split_kernel<bf16x3>" : "gemm_split_kernel<bf16>");
    return check_launch(what);
}

}  // namespace t2h
