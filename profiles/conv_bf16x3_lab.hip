// Lab behind DESIGN.md section 4 "3x3 convolution on bf16 MFMA with an exact 3-way operand split" (VERDICT r03 item 2).
// Not part of the product (profiles/ is evidence, csrc/ is what ships).  One file, no torch; the fp32 baseline is the SHIPPED
// kernel, called through the C ABI of tomosar2height_amd/libt2h_hip.so in the same process on the same data.
//
//   hipcc -w --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off profiles/conv_bf16x3_lab.hip \
//         -Iinclude -Ltomosar2height_amd -lt2h_hip -Wl,-rpath,'$ORIGIN/../tomosar2height_amd' -o profiles/conv_bf16x3_lab
//   ./profiles/conv_bf16x3_lab [H=512] [Cin=64] [Cout=128] [iters=20]
//
// Arithmetic.  x = x1 + x2 + x3 exactly with x1 = bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2) (round to nearest even;
// the two differences are exact in fp32: 24 = 8 + 8 + 8 significant bits, the residuals carry their own signs).  A product
// a b = sum_{i,j} a_i b_j; the six terms with i + j <= 4 are kept (a1b1, a1b2, a2b1, a1b3, a2b2, a3b1), the dropped ones are
// below 2^-25 |a b|.  Every bf16 x bf16 product is exact in fp32; v_mfma_f32_32x32x16_bf16 accumulates in fp32.  So the
// result differs from the fp32 fma chain of v_mfma_f32_32x32x2_f32 only in rounding ORDER -- measured below against float64.
// Rate: 6 bf16 MFMAs of 32 cycles cover 16 k of a 32 x 32 tile = 12 cycles per k; the fp32 MFMA needs 32: ceiling 2.67x.
//
// Structure (forward; the data gradient is the same kernel on flipped, transposed weights):
//   * a workgroup owns TH = 4 image rows x 32 columns x BN output channels; 4 waves = 2 (row pairs) x 2 (channel halves)
//   * per 32-input-channel chunk the (TH + 2) x 34 pixel HALO tile is read once (fp32), split once, and kept in LDS as three
//     bf16 images with an 80-byte pixel stride (odd multiple of 16 B: conflict-free ds_read_b128 fragments for every tap)
//     -- the nine taps re-read it from LDS, never from L2, and the inner loop carries no conversion
//   * weights are split ONCE (prep kernel; in the product: once per optimizer step) into the exact byte order of the MFMA
//     B fragments, so a (tap, 16-channel) slab is one linear 12 KB run: LDS-DMA (global_load_lds_dwordx4), double buffered
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
#include "t2h.h"

using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
using f32x2 = __attribute__((ext_vector_type(2))) float;
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

// ---- the split --------------------------------------------------------------------------------------------------------
__device__ inline unsigned pack_bf16x2(float a, float b) {             // v_cvt_pk_bf16_f32 (round to nearest even)
    f32x2 v = {a, b};
    bf16x2 r = __builtin_convertvector(v, bf16x2);
    return *reinterpret_cast<unsigned *>(&r);
}
__device__ inline float bf_lo(unsigned p) { return __uint_as_float(p << 16); }
__device__ inline float bf_hi(unsigned p) { return __uint_as_float(p & 0xffff0000u); }
// two floats -> three packed bf16 pairs (hi, mid, lo)
__device__ inline void split3(float a, float b, unsigned &p1, unsigned &p2, unsigned &p3) {
    p1 = pack_bf16x2(a, b);
    const float ra = a - bf_lo(p1), rb = b - bf_hi(p1);                   // exact
    p2 = pack_bf16x2(ra, rb);
    const float sa = ra - bf_lo(p2), sb = rb - bf_hi(p2);                 // exact
    p3 = pack_bf16x2(sa, sb);
}

// ---- weight preparation: [Cout][9][Cin] fp32 -> MFMA B-fragment order, three bf16 planes ----------------------------------
// slab (chunk c of 32 channels, tap t, half q) -> [cout tile of 32][plane][lane][8 bf16]; lane (r = l & 31, h = l >> 5) holds
// B[k = 8 h + j][col r] = W[cout tile * 32 + r][tap][c * 32 + q * 16 + 8 h + j].  DGRAD: the same for the transposed conv:
// B[k = (tap, co)][n = ci] = W[co][8 - tap][ci], chunks over Cout.
template <bool DGRAD>
__global__ void prep_weights_kernel(const float *__restrict__ w, int Cin, int Cout, unsigned *__restrict__ wf, int cc) {
    const int Kc = DGRAD ? Cout : Cin, Nc = DGRAD ? Cin : Cout;         // reduction channels, output channels
    const int ntile = Nc / 32;
    const long long total = (long long)(Kc / 16) * 9 * ntile * 64;        // one thread per (slab, tile, lane)
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int lane = (int)(t & 63);
    const int tile = (int)((t >> 6) % ntile);
    const long long slab = (t >> 6) / ntile;                              // ((c * 9 + tap) * nq + q), nq = cc / 16
    const int nq = cc / 16;
    const int q = (int)(slab % nq), tap = (int)((slab / nq) % 9), c = (int)((slab / nq) / 9);
    const int r = lane & 31, h = lane >> 5;
    const int n = tile * 32 + r, k0 = c * cc + q * 16 + 8 * h;
    unsigned p1[4], p2[4], p3[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float a, b;
        if (!DGRAD) { a = w[((size_t)n * 9 + tap) * Cin + k0 + 2 * j]; b = w[((size_t)n * 9 + tap) * Cin + k0 + 2 * j + 1]; }
        else { a = w[((size_t)(k0 + 2 * j) * 9 + (8 - tap)) * Cin + n]; b = w[((size_t)(k0 + 2 * j + 1) * 9 + (8 - tap)) * Cin + n]; }
        split3(a, b, p1[j], p2[j], p3[j]);
    }
    uint4 *dst = reinterpret_cast<uint4 *>(wf) + ((slab * ntile + tile) * 3) * 64 + lane;
    dst[0] = make_uint4(p1[0], p1[1], p1[2], p1[3]);
    dst[64] = make_uint4(p2[0], p2[1], p2[2], p2[3]);
    dst[128] = make_uint4(p3[0], p3[1], p3[2], p3[3]);
}

// ---- the convolution ------------------------------------------------------------------------------------------------------
struct ConvArgs {
    const float *x;          // [B,H,W,Cin] fp32
    const unsigned *wf;      // prepared weights
    const float *bias;       // [Cout] or null
    float *y;                // [B,H,W,Cout]
    int H, W, Cin, Cout, relu;
};

constexpr int TW = 32;

// ORD: 0 = fragments loaded tile-major, smallest terms first, all reads before the first MFMA (the first shipped form);
//      1 = fragments loaded PLANE-major and the products largest first (a1b1 needs only the first TM + TN reads), reads still fenced;
//      2 = as 1 without the fence (the compiler may sink reads between the MFMAs)
// PRIO: 1 = static s_setprio by the parity of the wave's slot on its SIMD (HW_ID.WAVE_ID): the two co-resident waves of a SIMD get
//       different priorities, so their MFMA bursts do not interleave evenly (both then finish together and leave the matrix pipe idle
//       while both read LDS / issue DMA / wait at their barriers) but run one after the other; 2 = by (blockIdx.x >> 8) & 1
template <int TH, int BN, int WAVES_M, int WAVES_N, int CC, int ORD = 0, int PRIO = 0>
__global__ __launch_bounds__(256, 2) void conv3x3_bf16x3_kernel(ConvArgs p) {
    if (PRIO == 1) {
        const unsigned slot = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4);      // HW_REG_HW_ID, bits [3:0]: WAVE_ID
        if (slot & 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
    } else if (PRIO == 2) {
        if ((blockIdx.x >> 8) & 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
    }
    constexpr int PXB = CC * 2 + 16;              // bytes per pixel and plane in the halo image: CC bf16 + 16 B pad (odd multiple of 16 B)
    constexpr int NQ = CC / 16, F4 = CC / 4;
    constexpr int TM = TH / WAVES_M, TN = BN / (32 * WAVES_N);
    constexpr int HP = (TH + 2) * (TW + 2);                              // halo pixels
    constexpr int PLANE = HP * PXB;                                      // bytes per bf16 plane
    constexpr int BSLAB = (BN / 32) * 3 * 1024;                          // bytes per weight slab
    constexpr int HALO_BYTES = 3 * PLANE;
    constexpr int LDS_BYTES = HALO_BYTES + 2 * BSLAB;
    static_assert(LDS_BYTES >= 4 * 32 * 36 * 4, "epilogue patches");
    __shared__ __attribute__((aligned(1024))) unsigned char lds[LDS_BYTES];
    unsigned char *halo = lds, *bbuf = lds + HALO_BYTES;
    static_assert(HALO_BYTES % 1024 == 0 || true, "");

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    // XCD-aware order: consecutive tiles (same image rows, neighbouring columns share halo columns in L2) on one XCD
    const unsigned nb = gridDim.x, bid = blockIdx.x;
    const unsigned qq = nb / 8, rr = nb % 8, xx = bid % 8, i8 = bid / 8;
    unsigned t = (xx < rr ? xx * (qq + 1) : rr * (qq + 1) + (xx - rr) * qq) + i8;
    const int ntn = p.Cout / BN;
    const int tn = t % ntn; t /= ntn;
    const int tiles_x = p.W / TW;
    const int tx = t % tiles_x; t /= tiles_x;
    const int tiles_y = p.H / TH;
    const int ty = t % tiles_y, b = t / tiles_y;
    const int x0 = tx * TW, y0 = ty * TH, n0 = tn * BN;
    const int nchunk = p.Cin / CC;

    // halo staging: float4 = 4 channels; 8 float4 per pixel and chunk
    constexpr int NF4 = HP * F4, PER = (NF4 + 255) / 256;
    float4 hreg[PER];
    auto halo_load = [&](int c) {
#pragma unroll
        for (int f = 0; f < PER; ++f) {
            const int idx = tid + f * 256;
            const int px = idx / F4, c4 = idx % F4;
            const int hy = px / (TW + 2), hx = px - hy * (TW + 2);
            const int gy = y0 + hy - 1, gx = x0 + hx - 1;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (idx < NF4 && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W)
                v = *reinterpret_cast<const float4 *>(p.x + (((size_t)b * p.H + gy) * p.W + gx) * p.Cin + c * CC + c4 * 4);
            hreg[f] = v;
        }
    };
    auto halo_store = [&]() {
#pragma unroll
        for (int f = 0; f < PER; ++f) {
            const int idx = tid + f * 256;
            if (idx < NF4) {
                const int px = idx / F4, c4 = idx % F4;
                unsigned a1, a2, a3, b1, b2, b3;
                split3(hreg[f].x, hreg[f].y, a1, a2, a3);
                split3(hreg[f].z, hreg[f].w, b1, b2, b3);
                unsigned char *d = halo + px * PXB + c4 * 8;
                *reinterpret_cast<uint2 *>(d) = make_uint2(a1, b1);
                *reinterpret_cast<uint2 *>(d + PLANE) = make_uint2(a2, b2);
                *reinterpret_cast<uint2 *>(d + 2 * PLANE) = make_uint2(a3, b3);
            }
        }
    };
    // weight slab s = (chunk * 9 + tap) * 2 + q: (BN / 32) * 3 KB, linear
    const unsigned char *wbase = reinterpret_cast<const unsigned char *>(p.wf);
    const size_t slab_stride = (size_t)(p.Cout / 32) * 3 * 1024;
    auto issue_b = [&](int s, unsigned char *dst) {
        const unsigned char *src = wbase + (size_t)s * slab_stride + (size_t)(n0 / 32) * 3 * 1024;
        constexpr int PIECES = BSLAB / 1024;                             // 1 KB per wave instruction, dealt round-robin to the waves
#pragma unroll
        for (int j = 0; j < (PIECES + 3) / 4; ++j)
            if (j * 4 + wave < PIECES)
                __builtin_amdgcn_global_load_lds((glb_void *)(src + (j * 256 + tid) * 16), (lds_void *)(dst + (j * 4 + wave) * 1024), 16, 0, 0);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int z = 0; z < 16; ++z) acc[i][j][z] = 0.0f;

    const int r = lane & 31, h = lane >> 5;
    halo_load(0);
    int s = 0;                                                           // global slab counter
    for (int c = 0; c < nchunk; ++c) {
        halo_store();                                                    // (the previous chunk's last barrier has passed)
        if (c == 0) issue_b(0, bbuf);                                    // (later chunks: issued by the previous chunk's last step)
        if (c + 1 < nchunk) halo_load(c + 1);                            // in flight under this chunk's MFMAs
        __syncthreads();
#pragma unroll 1
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap - 3 * ky;
#pragma unroll
            for (int q = 0; q < NQ; ++q, ++s) {
                const unsigned char *cur = bbuf + (s & 1) * BSLAB;
                if (!(c == nchunk - 1 && tap == 8 && q == NQ - 1)) issue_b(s + 1, bbuf + ((s + 1) & 1) * BSLAB);
                uint4 af[TM][3], bf[TN][3];
                if (ORD == 0) {
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        const int px = (wm * TM + i + ky) * (TW + 2) + r + kx;
                        const unsigned char *a = halo + px * PXB + q * 32 + h * 16;
#pragma unroll
                        for (int pl = 0; pl < 3; ++pl) af[i][pl] = *reinterpret_cast<const uint4 *>(a + pl * PLANE);
                    }
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int pl = 0; pl < 3; ++pl)
                            bf[j][pl] = *reinterpret_cast<const uint4 *>(cur + ((wn * TN + j) * 3 + pl) * 1024 + lane * 16);
                } else {
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
                        for (int i = 0; i < TM; ++i) {
                            const int px = (wm * TM + i + ky) * (TW + 2) + r + kx;
                            af[i][pl] = *reinterpret_cast<const uint4 *>(halo + px * PXB + q * 32 + h * 16 + pl * PLANE);
                        }
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            bf[j][pl] = *reinterpret_cast<const uint4 *>(cur + ((wn * TN + j) * 3 + pl) * 1024 + lane * 16);
                    }
                }
                if (ORD != 2) __builtin_amdgcn_sched_barrier(0);
                // ORD 0: smallest terms first: a3b1, a1b3, a2b2, a2b1, a1b2, a1b1; else largest first: a1b1, a1b2, a2b1, a2b2, a1b3, a3b1
                constexpr int ia0[6] = {2, 0, 1, 1, 0, 0}, ib0[6] = {0, 2, 1, 0, 1, 0};
                constexpr int ia1[6] = {0, 0, 1, 1, 0, 2}, ib1[6] = {0, 1, 0, 1, 2, 0};
#pragma unroll
                for (int e = 0; e < 6; ++e)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<bf16x8 *>(&af[i][ORD ? ia1[e] : ia0[e]]),
                                                                                 *reinterpret_cast<bf16x8 *>(&bf[j][ORD ? ib1[e] : ib0[e]]), acc[i][j], 0, 0, 0);
                __syncthreads();                                         // (drains the next slab's DMA: vmcnt(0))
            }
        }
    }

    // epilogue: per 32 x 32 tile through a private LDS patch -> float4 rows along the output channels
    float *patch = reinterpret_cast<float *>(lds) + wave * (32 * 36);
    const int er = lane >> 3, ec = (lane & 7) * 4;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + (wn * TN + j) * 32 + ec;
#pragma unroll
            for (int z = 0; z < 16; ++z) patch[((z & 3) + 8 * (z >> 2) + 4 * h) * 36 + r] = acc[i][j][z];
            float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p.bias) bv = *reinterpret_cast<const float4 *>(p.bias + col);
            const size_t pix0 = ((size_t)b * p.H + y0 + wm * TM + i) * p.W + x0;
#pragma unroll
            for (int pass = 0; pass < 4; ++pass) {
                float4 v = *reinterpret_cast<const float4 *>(patch + (pass * 8 + er) * 36 + ec);
                v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
                if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                *reinterpret_cast<float4 *>(p.y + (pix0 + pass * 8 + er) * p.Cout + col) = v;
            }
        }
}


// ---- "wide" variant: ONE 512-thread workgroup per CU (8 waves as 4 x 2, 8 image rows x 32 columns x 128 channels), 32 reduction
// channels per staged chunk AND per barrier (both 16-channel halves of a tap in one step: 48 MFMAs per wave between barriers instead
// of 24, weight slabs of 24 KB).  LDS: halo 81.6 KB + 2 x 24 KB.  The two waves of a SIMD belong to the same barrier domain.
template <int BN>
__global__ __launch_bounds__(512, 2) void conv3x3_bf16x3_wide_kernel(ConvArgs p) {
    constexpr int TH = 8, WAVES_M = 4, WAVES_N = 2, CC = 32, NTH = 512, NWV = 8;
    constexpr int PXB = CC * 2 + 16, F4 = CC / 4;
    constexpr int TM = TH / WAVES_M, TN = BN / (32 * WAVES_N);
    constexpr int HP = (TH + 2) * (TW + 2);
    constexpr int PLANE = HP * PXB;
    constexpr int BSLAB = 2 * (BN / 32) * 3 * 1024;                      // both halves of a tap
    constexpr int HALO_BYTES = 3 * PLANE;
    constexpr int LDS_BYTES = HALO_BYTES + 2 * BSLAB;
    __shared__ __attribute__((aligned(1024))) unsigned char lds[LDS_BYTES];
    unsigned char *halo = lds, *bbuf = lds + HALO_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const unsigned nb = gridDim.x, bid = blockIdx.x;
    const unsigned qq = nb / 8, rr = nb % 8, xx = bid % 8, i8 = bid / 8;
    unsigned t = (xx < rr ? xx * (qq + 1) : rr * (qq + 1) + (xx - rr) * qq) + i8;
    const int ntn = p.Cout / BN;
    const int tn = t % ntn; t /= ntn;
    const int tiles_x = p.W / TW;
    const int tx = t % tiles_x; t /= tiles_x;
    const int tiles_y = p.H / TH;
    const int ty = t % tiles_y, b = t / tiles_y;
    const int x0 = tx * TW, y0 = ty * TH, n0 = tn * BN;
    const int nchunk = p.Cin / CC;
    constexpr int NF4 = HP * F4, PER = (NF4 + NTH - 1) / NTH;
    float4 hreg[PER];
    auto halo_load = [&](int c) {
#pragma unroll
        for (int f = 0; f < PER; ++f) {
            const int idx = tid + f * NTH;
            const int px = idx / F4, c4 = idx % F4;
            const int hy = px / (TW + 2), hx = px - hy * (TW + 2);
            const int gy = y0 + hy - 1, gx = x0 + hx - 1;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (idx < NF4 && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W)
                v = *reinterpret_cast<const float4 *>(p.x + (((size_t)b * p.H + gy) * p.W + gx) * p.Cin + c * CC + c4 * 4);
            hreg[f] = v;
        }
    };
    auto halo_store = [&]() {
#pragma unroll
        for (int f = 0; f < PER; ++f) {
            const int idx = tid + f * NTH;
            if (idx < NF4) {
                const int px = idx / F4, c4 = idx % F4;
                unsigned a1, a2, a3, b1, b2, b3;
                split3(hreg[f].x, hreg[f].y, a1, a2, a3);
                split3(hreg[f].z, hreg[f].w, b1, b2, b3);
                unsigned char *d = halo + px * PXB + c4 * 8;
                *reinterpret_cast<uint2 *>(d) = make_uint2(a1, b1);
                *reinterpret_cast<uint2 *>(d + PLANE) = make_uint2(a2, b2);
                *reinterpret_cast<uint2 *>(d + 2 * PLANE) = make_uint2(a3, b3);
            }
        }
    };
    // weights in the lab's CC = 32 order: slab s = (c * 9 + tap) * 2 + q; a step = the two consecutive slabs of a tap
    const unsigned char *wbase = reinterpret_cast<const unsigned char *>(p.wf);
    const size_t slab_stride = (size_t)(p.Cout / 32) * 3 * 1024;
    auto issue_b = [&](int step, unsigned char *dst) {
        constexpr int HALF = (BN / 32) * 3;                              // 1 KB pieces per half
#pragma unroll
        for (int j = 0; j < (2 * HALF + NWV - 1) / NWV; ++j) {
            const int piece = j * NWV + wave;
            if (piece < 2 * HALF) {
                const int q = piece / HALF, pp = piece - q * HALF;
                const unsigned char *src = wbase + (size_t)(2 * step + q) * slab_stride + (size_t)(n0 / 32) * 3 * 1024 + pp * 1024 + lane * 16;
                __builtin_amdgcn_global_load_lds((glb_void *)src, (lds_void *)(dst + piece * 1024), 16, 0, 0);
            }
        }
    };
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int z = 0; z < 16; ++z) acc[i][j][z] = 0.0f;
    const int r = lane & 31, h = lane >> 5;
    halo_load(0);
    int s = 0;
    const int nstep = nchunk * 9;
    for (int c = 0; c < nchunk; ++c) {
        halo_store();
        if (c == 0) issue_b(0, bbuf);
        if (c + 1 < nchunk) halo_load(c + 1);
        __syncthreads();
#pragma unroll 1
        for (int tap = 0; tap < 9; ++tap, ++s) {
            const int ky = tap / 3, kx = tap - 3 * ky;
            const unsigned char *cur = bbuf + (s & 1) * BSLAB;
            if (s + 1 < nstep) issue_b(s + 1, bbuf + ((s + 1) & 1) * BSLAB);
            uint4 af[2][TM][3], bf[2][TN][3];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        const int px = (wm * TM + i + ky) * (TW + 2) + r + kx;
                        af[q][i][pl] = *reinterpret_cast<const uint4 *>(halo + px * PXB + q * 32 + h * 16 + pl * PLANE);
                    }
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        bf[q][j][pl] = *reinterpret_cast<const uint4 *>(cur + ((q * (BN / 32) + wn * TN + j) * 3 + pl) * 1024 + lane * 16);
                }
            }
            constexpr int ia1[6] = {0, 0, 1, 1, 0, 2}, ib1[6] = {0, 1, 0, 1, 2, 0};
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int e = 0; e < 6; ++e)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<bf16x8 *>(&af[q][i][ia1[e]]),
                                                                                 *reinterpret_cast<bf16x8 *>(&bf[q][j][ib1[e]]), acc[i][j], 0, 0, 0);
            __syncthreads();
        }
    }
    float *patch = reinterpret_cast<float *>(lds) + wave * (32 * 36);
    const int er = lane >> 3, ec = (lane & 7) * 4;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + (wn * TN + j) * 32 + ec;
#pragma unroll
            for (int z = 0; z < 16; ++z) patch[((z & 3) + 8 * (z >> 2) + 4 * h) * 36 + r] = acc[i][j][z];
            float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p.bias) bv = *reinterpret_cast<const float4 *>(p.bias + col);
            const size_t pix0 = ((size_t)b * p.H + y0 + wm * TM + i) * p.W + x0;
#pragma unroll
            for (int pass = 0; pass < 4; ++pass) {
                float4 v = *reinterpret_cast<const float4 *>(patch + (pass * 8 + er) * 36 + ec);
                v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
                if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                *reinterpret_cast<float4 *>(p.y + (pix0 + pass * 8 + er) * p.Cout + col) = v;
            }
        }
}

// ---- host -------------------------------------------------------------------------------------------------------------------
static float frand(unsigned &s) { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.0f - 1.0f; }

int main(int argc, char **argv) {
    const int H = argc > 1 ? atoi(argv[1]) : 512, Cin = argc > 2 ? atoi(argv[2]) : 64, Cout = argc > 3 ? atoi(argv[3]) : 128;
    const int iters = argc > 4 ? atoi(argv[4]) : 20;
    const int W = H, B = 1;
    const size_t nx = (size_t)B * H * W * Cin, ny = (size_t)B * H * W * Cout, nw = (size_t)Cout * 9 * Cin;
    std::vector<float> hx(nx), hw(nw), hb(Cout);
    unsigned seed = 12345;
    for (auto &v : hx) v = frand(seed) * (1.0f + 3.0f * fabsf(frand(seed)));       // full-range signs, some dynamic range
    const float ws = sqrtf(6.0f / (9.0f * (Cin + Cout)));                          // Xavier-uniform scale
    for (auto &v : hw) v = frand(seed) * ws;
    for (auto &v : hb) v = 0.1f * frand(seed);
    float *dx, *dw, *db, *dy32, *dy16;
    unsigned *dwf;
    CK(hipMalloc(&dx, nx * 4)); CK(hipMalloc(&dw, nw * 4)); CK(hipMalloc(&db, Cout * 4));
    CK(hipMalloc(&dy32, ny * 4)); CK(hipMalloc(&dy16, ny * 4)); CK(hipMalloc(&dwf, nw * 6));
    CK(hipMemcpy(dx, hx.data(), nx * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dw, hw.data(), nw * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, hb.data(), Cout * 4, hipMemcpyHostToDevice));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));

    // fp32 baseline: the shipped kernel through the C ABI
    const size_t wsb = t2h_conv3x3_fwd_workspace_bytes(B, H, W, Cin, Cout);
    void *wsp = nullptr;
    if (wsb) CK(hipMalloc(&wsp, wsb));
    auto run32 = [&]() { int rc = t2h_conv3x3_fwd(dx, dw, db, dy32, B, H, W, Cin, Cout, T2H_RELU_OUT, wsp, wsb, st); if (rc) { printf("t2h_conv3x3_fwd: %s\n", t2h_last_error_string()); exit(1); } };

    ConvArgs a{dx, dwf, db, dy16, H, W, Cin, Cout, 1};
    const int BN = Cout >= 128 ? 128 : (Cout >= 64 ? 64 : 32);
    struct Variant { const char *name; int th, cc; void (*launch)(ConvArgs, int, hipStream_t); };
#define V(TH_, BN_, WM_, WN_, CC_) VO(TH_, BN_, WM_, WN_, CC_, 0)
#define VO(TH_, BN_, WM_, WN_, CC_, ORD_) VP(TH_, BN_, WM_, WN_, CC_, ORD_, 0)
#define VP(TH_, BN_, WM_, WN_, CC_, ORD_, PRIO_) Variant{"TH=" #TH_ " BN=" #BN_ " waves " #WM_ "x" #WN_ " CC=" #CC_ " ORD=" #ORD_ " PRIO=" #PRIO_, TH_, CC_, \
    [](ConvArgs q, int grid, hipStream_t s_) { hipLaunchKernelGGL((conv3x3_bf16x3_kernel<TH_, BN_, WM_, WN_, CC_, ORD_, PRIO_>), dim3(grid), dim3(256), 0, s_, q); }}
    std::vector<Variant> vars;
    Variant wide128{"WIDE 512 threads TH=8 BN=128 CC=32 (48 MFMAs / barrier)", 8, 32,
        [](ConvArgs q, int grid, hipStream_t s_) { hipLaunchKernelGGL((conv3x3_bf16x3_wide_kernel<128>), dim3(grid), dim3(512), 0, s_, q); }};
    Variant wide64{"WIDE 512 threads TH=8 BN=64 CC=32", 8, 32,
        [](ConvArgs q, int grid, hipStream_t s_) { hipLaunchKernelGGL((conv3x3_bf16x3_wide_kernel<64>), dim3(grid), dim3(512), 0, s_, q); }};
    if (BN == 128) vars = {VP(4, 128, 2, 2, 32, 0, 1), VP(4, 128, 2, 2, 32, 0, 2), VP(8, 128, 2, 2, 16, 0, 1), VP(8, 128, 2, 2, 16, 0, 2), wide128, V(4, 128, 2, 2, 32), VO(4, 128, 2, 2, 32, 1), VO(4, 128, 2, 2, 32, 2), V(8, 128, 2, 2, 16), VO(8, 128, 2, 2, 16, 1), VO(8, 128, 2, 2, 16, 2)};
    else if (BN == 64) vars = {VP(4, 64, 4, 1, 32, 0, 1), VP(8, 64, 4, 1, 16, 0, 1), VP(8, 64, 4, 1, 16, 0, 2), wide64, V(4, 64, 4, 1, 32), VO(4, 64, 4, 1, 32, 1), VO(4, 64, 4, 1, 32, 2), V(8, 64, 4, 1, 16), VO(8, 64, 4, 1, 16, 1), VO(8, 64, 4, 1, 16, 2)};
    else vars = {V(4, 32, 4, 1, 32), VO(4, 32, 4, 1, 32, 1), VO(4, 32, 4, 1, 32, 2), V(8, 32, 4, 1, 16), VO(8, 32, 4, 1, 16, 1)};
    int cur_cc = 0;
    auto prep = [&]() {
        const long long total = (long long)(Cin / 16) * 9 * (Cout / 32) * 64;
        hipLaunchKernelGGL(prep_weights_kernel<false>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, dw, Cin, Cout, dwf, cur_cc);
    };
    auto time = [&](auto fn, const char *name) {
        for (int i = 0; i < 3; ++i) fn();
        CK(hipStreamSynchronize(st));
        float best = 1e30f, tot = 0.f;
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < iters; ++i) fn();
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            tot += ms; if (ms < best) best = ms;
        }
        const double us = 1e3 * best / iters, fl = 2.0 * 9 * Cin * (double)Cout * B * H * W;
        printf("%-34s %8.1f us (best of 5 x %d; mean %.1f)  %6.1f TF fp32-equivalent\n", name, us, iters, 1e3 * tot / 5 / iters, fl / us * 1e-6);
        return us;
    };
    // interleaved rounds in one process (cdna_hip_programming.md rule 24)
    double t32 = 1e30, t16 = 1e30;
    std::vector<double> tv(vars.size(), 1e30);
    std::vector<float> y32(ny), y16(ny);
    for (int round = 0; round < 2; ++round) {
        t32 = fmin(t32, time(run32, "fp32 MFMA (shipped)"));
        for (size_t v = 0; v < vars.size(); ++v) {
            if (H % vars[v].th) continue;
            cur_cc = vars[v].cc;
            prep();
            const int grid = B * (H / vars[v].th) * (W / TW) * (Cout / BN);
            CK(hipMemsetAsync(dy16, 0xff, ny * 4, st));
            tv[v] = fmin(tv[v], time([&]() { vars[v].launch(a, grid, st); }, vars[v].name));
            CK(hipGetLastError());
            if (round == 0) {                                            // every variant against the fp32 kernel's output
                CK(hipMemcpy(y32.data(), dy32, ny * 4, hipMemcpyDeviceToHost));
                CK(hipMemcpy(y16.data(), dy16, ny * 4, hipMemcpyDeviceToHost));
                double d = 0, m = 0;
                for (size_t i = 0; i < ny; ++i) { d = fmax(d, fabs((double)y32[i] - y16[i])); m = fmax(m, fabs((double)y32[i])); }
                printf("    max |fp32 - variant| = %.3e (max output %.2f)%s\n", d, m, d < 1e-4 * fmax(m, 1.0) ? "" : "   <-- MISMATCH");
            }
        }
    }
    size_t best_v = 0;
    for (size_t v = 1; v < vars.size(); ++v) if (tv[v] < tv[best_v]) best_v = v;
    t16 = tv[best_v];
    cur_cc = vars[best_v].cc;
    const double tprep = time(prep, "weight split (prep)");
    { const int grid = B * (H / vars[best_v].th) * (W / TW) * (Cout / BN); vars[best_v].launch(a, grid, st); CK(hipStreamSynchronize(st)); }
    printf("shape %dx%d %d->%d: best variant [%s] speedup %.2fx (fp32 %.1f us, bf16x3 %.1f us, prep %.1f us once per optimizer step)\n", H, W, Cin, Cout,
           vars[best_v].name, t32 / t16, t32, t16, tprep);
    CK(hipGetLastError());

    // accuracy against float64 on sampled outputs (both kernels)
    CK(hipMemcpy(y32.data(), dy32, ny * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(y16.data(), dy16, ny * 4, hipMemcpyDeviceToHost));
    double e32 = 0, e16 = 0, ymax = 0, d3216 = 0;
    unsigned s2 = 777;
    const int nsamp = 20000;
    for (int it = 0; it < nsamp; ++it) {
        s2 = s2 * 1664525u + 1013904223u; int y = (it < 64) ? (it & 1 ? H - 1 : 0) : (s2 >> 8) % H;
        s2 = s2 * 1664525u + 1013904223u; int x = (it < 64) ? ((it >> 1) & 1 ? W - 1 : (it * 7) % W) : (s2 >> 8) % W;
        s2 = s2 * 1664525u + 1013904223u; int co = (s2 >> 8) % Cout;
        double acc = hb[co], mag = fabs((double)hb[co]);
        for (int ky = 0; ky < 3; ++ky)
            for (int kx = 0; kx < 3; ++kx) {
                int yy = y + ky - 1, xq = x + kx - 1;
                if (yy < 0 || yy >= H || xq < 0 || xq >= W) continue;
                const float *xp = &hx[((size_t)yy * W + xq) * Cin], *wp = &hw[((size_t)co * 9 + ky * 3 + kx) * Cin];
                for (int ci = 0; ci < Cin; ++ci) { acc += (double)xp[ci] * wp[ci]; mag += fabs((double)xp[ci] * wp[ci]); }
            }
        const double want = acc > 0 ? acc : 0;
        const size_t o = ((size_t)y * W + x) * Cout + co;
        e32 = fmax(e32, fabs(y32[o] - want) / mag); e16 = fmax(e16, fabs(y16[o] - want) / mag);
        ymax = fmax(ymax, want);
    }
    for (size_t i = 0; i < ny; ++i) d3216 = fmax(d3216, fabs((double)y32[i] - y16[i]));
    printf("max |err| / sum|a b| vs float64 over %d sampled outputs: fp32 MFMA %.3e, bf16x3 %.3e;  max |fp32 - bf16x3| over all %zu outputs %.3e (max output %.2f)\n",
           nsamp, e32, e16, ny, d3216, ymax);
    return (e16 < 4e-7 && d3216 < 1e-4 * fmax(ymax, 1.0)) ? 0 : 2;
}
