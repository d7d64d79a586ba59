// Ablation lab behind DESIGN.md section 4 ("standalone ablation of the 512 -> 1024 forward loop"): one file, one variant
// per -D flag, no torch.  Not part of the product (profiles/ is evidence, csrc/ is what ships).
//
//   for v in BASE PIPE NO_GLOBAL NO_LDS_WRITE NO_BARRIER NO_LDS_READ ROWL FULLLINE FULLA SPLITW GLDS GLDS2 GLDS3 GLDS4 GLDS5 GLDS6; do
//     hipcc -w --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -D$v -DGLDS_MINW=4 -DVARIANT="\"$v\"" \
//           profiles/gemm_lab.hip -o lab_$v && ./lab_$v; done
//
//   BASE      the register-staged 128x128x16 loop of csrc/gemm.hip (plain dword epilogue)         113 TF
//   PIPE      + LDS fragment reads of k-pair kp+1 issued before the MFMAs of kp (sched_barrier)   114-115
//   NO_*      BASE with the global loads / LDS stores / barrier / LDS reads removed               134 / 127 / 115 / 140
//   ROWL      row-major b128 LDS image through registers (spills at the 128-VGPR budget)          78
//   FULLLINE  full 128-byte-line global loads for both operands (spills)                          82
//   FULLA     full-line loads for A only                                                          114.6
//   SPLITW    LDS stores interleaved behind the MFMAs                                             112.9
//   GLDS      LDS-DMA staging, swizzled lane-linear image, b128 fragments                         116-118
//   GLDS2     + all fragment reads of a slab before its first MFMA (= csrc/gemm_dma.hip)          122.8
//   GLDS3     GLDS2 on a 256 x 128 tile with 8 waves (2 workgroups / CU)                          118-120
//   GLDS4     GLDS2 with s_setprio(1) around the MFMA cluster                                     111.6
//   GLDS5     three LDS stages + a second fragment register set: a wave reads slab kt+1's fragments
//             under its own MFMAs of slab kt (155 VGPRs -> 3 waves / SIMD, 3 workgroups / CU)         114 (GLDS2 that day: 121)
//   GLDS6     GLDS2 persistent: 1024 resident workgroups walk the tiles, the next tile's first slab is
//             requested before the current tile's stores (the four workgroups of a CU stay in lock-step)  110 (GLDS2: 122)
// (MI355X, M = 131072, K = 512, N = 1024, random data; combine -DNO_LDS_WRITE -DNO_BARRIER -DNO_GLOBAL [-DNO_LDS_READ]
// for the "only LDS reads + MFMA" (133) and "only MFMA" (140) points.)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
using f32x16 = __attribute__((ext_vector_type(16))) float;
constexpr int kPad = 4, BK = 16, BM = 128, BN = 128, NT = 256, TM = 2, TN = 2, SA = BM + kPad, SB = BN + kPad;

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;
struct Args { const float *A, *B; float *C; int M, N, K; };

__device__ inline void xcd(int &tm, int &tn) {
    const unsigned nb = gridDim.x * gridDim.y, b = blockIdx.y * gridDim.x + blockIdx.x;
    const unsigned q = nb / 8, r = nb % 8, x = b % 8, i = b / 8;
    const unsigned t = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
    tn = t % gridDim.x; tm = t / gridDim.x;
}

#ifdef ROWL
constexpr int RS = 20;
#endif
__global__ __launch_bounds__(NT, 4) void k_nt(Args p) {
#ifdef ROWL
    __shared__ __attribute__((aligned(16))) float lds[2 * (BM + BN) * RS];
    constexpr int BUF = (BM + BN) * RS;
#else
    __shared__ __attribute__((aligned(16))) float lds[2 * BK * (SA + SB)];
    constexpr int BUF = BK * (SA + SB);
#endif
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    int tile_m, tile_n; xcd(tile_m, tile_n);
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    float4 ra[2], rb[2];
    f32x16 acc[TM][TN];
    for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
    const int nk = p.K / BK;
    auto load = [&](int k0) {
#ifndef NO_GLOBAL
        for (int f = 0; f < 2; ++f) {
            int idx = tid + f * NT, i = idx >> 2, kc = idx & 3;
            ra[f] = *reinterpret_cast<const float4 *>(p.A + (size_t)(m0 + i) * p.K + k0 + kc * 4);
            rb[f] = *reinterpret_cast<const float4 *>(p.B + (size_t)(n0 + i) * p.K + k0 + kc * 4);
        }
#else
        for (int f = 0; f < 2; ++f) { ra[f] = make_float4(k0, 1, 2, 3); rb[f] = make_float4(3, 2, 1, k0); }
#endif
    };
    auto store = [&](float *buf) {
#ifdef ROWL
        for (int f = 0; f < 2; ++f) {
            int idx = tid + f * NT, i = idx >> 2, kc = idx & 3;
            *reinterpret_cast<float4 *>(buf + i * RS + kc * 4) = ra[f];
            *reinterpret_cast<float4 *>(buf + BM * RS + i * RS + kc * 4) = rb[f];
        }
#elif !defined(NO_LDS_WRITE)
        for (int f = 0; f < 2; ++f) {
            int idx = tid + f * NT, i = idx >> 2, kc = idx & 3;
            float *q = buf + kc * 4 * SA + i;
            q[0] = ra[f].x; q[SA] = ra[f].y; q[2 * SA] = ra[f].z; q[3 * SA] = ra[f].w;
            float *r = buf + BK * SA + kc * 4 * SB + i;
            r[0] = rb[f].x; r[SB] = rb[f].y; r[2 * SB] = rb[f].z; r[3 * SB] = rb[f].w;
        }
#else
        if (ra[0].x == 12345.f) buf[tid] = ra[0].y + ra[1].z + rb[0].x + rb[1].w;
#endif
    };
    load(0); store(lds);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) load((kt + 1) * BK);
        const float *a_base = lds + cur * BUF + (lane >> 5) * SA + wm * 64 + (lane & 31);
        const float *b_base = lds + cur * BUF + BK * SA + (lane >> 5) * SB + wn * 64 + (lane & 31);
#ifdef SPLITW
        float a[2][TM], b[2][TN];
        for (int i = 0; i < TM; ++i) a[0][i] = a_base[i * 32];
        for (int j = 0; j < TN; ++j) b[0][j] = b_base[j * 32];
        float *nbuf = lds + (cur ^ 1) * BUF;
#pragma unroll
        for (int kp = 0; kp < BK / 2; ++kp) {
            const int c = kp & 1, n = c ^ 1;
            if (kp + 1 < BK / 2) {
                for (int i = 0; i < TM; ++i) a[n][i] = a_base[(kp + 1) * 2 * SA + i * 32];
                for (int j = 0; j < TN; ++j) b[n][j] = b_base[(kp + 1) * 2 * SB + j * 32];
            }
            if (kp >= 4 && kt + 1 < nk) {       // one quarter of the next slab's LDS image behind each of the last 4 MFMA groups
                const int part = kp - 4, f = part >> 1;
                int idx = tid + f * NT, i = idx >> 2, kc = idx & 3;
                if ((part & 1) == 0) {
                    float *q = nbuf + kc * 4 * SA + i;
                    q[0] = ra[f].x; q[SA] = ra[f].y; q[2 * SA] = ra[f].z; q[3 * SA] = ra[f].w;
                } else {
                    float *r = nbuf + BK * SA + kc * 4 * SB + i;
                    r[0] = rb[f].x; r[SB] = rb[f].y; r[2 * SB] = rb[f].z; r[3 * SB] = rb[f].w;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c][i], b[c][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
#elif defined(ROWL)
        {
            const float *ar = lds + cur * BUF + (wm * 64 + (lane & 31)) * RS + (lane >> 5) * 4;
            const float *br = lds + cur * BUF + BM * RS + (wn * 64 + (lane & 31)) * RS + (lane >> 5) * 4;
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                float4 a4[TM], b4[TN];
                for (int i = 0; i < TM; ++i) a4[i] = *reinterpret_cast<const float4 *>(ar + i * 32 * RS + g * 8);
                for (int j = 0; j < TN; ++j) b4[j] = *reinterpret_cast<const float4 *>(br + j * 32 * RS + g * 8);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) {
                        float av = e == 0 ? a4[i].x : e == 1 ? a4[i].y : e == 2 ? a4[i].z : a4[i].w;
                        float bv = e == 0 ? b4[j].x : e == 1 ? b4[j].y : e == 2 ? b4[j].z : b4[j].w;
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][j], 0, 0, 0);
                    }
            }
        }
#elif defined(PIPE)
        float a[2][TM], b[2][TN];
        for (int i = 0; i < TM; ++i) a[0][i] = a_base[i * 32];
        for (int j = 0; j < TN; ++j) b[0][j] = b_base[j * 32];
#pragma unroll
        for (int kp = 0; kp < BK / 2; ++kp) {
            const int c = kp & 1, n = c ^ 1;
            if (kp + 1 < BK / 2) {
                for (int i = 0; i < TM; ++i) a[n][i] = a_base[(kp + 1) * 2 * SA + i * 32];
                for (int j = 0; j < TN; ++j) b[n][j] = b_base[(kp + 1) * 2 * SB + j * 32];
            }
            __builtin_amdgcn_sched_barrier(0);
            for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c][i], b[c][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
#elif defined(NO_LDS_READ)
#pragma unroll
        for (int kp = 0; kp < BK / 2; ++kp) {
            float a[TM], b[TN];
            for (int i = 0; i < TM; ++i) a[i] = (float)(kp + i + lane);
            for (int j = 0; j < TN; ++j) b[j] = (float)(kp - j + lane);
            for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
#else
#pragma unroll
        for (int kp = 0; kp < BK / 2; ++kp) {
            float a[TM], b[TN];
            for (int i = 0; i < TM; ++i) a[i] = a_base[kp * 2 * SA + i * 32];
            for (int j = 0; j < TN; ++j) b[j] = b_base[kp * 2 * SB + j * 32];
            for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
#endif
#ifndef SPLITW
        if (kt + 1 < nk) store(lds + (cur ^ 1) * BUF);
#endif
#ifndef NO_BARRIER
        __syncthreads();
#endif
    }
    // plain epilogue (dword stores; same for every variant)
    for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int q = 0; q < 16; ++q) {
        int row = m0 + wm * 64 + i * 32 + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5), col = n0 + wn * 64 + j * 32 + (lane & 31);
        p.C[(size_t)row * p.N + col] = acc[i][j][q];
    }
}

#ifdef FULLLINE
// full 128-byte-line global loads (8 rows x 128 B per wave instruction) for 32 k at a time, LDS slabs stay 16 deep
__global__ __launch_bounds__(NT, 4) void k_nt2(Args p) {
    __shared__ __attribute__((aligned(16))) float lds[2 * BK * (SA + SB)];
    constexpr int BUF = BK * (SA + SB);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    int tile_m, tile_n; xcd(tile_m, tile_n);
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int half = lane >> 5, q = lane & 31;
    float4 ra[4], rb[4];
    f32x16 acc[TM][TN];
    for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int z = 0; z < 16; ++z) acc[i][j][z] = 0.f;
    const int nk = p.K / BK;
    auto load = [&](int k0) {        // k0 multiple of 32
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            int row = (f * 4 + wave) * 8 + (q >> 2), c8 = ((half ^ (f & 1)) << 2) | (q & 3);
            ra[f] = *reinterpret_cast<const float4 *>(p.A + (size_t)(m0 + row) * p.K + k0 + c8 * 4);
            rb[f] = *reinterpret_cast<const float4 *>(p.B + (size_t)(n0 + row) * p.K + k0 + c8 * 4);
        }
    };
    auto store = [&](float *buf, int s) {
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
            const bool first = half == s;
            float4 va = first ? ra[2 * pr] : ra[2 * pr + 1], vb = first ? rb[2 * pr] : rb[2 * pr + 1];
            int fsel = first ? 2 * pr : 2 * pr + 1;
            int row = (fsel * 4 + wave) * 8 + (q >> 2), kc = q & 3;
            float *qa = buf + kc * 4 * SA + row;
            qa[0] = va.x; qa[SA] = va.y; qa[2 * SA] = va.z; qa[3 * SA] = va.w;
            float *qb = buf + BK * SA + kc * 4 * SB + row;
            qb[0] = vb.x; qb[SB] = vb.y; qb[2 * SB] = vb.z; qb[3 * SB] = vb.w;
        }
    };
    load(0); store(lds, 0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if ((kt & 1) && kt + 1 < nk) load((kt + 1) * BK);      // odd iteration: fetch the next 32 k
        const float *a_base = lds + cur * BUF + (lane >> 5) * SA + wm * 64 + (lane & 31);
        const float *b_base = lds + cur * BUF + BK * SA + (lane >> 5) * SB + wn * 64 + (lane & 31);
#pragma unroll
        for (int kp = 0; kp < BK / 2; ++kp) {
            float a[TM], b[TN];
            for (int i = 0; i < TM; ++i) a[i] = a_base[kp * 2 * SA + i * 32];
            for (int j = 0; j < TN; ++j) b[j] = b_base[kp * 2 * SB + j * 32];
            for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) store(lds + (cur ^ 1) * BUF, (kt + 1) & 1);
        __syncthreads();
    }
    for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int z = 0; z < 16; ++z) {
        int row = m0 + wm * 64 + i * 32 + (z & 3) + 8 * (z >> 2) + 4 * (lane >> 5), col = n0 + wn * 64 + j * 32 + (lane & 31);
        p.C[(size_t)row * p.N + col] = acc[i][j][z];
    }
}
#define k_nt k_nt2
#endif
#ifdef FULLA
// full 128-byte-line global loads (8 rows x 128 B per wave instruction) for 32 k at a time, LDS slabs stay 16 deep
__global__ __launch_bounds__(NT, 4) void k_nt3(Args p) {
    __shared__ __attribute__((aligned(16))) float lds[2 * BK * (SA + SB)];
    constexpr int BUF = BK * (SA + SB);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    int tile_m, tile_n; xcd(tile_m, tile_n);
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int half = lane >> 5, q = lane & 31;
    float4 ra[4], rb[2];
    f32x16 acc[TM][TN];
    for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int z = 0; z < 16; ++z) acc[i][j][z] = 0.f;
    const int nk = p.K / BK;
    auto load = [&](int k0) {        // k0 multiple of 32
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            int row = (f * 4 + wave) * 8 + (q >> 2), c8 = ((half ^ (f & 1)) << 2) | (q & 3);
            ra[f] = *reinterpret_cast<const float4 *>(p.A + (size_t)(m0 + row) * p.K + k0 + c8 * 4);
        }
    };
    auto loadb = [&](int k0) {       // B: the fragment-shaped 16-deep loads of the baseline
        for (int f = 0; f < 2; ++f) {
            int idx = tid + f * NT, i = idx >> 2, kc = idx & 3;
            rb[f] = *reinterpret_cast<const float4 *>(p.B + (size_t)(n0 + i) * p.K + k0 + kc * 4);
        }
    };
    auto store = [&](float *buf, int s) {
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
            const bool first = half == s;
            float4 va = first ? ra[2 * pr] : ra[2 * pr + 1];
            int fsel = first ? 2 * pr : 2 * pr + 1;
            int row = (fsel * 4 + wave) * 8 + (q >> 2), kc = q & 3;
            float *qa = buf + kc * 4 * SA + row;
            qa[0] = va.x; qa[SA] = va.y; qa[2 * SA] = va.z; qa[3 * SA] = va.w;
        }
        for (int f = 0; f < 2; ++f) {
            int idx = tid + f * NT, i = idx >> 2, kc = idx & 3;
            float *r = buf + BK * SA + kc * 4 * SB + i;
            r[0] = rb[f].x; r[SB] = rb[f].y; r[2 * SB] = rb[f].z; r[3 * SB] = rb[f].w;
        }
    };
    load(0); loadb(0); store(lds, 0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if ((kt & 1) && kt + 1 < nk) load((kt + 1) * BK);      // odd iteration: fetch the next 32 k
        if (kt + 1 < nk) loadb((kt + 1) * BK);
        const float *a_base = lds + cur * BUF + (lane >> 5) * SA + wm * 64 + (lane & 31);
        const float *b_base = lds + cur * BUF + BK * SA + (lane >> 5) * SB + wn * 64 + (lane & 31);
#pragma unroll
        for (int kp = 0; kp < BK / 2; ++kp) {
            float a[TM], b[TN];
            for (int i = 0; i < TM; ++i) a[i] = a_base[kp * 2 * SA + i * 32];
            for (int j = 0; j < TN; ++j) b[j] = b_base[kp * 2 * SB + j * 32];
            for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) store(lds + (cur ^ 1) * BUF, (kt + 1) & 1);
        __syncthreads();
    }
    for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int z = 0; z < 16; ++z) {
        int row = m0 + wm * 64 + i * 32 + (z & 3) + 8 * (z >> 2) + 4 * (lane >> 5), col = n0 + wn * 64 + j * 32 + (lane & 31);
        p.C[(size_t)row * p.N + col] = acc[i][j][z];
    }
}
#define k_nt k_nt3
#endif

#ifdef GLDS
// LDS-DMA staging: global_load_lds_dwordx4 into a lane-linear [row][16] image, XOR-swizzled through the SOURCE address,
// fragments by ds_read_b128 with a k permutation (lane half h, group g reads k = 4*(2g+h) .. +3)
__global__ __launch_bounds__(NT, GLDS_MINW) void k_nt4(Args p) {
    __shared__ __attribute__((aligned(1024))) float lds[2 * 2 * 128 * 16];
    constexpr int STG = 2 * 128 * 16, OPB = 128 * 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    int tile_m, tile_n; xcd(tile_m, tile_n);
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    f32x16 acc[TM][TN];
    for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int z = 0; z < 16; ++z) acc[i][j][z] = 0.f;
    const int nk = p.K / BK;
    auto issue = [&](int kt, int stage) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int blk = wave * 2 + j, row = blk * 16 + (lane >> 2), c = (lane & 3) ^ ((row >> 2) & 3);
            const float *ga = p.A + (size_t)(m0 + row) * p.K + kt * BK + c * 4;
            const float *gb = p.B + (size_t)(n0 + row) * p.K + kt * BK + c * 4;
            __builtin_amdgcn_global_load_lds((glb_void *)ga, (lds_void *)(lds + stage * STG + blk * 256), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((glb_void *)gb, (lds_void *)(lds + stage * STG + OPB + blk * 256), 16, 0, 0);
        }
    };
    issue(0, 0);
    __syncthreads();
    const int half = lane >> 5, m = lane & 31, sw = (m >> 2) & 3;
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) issue(kt + 1, cur ^ 1);
        const float *ar = lds + cur * STG + (wm * 64 + m) * 16;
        const float *br = lds + cur * STG + OPB + (wn * 64 + m) * 16;
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const int off = (((2 * g + half) ^ sw) * 4);
            float4 a4[TM], b4[TN];
            for (int i = 0; i < TM; ++i) a4[i] = *reinterpret_cast<const float4 *>(ar + i * 32 * 16 + off);
            for (int j = 0; j < TN; ++j) b4[j] = *reinterpret_cast<const float4 *>(br + j * 32 * 16 + off);
#pragma unroll
            for (int e = 0; e < 4; ++e)
                for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) {
                    float av = e == 0 ? a4[i].x : e == 1 ? a4[i].y : e == 2 ? a4[i].z : a4[i].w;
                    float bv = e == 0 ? b4[j].x : e == 1 ? b4[j].y : e == 2 ? b4[j].z : b4[j].w;
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][j], 0, 0, 0);
                }
        }
        __syncthreads();
    }
    for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int z = 0; z < 16; ++z) {
        int row = m0 + wm * 64 + i * 32 + (z & 3) + 8 * (z >> 2) + 4 * (lane >> 5), col = n0 + wn * 64 + j * 32 + (lane & 31);
        p.C[(size_t)row * p.N + col] = acc[i][j][z];
    }
}
#define k_nt k_nt4
#endif
#ifdef GLDS2
// LDS-DMA staging: global_load_lds_dwordx4 into a lane-linear [row][16] image, XOR-swizzled through the SOURCE address,
// fragments by ds_read_b128 with a k permutation (lane half h, group g reads k = 4*(2g+h) .. +3)
__global__ __launch_bounds__(NT, 4) void k_nt5(Args p) {
    __shared__ __attribute__((aligned(1024))) float lds[2 * 2 * 128 * 16];
    constexpr int STG = 2 * 128 * 16, OPB = 128 * 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    int tile_m, tile_n; xcd(tile_m, tile_n);
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    f32x16 acc[TM][TN];
    for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int z = 0; z < 16; ++z) acc[i][j][z] = 0.f;
    const int nk = p.K / BK;
    auto issue = [&](int kt, int stage) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int blk = wave * 2 + j, row = blk * 16 + (lane >> 2), c = (lane & 3) ^ ((row >> 2) & 3);
            const float *ga = p.A + (size_t)(m0 + row) * p.K + kt * BK + c * 4;
            const float *gb = p.B + (size_t)(n0 + row) * p.K + kt * BK + c * 4;
            __builtin_amdgcn_global_load_lds((glb_void *)ga, (lds_void *)(lds + stage * STG + blk * 256), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((glb_void *)gb, (lds_void *)(lds + stage * STG + OPB + blk * 256), 16, 0, 0);
        }
    };
    issue(0, 0);
    __syncthreads();
    const int half = lane >> 5, m = lane & 31, sw = (m >> 2) & 3;
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) issue(kt + 1, cur ^ 1);
        const float *ar = lds + cur * STG + (wm * 64 + m) * 16;
        const float *br = lds + cur * STG + OPB + (wn * 64 + m) * 16;
        float4 a4[2][TM], b4[2][TN];
        {
            const int off = ((half ^ sw) * 4);
            for (int i = 0; i < TM; ++i) a4[0][i] = *reinterpret_cast<const float4 *>(ar + i * 32 * 16 + off);
            for (int j = 0; j < TN; ++j) b4[0][j] = *reinterpret_cast<const float4 *>(br + j * 32 * 16 + off);
        }
        {
            const int off = (((2 + half) ^ sw) * 4);
            for (int i = 0; i < TM; ++i) a4[1][i] = *reinterpret_cast<const float4 *>(ar + i * 32 * 16 + off);
            for (int j = 0; j < TN; ++j) b4[1][j] = *reinterpret_cast<const float4 *>(br + j * 32 * 16 + off);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < 2; ++g) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) {
                    float av = e == 0 ? a4[g][i].x : e == 1 ? a4[g][i].y : e == 2 ? a4[g][i].z : a4[g][i].w;
                    float bv = e == 0 ? b4[g][j].x : e == 1 ? b4[g][j].y : e == 2 ? b4[g][j].z : b4[g][j].w;
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][j], 0, 0, 0);
                }
        }
        __syncthreads();
    }
    for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int z = 0; z < 16; ++z) {
        int row = m0 + wm * 64 + i * 32 + (z & 3) + 8 * (z >> 2) + 4 * (lane >> 5), col = n0 + wn * 64 + j * 32 + (lane & 31);
        p.C[(size_t)row * p.N + col] = acc[i][j][z];
    }
}
#define k_nt k_nt5
#endif
#ifdef GLDS4
// LDS-DMA staging: global_load_lds_dwordx4 into a lane-linear [row][16] image, XOR-swizzled through the SOURCE address,
// fragments by ds_read_b128 with a k permutation (lane half h, group g reads k = 4*(2g+h) .. +3)
__global__ __launch_bounds__(NT, 4) void k_nt7(Args p) {
    __shared__ __attribute__((aligned(1024))) float lds[2 * 2 * 128 * 16];
    constexpr int STG = 2 * 128 * 16, OPB = 128 * 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    int tile_m, tile_n; xcd(tile_m, tile_n);
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    f32x16 acc[TM][TN];
    for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int z = 0; z < 16; ++z) acc[i][j][z] = 0.f;
    const int nk = p.K / BK;
    auto issue = [&](int kt, int stage) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int blk = wave * 2 + j, row = blk * 16 + (lane >> 2), c = (lane & 3) ^ ((row >> 2) & 3);
            const float *ga = p.A + (size_t)(m0 + row) * p.K + kt * BK + c * 4;
            const float *gb = p.B + (size_t)(n0 + row) * p.K + kt * BK + c * 4;
            __builtin_amdgcn_global_load_lds((glb_void *)ga, (lds_void *)(lds + stage * STG + blk * 256), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((glb_void *)gb, (lds_void *)(lds + stage * STG + OPB + blk * 256), 16, 0, 0);
        }
    };
    issue(0, 0);
    __syncthreads();
    const int half = lane >> 5, m = lane & 31, sw = (m >> 2) & 3;
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) issue(kt + 1, cur ^ 1);
        const float *ar = lds + cur * STG + (wm * 64 + m) * 16;
        const float *br = lds + cur * STG + OPB + (wn * 64 + m) * 16;
        float4 a4[2][TM], b4[2][TN];
        {
            const int off = ((half ^ sw) * 4);
            for (int i = 0; i < TM; ++i) a4[0][i] = *reinterpret_cast<const float4 *>(ar + i * 32 * 16 + off);
            for (int j = 0; j < TN; ++j) b4[0][j] = *reinterpret_cast<const float4 *>(br + j * 32 * 16 + off);
        }
        {
            const int off = (((2 + half) ^ sw) * 4);
            for (int i = 0; i < TM; ++i) a4[1][i] = *reinterpret_cast<const float4 *>(ar + i * 32 * 16 + off);
            for (int j = 0; j < TN; ++j) b4[1][j] = *reinterpret_cast<const float4 *>(br + j * 32 * 16 + off);
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int g = 0; g < 2; ++g) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) {
                    float av = e == 0 ? a4[g][i].x : e == 1 ? a4[g][i].y : e == 2 ? a4[g][i].z : a4[g][i].w;
                    float bv = e == 0 ? b4[g][j].x : e == 1 ? b4[g][j].y : e == 2 ? b4[g][j].z : b4[g][j].w;
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][j], 0, 0, 0);
                }
        }
        __builtin_amdgcn_s_setprio(0);
        __syncthreads();
    }
    for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int z = 0; z < 16; ++z) {
        int row = m0 + wm * 64 + i * 32 + (z & 3) + 8 * (z >> 2) + 4 * (lane >> 5), col = n0 + wn * 64 + j * 32 + (lane & 31);
        p.C[(size_t)row * p.N + col] = acc[i][j][z];
    }
}
#define k_nt k_nt7
#endif
#ifdef GLDS5
// GLDS2 restructured so that a wave overlaps its own LDS fragment reads with its own MFMAs: three LDS stages, the
// fragments of slab kt+1 are read into a second register set right after the barrier and BEFORE the MFMAs of slab kt
// are issued; ~150 VGPRs -> 3 waves / SIMD, 48 KB LDS -> 3 workgroups / CU.
#ifndef GLDS5_MINW
#define GLDS5_MINW 3
#endif
struct Frag { float4 a[2][TM], b[2][TN]; };
__global__ __launch_bounds__(NT, GLDS5_MINW) void k_nt8(Args p) {
    __shared__ __attribute__((aligned(1024))) float lds[3 * 2 * 128 * 16];
    constexpr int STG = 2 * 128 * 16, OPB = 128 * 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    int tile_m, tile_n; xcd(tile_m, tile_n);
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    f32x16 acc[TM][TN];
    for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int z = 0; z < 16; ++z) acc[i][j][z] = 0.f;
    const int nk = p.K / BK;
    const float *ga[2], *gb[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int blk = wave * 2 + j, row = blk * 16 + (lane >> 2), c = (lane & 3) ^ ((row >> 2) & 3);
        ga[j] = p.A + (size_t)(m0 + row) * p.K + c * 4;
        gb[j] = p.B + (size_t)(n0 + row) * p.K + c * 4;
    }
    auto issue = [&](int kt) {
        float *st = lds + (kt % 3) * STG;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int blk = wave * 2 + j;
            __builtin_amdgcn_global_load_lds((glb_void *)(ga[j] + kt * BK), (lds_void *)(st + blk * 256), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((glb_void *)(gb[j] + kt * BK), (lds_void *)(st + OPB + blk * 256), 16, 0, 0);
        }
    };
    const int half = lane >> 5, m = lane & 31, sw = (m >> 2) & 3;
    const int off0 = ((half ^ sw) * 4), off1 = (((2 + half) ^ sw) * 4);
    auto read = [&](int kt, Frag &f) {
        const float *st = lds + (kt % 3) * STG;
        const float *ar = st + (wm * 64 + m) * 16, *br = st + OPB + (wn * 64 + m) * 16;
#pragma unroll
        for (int i = 0; i < TM; ++i) { f.a[0][i] = *reinterpret_cast<const float4 *>(ar + i * 32 * 16 + off0); f.a[1][i] = *reinterpret_cast<const float4 *>(ar + i * 32 * 16 + off1); }
#pragma unroll
        for (int j = 0; j < TN; ++j) { f.b[0][j] = *reinterpret_cast<const float4 *>(br + j * 32 * 16 + off0); f.b[1][j] = *reinterpret_cast<const float4 *>(br + j * 32 * 16 + off1); }
    };
    auto mma = [&](const Frag &f) {
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        float av = e == 0 ? f.a[g][i].x : e == 1 ? f.a[g][i].y : e == 2 ? f.a[g][i].z : f.a[g][i].w;
                        float bv = e == 0 ? f.b[g][j].x : e == 1 ? f.b[g][j].y : e == 2 ? f.b[g][j].z : f.b[g][j].w;
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][j], 0, 0, 0);
                    }
    };
    Frag f0, f1;
    issue(0);
    if (nk > 1) issue(1);
    if (nk > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    read(0, f0);
    for (int kt = 0; kt < nk; kt += 2) {
        // slab kt + 1 was issued either in the prologue (kt == 0) or by step(kt - 1)
        {
            const int k = kt;
            if (k + 2 < nk) { issue(k + 2); asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory"); }
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (k + 1 < nk) read(k + 1, f1);
            __builtin_amdgcn_sched_barrier(0);
            mma(f0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (kt + 1 < nk) {
            const int k = kt + 1;
            if (k + 2 < nk) { issue(k + 2); asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory"); }
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (k + 1 < nk) read(k + 1, f0);
            __builtin_amdgcn_sched_barrier(0);
            mma(f1);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int z = 0; z < 16; ++z) {
        int row = m0 + wm * 64 + i * 32 + (z & 3) + 8 * (z >> 2) + 4 * (lane >> 5), col = n0 + wn * 64 + j * 32 + (lane & 31);
        p.C[(size_t)row * p.N + col] = acc[i][j][z];
    }
}
#define k_nt k_nt8
#endif
#ifdef GLDS6
// GLDS2 as a persistent kernel: 1024 resident workgroups walk the tiles (same XCD-aware order, stride = grid size) and
// request the next tile's first slab BEFORE the epilogue of the current one, so the prologue latency of every tile but
// the first runs under the previous tile's stores.
__global__ __launch_bounds__(NT, 4) void k_nt9(Args p) {
    __shared__ __attribute__((aligned(1024))) float lds[2 * 2 * 128 * 16];
    constexpr int STG = 2 * 128 * 16, OPB = 128 * 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const unsigned tiles_n = p.N / BN, ntiles = tiles_n * (p.M / BM);
    const unsigned nb = gridDim.x;
    const int nk = p.K / BK;
    const int half = lane >> 5, m = lane & 31, sw = (m >> 2) & 3;
    auto tile_of = [&](unsigned t, int &m0, int &n0) {       // XCD-aware order over ALL tiles, as xcd() does per launch
        const unsigned q = ntiles / 8, r = ntiles % 8, x = t % 8, i = t / 8;
        const unsigned u = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
        n0 = (u % tiles_n) * BN; m0 = (u / tiles_n) * BM;
    };
    auto issue = [&](int m0, int n0, int kt, int stage) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int blk = wave * 2 + j, row = blk * 16 + (lane >> 2), c = (lane & 3) ^ ((row >> 2) & 3);
            const float *ga = p.A + (size_t)(m0 + row) * p.K + kt * BK + c * 4;
            const float *gb = p.B + (size_t)(n0 + row) * p.K + kt * BK + c * 4;
            __builtin_amdgcn_global_load_lds((glb_void *)ga, (lds_void *)(lds + stage * STG + blk * 256), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((glb_void *)gb, (lds_void *)(lds + stage * STG + OPB + blk * 256), 16, 0, 0);
        }
    };
    unsigned t = blockIdx.x;
    int m0, n0;
    if (t >= ntiles) return;
    tile_of(t, m0, n0);
    issue(m0, n0, 0, 0);
    for (;;) {
        f32x16 acc[TM][TN];
        for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int z = 0; z < 16; ++z) acc[i][j][z] = 0.f;
        __syncthreads();                                     // slab 0 of this tile has landed (vmcnt(0) + barrier)
        for (int kt = 0; kt < nk; ++kt) {
            const int cur = kt & 1;
            if (kt + 1 < nk) issue(m0, n0, kt + 1, cur ^ 1);
            const float *ar = lds + cur * STG + (wm * 64 + m) * 16;
            const float *br = lds + cur * STG + OPB + (wn * 64 + m) * 16;
            float4 a4[2][TM], b4[2][TN];
            {
                const int off = ((half ^ sw) * 4);
                for (int i = 0; i < TM; ++i) a4[0][i] = *reinterpret_cast<const float4 *>(ar + i * 32 * 16 + off);
                for (int j = 0; j < TN; ++j) b4[0][j] = *reinterpret_cast<const float4 *>(br + j * 32 * 16 + off);
            }
            {
                const int off = (((2 + half) ^ sw) * 4);
                for (int i = 0; i < TM; ++i) a4[1][i] = *reinterpret_cast<const float4 *>(ar + i * 32 * 16 + off);
                for (int j = 0; j < TN; ++j) b4[1][j] = *reinterpret_cast<const float4 *>(br + j * 32 * 16 + off);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g = 0; g < 2; ++g) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) {
                        float av = e == 0 ? a4[g][i].x : e == 1 ? a4[g][i].y : e == 2 ? a4[g][i].z : a4[g][i].w;
                        float bv = e == 0 ? b4[g][j].x : e == 1 ? b4[g][j].y : e == 2 ? b4[g][j].z : b4[g][j].w;
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][j], 0, 0, 0);
                    }
            }
            if (kt + 1 < nk) __syncthreads();
        }
        // next tile: its first slab goes to stage 0 -- nk is even, so the last slab of this tile was read from stage 1 and
        // stage 0 was last read one barrier ago
        const int pm0 = m0, pn0 = n0;
        t += nb;
        const bool more = t < ntiles;
        if (more) { tile_of(t, m0, n0); issue(m0, n0, 0, 0); }
        for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int z = 0; z < 16; ++z) {
            int row = pm0 + wm * 64 + i * 32 + (z & 3) + 8 * (z >> 2) + 4 * (lane >> 5), col = pn0 + wn * 64 + j * 32 + (lane & 31);
            p.C[(size_t)row * p.N + col] = acc[i][j][z];
        }
        if (!more) break;
    }
}
#define k_nt k_nt9
#define PERSISTENT_GRID 1024
#endif
#ifdef GLDS3
// LDS-DMA staging: global_load_lds_dwordx4 into a lane-linear [row][16] image, XOR-swizzled through the SOURCE address,
// fragments by ds_read_b128 with a k permutation (lane half h, group g reads k = 4*(2g+h) .. +3)
__global__ __launch_bounds__(512, 2) void k_nt6(Args p) {
    __shared__ __attribute__((aligned(1024))) float lds[2 * 3 * 128 * 16];
    constexpr int STG = 3 * 128 * 16, OPB = 256 * 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;   // 8 waves: wm 0..3
    int tile_m, tile_n; xcd(tile_m, tile_n);
    const int m0 = tile_m * 256, n0 = tile_n * BN;
    f32x16 acc[TM][TN];
    for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int z = 0; z < 16; ++z) acc[i][j][z] = 0.f;
    const int nk = p.K / BK;
    auto issue = [&](int kt, int stage) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int blk = wave * 2 + j, row = blk * 16 + (lane >> 2), c = (lane & 3) ^ ((row >> 2) & 3);
            const float *ga = p.A + (size_t)(m0 + row) * p.K + kt * BK + c * 4;
            __builtin_amdgcn_global_load_lds((glb_void *)ga, (lds_void *)(lds + stage * STG + blk * 256), 16, 0, 0);
        }
        {
            const int blk = wave, row = blk * 16 + (lane >> 2), c = (lane & 3) ^ ((row >> 2) & 3);
            const float *gb = p.B + (size_t)(n0 + row) * p.K + kt * BK + c * 4;
            __builtin_amdgcn_global_load_lds((glb_void *)gb, (lds_void *)(lds + stage * STG + OPB + blk * 256), 16, 0, 0);
        }
    };
    issue(0, 0);
    __syncthreads();
    const int half = lane >> 5, m = lane & 31, sw = (m >> 2) & 3;
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) issue(kt + 1, cur ^ 1);
        const float *ar = lds + cur * STG + (wm * 64 + m) * 16;
        const float *br = lds + cur * STG + OPB + (wn * 64 + m) * 16;
        float4 a4[2][TM], b4[2][TN];
        {
            const int off = ((half ^ sw) * 4);
            for (int i = 0; i < TM; ++i) a4[0][i] = *reinterpret_cast<const float4 *>(ar + i * 32 * 16 + off);
            for (int j = 0; j < TN; ++j) b4[0][j] = *reinterpret_cast<const float4 *>(br + j * 32 * 16 + off);
        }
        {
            const int off = (((2 + half) ^ sw) * 4);
            for (int i = 0; i < TM; ++i) a4[1][i] = *reinterpret_cast<const float4 *>(ar + i * 32 * 16 + off);
            for (int j = 0; j < TN; ++j) b4[1][j] = *reinterpret_cast<const float4 *>(br + j * 32 * 16 + off);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < 2; ++g) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) {
                    float av = e == 0 ? a4[g][i].x : e == 1 ? a4[g][i].y : e == 2 ? a4[g][i].z : a4[g][i].w;
                    float bv = e == 0 ? b4[g][j].x : e == 1 ? b4[g][j].y : e == 2 ? b4[g][j].z : b4[g][j].w;
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][j], 0, 0, 0);
                }
        }
        __syncthreads();
    }
    for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int z = 0; z < 16; ++z) {
        int row = m0 + wm * 64 + i * 32 + (z & 3) + 8 * (z >> 2) + 4 * (lane >> 5), col = n0 + wn * 64 + j * 32 + (lane & 31);
        p.C[(size_t)row * p.N + col] = acc[i][j][z];
    }
}
#define k_nt k_nt6
#define WG_THREADS 512
#define TILE_M 256
#endif

int main() {
    const int M = 131072, K = 512, N = 1024;
    float *A, *B, *C;
    hipMalloc(&A, (size_t)M * K * 4); hipMalloc(&B, (size_t)N * K * 4); hipMalloc(&C, (size_t)M * N * 4);
    std::vector<float> h((size_t)M * K);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)rand() / RAND_MAX - 0.5f;
    hipMemcpy(A, h.data(), (size_t)M * K * 4, hipMemcpyHostToDevice);
    hipMemcpy(B, h.data(), (size_t)N * K * 4, hipMemcpyHostToDevice);
    Args a{A, B, C, M, N, K};
#ifndef TILE_M
#define TILE_M BM
#define WG_THREADS NT
#endif
#ifdef PERSISTENT_GRID
    dim3 grid(PERSISTENT_GRID, 1);
#else
    dim3 grid(N / BN, M / TILE_M);
#endif
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k_nt, grid, dim3(WG_THREADS), 0, 0, a);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    const int reps = 20;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k_nt, grid, dim3(WG_THREADS), 0, 0, a);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double us = ms * 1e3 / reps;
    std::vector<float> hc(4096);
    hipMemcpy(hc.data(), C + 12345 * (size_t)N, 4096 * 4, hipMemcpyDeviceToHost);
    double cs = 0; for (float v : hc) cs += v;
    printf("%-28s %8.1f us  %6.1f TF  checksum %.6f\n", VARIANT, us, 2.0 * M * K * N / us / 1e6, cs);
    return 0;
}
