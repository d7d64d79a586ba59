// Shared host/device helpers of libt2h_hip.so (gfx950 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/t2h.h"

#define T2H_API extern "C" __attribute__((visibility("default")))

namespace t2h {

constexpr int kWave = 64;

// ---- error reporting (thread local, never throws) -------------------------------------------
char *err_buf();
int fail(int code, const char *fmt, ...);
int check_launch(const char *what);
// name of the device kernel an entry point just launched (its main kernel, not helper reductions): thread local,
// read back through t2h_last_kernel_name() by profilers that aggregate per kernel symbol like rocprofv3 does
void note_kernel(const char *name);
const char *noted_kernel();

inline hipStream_t as_stream(t2h_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// (B, N) of the point-side entry points: B tiles of N rows each -- or, N < 0, a RAGGED batch of B tiles with -N rows in all
// (t2h_tile_build_ragged: tile boundaries are in off0 / the cell codes, and a row's tile index is kept as int bits in the LAST
// float of its pts row).  The reference trains with batch = 1 because N varies per tile (tomosar2height.yaml:40); the tiles of
// its 64-tile accumulation window are independent, so they can share launches.
inline int64_t rows_of(int B, int N) { return N >= 0 ? (int64_t)B * N : -(int64_t)N; }
inline int rows_per_tile(int B, int N) { return N >= 0 ? N : (int)(-(int64_t)N / (B > 0 ? B : 1)); }

// ---- Morton helpers: x in even bits, y in odd bits (up to 16 bits per coordinate) --------------
__host__ __device__ inline uint32_t part1by1(uint32_t v) {
    v &= 0x0000ffffu;
    v = (v | (v << 8)) & 0x00ff00ffu;
    v = (v | (v << 4)) & 0x0f0f0f0fu;
    v = (v | (v << 2)) & 0x33333333u;
    v = (v | (v << 1)) & 0x55555555u;
    return v;
}
__host__ __device__ inline uint32_t compact1by1(uint32_t v) {
    v &= 0x55555555u;
    v = (v | (v >> 1)) & 0x33333333u;
    v = (v | (v >> 2)) & 0x0f0f0f0fu;
    v = (v | (v >> 4)) & 0x00ff00ffu;
    v = (v | (v >> 8)) & 0x0000ffffu;
    return v;
}
__host__ __device__ inline uint32_t morton2(uint32_t x, uint32_t y) { return part1by1(x) | (part1by1(y) << 1); }

// ---- channel vector access: VEC = 4 (16-byte rows, C % 4 == 0) or 1 (any C) -----------------------
template <int VEC> struct Vec;
template <> struct Vec<4> {
    float v[4];
    __device__ static Vec load(const float *p) {
        float4 t = *reinterpret_cast<const float4 *>(p);
        Vec r; r.v[0] = t.x; r.v[1] = t.y; r.v[2] = t.z; r.v[3] = t.w; return r;
    }
    __device__ void store(float *p) const { *reinterpret_cast<float4 *>(p) = make_float4(v[0], v[1], v[2], v[3]); }
};
template <> struct Vec<1> {
    float v[1];
    __device__ static Vec load(const float *p) { Vec r; r.v[0] = *p; return r; }
    __device__ void store(float *p) const { *p = v[0]; }
};

// lanes per row group for C channels at VEC channels per lane: smallest power of two >= C/VEC, capped at 64
inline int group_log2(int C, int vec) {
    int lanes = (C + vec - 1) / vec;
    int lg = 0;
    while ((1 << lg) < lanes && lg < 6) ++lg;
    return lg;
}

// grid_sample's unnormalise + border clip, align_corners=True (ATen grid_sampler_2d), on the
// reference's vgrid = 2*xy - 1 (alto.py:94).  Kept as separate roundings (no contraction).
__device__ inline float unnormalize_clip(float x01, int size) {
    float g = __fsub_rn(__fmul_rn(2.0f, x01), 1.0f);
    // ATen divides by 2: multiplying by 0.5f rounds the same real number, so it is bit-identical -- and an order of
    // magnitude cheaper than the IEEE division sequence (this runs per visited row in the sample backward's gather)
    float ix = __fmul_rn(__fmul_rn(__fadd_rn(g, 1.0f), 0.5f), (float)(size - 1));
    ix = fminf(fmaxf(ix, 0.0f), (float)(size - 1));
    return ix;
}

}  // namespace t2h
