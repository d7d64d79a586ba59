// scratch: synthetic aggressors (one instruction family each) to run beside a victim kernel
#include <hip/hip_runtime.h>
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
extern "C" {
__global__ __launch_bounds__(256, 2) void aggr_mfma_f16_32x32x16(float *out, int iters) {
    f16x8 a, b; for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
    f32x16 acc[4]; for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[j], 0, 0, 0);
    float s = 0; for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) s += acc[j][i];
    if (s == 12345.f) out[0] = s;
}
__global__ __launch_bounds__(256, 2) void aggr_mfma_bf16_32x32x16(float *out, int iters) {
    bf16x8 a, b; for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(i * 0.5f); }
    f32x16 acc[4]; for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);
    float s = 0; for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) s += acc[j][i];
    if (s == 12345.f) out[0] = s;
}
__global__ __launch_bounds__(256, 2) void aggr_mfma_f32_32x32x2(float *out, int iters) {
    float a = threadIdx.x * 0.001f, b = 0.5f;
    f32x16 acc[4]; for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
    float s = 0; for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) s += acc[j][i];
    if (s == 12345.f) out[0] = s;
}
__global__ __launch_bounds__(256, 2) void aggr_mfma_f16_16x16x32(float *out, int iters) {
    f16x8 a, b; for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
    f32x4 acc[8]; for (int j = 0; j < 8; ++j) for (int i = 0; i < 4; ++i) acc[j][i] = 0.f;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[j], 0, 0, 0);
    float s = 0; for (int j = 0; j < 8; ++j) for (int i = 0; i < 4; ++i) s += acc[j][i];
    if (s == 12345.f) out[0] = s;
}
// many VALU compares into SGPR pairs + selects
__global__ __launch_bounds__(256, 2) void aggr_cmp(float *out, int iters) {
    float v[16]; for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 0.37f + i;
    float s = 0;
    for (int it = 0; it < iters * 8; ++it)
#pragma unroll
        for (int i = 0; i < 16; ++i) { s += (v[i] > (float)(it & 63)) ? v[i] : 0.5f; v[i] = v[i] * 1.0001f; }
    if (s == 12345.f) out[0] = s;
}
// big LDS footprint + LDS traffic, 200+ VGPRs worth of state is not needed: LDS only
__global__ __launch_bounds__(256, 2) void aggr_lds(float *out, int iters) {
    __shared__ float4 buf[2048];
    for (int i = threadIdx.x; i < 2048; i += 256) buf[i] = make_float4(i, i, i, i);
    __syncthreads();
    float4 s = make_float4(0, 0, 0, 0);
    for (int it = 0; it < iters * 4; ++it) { const float4 v = buf[(threadIdx.x * 7 + it * 13) & 2047]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
    if (s.x == 12345.f) out[0] = s.x;
}
// r06 (coresidency_hunt.py): integer checksum of a device block, added into *out (64-bit integer adds: order-free, deterministic)
__global__ __launch_bounds__(256) void aggr_checksum_kernel(const unsigned *p, long long nwords, unsigned long long *out) {
    unsigned long long s = 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nwords; i += (long long)gridDim.x * 256)
        s += (unsigned long long)p[i] * (unsigned long long)((i & 1023) + 1);
    for (int d = 32; d > 0; d >>= 1) s += __shfl_down(s, d, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(out, s);
}
int aggr_checksum(const void *p, long long nbytes, void *out, void *stream) {
    const long long nwords = nbytes / 4;
    long long blocks = (nwords + 256 * 16 - 1) / (256 * 16);
    if (blocks < 1) blocks = 1;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(aggr_checksum_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const unsigned *)p, nwords,
                       (unsigned long long *)out);
    return (int)hipGetLastError();
}
int aggr_launch(int which, float *out, int blocks, int iters, void *stream) {
    hipStream_t s = (hipStream_t)stream;
    if (which == 0) hipLaunchKernelGGL(aggr_mfma_f16_32x32x16, dim3(blocks), dim3(256), 0, s, out, iters);
    else if (which == 1) hipLaunchKernelGGL(aggr_mfma_bf16_32x32x16, dim3(blocks), dim3(256), 0, s, out, iters);
    else if (which == 2) hipLaunchKernelGGL(aggr_mfma_f32_32x32x2, dim3(blocks), dim3(256), 0, s, out, iters);
    else if (which == 3) hipLaunchKernelGGL(aggr_mfma_f16_16x16x32, dim3(blocks), dim3(256), 0, s, out, iters);
    else if (which == 4) hipLaunchKernelGGL(aggr_cmp, dim3(blocks), dim3(256), 0, s, out, iters);
    else hipLaunchKernelGGL(aggr_lds, dim3(blocks), dim3(256), 0, s, out, iters);
    return (int)hipGetLastError();
}
}
