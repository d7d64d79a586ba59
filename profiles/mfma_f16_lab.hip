// Lab for the fp16 two-way split (round 4): (1) what the chip sustains on v_mfma_f32_32x32x16_f16 against ..._bf16 with random operands
// (same loop, same data bits reinterpreted -- DVFS reacts to toggling, MI355X_MICROARCH.md), (2) whether the f16 MFMA flushes
// subnormal INPUTS (the low part of a split element can be subnormal).
//   hipcc --offload-arch=gfx950 -O3 -o profiles/mfma_f16_lab profiles/mfma_f16_lab.hip && ./profiles/mfma_f16_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <bool F16>
__global__ __launch_bounds__(256, 2) void loop_kernel(const uint4 *src, float *out, int iters) {
    const int lane = threadIdx.x & 63;
    uint4 a[4], b[2];
    for (int i = 0; i < 4; ++i) a[i] = src[(lane + 64 * i) & 1023];
    for (int i = 0; i < 2; ++i) b[i] = src[(lane + 64 * (4 + i)) & 1023];
    f32x16 acc[8];
    for (int t = 0; t < 8; ++t) for (int z = 0; z < 16; ++z) acc[t][z] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            if (F16) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<f16x8 *>(&a[t & 3]), *reinterpret_cast<f16x8 *>(&b[t >> 2]), acc[t], 0, 0, 0);
            else acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<bf16x8 *>(&a[t & 3]), *reinterpret_cast<bf16x8 *>(&b[t >> 2]), acc[t], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int t = 0; t < 8; ++t) for (int z = 0; z < 16; ++z) s += acc[t][z];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ void denorm_kernel(float *out, unsigned short abits, unsigned short bbits) {
    typedef __attribute__((ext_vector_type(8))) unsigned short u16x8;
    u16x8 av, bv;
    for (int i = 0; i < 8; ++i) { av[i] = abits; bv[i] = bbits; }
    f32x16 acc;
    for (int z = 0; z < 16; ++z) acc[z] = 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<f16x8 *>(&av), *reinterpret_cast<f16x8 *>(&bv), acc, 0, 0, 0);
    if (threadIdx.x == 0) out[0] = acc[0];
}

int main() {
    const int nblk = 512, iters = 20000;
    std::vector<uint32_t> h(4096);
    // random f16 / bf16 bit patterns with moderate exponents (no inf / nan): sign random, exponent field mid-range, mantissa random
    for (auto &w : h) {
        auto one = [&]() { uint32_t m = rand() & 0x3ff, e = 12 + rand() % 6, s = rand() & 1; return (s << 15) | (e << 10) | m; };
        w = one() | (one() << 16);
    }
    uint4 *src; float *out;
    hipMalloc(&src, 4096 * 4); hipMalloc(&out, nblk * 256 * 4 + 64);
    hipMemcpy(src, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep)
        for (int f = 0; f < 2; ++f) {
            hipEventRecord(e0);
            if (f) hipLaunchKernelGGL(loop_kernel<true>, dim3(nblk), dim3(256), 0, 0, src, out, iters);
            else hipLaunchKernelGGL(loop_kernel<false>, dim3(nblk), dim3(256), 0, 0, src, out, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double tf = (double)nblk * 4 * iters * 8 * 2.0 * 32 * 32 * 16 / (ms * 1e-3) / 1e12;
            printf("rep %d  %s  %.2f ms  %.1f TF\n", rep, f ? "f16 " : "bf16", ms, tf);
        }
    // subnormal inputs: a = 2^-20 (f16 subnormal 0x0010), b = 2^10 (0x6400): sum of 16 products = 2^-6; a = smallest normal 2^-14 (0x0400)
    float r;
    hipLaunchKernelGGL(denorm_kernel, dim3(1), dim3(64), 0, 0, out, (unsigned short)0x0010, (unsigned short)0x6400);
    hipMemcpy(&r, out, 4, hipMemcpyDeviceToHost);
    printf("subnormal a = 2^-20, b = 2^10, K = 16: got %.9g, exact %.9g  -> inputs %s\n", r, 16 * 9.5367431640625e-07 * 1024.0, r == 0.f ? "FLUSHED" : "kept");
    hipLaunchKernelGGL(denorm_kernel, dim3(1), dim3(64), 0, 0, out, (unsigned short)0x0400, (unsigned short)0x6400);
    hipMemcpy(&r, out, 4, hipMemcpyDeviceToHost);
    printf("normal    a = 2^-14, b = 2^10, K = 16: got %.9g, exact %.9g\n", r, 16 * 6.103515625e-05 * 1024.0);
    return 0;
}
