// Does the f32 MFMA shape change the throughput the chip sustains (MI355X_MICROARCH.md, DVFS item 7: for bf16 the 16x16
// shape delivers ~1.15x the FLOP/s of the 32x32 shape at equal cycles per FLOP, because the clock held under load differs)?
// Bare loops on random operands held in registers, one / two waves per SIMD, every CU busy.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_shape_lab profiles/mfma_shape_lab.hip && /tmp/mfma_shape_lab
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

// 4 accumulators of 32x32 (64 regs): per iteration 4 MFMAs x 4096 FLOP... = 32x32x2x2 flop each
__global__ __launch_bounds__(256) void k32(const float *in, float *out, int iters) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    float a0 = in[t], a1 = in[t + 1], b0 = in[t + 2], b1 = in[t + 3];
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int q = 0; q < 16; ++q) acc[i][q] = 0.f;
    for (int it = 0; it < iters; ++it) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[3], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int q = 0; q < 16; ++q) s += acc[i][q];
    out[t] = s;
}

// the same 64 x 64 output block per wave as 16 accumulators of 16x16 (64 regs): per iteration 16 MFMAs (k = 4)
__global__ __launch_bounds__(256) void k16(const float *in, float *out, int iters) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    float a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = in[t + i]; b[i] = in[t + 4 + i]; }
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) for (int q = 0; q < 4; ++q) acc[i][q] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i * 4 + j], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) for (int q = 0; q < 4; ++q) s += acc[i][q];
    out[t] = s;
}

int main() {
    const int blocks = 256 * 2, iters = 20000;              // 2 workgroups per CU = 2 waves per SIMD
    float *in, *out;
    hipMalloc(&in, (blocks * 256 + 16) * sizeof(float));
    hipMalloc(&out, blocks * 256 * sizeof(float));
    std::vector<float> h(blocks * 256 + 16);
    for (auto &v : h) v = (float)rand() / RAND_MAX - 0.5f;
    hipMemcpy(in, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wgs = 1; wgs <= 2; ++wgs) {
        const int nb = 256 * wgs;
        for (int rep = 0; rep < 3; ++rep) {
            float ms;
            // 32x32x2: per wave per iteration 4 MFMAs x (32*32*2*2) flop
            hipEventRecord(e0); for (int r = 0; r < 20; ++r) hipLaunchKernelGGL(k32, dim3(nb), dim3(256), 0, 0, in, out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
            double f32 = 20.0 * nb * 4 * iters * 4.0 * 4096.0 / (ms * 1e-3) / 1e12;
            // 16x16x4: per wave per iteration 16 MFMAs x (16*16*4*2) flop
            hipEventRecord(e0); for (int r = 0; r < 20; ++r) hipLaunchKernelGGL(k16, dim3(nb), dim3(256), 0, 0, in, out, iters / 2); hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
            double f16 = 20.0 * nb * 4 * (iters / 2) * 16.0 * 2048.0 / (ms * 1e-3) / 1e12;
            printf("%d workgroup(s)/CU  rep %d:  32x32x2 %.1f TF   16x16x4 %.1f TF   ratio %.3f\n", wgs, rep, f32, f16, f16 / f32);
        }
    }
    return 0;
}
