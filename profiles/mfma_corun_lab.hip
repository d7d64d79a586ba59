// What does a wave that is NOT issuing MFMAs get beside waves that are?  The 128 x 128 GEMM epilogue (one wave's share:
// ~400 vector / LDS instructions) takes 2.3 us when its workgroup has the CU to itself and 12-17 us beside three
// main-loop workgroups (in-kernel stamps); neither dropping its stores nor s_setprio changed that.
// This lab times a "victim" wave running a fixed instruction stream beside 0 / 1 / 3 "aggressor" waves per SIMD that
// issue v_mfma_f32_32x32x2_f32 (or 16x16x4) back to back.  Result (MI355X): beside 0 or 1 such wave the victim is
// unaffected; beside 3 it finishes when they do, whatever it executes -- VALU, LDS, global stores or plain scalar adds --
// and whatever its s_setprio: the wave as a whole is not scheduled.
//   hipcc --offload-arch=gfx950 -O3 -o scratch/mfma_corun_lab profiles/mfma_corun_lab.hip && scratch/mfma_corun_lab
#include <hip/hip_runtime.h>
#pragma clang diagnostic ignored "-Wunused-value"
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

enum Victim { V_VALU = 0, V_LDS_READ = 1, V_LDS_WRITE = 2, V_MIXED = 3, V_VALU_DEP = 4, V_VMEM_STORE = 5, V_SALU = 6 };

// waves 0..3: victims (one per SIMD); waves 4..: aggressors.  out[block * 4 + wave] = victim cycles
template <int AGG_SHAPE>
__global__ void corun(const float *in, float *sink, unsigned long long *cycles, int victim, int victim_iters, int agg_iters, int prio) {
    __shared__ float lds[8192];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int i = tid; i < 8192; i += blockDim.x) lds[i] = in[i & 1023];
    __syncthreads();
    if (wave >= 4) {
        float a0 = in[tid], a1 = in[tid + 1], b0 = in[tid + 2], b1 = in[tid + 3];
        float s = 0.f;
        if (AGG_SHAPE == 32) {
            f32x16 acc[4];
            for (int i = 0; i < 4; ++i) for (int q = 0; q < 16; ++q) acc[i][q] = 0.f;
            for (int it = 0; it < agg_iters; ++it) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[3], 0, 0, 0);
            }
            for (int i = 0; i < 4; ++i) for (int q = 0; q < 16; ++q) s += acc[i][q];
        } else {
            f32x4 acc[8];
            for (int i = 0; i < 8; ++i) for (int q = 0; q < 4; ++q) acc[i][q] = 0.f;
            for (int it = 0; it < agg_iters; ++it) {
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(i & 1 ? a0 : a1, i & 2 ? b0 : b1, acc[i], 0, 0, 0);
            }
            for (int i = 0; i < 8; ++i) for (int q = 0; q < 4; ++q) s += acc[i][q];
        }
        sink[blockIdx.x * blockDim.x + tid] = s;
        return;
    }
    // victim: let the aggressors get going first
    __builtin_amdgcn_s_sleep(64);
    if (prio) __builtin_amdgcn_s_setprio(3);
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = in[tid + i];
    const float ca = in[5], cb = in[6];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < victim_iters; ++it) {
        if (victim == V_VALU) {                    // 64 independent fmas
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(ca), "v"(cb));
        } else if (victim == V_VALU_DEP) {         // 64 fmas, one dependent chain
#pragma unroll
            for (int r = 0; r < 64; ++r) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[0]) : "v"(ca), "v"(cb));
        } else if (victim == V_LDS_READ) {         // 16 ds_read_b128, all in flight, then one wait
            f32x4 v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = *reinterpret_cast<volatile f32x4 *>(lds + ((lane * 4 + r * 260) & 8188));
#pragma unroll
            for (int r = 0; r < 16; ++r) x[r & 7] += v[r].x;
        } else if (victim == V_LDS_WRITE) {        // 64 ds_write_b32
#pragma unroll
            for (int r = 0; r < 64; ++r) *reinterpret_cast<volatile float *>(lds + ((lane + r * 68) & 8191)) = x[r & 7];
        } else if (victim == V_VMEM_STORE) {       // 32 global_store_dword, addresses = one VGPR pair + immediate offsets
            float *dst = sink + (size_t)blockIdx.x * blockDim.x + tid;
#pragma unroll
            for (int r = 0; r < 32; ++r) asm volatile("global_store_dword %0, %1, off offset:%2" :: "v"(dst), "v"(x[r & 7]), "n"(r * 64) : "memory");
        } else if (victim == V_SALU) {             // 64 scalar adds
            int sa = it;
#pragma unroll
            for (int r = 0; r < 64; ++r) asm volatile("s_add_i32 %0, %0, 1" : "+s"(sa));
            if (sa == -1) x[0] += 1.f;
        } else {                                   // epilogue-like: 16 ds_write_b32, then 4 x (ds_read_b128, wait, 12 VALU)
#pragma unroll
            for (int r = 0; r < 16; ++r) *reinterpret_cast<volatile float *>(lds + wave * 1152 + ((lane & 31) + r * 36 + (lane >> 5) * 144) % 1152) = x[r & 7];
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                f32x4 v = *reinterpret_cast<volatile f32x4 *>(lds + wave * 1152 + ((lane >> 3) + p * 8) * 36 + (lane & 7) * 4);
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x[0]) : "v"(v.x), "v"(ca));
                    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x[1]) : "v"(v.y), "v"(ca));
                    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x[2]) : "v"(v.z), "v"(ca));
                    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x[3]) : "v"(v.w), "v"(ca));
                }
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += x[i];
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    sink[blockIdx.x * blockDim.x + tid] = s;
    if (lane == 0) cycles[blockIdx.x * 4 + wave] = t1 - t0;
}

int main() {
    const int blocks = 256;
    float *in, *sink;
    unsigned long long *cyc;
    hipMalloc(&in, 8192 * sizeof(float));
    hipMalloc(&sink, (blocks * 1024 + 4096) * sizeof(float));
    hipMalloc(&cyc, blocks * 4 * sizeof(unsigned long long));
    std::vector<float> h(8192);
    for (auto &v : h) v = (float)rand() / RAND_MAX - 0.5f;
    hipMemcpy(in, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice);
    const char *names[] = {"64 indep v_fma", "16 ds_read_b128+wait", "64 ds_write_b32", "epilogue-like tile", "64 dependent v_fma",
                           "32 global_store_dword", "64 s_add_i32"};
    const int instrs[] = {64, 16, 64, 16 + 4 * 13, 64, 32, 64};
    const int victim_iters = 64;
    for (int victim : {0, 4, 1, 2, 3, 5, 6}) {
        for (int shape : {32, 16}) {
            for (int agg : {0, 1, 3}) {
                if (agg == 0 && shape == 16) continue;
                for (int prio : {0, 1}) {
                    if (agg == 0 && prio) continue;
                    const int threads = 256 + 256 * agg;
                    // aggressors must outlast the victims: 4 MFMAs x 64 cycles per iteration for 32x32, 8 x 32 for 16x16
                    const int agg_iters = 40000;
                    std::vector<unsigned long long> hc(blocks * 4);
                    for (int rep = 0; rep < 2; ++rep) {
                        if (shape == 32) hipLaunchKernelGGL(corun<32>, dim3(blocks), dim3(threads), 0, 0, in, sink, cyc, victim, victim_iters, agg_iters, prio);
                        else hipLaunchKernelGGL(corun<16>, dim3(blocks), dim3(threads), 0, 0, in, sink, cyc, victim, victim_iters, agg_iters, prio);
                        hipDeviceSynchronize();
                    }
                    hipMemcpy(hc.data(), cyc, hc.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
                    std::sort(hc.begin(), hc.end());
                    const double med = (double)hc[hc.size() / 2] / victim_iters;
                    printf("%-24s beside %d x mfma %s/SIMD%s: %8.1f cycles per pass = %6.1f per instruction\n", names[victim], agg,
                           shape == 32 ? "32x32x2 " : "16x16x4 ", prio ? " prio3" : "      ", med, med / instrs[victim]);
                }
            }
        }
    }
    return 0;
}
