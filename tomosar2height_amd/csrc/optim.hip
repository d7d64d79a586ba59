// AdamW over the whole parameter set in ONE launch (reference: train.py:97 `optim.AdamW(model.parameters(), lr)` stepped
// at trainer.py:78-79, torch defaults betas (0.9, 0.999), eps 1e-8, weight_decay 0.01).  torch's own step is ~10
// multi-tensor launches over ~150 tensors; here the gradients already live in one flat bucket (trainer.GradBucket), the
// moments in two flat buffers of the same layout, and a device-resident table maps 4096-element chunks to tensors, so one
// streaming pass reads p, g, m, v and writes p, m, v (28 B / parameter: 306 MB for Berlin cloud-only, HBM bound).
//
// Arithmetic follows torch/optim/adamw.py (single-tensor form, amsgrad = maximize = False), one rounding per operation, no
// contraction (the library is built with -ffp-contract=off).  It matches torch TO ROUNDING, not bit for bit: ATen's device
// `tensor / python_scalar` multiplies by the fp32 reciprocal of the scalar, the IEEE division sqrt(v) / sqrt(bc2) below can
// differ from that by one ulp (tests/test_hip_optim.py: 2e-6 relative):
//     p  = p * (1 - lr * wd)
//     m  = m + (g - m) * (1 - beta1)                      (Tensor.lerp_, weight < 0.5 branch)
//     v  = v * beta2 + (1 - beta2) * g * g                (mul_ then addcmul_)
//     p  = p + (-(lr / bc1)) * (m / (sqrt(v) / sqrt(bc2) + eps))
// The host passes the step-dependent scalars computed in double precision exactly as torch does.
#include <math.h>

#include "t2h_common.h"

namespace t2h {

struct AdamTensor {          // one row of the table: 5 x 8 bytes, filled by the host binding
    float *p;
    const float *g;
    float *m;
    float *v;
    long long n;
};

constexpr int kAdamChunk = 4096;     // elements per workgroup: 256 threads x 4 float4

struct AdamScalars { float decay, one_minus_beta1, beta2, one_minus_beta2, neg_step_size, bc2_sqrt, eps; };

__device__ inline void adam_one(float &p, float g, float &m, float &v, const AdamScalars &s) {
    p = __fmul_rn(p, s.decay);
    m = __fadd_rn(m, __fmul_rn(s.one_minus_beta1, __fsub_rn(g, m)));
    v = __fadd_rn(__fmul_rn(v, s.beta2), __fmul_rn(__fmul_rn(s.one_minus_beta2, g), g));
    float denom = __fadd_rn(__fdiv_rn(__fsqrt_rn(v), s.bc2_sqrt), s.eps);
    p = __fadd_rn(p, __fmul_rn(s.neg_step_size, __fdiv_rn(m, denom)));
}

__global__ __launch_bounds__(256) void adamw_flat_kernel(const AdamTensor *__restrict__ table,
                                                         const int2 *__restrict__ chunks, AdamScalars s, int zero_grad) {
    const int2 c = chunks[blockIdx.x];                       // (tensor index, first element)
    const AdamTensor t = table[c.x];
    const long long begin = c.y, end = min((long long)c.y + kAdamChunk, t.n);
    float *p = t.p + begin, *m = t.m + begin, *v = t.v + begin;
    float *g = const_cast<float *>(t.g) + begin;
    const int n = (int)(end - begin);
    const bool vec = (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0;
    if (vec) {
        const int n4 = n >> 2;
        for (int i = threadIdx.x; i < n4; i += 256) {
            float4 pp = reinterpret_cast<float4 *>(p)[i], gg = reinterpret_cast<const float4 *>(g)[i];
            float4 mm = reinterpret_cast<float4 *>(m)[i], vv = reinterpret_cast<float4 *>(v)[i];
            adam_one(pp.x, gg.x, mm.x, vv.x, s);
            adam_one(pp.y, gg.y, mm.y, vv.y, s);
            adam_one(pp.z, gg.z, mm.z, vv.z, s);
            adam_one(pp.w, gg.w, mm.w, vv.w, s);
            reinterpret_cast<float4 *>(p)[i] = pp;
            reinterpret_cast<float4 *>(m)[i] = mm;
            reinterpret_cast<float4 *>(v)[i] = vv;
            if (zero_grad) reinterpret_cast<float4 *>(g)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        for (int i = (n4 << 2) + threadIdx.x; i < n; i += 256) {
            float pp = p[i], mm = m[i], vv = v[i];
            adam_one(pp, g[i], mm, vv, s);
            p[i] = pp; m[i] = mm; v[i] = vv;
            if (zero_grad) g[i] = 0.f;
        }
    } else {
        for (int i = threadIdx.x; i < n; i += 256) {
            float pp = p[i], mm = m[i], vv = v[i];
            adam_one(pp, g[i], mm, vv, s);
            p[i] = pp; m[i] = mm; v[i] = vv;
            if (zero_grad) g[i] = 0.f;
        }
    }
}

}  // namespace t2h

using namespace t2h;

T2H_API int t2h_adamw_chunk_elems(void) { return kAdamChunk; }

T2H_API int t2h_adamw_flat_step(const void *table, const int32_t *chunks, int n_chunks, double lr, double beta1, double beta2,
                                double eps, double weight_decay, int64_t step, int zero_grad, t2h_stream_t stream) {
    if (!table || !chunks) return fail(T2H_ERR_ARG, "adamw_flat_step: null pointer");
    if (n_chunks < 0 || step < 1 || !(beta1 >= 0.0 && beta1 < 1.0) || !(beta2 >= 0.0 && beta2 < 1.0) || !(eps >= 0.0))
        return fail(T2H_ERR_ARG, "adamw_flat_step: bad hyper-parameters (step=%lld)", (long long)step);
    if (n_chunks == 0) return T2H_OK;
    // torch/optim/adamw.py (_single_tensor_adam with decoupled decay): python-float (double) scalars, rounded to fp32 when
    // they meet the fp32 tensors
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    AdamScalars s;
    s.decay = (float)(1.0 - lr * weight_decay);
    s.one_minus_beta1 = (float)(1.0 - beta1);
    s.beta2 = (float)beta2;
    s.one_minus_beta2 = (float)(1.0 - beta2);
    s.neg_step_size = (float)(-(lr / bc1));
    s.bc2_sqrt = (float)sqrt(bc2);
    s.eps = (float)eps;
    hipLaunchKernelGGL(adamw_flat_kernel, dim3((unsigned)n_chunks), dim3(256), 0, as_stream(stream),
                       static_cast<const AdamTensor *>(table), reinterpret_cast<const int2 *>(chunks), s, zero_grad);
    return check_launch("adamw_flat_step");
}
