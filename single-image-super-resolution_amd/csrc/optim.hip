// optim.hip -- fused multi-tensor Adam step: one launch updates every parameter of a network.
// Replaces the per-step optimizer passes of the reference, torch.optim.Adam(net.parameters(), lr, betas=(.9, .999))
// (config.py:292-294; stepped at train.py:75,108), with torch's semantics for amsgrad=False, maximize=False:
//     g' = g + wd * p ;  m = m + (g' - m) * (1 - b1) ;  v = v * b2 + (1 - b2) * g'^2
//     p -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// HBM-bound: 4 reads + 3 writes of 4 bytes per parameter.  The descriptor table lives in device memory; a workgroup
// finds its tensor by a binary search over the per-tensor first-block indices.
#include "sisr_dev.h"

#define ADAM_CHUNK (SISR_BLOCK * 16)          // elements per workgroup

__global__ void __launch_bounds__(SISR_BLOCK) adam_step_kernel(const SisrAdamDesc* __restrict__ table, int n, float step_size,
                                                                float omb1, float beta2, float omb2, float eps, float wd,
                                                                float inv_sqrt_bc2) {
    int lo = 0, hi = n - 1;                     // last tensor whose block_start <= blockIdx.x
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (table[mid].block_start <= (int64_t)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const SisrAdamDesc t = table[lo];
    const int64_t base = ((int64_t)blockIdx.x - t.block_start) * ADAM_CHUNK;
    if ((t.numel & 3) == 0) {
        const int64_t n4 = t.numel >> 2;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int64_t i = (base >> 2) + k * SISR_BLOCK + threadIdx.x;
            if (i < n4) {
                f32x4 p = reinterpret_cast<const f32x4*>(t.p)[i], g = reinterpret_cast<const f32x4*>(t.g)[i];
                f32x4 m = reinterpret_cast<const f32x4*>(t.m)[i], v = reinterpret_cast<const f32x4*>(t.v)[i];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float gg = g[j] + wd * p[j];
                    m[j] = m[j] + (gg - m[j]) * omb1;
                    v[j] = v[j] * beta2 + omb2 * gg * gg;
                    p[j] -= step_size * (m[j] / (sqrtf(v[j]) * inv_sqrt_bc2 + eps));
                }
                reinterpret_cast<f32x4*>(t.p)[i] = p;
                reinterpret_cast<f32x4*>(t.m)[i] = m;
                reinterpret_cast<f32x4*>(t.v)[i] = v;
            }
        }
    } else {
        for (int k = 0; k < 16; ++k) {
            const int64_t i = base + k * SISR_BLOCK + threadIdx.x;
            if (i < t.numel) {
                const float gg = t.g[i] + wd * t.p[i];
                const float m = t.m[i] + (gg - t.m[i]) * omb1;
                const float v = t.v[i] * beta2 + omb2 * gg * gg;
                t.m[i] = m; t.v[i] = v;
                t.p[i] -= step_size * (m / (sqrtf(v) * inv_sqrt_bc2 + eps));
            }
        }
    }
}

extern "C" int64_t sisr_adam_blocks(int64_t numel) { return numel <= 0 ? 0 : (numel + ADAM_CHUNK - 1) / ADAM_CHUNK; }

extern "C" int sisr_adam_step(const SisrAdamDesc* table_dev, int32_t n, int64_t total_blocks, double lr, double beta1,
                              double beta2, double eps, double weight_decay, double bias_corr1, double bias_corr2,
                              void* stream) {
    if (!table_dev || n <= 0 || total_blocks <= 0 || total_blocks >= (1ll << 31) || bias_corr1 <= 0.0 || bias_corr2 <= 0.0)
        return SISR_E_BADARG;
    // host scalars are doubles (as in torch): 1 - beta, lr / bias_corr1 ... are rounded to fp32 once, after the arithmetic
    hipLaunchKernelGGL(adam_step_kernel, dim3((unsigned)total_blocks), dim3(SISR_BLOCK), 0, reinterpret_cast<hipStream_t>(stream),
                       table_dev, n, (float)(lr / bias_corr1), (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2),
                       (float)eps, (float)weight_decay, (float)(1.0 / sqrt(bias_corr2)));
    SISR_CHECK_LAUNCH();
    return 0;
}
