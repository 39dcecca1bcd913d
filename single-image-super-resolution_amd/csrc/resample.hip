// resample.hip -- bicubic HR->LR degradation of utils.lr_from_hr (utils.py:16-31):
// F.interpolate(mode='bicubic', align_corners=True) followed by a clamp to [-1, 1].
// Restates ATen's UpSampleBicubic2d: scale = (in-1)/(out-1), src = scale*dst, 4x4 taps at
// floor(src)-1..+2 with indices clamped to the image, cubic-convolution weights with A = -0.75.
// The phase is non-integer ((in-1)/(out-1), e.g. 2.0105), so every output pixel has its own 16
// weights.  Pure bandwidth: planar NCHW in/out, one thread per output pixel, coalesced along x.
#include "sisr_dev.h"

#include <algorithm>

#define CUBIC_A (-0.75f)
__device__ __forceinline__ float cc1(float x) { return ((CUBIC_A + 2.f) * x - (CUBIC_A + 3.f)) * x * x + 1.f; }
__device__ __forceinline__ float cc2(float x) { return ((CUBIC_A * x - 5.f * CUBIC_A) * x + 8.f * CUBIC_A) * x - 4.f * CUBIC_A; }

__device__ __forceinline__ void cubic_setup(int dst, float scale, int n_in, int idx[4], float w[4]) {
    const float src = scale * (float)dst;
    const float fl = floorf(src);
    const int i0 = (int)fl;
    const float t = src - fl;
    w[0] = cc2(t + 1.f); w[1] = cc1(t); w[2] = cc1(1.f - t); w[3] = cc2(2.f - t);
#pragma unroll
    for (int k = 0; k < 4; ++k) idx[k] = min(max(i0 - 1 + k, 0), n_in - 1);
}

__global__ void bicubic_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int NC, int H, int W,
                                   int Ho, int Wo, float sy, float sx, int clampv) {
    const int64_t total = (int64_t)NC * Ho * Wo;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int ox = (int)(e % Wo);
        const int oy = (int)((e / Wo) % Ho);
        const int nc = (int)(e / ((int64_t)Wo * Ho));
        int iy[4], ix[4];
        float wy[4], wx[4];
        cubic_setup(oy, sy, H, iy, wy);
        cubic_setup(ox, sx, W, ix, wx);
        const float* p = x + (int64_t)nc * H * W;
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float* row = p + (int64_t)iy[i] * W;
            const float r = row[ix[0]] * wx[0] + row[ix[1]] * wx[1] + row[ix[2]] * wx[2] + row[ix[3]] * wx[3];
            acc += r * wy[i];
        }
        if (clampv) acc = fminf(fmaxf(acc, -1.f), 1.f);
        y[e] = acc;
    }
}

// transpose of the interpolation: scatter with float atomics (the tensors are 3-channel images)
__global__ void bicubic_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ yc, float* dx, int NC,
                                   int H, int W, int Ho, int Wo, float sy, float sx) {
    const int64_t total = (int64_t)NC * Ho * Wo;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        float g = dy[e];
        if (yc != nullptr) {   // clamp mask: gradient passes only strictly inside (-1, 1)
            const float v = yc[e];
            if (!(v > -1.f && v < 1.f)) g = 0.f;
        }
        if (g == 0.f) continue;
        const int ox = (int)(e % Wo);
        const int oy = (int)((e / Wo) % Ho);
        const int nc = (int)(e / ((int64_t)Wo * Ho));
        int iy[4], ix[4];
        float wy[4], wx[4];
        cubic_setup(oy, sy, H, iy, wy);
        cubic_setup(ox, sx, W, ix, wx);
        float* p = dx + (int64_t)nc * H * W;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) atomicAdd(p + (int64_t)iy[i] * W + ix[j], g * wy[i] * wx[j]);
    }
}

static inline float ac_scale(int n_in, int n_out) { return n_out > 1 ? (float)(n_in - 1) / (float)(n_out - 1) : 0.f; }

extern "C" int sisr_bicubic_fwd(const float* x, float* y, int32_t NC, int32_t H, int32_t W, int32_t Ho, int32_t Wo,
                                int32_t clampv, void* stream) {
    if (!x || !y || NC <= 0 || H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0) return SISR_E_BADARG;
    const int64_t total = (int64_t)NC * Ho * Wo;
    const int blocks = (int)std::min<int64_t>((total + 255) / 256, 4096);
    hipLaunchKernelGGL(bicubic_fwd_kernel, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, y,
                       NC, H, W, Ho, Wo, ac_scale(H, Ho), ac_scale(W, Wo), clampv);
    SISR_CHECK_LAUNCH();
    return 0;
}

extern "C" int sisr_bicubic_bwd(const float* dy, const float* y_clamped, float* dx, int32_t NC, int32_t H, int32_t W,
                                int32_t Ho, int32_t Wo, void* stream) {
    if (!dy || !dx || NC <= 0 || H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0) return SISR_E_BADARG;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    hipError_t e = hipMemsetAsync(dx, 0, (size_t)NC * H * W * sizeof(float), st);
    if (e != hipSuccess) return (int)e;
    const int64_t total = (int64_t)NC * Ho * Wo;
    const int blocks = (int)std::min<int64_t>((total + 255) / 256, 4096);
    hipLaunchKernelGGL(bicubic_bwd_kernel, dim3(blocks), dim3(256), 0, st, dy, y_clamped, dx, NC, H, W, Ho, Wo,
                       ac_scale(H, Ho), ac_scale(W, Wo));
    SISR_CHECK_LAUNCH();
    return 0;
}
