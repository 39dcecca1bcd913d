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

// weight with which output index `o` reads input index `i` along one axis (taps that the border clamp folds onto
// the same input index add up)
__device__ __forceinline__ float cubic_weight_of(int o, float scale, int n_in, int i) {
    int idx[4];
    float w[4];
    cubic_setup(o, scale, n_in, idx, w);
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) s += idx[k] == i ? w[k] : 0.f;
    return s;
}

// output indices whose 4 taps can reach input index i: floor(scale * o) in [i - 2, i + 1] (one index of margin
// on each side; cubic_weight_of() returns 0 for the ones that do not)
__device__ __forceinline__ void cubic_reach(int i, float scale, int n_out, int& lo, int& hi) {
    if (scale > 0.f) {
        lo = max(0, (int)floorf((float)(i - 2) / scale) - 1);
        hi = min(n_out - 1, (int)ceilf((float)(i + 2) / scale) + 1);
    } else {
        lo = 0; hi = n_out - 1;
    }
}

// transpose of the interpolation in GATHER form: one thread per input pixel sums, in a fixed order, the output
// gradients that read it -- deterministic (no atomics), which the unsupervised branch's replay tests rely on
__global__ void bicubic_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ yc, float* __restrict__ dx,
                                   int NC, int H, int W, int Ho, int Wo, float sy, float sx) {
    const int64_t total = (int64_t)NC * H * W;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int ix = (int)(e % W);
        const int iy = (int)((e / W) % H);
        const int nc = (int)(e / ((int64_t)W * H));
        int ylo, yhi, xlo, xhi;
        cubic_reach(iy, sy, Ho, ylo, yhi);
        cubic_reach(ix, sx, Wo, xlo, xhi);
        const float* g = dy + (int64_t)nc * Ho * Wo;
        const float* v = yc != nullptr ? yc + (int64_t)nc * Ho * Wo : nullptr;
        float acc = 0.f;
        for (int oy = ylo; oy <= yhi; ++oy) {
            const float wy = cubic_weight_of(oy, sy, H, iy);
            if (wy == 0.f) continue;
            float row = 0.f;
            for (int ox = xlo; ox <= xhi; ++ox) {
                const float wx = cubic_weight_of(ox, sx, W, ix);
                float gv = g[(int64_t)oy * Wo + ox];
                if (v != nullptr) {                // clamp mask: gradient passes only strictly inside (-1, 1)
                    const float c = v[(int64_t)oy * Wo + ox];
                    if (!(c > -1.f && c < 1.f)) gv = 0.f;
                }
                row += wx * gv;
            }
            acc += wy * row;
        }
        dx[e] = acc;
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
    const int64_t total = (int64_t)NC * H * W;
    const int blocks = (int)std::min<int64_t>((total + 255) / 256, 4096);
    hipLaunchKernelGGL(bicubic_bwd_kernel, dim3(blocks), dim3(256), 0, st, dy, y_clamped, dx, NC, H, W, Ho, Wo,
                       ac_scale(H, Ho), ac_scale(W, Wo));
    SISR_CHECK_LAUNCH();
    return 0;
}
