// resample.hip -- the image-side resampling of the path:
//   * the dataset transform of config.py:225-231 -- transforms.Resize (Pillow's anti-aliased BILINEAR ImagingResample on
//     8-bit images), ToTensor, Normalize(.5, .5) -- as ONE kernel over a batch of decoded uint8 images (row f4);
//   * bicubic HR->LR degradation of utils.lr_from_hr (utils.py:16-31):
// F.interpolate(mode='bicubic', align_corners=True) followed by a clamp to [-1, 1].
// Restates ATen's UpSampleBicubic2d: scale = (in-1)/(out-1), src = scale*dst, 4x4 taps at
// floor(src)-1..+2 with indices clamped to the image, cubic-convolution weights with A = -0.75.
// The phase is non-integer ((in-1)/(out-1), e.g. 2.0105), so every output pixel has its own 16
// weights.  Pure bandwidth: planar NCHW in/out, one thread per output pixel, coalesced along x.
#include "sisr_dev.h"

#include <algorithm>
#include <cmath>

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

// ---- dataset transform: Resize (Pillow BILINEAR, 8 bits per channel) + ToTensor + Normalize ----------------------------------
// Pillow resamples separably, horizontal pass first, and ROUNDS TO uint8 AFTER EACH PASS (Resample.c: coefficients with 22
// fractional bits, int32 accumulation from 2^21, clip8).  One thread per output pixel: for each of its <= ksize_y source rows
// it forms the horizontal result (<= ksize_x taps, rounded to 8 bits exactly as the intermediate image would hold it), then
// the vertical sum -- the horizontal work is repeated ksize_y times per output row, which at these sizes (1.8 MB of source
// bytes per batch, served by L1 / L2) costs less than a second launch with an intermediate image.  Integer arithmetic:
// bit-exact with Pillow.  Output: NCHW float32, (u / 255 - 0.5) / 0.5 as ToTensor + Normalize compute it.
#define PIL_PRECISION_BITS 22
__device__ __forceinline__ int pil_clip8(int v) { v >>= PIL_PRECISION_BITS; return v < 0 ? 0 : (v > 255 ? 255 : v); }

__global__ void resize_u8_normalize_kernel(const unsigned char* __restrict__ src, float* __restrict__ dst, int N, int H0, int W0,
                                           int C, int H, int W, const int* __restrict__ bx, const int* __restrict__ kx, int ksx,
                                           const int* __restrict__ by, const int* __restrict__ ky, int ksy, int pass_x,
                                           int pass_y, float mean, float stdv) {
    const int64_t total = (int64_t)N * H * W;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int ox = (int)(e % W), oy = (int)((e / W) % H), n = (int)(e / ((int64_t)W * H));
        const int xmin = pass_x ? bx[2 * ox] : ox, xn = pass_x ? bx[2 * ox + 1] : 1;
        const int ymin = pass_y ? by[2 * oy] : oy, yn = pass_y ? by[2 * oy + 1] : 1;
        const unsigned char* img = src + (int64_t)n * H0 * W0 * C;
        for (int c = 0; c < C; ++c) {
            int acc_y = 1 << (PIL_PRECISION_BITS - 1);
            for (int j = 0; j < yn; ++j) {
                const unsigned char* row = img + ((int64_t)(ymin + j) * W0 + xmin) * C + c;
                int h8;
                if (pass_x) {
                    int acc_x = 1 << (PIL_PRECISION_BITS - 1);
                    for (int i = 0; i < xn; ++i) acc_x += (int)row[(int64_t)i * C] * kx[ox * ksx + i];
                    h8 = pil_clip8(acc_x);
                } else {
                    h8 = row[0];
                }
                acc_y += pass_y ? h8 * ky[oy * ksy + j] : 0;
                if (!pass_y) acc_y = h8;
            }
            const int u8 = pass_y ? pil_clip8(acc_y) : acc_y;
            const float t = (float)u8 / 255.0f;                                  // ToTensor (correctly rounded division)
            dst[(((int64_t)n * C + c) * H + oy) * W + ox] = (t - mean) / stdv;   // Normalize
        }
    }
}

// precompute_coeffs + normalize_coeffs_8bpc of Pillow's Resample.c for the whole-image box and the BILINEAR filter
// (support 1.0).  Returns ksize (taps per output index); with bounds / kk non-null also fills bounds[out][2] = (first input
// index, tap count) and kk[out][ksize] (22-bit fixed point).  Host function: double precision, as Pillow.
extern "C" int sisr_resize_coeffs(int32_t in_size, int32_t out_size, int32_t* bounds, int32_t* kk) {
    if (in_size <= 0 || out_size <= 0) return SISR_E_BADARG;
    const double scale = (double)((float)in_size - 0.0f) / out_size;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = 1.0 * filterscale;
    const int ksize = (int)ceil(support) * 2 + 1;
    if (ksize > 64) return SISR_E_UNSUPPORTED;            // (a down-scale above ~31x; checked in the size query too, before a caller sizes tables)
    if (!bounds || !kk) return ksize;
    const double ss = 1.0 / filterscale;
    for (int xx = 0; xx < out_size; ++xx) {
        const double center = 0.0 + (xx + 0.5) * scale;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        double w[64], ww = 0.0;
        for (int x = 0; x < xmax; ++x) {
            double a = (x + xmin - center + 0.5) * ss;
            if (a < 0.0) a = -a;
            w[x] = a < 1.0 ? 1.0 - a : 0.0;
            ww += w[x];
        }
        for (int x = 0; x < xmax; ++x)
            if (ww != 0.0) w[x] /= ww;
        for (int x = xmax; x < ksize; ++x) w[x] = 0.0;
        for (int x = 0; x < ksize; ++x)
            kk[xx * ksize + x] = w[x] < 0 ? (int)(-0.5 + w[x] * (1 << PIL_PRECISION_BITS)) : (int)(0.5 + w[x] * (1 << PIL_PRECISION_BITS));
        bounds[2 * xx] = xmin;
        bounds[2 * xx + 1] = xmax;
    }
    return ksize;
}

// src: [N][H0][W0][C] uint8 (decoded images, Pillow's memory layout); dst: [N][C][H][W] float32.  bx / kx (by / ky): device
// copies of sisr_resize_coeffs(W0, W) (sisr_resize_coeffs(H0, H)); a pass whose size does not change is skipped, as
// ImagingResample does (its tables may be NULL then).
extern "C" int sisr_resize_u8_normalize(const unsigned char* src, float* dst, int32_t N, int32_t H0, int32_t W0, int32_t C,
                                        int32_t H, int32_t W, const int32_t* bx, const int32_t* kx, int32_t ksx,
                                        const int32_t* by, const int32_t* ky, int32_t ksy, float mean, float stdv, void* stream) {
    if (!src || !dst || N <= 0 || H0 <= 0 || W0 <= 0 || C <= 0 || H <= 0 || W <= 0 || stdv == 0.f) return SISR_E_BADARG;
    const int pass_x = W0 != W, pass_y = H0 != H;
    if ((pass_x && (!bx || !kx || ksx <= 0)) || (pass_y && (!by || !ky || ksy <= 0))) return SISR_E_BADARG;
    const int64_t total = (int64_t)N * H * W;
    const int blocks = (int)std::min<int64_t>((total + 255) / 256, 4096);
    hipLaunchKernelGGL(resize_u8_normalize_kernel, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), src, dst,
                       N, H0, W0, C, H, W, bx, kx, ksx, by, ky, ksy, pass_x, pass_y, mean, stdv);
    SISR_CHECK_LAUNCH();
    return 0;
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
