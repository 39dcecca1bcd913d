// layout_fc.hip -- NHWC<->NCHW materialisation (D's flatten, VGG taps) and the fully connected
// layers of the discriminator (nn.Linear(fc_in, 1024) + LeakyReLU + nn.Linear(1024, 1) + Sigmoid,
// model_discriminator.py:47-53).  The FC layers are pure weight streaming: W[1024][fc_in] is
// 75 MB (HR 96) / 302 MB (HR 192) and the batch is 16, so every kernel reads or writes W exactly
// once with 16-byte coalesced accesses along K and keeps the 16 batch rows in registers.
#include "sisr_dev.h"

#include <algorithm>

// ---- layout -------------------------------------------------------------------------------------
// 32x32 (pixel x channel) LDS transpose tiles: reads coalesced along C, writes coalesced along pixels
__global__ void __launch_bounds__(SISR_BLOCK) nhwc_to_nchw_kernel(const float* __restrict__ x,
                                                                  const float* __restrict__ pa,
                                                                  const float* __restrict__ pd,
                                                                  const float* slope_p, float slope,
                                                                  float* __restrict__ y, int64_t dst_stride,
                                                                  int HW, int C, int x_bf16) {
    __shared__ float tile[32][33];
    if (slope_p != nullptr) slope = slope_p[0];
    const int n = blockIdx.z;
    const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    for (int i = ty; i < 32; i += 8) {
        const int p = p0 + i, c = c0 + tx;
        float v = 0.f;
        if (p < HW && c < C) {
            v = ld_elem(x, ((int64_t)n * HW + p) * C + c, x_bf16 != 0);
            if (pa != nullptr) v = pa[c] * v + pd[c];
            v = lrelu(v, slope);
        }
        tile[i][tx] = v;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, p = p0 + tx;
        if (p < HW && c < C) y[(int64_t)n * dst_stride + (int64_t)c * HW + p] = tile[tx][i];
    }
}

__global__ void __launch_bounds__(SISR_BLOCK) nchw_to_nhwc_kernel(const float* __restrict__ x, int64_t src_stride,
                                                                  float* __restrict__ y, int HW, int C, int y_bf16) {
    __shared__ float tile[32][33];
    const int n = blockIdx.z;
    const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, p = p0 + tx;
        tile[i][tx] = (p < HW && c < C) ? x[(int64_t)n * src_stride + (int64_t)c * HW + p] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int p = p0 + i, c = c0 + tx;
        if (p < HW && c < C) st_elem(y, ((int64_t)n * HW + p) * C + c, y_bf16 != 0, tile[tx][i]);
    }
}

// ---- NCHW gradient image (few channels) -> NHWC padded to a multiple of 4 channels, with the tanh-backward
// transform fused: g[n][p][c] = dy * (1 - out^2) (or dy).  Feeds the bf16 weight-gradient kernel of the
// generator's last convolution (Cout = 3), whose dy operand would otherwise be a scalar NCHW gather.
__global__ void __launch_bounds__(SISR_BLOCK) nchw_grad_to_nhwc4_kernel(const float* __restrict__ dy,
                                                                        const float* __restrict__ out, float* __restrict__ g,
                                                                        int64_t HW, int C, int Cp, int64_t total) {
    for (int64_t i = (int64_t)blockIdx.x * SISR_BLOCK + threadIdx.x; i < total; i += (int64_t)gridDim.x * SISR_BLOCK) {
        const int64_t n = i / HW, p = i - n * HW;
        for (int c4 = 0; c4 < Cp; c4 += 4) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (c4 + j < C) {
                    const int64_t o = (n * C + c4 + j) * HW + p;
                    float t = dy[o];
                    if (out != nullptr) { const float y = out[o]; t *= 1.f - y * y; }
                    v[j] = t;
                }
            }
            *reinterpret_cast<f32x4*>(g + i * Cp + c4) = v;
        }
    }
}

// ---- FC forward: workgroup = FC_R weight rows, threads stride over K in float4 ----------------------
#define FC_R 4
#define FC_B 16
__global__ void __launch_bounds__(SISR_BLOCK) fc_forward_kernel(const float* __restrict__ x, float in_slope,
                                                                const float* __restrict__ W,
                                                                const float* __restrict__ bias,
                                                                float* __restrict__ y, int B, int K, int Nout,
                                                                int epi) {
    __shared__ float red[SISR_BLOCK / 64][FC_R * FC_B];
    const int n0 = blockIdx.x * FC_R;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float acc[FC_R][FC_B];
#pragma unroll
    for (int r = 0; r < FC_R; ++r)
#pragma unroll
        for (int b = 0; b < FC_B; ++b) acc[r][b] = 0.f;
    const int K4 = K >> 2;
    for (int k4 = tid; k4 < K4; k4 += SISR_BLOCK) {
        f32x4 wv[FC_R];
#pragma unroll
        for (int r = 0; r < FC_R; ++r)
            wv[r] = (n0 + r < Nout) ? *reinterpret_cast<const f32x4*>(W + (int64_t)(n0 + r) * K + k4 * 4)
                                    : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int b = 0; b < FC_B; ++b) {
            if (b < B) {
                f32x4 xv = *reinterpret_cast<const f32x4*>(x + (int64_t)b * K + k4 * 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) xv[j] = lrelu(xv[j], in_slope);
#pragma unroll
                for (int r = 0; r < FC_R; ++r)
                    acc[r][b] += wv[r][0] * xv[0] + wv[r][1] * xv[1] + wv[r][2] * xv[2] + wv[r][3] * xv[3];
            }
        }
    }
#pragma unroll
    for (int r = 0; r < FC_R; ++r)
#pragma unroll
        for (int b = 0; b < FC_B; ++b) {
            const float s = wave_sum(acc[r][b]);
            if (lane == 0) red[wave][r * FC_B + b] = s;
        }
    __syncthreads();
    if (tid < FC_R * FC_B) {
        const int r = tid / FC_B, b = tid - r * FC_B;
        if (n0 + r < Nout && b < B) {
            float v = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid] + (bias ? bias[n0 + r] : 0.f);
            if (epi == 1) v = 1.f / (1.f + expf(-v));
            y[(int64_t)b * Nout + n0 + r] = v;
        }
    }
}

// ---- FC dgrad: thread owns one float4 of K for all B rows; workgroups split Nout ---------------------
__global__ void __launch_bounds__(SISR_BLOCK) fc_dgrad_kernel(const float* __restrict__ dy,
                                                              const float* __restrict__ W, float* __restrict__ work,
                                                              int B, int K, int Nout, int rows_per_split) {
    extern __shared__ float dys[];   // [rows_per_split][FC_B]
    const int k4 = blockIdx.x * SISR_BLOCK + threadIdx.x;
    const int nb = blockIdx.y * rows_per_split;
    const int nrows = min(rows_per_split, Nout - nb);
    for (int i = threadIdx.x; i < rows_per_split * FC_B; i += SISR_BLOCK) {
        const int r = i / FC_B, b = i - r * FC_B;
        dys[i] = (r < nrows && b < B) ? dy[(int64_t)b * Nout + nb + r] : 0.f;
    }
    __syncthreads();
    if (k4 >= (K >> 2)) return;
    f32x4 acc[FC_B];
#pragma unroll
    for (int b = 0; b < FC_B; ++b) acc[b] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int r = 0; r < nrows; ++r) {
        const f32x4 wv = *reinterpret_cast<const f32x4*>(W + (int64_t)(nb + r) * K + k4 * 4);
#pragma unroll
        for (int b = 0; b < FC_B; ++b) acc[b] += wv * dys[r * FC_B + b];
    }
    for (int b = 0; b < B; ++b)
        *reinterpret_cast<f32x4*>(work + ((int64_t)blockIdx.y * B + b) * K + k4 * 4) = acc[b];
}

// ---- FC wgrad: dW[n][k4] = sum_b dy[b][n] * x[b][k4]; x rows in registers, loop over a slice of n ----
__global__ void __launch_bounds__(SISR_BLOCK) fc_wgrad_kernel(const float* __restrict__ dy,
                                                              const float* __restrict__ x, float in_slope,
                                                              float* __restrict__ dW, int B, int K, int Nout,
                                                              int rows_per_split) {
    extern __shared__ float dys[];
    const int k4 = blockIdx.x * SISR_BLOCK + threadIdx.x;
    const int nb = blockIdx.y * rows_per_split;
    const int nrows = min(rows_per_split, Nout - nb);
    for (int i = threadIdx.x; i < rows_per_split * FC_B; i += SISR_BLOCK) {
        const int r = i / FC_B, b = i - r * FC_B;
        dys[i] = (r < nrows && b < B) ? dy[(int64_t)b * Nout + nb + r] : 0.f;
    }
    __syncthreads();
    if (k4 >= (K >> 2)) return;
    f32x4 xv[FC_B];
#pragma unroll
    for (int b = 0; b < FC_B; ++b) {
        xv[b] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (b < B) {
            xv[b] = *reinterpret_cast<const f32x4*>(x + (int64_t)b * K + k4 * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) xv[b][j] = lrelu(xv[b][j], in_slope);
        }
    }
    for (int r = 0; r < nrows; ++r) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int b = 0; b < FC_B; ++b) s += xv[b] * dys[r * FC_B + b];
        *reinterpret_cast<f32x4*>(dW + (int64_t)(nb + r) * K + k4 * 4) = s;
    }
}

__global__ void fc_bias_grad_kernel(const float* __restrict__ dy, float* __restrict__ db, int B, int Nout) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n < Nout) {
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += dy[(int64_t)b * Nout + n];
        db[n] = s;
    }
}

__global__ void act_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ ref, float* __restrict__ out,
                               int64_t n, int kind, float slope) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float r = ref[i];
        out[i] = kind == 0 ? (r > 0.f ? dy[i] : slope * dy[i]) : dy[i] * r * (1.f - r);
    }
}

// partial sums over the Nout splits of fc_dgrad
__global__ void fc_split_reduce_kernel(const float* __restrict__ work, float* __restrict__ dx, int splits,
                                       int64_t elems) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (elems >> 2);
         i += (int64_t)gridDim.x * blockDim.x) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < splits; ++k) s += reinterpret_cast<const f32x4*>(work + (int64_t)k * elems)[i];
        reinterpret_cast<f32x4*>(dx)[i] = s;
    }
}

// ---- MaxPool2d(2,2) + fused ReLU backward (VGG19 features), NHWC float4 along C -------------------
template <bool XB, bool YB>
__global__ void maxpool2_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int H, int W, int C4,
                                    int Ho, int Wo, int64_t total) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(e % C4);
        int64_t t = e / C4;
        const int ox = (int)(t % Wo); t /= Wo;
        const int oy = (int)(t % Ho);
        const int n = (int)(t / Ho);
        const int64_t p = (((int64_t)n * H + 2 * oy) * W + 2 * ox) * C4 + c4;
        const f32x4 a = ld4<XB>(x, p), b = ld4<XB>(x, p + C4), c = ld4<XB>(x, p + (int64_t)W * C4),
                    d = ld4<XB>(x, p + (int64_t)W * C4 + C4);
        f32x4 m;
#pragma unroll
        for (int j = 0; j < 4; ++j) m[j] = fmaxf(fmaxf(a[j], b[j]), fmaxf(c[j], d[j]));
        st4<YB>(y, e, m);
    }
}

template <bool GB, bool XB, bool DB>
__global__ void maxpool2_relu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                         float* __restrict__ dx, int H, int W, int C4, int Ho, int Wo,
                                         int64_t total) {
    // one thread per (pooled pixel, channel group): writes its whole 2x2 window of dx
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(e % C4);
        int64_t t = e / C4;
        const int ox = (int)(t % Wo); t /= Wo;
        const int oy = (int)(t % Ho);
        const int n = (int)(t / Ho);
        const int64_t base = (((int64_t)n * H + 2 * oy) * W + 2 * ox) * C4 + c4;
        const int64_t offs[4] = {0, C4, (int64_t)W * C4, (int64_t)W * C4 + C4};
        f32x4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = ld4<XB>(x, base + offs[k]);
        const f32x4 g = ld4<GB>(dy, e);
        f32x4 o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int am = 0;
            float mv = v[0][j];
#pragma unroll
            for (int k = 1; k < 4; ++k)
                if (v[k][j] > mv) { mv = v[k][j]; am = k; }
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k][j] = (k == am && mv > 0.f) ? g[j] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) st4<DB>(dx, base + offs[k], o[k]);
    }
}

// odd H/W: the last row / column is not covered by any window -> zero gradient
__global__ void maxpool2_bwd_edge_kernel(float* __restrict__ dx, int N, int H, int W, int C, int dx_bf16) {
    const int64_t total = (int64_t)N * H * W * C;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t pix = e / C;
        const int xw = (int)(pix % W), yh = (int)((pix / W) % H);
        if (((H & 1) && yh == H - 1) || ((W & 1) && xw == W - 1)) st_elem(dx, e, dx_bf16 != 0, 0.f);
    }
}

__global__ void add_relu_masked_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                       const float* __restrict__ ref, float* __restrict__ out, int64_t n, int dt) {
    const bool ab = dt & 1, bb = dt & 2, rb = dt & 4, ob = dt & 8;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float m = ld_elem(ref, i, rb) > 0.f ? ld_elem(b, i, bb) : 0.f;
        st_elem(out, i, ob, (a != nullptr ? ld_elem(a, i, ab) : 0.f) + m);
    }
}

static inline hipStream_t S_(void* s) { return reinterpret_cast<hipStream_t>(s); }

extern "C" int sisr_maxpool2_fwd(const float* x, float* y, int32_t N, int32_t H, int32_t W, int32_t C, int32_t dt,
                                 void* stream) {
    if (!x || !y || N <= 0 || H < 2 || W < 2 || C <= 0 || (C & 3)) return SISR_E_BADARG;
    const int Ho = H / 2, Wo = W / 2, C4 = C / 4;
    const int64_t total = (int64_t)N * Ho * Wo * C4;
    const int blocks = (int)std::min<int64_t>((total + 255) / 256, 4096);
    switch (dt & 3) {
        case 0: hipLaunchKernelGGL((maxpool2_fwd_kernel<false, false>), dim3(blocks), dim3(256), 0, S_(stream), x, y, H, W, C4, Ho, Wo, total); break;
        case 1: hipLaunchKernelGGL((maxpool2_fwd_kernel<true, false>), dim3(blocks), dim3(256), 0, S_(stream), x, y, H, W, C4, Ho, Wo, total); break;
        case 2: hipLaunchKernelGGL((maxpool2_fwd_kernel<false, true>), dim3(blocks), dim3(256), 0, S_(stream), x, y, H, W, C4, Ho, Wo, total); break;
        default: hipLaunchKernelGGL((maxpool2_fwd_kernel<true, true>), dim3(blocks), dim3(256), 0, S_(stream), x, y, H, W, C4, Ho, Wo, total); break;
    }
    SISR_CHECK_LAUNCH();
    return 0;
}

extern "C" int sisr_maxpool2_relu_bwd(const float* dy, const float* x, float* dx, int32_t N, int32_t H, int32_t W,
                                      int32_t C, int32_t dt, void* stream) {
    if (!dy || !x || !dx || N <= 0 || H < 2 || W < 2 || C <= 0 || (C & 3)) return SISR_E_BADARG;
    const int Ho = H / 2, Wo = W / 2, C4 = C / 4;
    const int64_t total = (int64_t)N * Ho * Wo * C4;
    const int blocks = (int)std::min<int64_t>((total + 255) / 256, 4096);
    if ((dt & 7) == 0)
        hipLaunchKernelGGL((maxpool2_relu_bwd_kernel<false, false, false>), dim3(blocks), dim3(256), 0, S_(stream), dy, x, dx, H, W, C4, Ho, Wo, total);
    else if ((dt & 7) == 7)
        hipLaunchKernelGGL((maxpool2_relu_bwd_kernel<true, true, true>), dim3(blocks), dim3(256), 0, S_(stream), dy, x, dx, H, W, C4, Ho, Wo, total);
    else
        return SISR_E_UNSUPPORTED;         // all fp32 or all bf16
    SISR_CHECK_LAUNCH();
    if ((H & 1) || (W & 1)) {
        hipLaunchKernelGGL(maxpool2_bwd_edge_kernel, dim3(1024), dim3(256), 0, S_(stream), dx, N, H, W, C, (dt >> 2) & 1);
        SISR_CHECK_LAUNCH();
    }
    return 0;
}

extern "C" int sisr_add_relu_masked(const float* a, const float* b, const float* ref, float* out, int64_t n,
                                    int32_t dt, void* stream) {
    if (!b || !ref || !out || n <= 0) return SISR_E_BADARG;
    const int blocks = (int)std::min<int64_t>((n + 255) / 256, 4096);
    hipLaunchKernelGGL(add_relu_masked_kernel, dim3(blocks), dim3(256), 0, S_(stream), a, b, ref, out, n, dt);
    SISR_CHECK_LAUNCH();
    return 0;
}

extern "C" int sisr_nhwc_to_nchw(const float* x, const float* pa, const float* pd, const float* slope_p,
                                 float slope, float* y, int64_t dst_stride, int32_t N, int32_t H, int32_t W,
                                 int32_t C, int32_t x_bf16, void* stream) {
    if (!x || !y || N <= 0 || H <= 0 || W <= 0 || C <= 0 || (pa && !pd) || dst_stride < (int64_t)C * H * W)
        return SISR_E_BADARG;
    const int HW = H * W;
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3((HW + 31) / 32, (C + 31) / 32, N), dim3(SISR_BLOCK), 0, S_(stream),
                       x, pa, pd, slope_p, slope, y, dst_stride, HW, C, x_bf16);
    SISR_CHECK_LAUNCH();
    return 0;
}

extern "C" int sisr_nchw_to_nhwc(const float* x, int64_t src_stride, float* y, int32_t N, int32_t H, int32_t W,
                                 int32_t C, int32_t y_bf16, void* stream) {
    if (!x || !y || N <= 0 || H <= 0 || W <= 0 || C <= 0 || src_stride < (int64_t)C * H * W) return SISR_E_BADARG;
    const int HW = H * W;
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3((HW + 31) / 32, (C + 31) / 32, N), dim3(SISR_BLOCK), 0, S_(stream),
                       x, src_stride, y, HW, C, y_bf16);
    SISR_CHECK_LAUNCH();
    return 0;
}

extern "C" int sisr_nchw_grad_to_nhwc4(const float* dy, const float* out, float* g, int32_t N, int32_t C, int32_t H,
                                       int32_t W, int32_t Cpad, void* stream) {
    if (!dy || !g || N <= 0 || C <= 0 || H <= 0 || W <= 0 || Cpad < C || (Cpad & 3)) return SISR_E_BADARG;
    const int64_t HW = (int64_t)H * W, total = (int64_t)N * HW;
    const int grid = (int)std::min<int64_t>((total + SISR_BLOCK - 1) / SISR_BLOCK, 4096);
    hipLaunchKernelGGL(nchw_grad_to_nhwc4_kernel, dim3(grid), dim3(SISR_BLOCK), 0, S_(stream), dy, out, g, HW, C, Cpad, total);
    SISR_CHECK_LAUNCH();
    return 0;
}

extern "C" int sisr_fc_forward(const float* x, float in_slope, const float* W, const float* bias, float* y, int32_t B,
                               int32_t K, int32_t Nout, int32_t epi, void* stream) {
    if (!x || !W || !y || B <= 0 || B > FC_B || K <= 0 || (K & 3) || Nout <= 0) return SISR_E_BADARG;
    hipLaunchKernelGGL(fc_forward_kernel, dim3((Nout + FC_R - 1) / FC_R), dim3(SISR_BLOCK), 0, S_(stream), x,
                       in_slope, W, bias, y, B, K, Nout, epi);
    SISR_CHECK_LAUNCH();
    return 0;
}

static int fc_rows_per_split(int K, int Nout) {
    // enough workgroups to fill 256 CUs twice: blocks_k * splits >= 512
    const int blocks_k = ((K >> 2) + SISR_BLOCK - 1) / SISR_BLOCK;
    int splits = std::max(1, std::min(Nout, (512 + blocks_k - 1) / blocks_k));
    int rows = (Nout + splits - 1) / splits;
    return std::min(rows, 256);
}

extern "C" int sisr_fc_dgrad_splits(int32_t K, int32_t Nout) {
    const int rows = fc_rows_per_split(K, Nout);
    return (Nout + rows - 1) / rows;
}

extern "C" int sisr_fc_dgrad(const float* dy, const float* W, float* dx, float* work, int32_t B, int32_t K,
                             int32_t Nout, void* stream) {
    if (!dy || !W || !dx || !work || B <= 0 || B > FC_B || K <= 0 || (K & 3) || Nout <= 0) return SISR_E_BADARG;
    const int rows = fc_rows_per_split(K, Nout);
    const int splits = (Nout + rows - 1) / rows;
    const int blocks_k = ((K >> 2) + SISR_BLOCK - 1) / SISR_BLOCK;
    hipLaunchKernelGGL(fc_dgrad_kernel, dim3(blocks_k, splits), dim3(SISR_BLOCK), rows * FC_B * 4, S_(stream), dy, W,
                       work, B, K, Nout, rows);
    SISR_CHECK_LAUNCH();
    const int64_t elems = (int64_t)B * K;
    const int blocks = (int)std::min<int64_t>((elems / 4 + 255) / 256, 2048);
    hipLaunchKernelGGL(fc_split_reduce_kernel, dim3(blocks), dim3(256), 0, S_(stream), work, dx, splits, elems);
    SISR_CHECK_LAUNCH();
    return 0;
}

extern "C" int sisr_fc_wgrad(const float* dy, const float* x, float in_slope, float* dW, float* db, int32_t B,
                             int32_t K, int32_t Nout, void* stream) {
    if (!dy || !x || !dW || B <= 0 || B > FC_B || K <= 0 || (K & 3) || Nout <= 0) return SISR_E_BADARG;
    const int rows = fc_rows_per_split(K, Nout);
    const int splits = (Nout + rows - 1) / rows;
    const int blocks_k = ((K >> 2) + SISR_BLOCK - 1) / SISR_BLOCK;
    hipLaunchKernelGGL(fc_wgrad_kernel, dim3(blocks_k, splits), dim3(SISR_BLOCK), rows * FC_B * 4, S_(stream), dy, x,
                       in_slope, dW, B, K, Nout, rows);
    SISR_CHECK_LAUNCH();
    if (db != nullptr) {
        hipLaunchKernelGGL(fc_bias_grad_kernel, dim3((Nout + 255) / 256), dim3(256), 0, S_(stream), dy, db, B, Nout);
        SISR_CHECK_LAUNCH();
    }
    return 0;
}

extern "C" int sisr_act_bwd(const float* dy, const float* ref, float* out, int64_t n, int32_t kind, float slope,
                            void* stream) {
    if (!dy || !ref || !out || n <= 0 || kind < 0 || kind > 1) return SISR_E_BADARG;
    const int blocks = (int)std::min<int64_t>((n + 255) / 256, 2048);
    hipLaunchKernelGGL(act_bwd_kernel, dim3(blocks), dim3(256), 0, S_(stream), dy, ref, out, n, kind, slope);
    SISR_CHECK_LAUNCH();
    return 0;
}
