// conv_toimage.hip -- the generator's LAST convolution (model_generator.py:52-53: 3x3, 64 -> 3 channels, stride 1, pad 1,
// + Tanh) in the bf16 build: bf16 NHWC activations in, NCHW fp32 image out.
//
// With 3 output channels the usual mapping (N = couts) leaves 29 of 32 MFMA columns empty and still walks 9 taps x 64
// channels of K per pixel: the generic kernel (conv_bf16.hip) takes 66 us for 75 MB of input
// (profiles/r02_trace_step_order.txt).  Here the convolution is split the transposed-convolution way:
//     P[pixel][(co, ky, kx)] = sum_ci x[pixel][ci] * w[co][ci][ky][kx]          a 1x1 convolution: K = 64, N = 27 of 32
//     out[pixel][co]        = bias[co] + sum_(ky,kx) P[pixel + (ky - 1, kx - 1)][(co, ky, kx)]             (col2im)
//   * the GEMM needs NO halo staging: a lane's B fragment (its pixel, 8 consecutive channels) is ONE 16-byte global
//     load straight into the MFMA operand registers (the PReLU of the prologue applied in registers); 4 MFMAs per 32
//     pixels instead of 36;
//   * a workgroup (256 threads) computes P for a region of 16 x 32 pixels (one region row per MFMA tile) into LDS
//     (fp32, 28 floats per pixel) and then gathers the 14 x 30 interior outputs: 9 LDS reads per output value, bias,
//     tanh, coalesced NCHW stores.  The 1-pixel border of the region is recomputed by the neighbouring workgroups
//     (1.22x the input reads, served by L2);
//   * no persistence: ~1,600 short workgroups, two per CU, overlap each other's load / MFMA / gather phases.
// Requirements (sisr_conv2d_toimage_eligible): Cin = 64, Cout = 3, 3x3, stride 1, pad 1, bf16 NHWC input with prologue
// NONE or ACT, NCHW fp32 output, epilogue NONE or TANH, no residual / statistics.
#include "sisr_dev.h"

#include <cstdlib>

#include "sisr_bf16_stage.h"

#define TO_RW 32                  // region columns (= one MFMA pixel tile)
#define TO_RH 16                  // region rows: 4 per wave
#define TO_OW (TO_RW - 2)
#define TO_OH (TO_RH - 2)
#define TO_PSTR 28                // floats per region pixel: 27 (co, ky, kx) sums + 1

struct ToImageArgs {
    const void *x, *wpk;
    const float *bias, *slope_p;
    float* y;
    float slope;
    int N, H, W, CoutPad, tanh_epi;
    int CK, PS, KROWP;                // fp32 variant: layout of the packed fp32 weight image
    int tiles_x, tiles_y, total, per_xcd;
};

// col2im gather of the 14 x 30 interior of a region, x fastest (coalesced NCHW stores)
__device__ __forceinline__ void toimage_gather(const ToImageArgs& a, const float* P, int n, int ty, int tx) {
    for (int idx = threadIdx.x; idx < 3 * TO_OH * TO_OW; idx += 256) {
        const int co = idx / (TO_OH * TO_OW), rem = idx - co * (TO_OH * TO_OW), oy = rem / TO_OW, ox = rem - oy * TO_OW;
        const int Y = ty * TO_OH + oy, Xo = tx * TO_OW + ox;
        float s = a.bias != nullptr ? a.bias[co] : 0.f;
        const float* p0 = P + (oy * TO_RW + ox) * TO_PSTR + co * 9;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) s += p0[(ky * TO_RW + kx) * TO_PSTR + ky * 3 + kx];
        if (a.tanh_epi) s = tanhf(s);
        if (Y < a.H && Xo < a.W) a.y[((long long)(n * 3 + co) * a.H + Y) * a.W + Xo] = s;
    }
}

template <bool ACT>
__global__ void __launch_bounds__(256, 2) conv_toimage_kernel(const ToImageArgs a) {
    __shared__ __attribute__((aligned(16))) float P[TO_RH * TO_RW * TO_PSTR];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, kk = lane >> 5;
    // workgroup b runs on XCD b % 8: every XCD gets a contiguous eighth of the regions, so the border pixels two
    // neighbouring regions both read come out of ONE L2
    const int tile = (blockIdx.x & 7) * a.per_xcd + (blockIdx.x >> 3);
    if (tile >= a.total) return;
    const int tx = tile % a.tiles_x, t2 = tile / a.tiles_x, ty = t2 % a.tiles_y, n = t2 / a.tiles_y;
    const float slope = a.slope_p ? a.slope_p[0] : a.slope;

    // ---- A fragments: row m = co * 9 + tap of the 27 x 64 weight matrix, packed bf16 image [chunk 32][cout][tap * 32 + cl]
    const __amdgpu_buffer_rsrc_t rw = sisr_rsrc(a.wpk, (unsigned)(2 * a.CoutPad * 288) * 2u);
    bf16x8 wf[4];
    {
        const int co = l31 / 9, tap = l31 - 9 * co;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const unsigned off = (unsigned)(((((s >> 1) * a.CoutPad + co) * 9 + tap) * 32 + 16 * (s & 1) + 8 * kk) * 2);
            wf[s] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rw, l31 < 27 ? off : 0x80000000u, 0, 0));
        }
    }
    // ---- B fragments: region pixel (row 4 wave + mt, column l31) = image pixel (ty * 14 - 1 + row, tx * 30 - 1 + column);
    // outside the image: offset 2^31 -> zeros, i.e. the convolution's zero padding (lrelu(0) = 0)
    const __amdgpu_buffer_rsrc_t rx = sisr_rsrc(a.x, (unsigned)a.N * (unsigned)(a.H * a.W) * 128u);
    u32x4 xf[4][4];
    const int X = tx * TO_OW - 1 + l31;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const int Y = ty * TO_OH - 1 + 4 * wave + mt;
        const int ok = (int)((unsigned)Y < (unsigned)a.H) & (int)((unsigned)X < (unsigned)a.W);
        const unsigned base = (unsigned)(((n * a.H + Y) * a.W + X) * 128 + 16 * kk);
#pragma unroll
        for (int s = 0; s < 4; ++s) xf[mt][s] = __builtin_amdgcn_raw_buffer_load_b128(rx, ok ? base + 32u * s : 0x80000000u, 0, 0);
    }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            u32x4 v = xf[mt][s];
            if (ACT) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float v0 = __uint_as_float(v[j] << 16), v1 = __uint_as_float(v[j] & 0xFFFF0000u);
                    v[j] = pack_bf16x2(v0 > 0.f ? v0 : slope * v0, v1 > 0.f ? v1 : slope * v1);
                }
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[s], __builtin_bit_cast(bf16x8, v), acc, 0, 0, 0);
        }
        // D[m][pixel]: register group q of a lane = rows m = 8q + 4kk .. +3 of its pixel
        float* pp = P + ((4 * wave + mt) * TO_RW + l31) * TO_PSTR + 4 * kk;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (q < 3 || kk == 0) *reinterpret_cast<f32x4*>(pp + 8 * q) = f32x4{acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
    }
    __syncthreads();
    toimage_gather(a, P, n, ty, tx);
}

// ---- the same layer with fp32 tensors (fp32 parity build): exact v_mfma_f32_32x32x2_f32, K = 64 channels in 32 steps.
// The generic fp32 kernel pads the 3 couts to 32 MFMA columns over K = 9 x 64: 255 us; here 2.4 GFLOP and 151 MB.
// K order: step (j, i) pairs channel 8j + i (lanes 0-31) with channel 8j + 4 + i (lanes 32-63), so that a lane's 32
// operand values are eight 16-byte loads of its pixel; the weight fragments use the same pairing.
template <bool ACT>
__global__ void __launch_bounds__(256, 2) conv_toimage_f32_kernel(const ToImageArgs a) {
    __shared__ __attribute__((aligned(16))) float P[TO_RH * TO_RW * TO_PSTR];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, kk = lane >> 5;
    // workgroup b runs on XCD b % 8: every XCD gets a contiguous eighth of the regions, so the border pixels two
    // neighbouring regions both read come out of ONE L2
    const int tile = (blockIdx.x & 7) * a.per_xcd + (blockIdx.x >> 3);
    if (tile >= a.total) return;
    const int tx = tile % a.tiles_x, t2 = tile / a.tiles_x, ty = t2 % a.tiles_y, n = t2 / a.tiles_y;
    const float slope = a.slope_p ? a.slope_p[0] : a.slope;
    // A fragments from the packed fp32 image [chunk][ky][cout][kx * PS + cl]
    float wa[8][4];
    {
        const int co = l31 / 9, tap = l31 - 9 * co, ky = tap / 3, kx = tap - 3 * ky;
        const float* wp = reinterpret_cast<const float*>(a.wpk);
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ci = 8 * j + 4 * kk + i, chunk = ci / a.CK, cl = ci - chunk * a.CK;
                wa[j][i] = l31 < 27 ? wp[((chunk * 3 + ky) * a.CoutPad + co) * a.KROWP + kx * a.PS + cl] : 0.f;
            }
    }
    const __amdgpu_buffer_rsrc_t rx = sisr_rsrc(a.x, (unsigned)a.N * (unsigned)(a.H * a.W) * 256u);
    f32x4 xf[4][8];
    const int X = tx * TO_OW - 1 + l31;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const int Y = ty * TO_OH - 1 + 4 * wave + mt;
        const int ok = (int)((unsigned)Y < (unsigned)a.H) & (int)((unsigned)X < (unsigned)a.W);
        const unsigned base = (unsigned)(((n * a.H + Y) * a.W + X) * 256 + 16 * kk);
#pragma unroll
        for (int j = 0; j < 8; ++j)
            xf[mt][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, ok ? base + 32u * j : 0x80000000u, 0, 0));
    }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float v = xf[mt][j][i];
                if (ACT) v = v > 0.f ? v : slope * v;
                acc = mfma32(wa[j][i], v, acc);
            }
        float* pp = P + ((4 * wave + mt) * TO_RW + l31) * TO_PSTR + 4 * kk;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (q < 3 || kk == 0) *reinterpret_cast<f32x4*>(pp + 8 * q) = f32x4{acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
    }
    __syncthreads();
    toimage_gather(a, P, n, ty, tx);
}

static int toimage_common(const SisrConvDesc* d) {
    const char* sw = getenv("SISR_THIN");                       // A/B switch: SISR_THIN=0 keeps the generic kernels
    if ((sw && sw[0] == '0') || !d) return 0;
    if (d->Cin != 64 || d->Cout != 3 || d->KH != 3 || d->KW != 3 || d->stride != 1 || d->pad_y != 1 || d->pad_x != 1) return 0;
    if (d->x_mode != SISR_X_NHWC || (d->pro_mode != SISR_PRO_NONE && d->pro_mode != SISR_PRO_ACT)) return 0;
    if (d->y_mode != SISR_Y_NCHW || d->y_bf16 || d->res || d->stat_part || d->bnb_part) return 0;
    if (d->epi_act != SISR_EPI_NONE && d->epi_act != SISR_EPI_TANH) return 0;
    if (d->Ho != d->H || d->Wo != d->W || d->plan.CoutPad < 3) return 0;
    if ((int64_t)d->N * d->H * d->W * 256 >= (1ll << 31)) return 0;
    return 1;
}

// bf16 tensors (behind sisr_conv2d_bf16, descriptor planned by sisr_conv2d_plan_bf16)
extern "C" int sisr_conv2d_toimage_eligible(const SisrConvDesc* d) {
    return toimage_common(d) && d->x_bf16 && d->plan.CK == 32;
}
// fp32 tensors (behind sisr_conv2d_f32, descriptor planned by sisr_conv2d_plan)
extern "C" int sisr_conv2d_toimage_f32_eligible(const SisrConvDesc* d) {
    return toimage_common(d) && !d->x_bf16 && d->plan.CK >= 4 && (64 % d->plan.CK) == 0 && d->plan.PS >= d->plan.CK &&
           d->plan.KROWP >= 3 * d->plan.PS - (d->plan.PS - d->plan.CK) && d->plan.n_chunk * d->plan.CK == 64;
}

int sisr_conv2d_toimage_launch(const SisrConvDesc* d, hipStream_t st) {
    ToImageArgs a;
    a.x = d->x1; a.wpk = d->wpk; a.bias = d->bias; a.y = d->y;
    a.slope_p = d->pro_slope_p; a.slope = d->pro_slope;
    a.N = d->N; a.H = d->H; a.W = d->W; a.CoutPad = d->plan.CoutPad;
    a.CK = d->plan.CK; a.PS = d->plan.PS; a.KROWP = d->plan.KROWP;
    a.tanh_epi = d->epi_act == SISR_EPI_TANH;
    a.tiles_x = (d->W + TO_OW - 1) / TO_OW;
    a.tiles_y = (d->H + TO_OH - 1) / TO_OH;
    a.total = a.tiles_x * a.tiles_y * d->N;
    a.per_xcd = (a.total + 7) / 8;
    const dim3 grid(8 * a.per_xcd), block(256);
    const bool act = d->pro_mode == SISR_PRO_ACT;
    if (d->x_bf16) {
        if (act) hipLaunchKernelGGL(conv_toimage_kernel<true>, grid, block, 0, st, a);
        else hipLaunchKernelGGL(conv_toimage_kernel<false>, grid, block, 0, st, a);
    } else {
        if (act) hipLaunchKernelGGL(conv_toimage_f32_kernel<true>, grid, block, 0, st, a);
        else hipLaunchKernelGGL(conv_toimage_f32_kernel<false>, grid, block, 0, st, a);
    }
    SISR_CHECK_LAUNCH();
    return 0;
}
