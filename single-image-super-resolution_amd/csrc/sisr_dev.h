// sisr_dev.h -- device-side helpers shared by the gfx950 kernels (CDNA4: 64-lane wavefronts,
// v_mfma_f32_32x32x2_f32 for exact-fp32 contractions, 160 KiB LDS per CU).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/sisr_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define SISR_BLOCK 256

#define SISR_CHECK_LAUNCH()                          \
    do {                                             \
        hipError_t e__ = hipGetLastError();          \
        if (e__ != hipSuccess) return (int)e__;      \
    } while (0)

// leaky-relu family: PReLU (shared slope), LeakyReLU(0.01), ReLU (slope 0), identity (slope 1)
__device__ __forceinline__ float lrelu(float v, float slope) { return v > 0.f ? v : slope * v; }

// MFMA 32x32x2 fp32 lane maps (cdna_hip_programming.md section 3):
//   A operand: lane l holds A[i = l&31][k = l>>5];  B operand: lane l holds B[k = l>>5][j = l&31]
//   C/D: acc[reg] = C[row = (reg&3) + 8*(reg>>2) + 4*(l>>5)][col = l&31]
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ int mfma_row(int reg, int lane) {
    return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// block-wide sum for SISR_BLOCK threads; `scratch` >= 4 floats of LDS; result valid in all threads
__device__ __forceinline__ float block_sum(float v, float* scratch) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0.f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += scratch[w];
    return t;
}

// One input element of a (possibly lazily transformed) operand; see SISR_PRO_* in sisr_hip.h.
struct OperandView {
    const float *x1, *x2, *pa, *pb, *pd, *ps, *pt;
    int N, H, W, C;      // logical dims
    int mode;            // SISR_X_*
    int pro;             // SISR_PRO_*
    float slope;
};

__device__ __forceinline__ int64_t operand_offset(const OperandView& o, int n, int y, int x, int c) {
    if (o.mode == SISR_X_NHWC) return (((int64_t)n * o.H + y) * o.W + x) * o.C + c;
    if (o.mode == SISR_X_NCHW) return (((int64_t)n * o.C + c) * o.H + y) * o.W + x;
    // pixel-unshuffle(2) gather: logical channel c = ij*Cq + cc lives at (2y+(ij>>1), 2x+(ij&1), cc)
    const int Cq = o.C >> 2;
    const int ij = c / Cq, cc = c - ij * Cq;
    return (((int64_t)n * (2 * o.H) + 2 * y + (ij >> 1)) * (2 * o.W) + 2 * x + (ij & 1)) * Cq + cc;
}

__device__ __forceinline__ float operand_apply(const OperandView& o, float a, float b, int c) {
    switch (o.pro) {
        case SISR_PRO_NONE: return a;
        case SISR_PRO_ACT: return lrelu(a, o.slope);
        case SISR_PRO_AFFINE_ACT: return lrelu(o.pa[c] * a + o.pd[c], o.slope);
        case SISR_PRO_BNBWD: return o.pa[c] * a + o.pb[c] * b + o.pd[c];
        case SISR_PRO_BNACT_BWD: {
            const float z = o.ps[c] * b + o.pt[c];
            const float g = z > 0.f ? a : o.slope * a;
            return o.pa[c] * g + o.pb[c] * b + o.pd[c];
        }
        case SISR_PRO_ACT_BWD: return b > 0.f ? a : o.slope * a;
        case SISR_PRO_TANH_BWD: return a * (1.f - b * b);
    }
    return a;
}

__host__ __device__ __forceinline__ bool operand_needs_x2(int pro) {
    return pro == SISR_PRO_BNBWD || pro == SISR_PRO_BNACT_BWD || pro == SISR_PRO_ACT_BWD ||
           pro == SISR_PRO_TANH_BWD;
}

// Stage one channel chunk of an input halo tile into LDS as [pixel][PS] (PS odd => conflict-free
// per-lane b32 reads with lanes on consecutive pixels), applying the operand's prologue.  Pixels
// outside the image and channel slots >= CK (the pad slot) are written as 0.
//   tile pixels: TN x IH x IW, origin image n0, input row iy_org, col ix_org; tile columns
//   >= valid_w are forced to 0; `slack` (<= 64) extra floats after the tile are zeroed.
__device__ __forceinline__ void stage_operand_tile(const OperandView& o, float* lds, int PS, int CK,
                                                   int c0, int TN, int IH, int IW, int n0, int iy_org,
                                                   int ix_org, bool vec_ok, int valid_w, int slack) {
    const int tid = threadIdx.x;
    const int npix = TN * IH * IW;
    const bool need2 = operand_needs_x2(o.pro);
    if (vec_ok) {
        const int G = CK >> 2;
        const int items = npix * G;
        for (int it = tid; it < items; it += SISR_BLOCK) {
            const int pix = it / G, g = it - pix * G;
            const int tn = pix / (IH * IW), rem = pix - tn * (IH * IW);
            const int iyl = rem / IW, ixl = rem - iyl * IW;
            const int n = n0 + tn, iy = iy_org + iyl, ix = ix_org + ixl;
            const int c = c0 + g * 4;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (n < o.N && iy >= 0 && iy < o.H && ix >= 0 && ix < o.W && c < o.C && ixl < valid_w) {
                const int64_t off = operand_offset(o, n, iy, ix, c);
                const f32x4 a = *reinterpret_cast<const f32x4*>(o.x1 + off);
                f32x4 b = {0.f, 0.f, 0.f, 0.f};
                if (need2) b = *reinterpret_cast<const f32x4*>(o.x2 + off);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = operand_apply(o, a[j], b[j], c + j);
            }
            float* dst = lds + pix * PS + g * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) dst[j] = v[j];
            if (g == 0 && PS > CK) lds[pix * PS + CK] = 0.f;
        }
    } else {
        const int items = npix * PS;
        for (int it = tid; it < items; it += SISR_BLOCK) {
            const int pix = it / PS, cs = it - pix * PS;
            const int tn = pix / (IH * IW), rem = pix - tn * (IH * IW);
            const int iyl = rem / IW, ixl = rem - iyl * IW;
            const int n = n0 + tn, iy = iy_org + iyl, ix = ix_org + ixl;
            const int c = c0 + cs;
            float v = 0.f;
            if (cs < CK && c < o.C && n < o.N && iy >= 0 && iy < o.H && ix >= 0 && ix < o.W && ixl < valid_w) {
                const int64_t off = operand_offset(o, n, iy, ix, c);
                const float a = o.x1[off];
                const float b = need2 ? o.x2[off] : 0.f;
                v = operand_apply(o, a, b, c);
            }
            lds[it] = v;
        }
    }
    if (tid < slack) lds[npix * PS + tid] = 0.f;   // slack read by zero-weight / masked K tails
}

