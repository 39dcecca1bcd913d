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

// ---- host side: per-device state ------------------------------------------------------------------------------------
// One process normally drives one GPU (one rank per device), but nothing below assumes it: the CU count and the
// dynamic-LDS caps are kept per device id, so a host that switches devices (the reference's nn.DataParallel,
// config.py:114-118) gets correct grids and attributes on each.
#define SISR_MAX_DEVICES 64
int sisr_device_index();                   // misc.hip: current device id, clamped to [0, SISR_MAX_DEVICES)
// misc.hip: workgroup slots a persistent kernel may fill = CUs of the CURRENT device.  SISR_PERSIST_MAX_WG=<n> caps it
// (test knob: a small cap makes a small input walk many tiles per workgroup, the schedule of the full-size launches)
int sisr_cu_slots();
struct SisrLdsCap { int v[SISR_MAX_DEVICES]; };
// raise hipFuncAttributeMaxDynamicSharedMemorySize of `fn` on the current device when `bytes` exceeds what was set
// (`base`: the cap a kernel starts with -- 64 KB without the attribute, 0 forces the first call to set it)
static inline int sisr_raise_lds_cap(SisrLdsCap& cap, const void* fn, int bytes, int base = 0) {
    int& cur = cap.v[sisr_device_index()];
    if (cur < base) cur = base;
    if (bytes <= cur) return 0;
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return (int)e;
    cur = bytes;
    return 0;
}

// fixed-order sum of per-workgroup partial slabs (slab_reduce_kernel, conv_wgrad.hip; also carried by the combined
// BatchNorm-backward finishing launch of norm.hip): 16 columns of 16 bytes x 16 slab splits per 256-thread workgroup --
// many short independent load chains beat few long ones
#define SR_COLS 16
#define SR_SPLITS (SISR_BLOCK / SR_COLS)
// lead (a multiple of 4, 0: none): the first `lead` elements of every row are stored as bf16 at the row's start (the persistent
// bf16 weight-gradient kernel's slabs: half the bytes written and re-read); the rest of the row -- the bias partials -- as fp32
// at their usual float offset
__device__ __forceinline__ void slab_reduce_block(const float* __restrict__ slab, float* __restrict__ out, int n_slabs,
                                                  int64_t elems, int block, f32x4 (*sh)[SR_COLS], int64_t lead = 0) {
    const int col = threadIdx.x & (SR_COLS - 1), split = threadIdx.x / SR_COLS;
    const int64_t i4 = (int64_t)block * SR_COLS + col;
    const int64_t n4 = elems >> 2;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (i4 < n4) {
        if (i4 * 4 < lead) {
            typedef unsigned sr_u32x2 __attribute__((ext_vector_type(2)));
#pragma unroll 8
            for (int k = split; k < n_slabs; k += SR_SPLITS) {
                const sr_u32x2 w = *reinterpret_cast<const sr_u32x2*>(reinterpret_cast<const unsigned short*>(slab + (int64_t)k * elems) + i4 * 4);
                s[0] += __uint_as_float(w[0] << 16); s[1] += __uint_as_float(w[0] & 0xFFFF0000u);
                s[2] += __uint_as_float(w[1] << 16); s[3] += __uint_as_float(w[1] & 0xFFFF0000u);
            }
        } else {
#pragma unroll 8
            for (int k = split; k < n_slabs; k += SR_SPLITS)
                s += *reinterpret_cast<const f32x4*>(slab + (int64_t)k * elems + i4 * 4);
        }
    }
    sh[split][col] = s;
    __syncthreads();
    if (split == 0 && i4 < n4) {
        f32x4 t = sh[0][col];
#pragma unroll
        for (int j = 1; j < SR_SPLITS; ++j) t += sh[j][col];
        *reinterpret_cast<f32x4*>(out + i4 * 4) = t;
    }
}

// the same sum for FEW slabs (n_slabs <= SR_SPLITS: above, every split then holds exactly one slab and the final loop adds them in
// slab order): one 16-byte column per THREAD, the slabs added in slab order -- bit-identical to slab_reduce_block, with 256
// columns per workgroup instead of 16 and every thread loading (a batch of layers planned for a share of the chip each leaves
// 1 .. 64 slabs of up to 4.7 MB: with 16 columns per workgroup that was 73,000 workgroups at 1.3 TB/s)
#define SR_FEW_COLS SISR_BLOCK
__device__ __forceinline__ void slab_reduce_block_few(const float* __restrict__ slab, float* __restrict__ out, int n_slabs,
                                                      int64_t elems, int block, int64_t lead = 0) {
    const int64_t i4 = (int64_t)block * SR_FEW_COLS + threadIdx.x;
    if (i4 >= (elems >> 2)) return;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (i4 * 4 < lead) {
        typedef unsigned sr_u32x2 __attribute__((ext_vector_type(2)));
        for (int k0 = 0; k0 < n_slabs; k0 += 4) {
            sr_u32x2 w[4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                w[u] = *reinterpret_cast<const sr_u32x2*>(reinterpret_cast<const unsigned short*>(slab + (int64_t)min(k0 + u, n_slabs - 1) * elems) + i4 * 4);
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (k0 + u < n_slabs) {
                    s[0] += __uint_as_float(w[u][0] << 16); s[1] += __uint_as_float(w[u][0] & 0xFFFF0000u);
                    s[2] += __uint_as_float(w[u][1] << 16); s[3] += __uint_as_float(w[u][1] & 0xFFFF0000u);
                }
        }
    } else {
        for (int k0 = 0; k0 < n_slabs; k0 += 4) {
            f32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const f32x4*>(slab + (int64_t)min(k0 + u, n_slabs - 1) * elems + i4 * 4);
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (k0 + u < n_slabs) s += v[u];
        }
    }
    *reinterpret_cast<f32x4*>(out + i4 * 4) = s;
}

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

// raw buffer resource over [p, p + bytes): loads / stores take 32-bit byte offsets, out-of-range ones are dropped
__device__ __forceinline__ __amdgpu_buffer_rsrc_t sisr_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}

// n / d for n, d < 2^16 with the host-side reciprocal m = ceil(2^32 / d) (0 encodes d = 1): one multiply-high
// instead of the ~25-instruction integer division sequence
__host__ __device__ __forceinline__ uint32_t fdiv_magic(int d) {
    return d <= 1 ? 0u : (uint32_t)((0x100000000ull + (uint32_t)d - 1) / (uint32_t)d);
}
__device__ __forceinline__ int fdiv(int n, uint32_t m) { return m ? (int)__umulhi((unsigned)n, m) : n; }

// ---- bf16 storage (activations / gradients kept as bf16 in HBM by the bf16 build; sisr_hip.h *_bf16 flags) -------
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
typedef float f32x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float bf16_bits_to_f32(unsigned short h) { return __uint_as_float((unsigned)h << 16); }
// round-to-nearest-even via the hardware conversion (keeps NaN a NaN: MI355X_MICROARCH.md, correctness table)
__device__ __forceinline__ unsigned short f32_to_bf16_bits(float v) {
    const __bf16 h = (__bf16)v;
    return __builtin_bit_cast(unsigned short, h);
}
// element i of a tensor stored as f32 or bf16
__device__ __forceinline__ float ld_elem(const void* p, int64_t i, bool bf) {
    return bf ? bf16_bits_to_f32(reinterpret_cast<const unsigned short*>(p)[i]) : reinterpret_cast<const float*>(p)[i];
}
__device__ __forceinline__ void st_elem(void* p, int64_t i, bool bf, float v) {
    if (bf) reinterpret_cast<unsigned short*>(p)[i] = f32_to_bf16_bits(v);
    else reinterpret_cast<float*>(p)[i] = v;
}
// elements [4*i4, 4*i4+4) (8- or 16-byte access)
template <bool BF>
__device__ __forceinline__ f32x4 ld4(const void* p, int64_t i4) {
    if (BF) {
        const u16x4 h = reinterpret_cast<const u16x4*>(p)[i4];
        f32x4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = bf16_bits_to_f32(h[j]);
        return v;
    }
    return reinterpret_cast<const f32x4*>(p)[i4];
}
template <bool BF>
__device__ __forceinline__ void st4(void* p, int64_t i4, const f32x4 v) {
    if (BF) {
        u16x4 h;
#pragma unroll
        for (int j = 0; j < 4; ++j) h[j] = f32_to_bf16_bits(v[j]);
        reinterpret_cast<u16x4*>(p)[i4] = h;
    } else {
        reinterpret_cast<f32x4*>(p)[i4] = v;
    }
}
// elements [8*i8, 8*i8+8) of a bf16 tensor (one 16-byte access)
__device__ __forceinline__ f32x8 ld8_bf16(const void* p, int64_t i8) {
    const u16x8 h = reinterpret_cast<const u16x8*>(p)[i8];
    f32x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = bf16_bits_to_f32(h[j]);
    return v;
}
__device__ __forceinline__ void st8_bf16(void* p, int64_t i8, const f32x8 v) {
    u16x8 h;
#pragma unroll
    for (int j = 0; j < 8; ++j) h[j] = f32_to_bf16_bits(v[j]);
    reinterpret_cast<u16x8*>(p)[i8] = h;
}

// One input element of a (possibly lazily transformed) operand; see SISR_PRO_* in sisr_hip.h.
struct OperandView {
    const float *x1, *x2, *pa, *pb, *pd, *ps, *pt;
    int N, H, W, C;      // logical dims
    int mode;            // SISR_X_*
    int pro;             // SISR_PRO_*
    float slope;
    int bf16 = 0;        // x1 / x2 are bf16 tensors (pointers still typed float*: only the address is used)
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
           pro == SISR_PRO_TANH_BWD || pro == SISR_PRO_RES_AFFINE;
}

// ---- lean float4 staging: prologue fixed at compile time, the thread's 4 channels are the same for
// every item it stages (items advance by 64 lanes and G = CK/4 divides 64), so the per-channel
// constants live in registers; addressing is 32-bit and row-relative.
template <int PRO>
__device__ __forceinline__ f32x4 apply4(const f32x4 a, const f32x4 b, const f32x4 ka, const f32x4 kb,
                                        const f32x4 kd, const f32x4 ks, const f32x4 kt, const float slope) {
    f32x4 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (PRO == SISR_PRO_NONE) v[j] = a[j];
        else if (PRO == SISR_PRO_ACT) v[j] = lrelu(a[j], slope);
        else if (PRO == SISR_PRO_AFFINE_ACT) v[j] = lrelu(ka[j] * a[j] + kd[j], slope);
        else if (PRO == SISR_PRO_BNBWD) v[j] = ka[j] * a[j] + kb[j] * b[j] + kd[j];
        else if (PRO == SISR_PRO_BNACT_BWD) {
            const float z = ks[j] * b[j] + kt[j];
            const float g = z > 0.f ? a[j] : slope * a[j];
            v[j] = ka[j] * g + kb[j] * b[j] + kd[j];
        } else if (PRO == SISR_PRO_ACT_BWD) v[j] = b[j] > 0.f ? a[j] : slope * a[j];
        else v[j] = a[j] * (1.f - b[j] * b[j]);
    }
    return v;
}

template <int PRO, int SBQ, bool XBF = false>
__device__ __forceinline__ void stage_tile_vec(const OperandView& o, float* lds, int PS, int CK, int c0,
                                               int TN, int IH, int IW, int n0, int iy_org, int ix_org,
                                               int valid_w) {
    constexpr bool need2 = PRO == SISR_PRO_BNBWD || PRO == SISR_PRO_BNACT_BWD || PRO == SISR_PRO_ACT_BWD ||
                           PRO == SISR_PRO_TANH_BWD;
    const int tid = threadIdx.x;
    const int G = CK >> 2;
    int lg = 0;
    while ((1 << lg) < G) ++lg;
    const int g = tid & (G - 1);
    const int c = c0 + g * 4;
    const bool c_ok = c < o.C;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    f32x4 ka = zero, kb = zero, kd = zero, ks = zero, kt = zero;
    if (c_ok) {
        if (PRO == SISR_PRO_AFFINE_ACT || PRO == SISR_PRO_BNBWD || PRO == SISR_PRO_BNACT_BWD) {
            ka = *reinterpret_cast<const f32x4*>(o.pa + c);
            kd = *reinterpret_cast<const f32x4*>(o.pd + c);
        }
        if (PRO == SISR_PRO_BNBWD || PRO == SISR_PRO_BNACT_BWD) kb = *reinterpret_cast<const f32x4*>(o.pb + c);
        if (PRO == SISR_PRO_BNACT_BWD) {
            ks = *reinterpret_cast<const f32x4*>(o.ps + c);
            kt = *reinterpret_cast<const f32x4*>(o.pt + c);
        }
    }
    // physical layout of the tensor behind the logical (N,H,W,C) view
    int coff = c, ysh = 0, xsh = 0, Cp = o.C, Wp = o.W, Hp = o.H, mul = 1;
    if (o.mode == SISR_X_NHWC_UNSHUFFLE2) {
        const int Cq = o.C >> 2;
        const int ij = c / Cq;
        coff = c - ij * Cq; ysh = ij >> 1; xsh = ij & 1; Cp = Cq; Wp = 2 * o.W; Hp = 2 * o.H; mul = 2;
    }
    const int xstep = mul * Cp;
    // flat item loop with batched loads (see sisr_bf16_stage.h): ~one memory latency per tile
    constexpr int SB = need2 ? SBQ / 2 : SBQ;      // loads in flight per thread and operand
    const int ppi = SISR_BLOCK >> lg;
    const int step_rows = ppi / IW, step_cols = ppi - step_rows * IW;
    const int rows = TN * IH, npix = rows * IW;
    int pix = tid >> lg;
    int row = pix / IW, ixl = pix - row * IW;
    for (; pix < npix; ) {
        f32x4 a[SB], b[need2 ? SB : 1];
        int lds_off[SB];
        bool live[SB], ok[SB];
#pragma unroll
        for (int u = 0; u < SB; ++u) {
            live[u] = pix < npix;
            int tn = 0, iyl = row;
            if (TN > 1) { tn = row / IH; iyl = row - tn * IH; }
            const int n = n0 + tn, iy = iy_org + iyl, ix = ix_org + ixl;
            ok[u] = live[u] && c_ok && n < o.N && iy >= 0 && iy < o.H && ix >= 0 && ix < o.W && ixl < valid_w;
            lds_off[u] = pix * PS + g * 4;
            a[u] = zero;
            if (need2) b[u] = zero;
            if (ok[u]) {
                const int off = ((n * Hp + iy * mul + ysh) * Wp + xsh) * Cp + coff + ix * xstep;   // multiple of 4
                a[u] = ld4<XBF>(o.x1, off >> 2);
                if (need2) b[u] = ld4<XBF>(o.x2, off >> 2);
            }
            pix += ppi; row += step_rows; ixl += step_cols;
            if (ixl >= IW) { ixl -= IW; ++row; }
        }
#pragma unroll
        for (int u = 0; u < SB; ++u) {
            if (live[u]) {
                f32x4 v = zero;
                if (ok[u]) v = apply4<PRO>(a[u], need2 ? b[u] : zero, ka, kb, kd, ks, kt, o.slope);
                float* dst = lds + lds_off[u];
                dst[0] = v[0]; dst[1] = v[1]; dst[2] = v[2]; dst[3] = v[3];
                if (g == 0 && PS > CK) lds[lds_off[u] + CK] = 0.f;
            }
        }
    }
}

// Stage one channel chunk of an input halo tile into LDS as [pixel][PS] (PS odd => conflict-free
// per-lane b32 reads with lanes on consecutive pixels), applying the operand's prologue.  Pixels
// outside the image and channel slots >= CK (the pad slot) are written as 0.
//   tile pixels: TN x IH x IW, origin image n0, input row iy_org, col ix_org; tile columns
//   >= valid_w are forced to 0; `slack` (<= 64) extra floats after the tile are zeroed.
template <int SBQ = 8>
__device__ __forceinline__ void stage_operand_tile(const OperandView& o, float* lds, int PS, int CK,
                                                   int c0, int TN, int IH, int IW, int n0, int iy_org,
                                                   int ix_org, bool vec_ok, int valid_w, int slack) {
    const int tid = threadIdx.x;
    const int npix = TN * IH * IW;
    if (vec_ok) {
        switch (o.pro) {
#define SISR_STAGE_CASE(P)                                                                                       \
    case P:                                                                                                      \
        if (o.bf16) stage_tile_vec<P, SBQ, true>(o, lds, PS, CK, c0, TN, IH, IW, n0, iy_org, ix_org, valid_w);   \
        else stage_tile_vec<P, SBQ, false>(o, lds, PS, CK, c0, TN, IH, IW, n0, iy_org, ix_org, valid_w);         \
        break;
            SISR_STAGE_CASE(SISR_PRO_NONE)
            SISR_STAGE_CASE(SISR_PRO_ACT)
            SISR_STAGE_CASE(SISR_PRO_AFFINE_ACT)
            SISR_STAGE_CASE(SISR_PRO_BNBWD)
            SISR_STAGE_CASE(SISR_PRO_BNACT_BWD)
            SISR_STAGE_CASE(SISR_PRO_ACT_BWD)
            SISR_STAGE_CASE(SISR_PRO_TANH_BWD)
#undef SISR_STAGE_CASE
        }
    } else {
        // scalar path (NCHW images, 1/3/4/16-channel layers): only the VALID channels of the chunk are
        // fetched -- items = pixels x valid channels, pixel index fastest for NCHW so the loads coalesce --
        // the remaining slots of each pixel are zero-filled by a cheap second loop.
        const bool need2 = operand_needs_x2(o.pro);
        const int cv = max(0, min(CK, o.C - c0));          // valid channels in this chunk
        const int hw = IH * IW;
        for (int it = tid; it < npix * cv; it += SISR_BLOCK) {
            int pix, cs;
            if (o.mode == SISR_X_NCHW) { cs = it / npix; pix = it - cs * npix; }
            else { pix = it / cv; cs = it - pix * cv; }
            const int tn = pix / hw, rem = pix - tn * hw;
            const int iyl = rem / IW, ixl = rem - iyl * IW;
            const int n = n0 + tn, iy = iy_org + iyl, ix = ix_org + ixl;
            const int c = c0 + cs;
            float v = 0.f;
            if (n < o.N && iy >= 0 && iy < o.H && ix >= 0 && ix < o.W && ixl < valid_w) {
                const int64_t off = operand_offset(o, n, iy, ix, c);
                const float a = ld_elem(o.x1, off, o.bf16);
                const float b = need2 ? ld_elem(o.x2, off, o.bf16) : 0.f;
                v = operand_apply(o, a, b, c);
            }
            lds[pix * PS + cs] = v;
        }
        const int pad = PS - cv;
        for (int it = tid; it < npix * pad; it += SISR_BLOCK) {
            const int pix = it / pad, cs = cv + (it - pix * pad);
            lds[pix * PS + cs] = 0.f;
        }
    }
    if (tid < slack) lds[npix * PS + tid] = 0.f;   // slack read by zero-weight / masked K tails
}

// ---- deferred BatchNorm finalisation inside a consuming kernel (SisrConvDesc.fin_*) ---------------------------------------
// 512 threads, 64 channels.  Every workgroup merges the statistics rows ([rows][2][64] mean / M2, [rows] counts) of the
// BatchNorm whose apply its prologue is: shifted sums in double (shift = the first row's mean: N = sum n_b,
// S = sum n_b (mean_b - K), Q = sum M2_b + n_b (mean_b - K)^2; mean = K + S / N, M2 = Q - S^2 / N), thread = (channel,
// one of 8 row splits), splits combined through LDS in a fixed order.  Leaves scale / shift in kfin[0..63] / [64..127]
// (LDS); the `writer` workgroup also stores scale, shift, mean, invstd to k [4][64] and updates the running statistics as
// sisr_bn_finalize does (norm.hip: momentum, unbiased variance).  scratch: LDS, 32 * 64 * 3 doubles (48 KB).  Three barriers.
struct BnFinArgs {
    const float *stat, *cnt, *gamma, *beta;
    float *rm, *rv, *k;
    int rows;
    float momentum, eps;
};
// NT = threads of the workgroup (512: two waves per SIMD; 256: the one-wave-per-SIMD kernels): NT / 16 row splits, summed in
// 4 groups through LDS (scratch: NT / 16 x 64 x 3 doubles)
template <int NT = 512>
__device__ __forceinline__ void bn_finalize_in_kernel(const BnFinArgs& f, double* scratch, float* kfin, bool writer) {
    constexpr int NSPLIT = NT / 16, PER_GROUP = NSPLIT / 4;
    // thread = (4 consecutive channels c4, one of 32 row splits): 16-byte loads, ~8 rows per thread at 231 rows -- two
    // memory round trips instead of the eight of a (channel, 8 splits) mapping; the 32 partial sums of a channel are
    // then added in two fixed-order stages (4 groups of 8 through LDS)
    const int c4 = threadIdx.x & 15, split32 = threadIdx.x >> 4;
    const f32x4 K4 = *reinterpret_cast<const f32x4*>(f.stat + 4 * c4);
    double N4 = 0.0, S4[4] = {0.0, 0.0, 0.0, 0.0}, Q4[4] = {0.0, 0.0, 0.0, 0.0};
#ifndef BNFIN_UNROLL
#define BNFIN_UNROLL 4
#endif
#pragma unroll BNFIN_UNROLL
    for (int t = split32; t < f.rows; t += NSPLIT) {
        const double nb = (double)f.cnt[t];
        const f32x4 mb = *reinterpret_cast<const f32x4*>(f.stat + (int64_t)t * 128 + 4 * c4);
        const f32x4 qb = *reinterpret_cast<const f32x4*>(f.stat + (int64_t)t * 128 + 64 + 4 * c4);
        N4 += nb;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const double dm = (double)mb[j] - (double)K4[j];
            S4[j] += nb * dm;
            Q4[j] += (double)qb[j] + nb * dm * dm;
        }
    }
    // stage 1: LDS [32 splits][64 channels][3]; thread (channel c, group of 8 splits) adds its 8 in order
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        double* w = scratch + (split32 * 64 + 4 * c4 + j) * 3;
        w[0] = N4; w[1] = S4[j]; w[2] = Q4[j];
    }
    __syncthreads();
    const int c = threadIdx.x & 63, split = threadIdx.x >> 6;          // split = group of PER_GROUP row splits (groups 4 .. idle)
    const double K = (double)f.stat[c];
    double N = 0.0, S = 0.0, Q = 0.0;
    if (split < 4) {
#pragma unroll
        for (int j = 0; j < PER_GROUP; ++j) {
            const double* r = scratch + ((PER_GROUP * split + j) * 64 + c) * 3;
            N += r[0]; S += r[1]; Q += r[2];
        }
    }
    __syncthreads();                                                   // stage-1 rows are consumed: reuse the scratch
    double* my = scratch + (split * 64 + c) * 3;
    my[0] = N; my[1] = S; my[2] = Q;
    __syncthreads();
    if (threadIdx.x < 64) {
        N = 0.0; S = 0.0; Q = 0.0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const double* r = scratch + (j * 64 + c) * 3;
            N += r[0]; S += r[1]; Q += r[2];
        }
        const double mean = K + S / N;
        const double m2 = Q - S * S / N;
        const double var = m2 / N;                                     // biased (normalisation)
        const float invstd = (float)(1.0 / sqrt(var + (double)f.eps));
        const float sc = f.gamma[c] * invstd;
        const float sh = f.beta[c] - (float)mean * sc;
        kfin[c] = sc;
        kfin[64 + c] = sh;
        if (writer) {
            f.k[c] = sc; f.k[64 + c] = sh; f.k[128 + c] = (float)mean; f.k[192 + c] = invstd;
            const double unb = N > 1.0 ? m2 / (N - 1.0) : var;         // unbiased (running estimate)
            f.rm[c] = (1.f - f.momentum) * f.rm[c] + f.momentum * (float)mean;
            f.rv[c] = (1.f - f.momentum) * f.rv[c] + f.momentum * (float)unb;
        }
    }
    __syncthreads();
}
