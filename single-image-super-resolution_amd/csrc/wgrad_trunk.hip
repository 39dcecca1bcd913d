// wgrad_trunk.hip -- weight gradient of the generator's trunk geometry (3x3, 64 -> 64, stride 1, pad 1, bf16 NHWC
// tensors, H % 8 == 0, W % 16 == 0), the third role of the persistent trunk kernels (conv_trunk.hip holds the forward
// and data-gradient roles).  Reference path: the autograd weight gradient of every nn.Conv2d(64, 64, 3, 1, 1) of
// model_generator.py:29-55 (residual blocks) and :89-93 (trunk end).
//
//   dW[co][tap][ci] = sum over pixels p of dy[p][co] * x[p + tap][ci],      db[co] = sum_p dy[p][co]
//
// both operands lazy: x = conv input under a NONE / ACT / AFFINE_ACT prologue, dy = the gradient under the two-tensor
// BatchNorm-backward prologue (BNBWD / BNACT_BWD), exactly as the generic kernel (wgrad_bf16.hip) takes them.
//
// Why a second kernel: the generic kernel splits the input channels over workgroups (dy staged twice), starts 512
// workgroups that each write a 147 KB partial slab and serialises staging and MFMA inside a workgroup.  Here
//   * one workgroup per CU walks its share of the 8 x 16 pixel tiles and keeps the WHOLE 64 x 576 gradient in the
//     accumulators of its four consumer waves (9 taps x 16 registers each) across all of its tiles: one slab per
//     workgroup (<= 256), written once;
//   * four producer waves stage the next tile (both prologues, bias sums on the fly) into the other LDS buffer while
//     the consumers multiply the current one: one barrier per tile;
//   * the contraction runs over pixels, so both MFMA operands are fetched with transposing LDS reads from pixel-major
//     images (pixel stride 192 bytes: the 4 pixel rows of a read sit 48 banks apart).  An x fragment (halo row R,
//     column shift kx) serves the three taps (ky, kx) of tile rows R - ky: 76 reads feed the 72 MFMAs of a tile.
// Slab layout as the generic kernel's ([chunk][tap][ci 32][64 co] + bias row), so the reduction and un-packing
// kernels are shared.
#include "sisr_bf16_stage.h"

#include <algorithm>
#include <cstdlib>

#define WT_TH 8
#define WT_TW 16
#define WT_IH (WT_TH + 2)
#define WT_IW (WT_TW + 2)
#define WT_NPIX (WT_IH * WT_IW)            // 180 halo pixels
#define WT_PS 192                           // LDS bytes per pixel: 64 bf16 + 64 bytes (bank spread of the transposing reads)
#define WT_XBYTES (WT_NPIX * WT_PS)         // 34560
#define WT_DBYTES (128 * WT_PS)             // 24576
#define WT_XITEMS ((WT_NPIX * 8 + 255) / 256)   // 6
#define WT_THREADS 512

struct WTrunkArgs {
    const void *x1, *g1, *g2;
    const float *pa, *pd;                   // x prologue (AFFINE_ACT)
    const float* xslope_p; float xslope;
    const float *qa, *qb, *qd, *qs, *qt;    // dy prologue
    const float* gslope_p; float gslope;
    float *slab, *bias_slab;
    int64_t slab_stride;
    int N, H, W;
    int tiles_x, per_img, total;
    uint32_t m_tiles_x, m_per_img;
    int xpro;
};

__device__ __forceinline__ bf16x8 wt_frag(const unsigned char* p) {
    const s16x4 lo = lds_tr16(reinterpret_cast<const __bf16*>(p)), hi = lds_tr16(reinterpret_cast<const __bf16*>(p + 4 * WT_PS));
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

template <int GPRO>
__global__ void __launch_bounds__(WT_THREADS, 2) wgrad_trunk_kernel(const WTrunkArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    // [2 buffers][x halo image | dy image], then the prologue constants
    float* kst = reinterpret_cast<float*>(lds + 2 * (WT_XBYTES + WT_DBYTES));      // xa, xd, qa, qb, qd, qs, qt: [7][64]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool consumer = wave < 4;
    const int l31 = lane & 31;
    const int h = wave & 1, gq = (wave >> 1) & 1;             // consumer: output-channel half, input-channel chunk
    const unsigned xbytes = (unsigned)a.N * (unsigned)a.H * (unsigned)a.W * 128u;
    auto tile_coords = [&](int T, int& n, int& ty, int& tx) {
        n = fdiv(T, a.m_per_img);
        const int rem = T - n * a.per_img;
        ty = fdiv(rem, a.m_tiles_x);
        tx = rem - ty * a.tiles_x;
    };

    if (tid < 64) {
        const bool aff = a.xpro == SISR_PRO_AFFINE_ACT;
        kst[tid] = aff ? a.pa[tid] : 1.f;
        kst[64 + tid] = aff ? a.pd[tid] : 0.f;
        kst[128 + tid] = a.qa[tid]; kst[192 + tid] = a.qb[tid]; kst[256 + tid] = a.qd[tid];
        kst[320 + tid] = GPRO == SISR_PRO_BNACT_BWD ? a.qs[tid] : 0.f;
        kst[384 + tid] = GPRO == SISR_PRO_BNACT_BWD ? a.qt[tid] : 0.f;
    }
    __syncthreads();

    // ---- consumer state: the whole gradient of (32 output channels) x (9 taps x 32 input channels) --------------------
    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    // transposing-read lane roles: 16-lane group grp -> (channel half grp & 1, pixel half grp >> 1 of the 16-pixel K
    // step); inside the group lane 4q + p addresses (pixel row q, channels 4p .. 4p + 3)
    const int grp = lane >> 4, tq = (lane & 15) >> 2, tp = lane & 3;
    const int rd_pix = (8 * (grp >> 1) + tq) * WT_PS + (16 * (grp & 1) + 4 * tp) * 2;
    // ---- producer state ------------------------------------------------------------------------------------------------
    const int ptid = tid & 255;
    float xslope = 1.f, gslope = 1.f;
    f32x8 bsum = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};        // bias gradient: this thread's 8 channels, its pixels
    u32x4 sx[WT_XITEMS], s1[4], s2[4];
    unsigned sok = 0;
    if (!consumer) {
        if (a.xpro != SISR_PRO_NONE) xslope = a.xslope_p ? a.xslope_p[0] : a.xslope;
        gslope = a.gslope_p ? a.gslope_p[0] : a.gslope;
    }

    // Producer schedule of one tile: every load of the tile in flight (56 registers), then the commits in issue order.
    int pn = 0, pty = 0, ptx = 0;
    auto issue_x = [&](int T) {
        const __amdgpu_buffer_rsrc_t rx = bf_rsrc(a.x1, xbytes);
        tile_coords(T, pn, pty, ptx);
        sok = 0;
        int pt_ = ptid;                              // (opaque: per-item index arithmetic stays inside the tile loop)
        asm volatile("" : "+v"(pt_));
        const int oct = pt_ & 7;
#pragma unroll
        for (int k = 0; k < WT_XITEMS; ++k) {
            const int px = (pt_ + k * 256) >> 3;
            const int py = px / WT_IW, pxx = px - py * WT_IW;
            const int iy = pty * WT_TH - 1 + py, ix = ptx * WT_TW - 1 + pxx;
            const bool ok = px < WT_NPIX && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
            const unsigned voff = ok ? (unsigned)(((pn * a.H + iy) * a.W + ix) * 128 + oct * 16) : 0x80000000u;
            sx[k] = __builtin_amdgcn_raw_buffer_load_b128(rx, voff, 0, 0);
            sok |= ok ? (1u << k) : 0u;
        }
    };
    auto issue_g = [&](int half) {
        const __amdgpu_buffer_rsrc_t r1 = bf_rsrc(a.g1, xbytes), r2 = bf_rsrc(a.g2, xbytes);
        int pt_ = ptid;
        asm volatile("" : "+v"(pt_));
        const int oct = pt_ & 7;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int m = (pt_ + (2 * half + k) * 256) >> 3;               // tile pixel: row m >> 4, column m & 15
            const unsigned voff = (unsigned)(((pn * a.H + pty * WT_TH + (m >> 4)) * a.W + ptx * WT_TW + (m & 15)) * 128 + oct * 16);
            s1[2 * half + k] = __builtin_amdgcn_raw_buffer_load_b128(r1, voff, 0, 0);
            s2[2 * half + k] = __builtin_amdgcn_raw_buffer_load_b128(r2, voff, 0, 0);
        }
    };
    auto commit_x = [&](int b) {
        unsigned char* xi = lds + b * (WT_XBYTES + WT_DBYTES);
        int pt_ = ptid;
        asm volatile("" : "+v"(pt_));
        const int oct = pt_ & 7;
        // lrelu(a x + d) (a = 1, d = 0, slope = 1 degenerate to ACT / NONE), zero outside the image
        const f32x8 ka = *reinterpret_cast<const f32x8*>(kst + oct * 8), kd = *reinterpret_cast<const f32x8*>(kst + 64 + oct * 8);
#pragma unroll
        for (int k = 0; k < WT_XITEMS; ++k) {
            const int px = (pt_ + k * 256) >> 3;
            const bool ok = (sok >> k) & 1u;
            u32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float v0 = __uint_as_float(sx[k][j] << 16), v1 = __uint_as_float(sx[k][j] & 0xFFFF0000u);
                const float r0 = lrelu(ka[2 * j] * v0 + kd[2 * j], xslope), r1 = lrelu(ka[2 * j + 1] * v1 + kd[2 * j + 1], xslope);
                o[j] = ok ? (f32_to_bf16_bits(r0) | (f32_to_bf16_bits(r1) << 16)) : 0u;
            }
            if (px < WT_NPIX) *reinterpret_cast<u32x4*>(xi + px * WT_PS + oct * 16) = o;
        }
    };
    auto commit_g = [&](int b, int half) {
        unsigned char* di = lds + b * (WT_XBYTES + WT_DBYTES) + WT_XBYTES;
        int pt_ = ptid;
        asm volatile("" : "+v"(pt_));
        const int oct = pt_ & 7;
        // BatchNorm backward (through the activation: sign of the re-derived pre-activation first, as a bit mask, so that
        // at most 24 constant registers are live at a time)
        unsigned zm = 0u;
        if (GPRO == SISR_PRO_BNACT_BWD) {
            const f32x8 ks = *reinterpret_cast<const f32x8*>(kst + 320 + oct * 8), kt = *reinterpret_cast<const f32x8*>(kst + 384 + oct * 8);
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const unsigned w = s2[2 * half + k][j];
                    const float b0 = __uint_as_float(w << 16), b1 = __uint_as_float(w & 0xFFFF0000u);
                    const unsigned p0 = ks[2 * j] * b0 + kt[2 * j] > 0.f ? 1u : 0u, p1 = ks[2 * j + 1] * b1 + kt[2 * j + 1] > 0.f ? 1u : 0u;
                    zm |= (p0 | (p1 << 1)) << (k * 8 + 2 * j);
                }
        }
        const f32x8 qa = *reinterpret_cast<const f32x8*>(kst + 128 + oct * 8), qb = *reinterpret_cast<const f32x8*>(kst + 192 + oct * 8),
                    qd = *reinterpret_cast<const f32x8*>(kst + 256 + oct * 8);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int m = (pt_ + (2 * half + k) * 256) >> 3;
            u32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned w1 = s1[2 * half + k][j], w2 = s2[2 * half + k][j];
                const float a0 = __uint_as_float(w1 << 16), a1 = __uint_as_float(w1 & 0xFFFF0000u);
                const float b0 = __uint_as_float(w2 << 16), b1 = __uint_as_float(w2 & 0xFFFF0000u);
                float g0 = a0, g1 = a1;
                if (GPRO == SISR_PRO_BNACT_BWD) {
                    g0 = (zm >> (k * 8 + 2 * j)) & 1u ? a0 : gslope * a0;
                    g1 = (zm >> (k * 8 + 2 * j + 1)) & 1u ? a1 : gslope * a1;
                }
                const float r0 = qa[2 * j] * g0 + qb[2 * j] * b0 + qd[2 * j];
                const float r1 = qa[2 * j + 1] * g1 + qb[2 * j + 1] * b1 + qd[2 * j + 1];
                bsum[2 * j] += r0; bsum[2 * j + 1] += r1;
                o[j] = f32_to_bf16_bits(r0) | (f32_to_bf16_bits(r1) << 16);
            }
            *reinterpret_cast<u32x4*>(di + m * WT_PS + oct * 16) = o;
        }
    };
    auto produce = [&](int T, int b) {
        issue_x(T);
        issue_g(0);
        issue_g(1);
        commit_x(b);
        commit_g(b, 0);
        commit_g(b, 1);
    };

    // Two role-specific tile loops with matching barrier counts (a barrier only counts arriving waves): written as ONE
    // loop with a role branch inside, the register allocator has to carry the consumers' 144 accumulator registers through
    // the producers' code, and spills them.
    float* sl = a.slab + (int64_t)blockIdx.x * a.slab_stride;
    if (!consumer) {
        int T = blockIdx.x;
        if (T < a.total) produce(T, 0);
        __syncthreads();
        int cur = 0;
        for (; T < a.total; T += gridDim.x, cur ^= 1) {
            const int Tn = T + gridDim.x;
            if (Tn < a.total) produce(Tn, cur ^ 1);
            __syncthreads();      // the next tile's images are complete; the consumers have finished reading this one
        }
        if (a.bias_slab != nullptr) *reinterpret_cast<f32x8*>(lds + ptid * 32) = bsum;     // the images are free by now
    } else {
        __syncthreads();
        int cur = 0;
        for (int T = blockIdx.x; T < a.total; T += gridDim.x, cur ^= 1) {
            // halo rows R = 0 .. 9: the three column shifts of row R against the gradient rows R, R - 1, R - 2
            const unsigned char* xb = lds + cur * (WT_XBYTES + WT_DBYTES) + rd_pix + 64 * gq;
            const unsigned char* db = lds + cur * (WT_XBYTES + WT_DBYTES) + WT_XBYTES + rd_pix + 64 * h;
            bf16x8 dyf[WT_TH];
#pragma unroll
            for (int R = 0; R < WT_IH; ++R) {
                if (R < WT_TH) dyf[R] = wt_frag(db + R * 16 * WT_PS);
                bf16x8 xf[3];
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) xf[kx] = wt_frag(xb + (R * WT_IW + kx) * WT_PS);
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const int r = R - ky;
                    if (r < 0 || r >= WT_TH) continue;
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx)
                        acc[ky * 3 + kx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf[kx], dyf[r], acc[ky * 3 + kx], 0, 0, 0);
                }
            }
            __syncthreads();
        }
        // one slab per workgroup: [chunk][tap][ci 32][64 co]
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) sl[((gq * 9 + t) * 32 + mfma_row(i, lane)) * 64 + 32 * h + l31] = acc[t][i];
    }
    if (a.bias_slab != nullptr) {                       // ... and its bias row
        __syncthreads();
        if (tid < 64) {
            const float* bs = reinterpret_cast<const float*>(lds);
            const int oct = tid >> 3, j = tid & 7;
            float s = 0.f;
            for (int i = 0; i < 32; ++i) s += bs[(oct + 8 * i) * 8 + j];                 // fixed order: deterministic
            a.bias_slab[(int64_t)blockIdx.x * a.slab_stride + tid] = s;
        }
    }
}

// ---- host ----------------------------------------------------------------------------------------------------------
static int wtrunk_grid(const SisrWgradDesc* d) {
    const int total = d->N * (d->H / WT_TH) * (d->W / WT_TW);
    static int cus = 0;                         // (one process drives one GPU: queried once)
    if (cus == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0)
            cus = v;
        else
            cus = 256;
    }
    const int rounds = (total + cus - 1) / cus;
    return (total + rounds - 1) / rounds;       // equal shares
}

extern "C" int sisr_wgrad_trunk_eligible(const SisrWgradDesc* d) {
    const char* sw = getenv("SISR_TRUNK");                      // A/B switch: SISR_TRUNK=0 keeps the generic kernel
    if (!d || (sw && sw[0] == '0')) return 0;
    const char* sw2 = getenv("SISR_TRUNK_WGRAD");
    if (sw2 && sw2[0] == '0') return 0;
    if (d->Cin != 64 || d->Cout != 64 || d->KH != 3 || d->KW != 3 || d->stride != 1 || d->pad_y != 1 || d->pad_x != 1) return 0;
    if (d->x_mode != SISR_X_NHWC || d->g_mode != SISR_X_NHWC || !d->x_bf16 || !d->g_bf16) return 0;
    if (d->Ho != d->H || d->Wo != d->W || (d->H % WT_TH) || (d->W % WT_TW)) return 0;
    if (d->CoutPad != 64 || d->n_chunk != 2 || d->KROWP != 9 * 32) return 0;
    if ((int64_t)d->N * d->H * d->W * 128 >= (1ll << 31)) return 0;
    if (d->N * (d->H / WT_TH) * (d->W / WT_TW) >= 65536) return 0;
    const bool xp = d->pro_mode == SISR_PRO_NONE || d->pro_mode == SISR_PRO_ACT || d->pro_mode == SISR_PRO_AFFINE_ACT;
    const bool gp = d->gpro_mode == SISR_PRO_BNBWD || d->gpro_mode == SISR_PRO_BNACT_BWD;
    return xp && gp ? 1 : 0;
}

// slabs a launch of this descriptor writes (rows of `slab` at slab_stride): one per workgroup
extern "C" int sisr_wgrad_bf16_slabs(const SisrWgradDesc* d) {
    if (!d) return SISR_E_BADARG;
    return sisr_wgrad_trunk_eligible(d) ? wtrunk_grid(d) : d->n_slabs;
}

template <int GPRO>
static int launch_wtrunk(const WTrunkArgs& a, int grid, hipStream_t st) {
    constexpr int lds_bytes = 2 * (WT_XBYTES + WT_DBYTES) + 7 * 64 * 4;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_trunk_kernel<GPRO>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    hipLaunchKernelGGL((wgrad_trunk_kernel<GPRO>), dim3(grid), dim3(WT_THREADS), lds_bytes, st, a);
    SISR_CHECK_LAUNCH();
    return 0;
}

// called by sisr_conv2d_wgrad_bf16 for eligible descriptors
int sisr_wgrad_trunk_launch(const SisrWgradDesc* d, hipStream_t st) {
    if (operand_needs_x2(d->gpro_mode) && !d->g2) return SISR_E_BADARG;
    if (d->pro_mode == SISR_PRO_AFFINE_ACT && (!d->pa || !d->pd)) return SISR_E_BADARG;
    if (!d->qa || !d->qb || !d->qd || (d->gpro_mode == SISR_PRO_BNACT_BWD && (!d->qs || !d->qt))) return SISR_E_BADARG;
    WTrunkArgs a;
    a.x1 = d->x1; a.g1 = d->g1; a.g2 = d->g2;
    a.pa = d->pa; a.pd = d->pd; a.xslope_p = d->pro_slope_p; a.xslope = d->pro_slope;
    a.qa = d->qa; a.qb = d->qb; a.qd = d->qd; a.qs = d->qs; a.qt = d->qt;
    a.gslope_p = d->gpro_slope_p; a.gslope = d->gpro_slope;
    a.slab = d->slab; a.bias_slab = d->bias_slab; a.slab_stride = d->slab_stride;
    a.N = d->N; a.H = d->H; a.W = d->W;
    a.tiles_x = d->W / WT_TW; a.per_img = (d->H / WT_TH) * a.tiles_x; a.total = d->N * a.per_img;
    a.m_tiles_x = fdiv_magic(a.tiles_x); a.m_per_img = fdiv_magic(a.per_img);
    a.xpro = d->pro_mode;
    const int grid = wtrunk_grid(d);
    if (d->gpro_mode == SISR_PRO_BNBWD) return launch_wtrunk<SISR_PRO_BNBWD>(a, grid, st);
    return launch_wtrunk<SISR_PRO_BNACT_BWD>(a, grid, st);
}
