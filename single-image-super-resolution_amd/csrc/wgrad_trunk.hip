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
//     the consumers multiply the current one: one barrier per tile (the consumers stage the x operand themselves, after
//     their MFMAs: see the kernel);
//   * the contraction runs over pixels, so both MFMA operands are fetched with transposing LDS reads from pixel-major
//     images (pixel stride 192 bytes: the 4 pixel rows of a read sit 48 banks apart).  An x fragment (halo row R,
//     column shift kx) serves the three taps (ky, kx) of tile rows R - ky: 76 reads feed the 72 MFMAs of a tile.
// Slab layout as the generic kernel's ([chunk][tap][ci 32][64 co] + bias row), so the reduction and un-packing
// kernels are shared.
#include "sisr_bf16_stage.h"

#include <algorithm>
#include <cstring>
#include <type_traits>
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

// phase timeline, developer build only (make trace; tools/trace_trunk.py with ROLE=wgrad)
#ifdef SISR_CONV_TRACE
__device__ unsigned long long sisr_wttrace_buf[512 * 128];
#define WTT(k) do { if (threadIdx.x == 0 && blockIdx.x < 512 && (k) < 64) sisr_wttrace_buf[blockIdx.x * 128 + (k)] = wall_clock64(); } while (0)
#define WTTP(k) do { if (threadIdx.x == 256 && blockIdx.x < 512 && (k) < 64) sisr_wttrace_buf[blockIdx.x * 128 + 64 + (k)] = wall_clock64(); } while (0)
extern "C" int sisr_wttrace_read(void* dst, int n_u64) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(sisr_wttrace_buf), (size_t)n_u64 * 8, 0, hipMemcpyDeviceToHost);
}
#else
#define WTT(k)
#define WTTP(k)
#endif

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
    // Cout = 256 with the gradient stored shuffled (the upscale conv, model_generator.py:43-48): four cout groups = the four
    // PixelShuffle phases, workgroup b serves group b % 4 on tile stream b / 4; the gradient operand of group (i, j) is the
    // strided view pixel (2 y + i, 2 x + j) of the [N][2H][2W][64] tensor.  A stream's four workgroups share one slab.
    int glog, cout_pad, gshuffle;
    int slab_bf16;                          // the gradient part of a slab row is stored as bf16 at the row's start (sisr_wgrad_bf16_slab_lead)
};

__device__ __forceinline__ bf16x8 wt_frag(const unsigned char* p) {
    const s16x4 lo = lds_tr16(reinterpret_cast<const __bf16*>(p)), hi = lds_tr16(reinterpret_cast<const __bf16*>(p + 4 * WT_PS));
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

// x prologue of one thread's staged halo items: lrelu(a x + d) (a = 1, d = 0, slope = 1 degenerate to ACT / NONE), zero
// outside the image; dst0 = the thread's first LDS slot, item k sits 32 pixels further
template <int EASY>
__device__ __forceinline__ void wt_commit_x(const u32x4 (&sx)[WT_XITEMS], unsigned bad, unsigned xflags, const float* kst, int oct,
                                            float xslope, unsigned char* dst0) {
    const f32x8 ka = *reinterpret_cast<const f32x8*>(kst + oct * 8), kd = *reinterpret_cast<const f32x8*>(kst + 64 + oct * 8);
#pragma unroll
    for (int k = 0; k < WT_XITEMS; ++k) {
        const bool ok = ((bad >> (5 * k)) & 31u) == 0u;
        u32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float v0 = __uint_as_float(sx[k][j] << 16), v1 = __uint_as_float(sx[k][j] & 0xFFFF0000u);
            const float r0 = lrelu_t<EASY>(ka[2 * j] * v0 + kd[2 * j], xslope), r1 = lrelu_t<EASY>(ka[2 * j + 1] * v1 + kd[2 * j + 1], xslope);
            const unsigned pk = pack_bf16x2(r0, r1);
            o[j] = ok ? pk : 0u;
        }
        // (only the last item of a thread can lie beyond the 180 halo pixels)
        if (k < WT_XITEMS - 1 || !((xflags >> (5 * k + 4)) & 1u)) *reinterpret_cast<u32x4*>(dst0 + k * 32 * WT_PS) = o;
    }
}

// wg_index / wg_count: this workgroup's place among the workgroups that serve `a` (the grid of a launch of one layer; a
// contiguous range of the grid of a batch of layers, wgrad_trunk_table_kernel)
template <int GPRO>
__device__ __forceinline__ void wgrad_trunk_body(const WTrunkArgs& a, const int wg_index, const int wg_count, unsigned char* lds) {
    // [2 buffers][x halo image | dy image], then the x prologue constants
    float* kst = reinterpret_cast<float*>(lds + 2 * (WT_XBYTES + WT_DBYTES));      // xa, xd: [2][64]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool consumer = wave < 4;
    const int l31 = lane & 31;
    const int h = wave & 1, gq = (wave >> 1) & 1;             // consumer: output-channel half, input-channel chunk
    const unsigned xbytes = (unsigned)a.N * (unsigned)a.H * (unsigned)a.W * 128u;
    const int cg = wg_index & ((1 << a.glog) - 1);                      // cout group (shuffle phase) of this workgroup
    const int t_first = wg_index >> a.glog, t_step = wg_count >> a.glog;
    auto tile_coords = [&](int T, int& n, int& ty, int& tx) {
        n = fdiv(T, a.m_per_img);
        const int rem = T - n * a.per_img;
        ty = fdiv(rem, a.m_tiles_x);
        tx = rem - ty * a.tiles_x;
    };
    auto tile_origin = [&](int T, int& ty, int& tx) {
        int n;
        tile_coords(T, n, ty, tx);
        return (unsigned)(((n * a.H + ty * WT_TH) * a.W + tx * WT_TW) * 128);
    };

    if (tid < 64) {
        const bool aff = a.xpro == SISR_PRO_AFFINE_ACT;
        kst[tid] = aff ? a.pa[tid] : 1.f;
        kst[64 + tid] = aff ? a.pd[tid] : 0.f;
    }
    __syncthreads();

    // Staging items: thread pt (0 .. 255 of its role), item k = pixel pt / 8 + 32 k of the halo (x, 6 items, the last one
    // partly beyond the 180 halo pixels) or of the tile (gradient, 4 items), channel octet pt % 8.  Both roles are bound by
    // VALU issue while they stage, so everything about an item that does not depend on the tile is computed once.
    const int pt = tid & 255, oct = tid & 7, m0 = pt >> 3;
    float* sl = a.slab + (int64_t)t_first * a.slab_stride;

    // Two role-specific tile loops with matching barrier counts (a barrier only counts arriving waves; written as one loop
    // with a role branch inside, the allocator carries the accumulators through the producers' code and spills them).
    // Division of labour per tile T (traced: with the producers staging both operands, the consumers idled 1.5-2.5 us per
    // tile at the barrier):
    //   producers   gradient operand of tile T + 1 (two-tensor BatchNorm-backward prologue, bias sums) -> LDS
    //   consumers   request the x halo of tile T + 1, 72 MFMAs on tile T while those loads fly, then x prologue -> LDS
    if (!consumer) {
        // ---- producers --------------------------------------------------------------------------------------------------
        struct GStage { u32x4 g1[4], g2[4]; };
        GStage gA, gB;
        const float gslope = a.gslope_p ? a.gslope_p[0] : a.gslope;
        const f32x8 zero8 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        f32x8 bsum = zero8;                                          // bias gradient: this thread's 8 channels, its pixels
        f32x8 qa, qb, qd, qs = zero8, qt = zero8;
        if (GPRO != SISR_PRO_ACT_BWD) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                qa[j] = a.qa[oct * 8 + j]; qb[j] = a.qb[oct * 8 + j]; qd[j] = a.qd[oct * 8 + j];
                if (GPRO == SISR_PRO_BNACT_BWD) { qs[j] = a.qs[oct * 8 + j]; qt[j] = a.qt[oct * 8 + j]; }
            }
        } else {
            qa = zero8; qb = zero8; qd = zero8;
        }
        // tile pixel m0 + 32 k: two rows further down per item; a shuffled gradient is read through the strided view of
        // this workgroup's phase (pixel pitch 2, row pitch 2 * 2W)
        const int gsc = a.gshuffle ? 2 : 1, gW = gsc * a.W;
        const int grel0 = ((m0 >> 4) * gsc * gW + (m0 & 15) * gsc) * 128 + oct * 16;
        const int gstep = 2 * gsc * gW * 128;
        const unsigned gbytes = (unsigned)(gsc * gsc) * xbytes;
        const int glds0 = WT_XBYTES + m0 * WT_PS + oct * 16;
        // (always executed, so that the tile loop below stays free of control flow around loads: past the last tile the
        // offsets are out of range, which costs an instruction and no memory traffic.  With a branch around the loads the
        // compiler's wait-count bookkeeping gives up at the merge and drains every load before the next commit.)
        auto issue_g = [&](int T, GStage& st) {
            const __amdgpu_buffer_rsrc_t r1 = bf_rsrc(a.g1, gbytes), r2 = bf_rsrc(a.g2, gbytes);
            int n, ty, tx;
            tile_coords(T, n, ty, tx);
            const unsigned org = (unsigned)(((n * gsc * a.H + gsc * ty * WT_TH + (a.gshuffle ? cg >> 1 : 0)) * gW +
                                             gsc * tx * WT_TW + (a.gshuffle ? cg & 1 : 0)) * 128);
            const unsigned origin = T < a.total ? org : 0x80000000u;                      // (a scalar select, not a branch)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const unsigned voff = origin + (unsigned)(grel0 + k * gstep);
                st.g1[k] = __builtin_amdgcn_raw_buffer_load_b128(r1, voff, 0, 0);
                st.g2[k] = __builtin_amdgcn_raw_buffer_load_b128(r2, voff, 0, 0);
            }
        };
        auto commit_g = [&](int b, const GStage& st) {
            unsigned char* img = lds + b * (WT_XBYTES + WT_DBYTES);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                u32x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const unsigned w1 = st.g1[k][j], w2 = st.g2[k][j];
                    const float a0 = __uint_as_float(w1 << 16), a1 = __uint_as_float(w1 & 0xFFFF0000u);
                    const float b0 = __uint_as_float(w2 << 16), b1 = __uint_as_float(w2 & 0xFFFF0000u);
                    float g0 = a0, g1 = a1;
                    if (GPRO == SISR_PRO_BNACT_BWD) {
                        g0 = qs[2 * j] * b0 + qt[2 * j] > 0.f ? a0 : gslope * a0;
                        g1 = qs[2 * j + 1] * b1 + qt[2 * j + 1] > 0.f ? a1 : gslope * a1;
                    }
                    float r0, r1;
                    if (GPRO == SISR_PRO_ACT_BWD) {                    // act'(pre-activation) * gradient
                        r0 = b0 > 0.f ? a0 : gslope * a0;
                        r1 = b1 > 0.f ? a1 : gslope * a1;
                    } else {
                        r0 = qa[2 * j] * g0 + qb[2 * j] * b0 + qd[2 * j];
                        r1 = qa[2 * j + 1] * g1 + qb[2 * j + 1] * b1 + qd[2 * j + 1];
                    }
                    bsum[2 * j] += r0; bsum[2 * j + 1] += r1;
                    o[j] = pack_bf16x2(r0, r1);
                }
                *reinterpret_cast<u32x4*>(img + glds0 + k * 32 * WT_PS) = o;
            }
        };
        // two register sets: the loads of tile T + 2 fly while tile T + 1 is transformed.  The loop is unrolled by two so
        // that each set has a fixed name in each half.
        int T = t_first;
        issue_g(T, gA);
        issue_g(T + t_step, gB);
        if (T < a.total) commit_g(0, gA);
        WTTP(2);
        __syncthreads();
        int cur = 0;
        [[maybe_unused]] int it = 0;
        while (T < a.total) {
            WTTP(4 + 6 * it);
            issue_g(T + 2 * t_step, gA);                       // gB holds tile T + grid
            WTTP(5 + 6 * it);
            if (T + t_step < a.total) commit_g(cur ^ 1, gB);
            WTTP(8 + 6 * it);
            __syncthreads();      // the next tile's images are complete; the consumers have finished reading this one
            WTTP(9 + 6 * it);
            T += t_step; cur ^= 1; ++it;
            if (T >= a.total) break;
            WTTP(4 + 6 * it);
            issue_g(T + 2 * t_step, gB);                       // gA holds tile T + grid
            WTTP(5 + 6 * it);
            if (T + t_step < a.total) commit_g(cur ^ 1, gA);
            WTTP(8 + 6 * it);
            __syncthreads();
            WTTP(9 + 6 * it);
            T += t_step; cur ^= 1; ++it;
        }
        if (a.bias_slab != nullptr) *reinterpret_cast<f32x8*>(lds + pt * 32) = bsum;     // the images are free by now
    } else {
        // ---- consumers: the whole gradient of (32 output channels) x (9 taps x 32 input channels) in accumulators -----------
        f32x16 acc[9];
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
        // transposing-read lane roles: 16-lane group grp -> (channel half grp & 1, pixel half grp >> 1 of the 16-pixel K
        // step); inside the group lane 4q + p addresses (pixel row q, channels 4p .. 4p + 3)
        const int grp = lane >> 4, tq = (lane & 15) >> 2, tp = lane & 3;
        const int rd_pix = (8 * (grp >> 1) + tq) * WT_PS + (16 * (grp & 1) + 4 * tp) * 2;
        // x staging: rel[k] byte offset of item k from the tile's first pixel (negative in the top / left halo); flags: 5
        // bits per item (halo row 0, row 9, column 0, column 17, beyond the halo) against the tile's edge pattern
        u32x4 sx[WT_XITEMS];
        unsigned bad = 0, xflags = 0;
        int xrel[WT_XITEMS];
#pragma unroll
        for (int k = 0; k < WT_XITEMS; ++k) {
            const int px = m0 + 32 * k;
            const int py = px / WT_IW, pxx = px - py * WT_IW;
            xrel[k] = ((py - 1) * a.W + (pxx - 1)) * 128 + oct * 16;
            const unsigned f = (py == 0 ? 1u : 0u) | (py == WT_IH - 1 ? 2u : 0u) | (pxx == 0 ? 4u : 0u) | (pxx == WT_IW - 1 ? 8u : 0u) |
                               (px >= WT_NPIX ? 16u : 0u);
            xflags |= f << (5 * k);
        }
        const int xlds0 = m0 * WT_PS + oct * 16;
        const int tiles_y = a.per_img / a.tiles_x;
        const float xslope = a.xpro != SISR_PRO_NONE ? (a.xslope_p ? a.xslope_p[0] : a.xslope) : 1.f;
        const bool easy_slope = xslope >= 0.f && xslope <= 1.f;
        auto issue_x = [&](int T) {
            const __amdgpu_buffer_rsrc_t rx = bf_rsrc(a.x1, xbytes);
            int ty, tx;
            const unsigned origin = tile_origin(T, ty, tx);
            const unsigned e = (ty == 0 ? 1u : 0u) | (ty == tiles_y - 1 ? 2u : 0u) | (tx == 0 ? 4u : 0u) | (tx == a.tiles_x - 1 ? 8u : 0u) | 16u;
            bad = xflags & (e * 0x02108421u);                                // edge pattern replicated over the 6 items
#pragma unroll
            for (int k = 0; k < WT_XITEMS; ++k) {
                const bool ok = ((bad >> (5 * k)) & 31u) == 0u;
                sx[k] = __builtin_amdgcn_raw_buffer_load_b128(rx, ok ? origin + (unsigned)xrel[k] : 0x80000000u, 0, 0);
            }
        };
        auto commit_x = [&](int b) {
            unsigned char* dst0 = lds + b * (WT_XBYTES + WT_DBYTES) + xlds0;
            if (a.xpro == SISR_PRO_NONE) {
                // no prologue: the staged bf16 items go to LDS as they are (outside the image the loads returned zeros)
#pragma unroll
                for (int k = 0; k < WT_XITEMS; ++k)
                    if (k < WT_XITEMS - 1 || !((xflags >> (5 * k + 4)) & 1u)) *reinterpret_cast<u32x4*>(dst0 + k * 32 * WT_PS) = sx[k];
            } else if (easy_slope) wt_commit_x<1>(sx, bad, xflags, kst, oct, xslope, dst0);
            else wt_commit_x<0>(sx, bad, xflags, kst, oct, xslope, dst0);
        };
        WTT(0);
        if (t_first < a.total) {
            issue_x(t_first);
            commit_x(0);
        }
        __syncthreads();
        int cur = 0, it = 0;
        for (int T = t_first; T < a.total; T += t_step, cur ^= 1, ++it) {
            const int Tn = T + t_step;
            WTT(4 + 6 * it);
            if (Tn < a.total) issue_x(Tn);
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
            WTT(6 + 6 * it);
            if (Tn < a.total) commit_x(cur ^ 1);
            WTT(8 + 6 * it);
            __syncthreads();
            WTT(9 + 6 * it);
        }
        WTT(3);
        // one slab per workgroup: [chunk][tap][ci 32][64 co]
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int idx = ((gq * 9 + t) * 32 + mfma_row(i, lane)) * a.cout_pad + 64 * cg + 32 * h + l31;
                if (a.slab_bf16) reinterpret_cast<__bf16*>(sl)[idx] = (__bf16)acc[t][i];
                else sl[idx] = acc[t][i];
            }
    }
    if (a.bias_slab != nullptr) {                       // ... and its bias row
        __syncthreads();
        if (tid < 64) {
            const float* bs = reinterpret_cast<const float*>(lds);
            const int o8 = tid >> 3, j = tid & 7;
            float s = 0.f;
            for (int i = 0; i < 32; ++i) s += bs[(o8 + 8 * i) * 8 + j];                  // fixed order: deterministic
            a.bias_slab[(int64_t)t_first * a.slab_stride + 64 * cg + tid] = s;
        }
    }
    WTT(63);
}

template <int GPRO>
__global__ void __launch_bounds__(WT_THREADS, 2) wgrad_trunk_kernel(const WTrunkArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    wgrad_trunk_body<GPRO>(a, blockIdx.x, gridDim.x, lds);
}

// Several layers of one gradient-prologue kind in ONE launch: workgroups [z * wpl, (z + 1) * wpl) serve table[z].  The generator's 33
// trunk layers alone are 33 launches of 231 workgroups x 5 tiles (1,152 tiles do not divide by 256 CUs: a tenth of the chip idles, and
// every launch pays its pipeline fill and 231 slabs); as two batches (17 + 16 layers, 15 / 16 workgroups each) every workgroup walks
// 72-77 tiles of ONE layer back to back and writes one slab: 15-16 slabs per layer instead of 231.
template <int GPRO>
__global__ void __launch_bounds__(WT_THREADS, 2) wgrad_trunk_table_kernel(const WTrunkArgs* __restrict__ table, const int wpl) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int z = (int)blockIdx.x / wpl;
    const WTrunkArgs a = table[z];
    wgrad_trunk_body<GPRO>(a, (int)blockIdx.x - z * wpl, wpl, lds);
}

// ---- host ----------------------------------------------------------------------------------------------------------
static int wtrunk_grid(const SisrWgradDesc* d) {
    const int total = d->N * (d->H / WT_TH) * (d->W / WT_TW);
    const int cus = sisr_cu_slots();
    const int G = d->Cout == 256 ? 4 : 1;       // cout groups: each tile stream is served by G workgroups
    const int slots = std::max(1, cus / G);
    const int rounds = (total + slots - 1) / slots;
    return G * ((total + rounds - 1) / rounds);  // equal shares
}

extern "C" int sisr_wgrad_trunk_eligible(const SisrWgradDesc* d) {
    const char* sw = getenv("SISR_TRUNK");                      // A/B switch: SISR_TRUNK=0 keeps the generic kernel
    if (!d || (sw && sw[0] == '0')) return 0;
    const char* sw2 = getenv("SISR_TRUNK_WGRAD");
    if (sw2 && sw2[0] == '0') return 0;
    if (d->Cin != 64 || d->KH != 3 || d->KW != 3 || d->stride != 1 || d->pad_y != 1 || d->pad_x != 1) return 0;
    // Cout = 64 (trunk: BatchNorm-backward gradient prologues), or 256 with the gradient stored shuffled and an
    // activation-backward prologue -- the upscale conv
    const char* swu = getenv("SISR_TRUNK_UP");                 // A/B switch for the upscale conv alone
    const bool up = !(swu && swu[0] == '0') && d->Cout == 256 && d->g_mode == SISR_X_NHWC_UNSHUFFLE2 && d->CoutPad == 256 &&
                    d->gpro_mode == SISR_PRO_ACT_BWD && (int64_t)d->N * d->H * d->W * 512 < (1ll << 31);
    if (!up && (d->Cout != 64 || d->g_mode != SISR_X_NHWC || d->CoutPad != 64)) return 0;
    if (d->x_mode != SISR_X_NHWC || !d->x_bf16 || !d->g_bf16) return 0;
    if (d->Ho != d->H || d->Wo != d->W || (d->H % WT_TH) || (d->W % WT_TW)) return 0;
    if (d->n_chunk != 2 || d->KROWP != 9 * 32) return 0;
    if ((int64_t)d->N * d->H * d->W * 128 >= (1ll << 31)) return 0;
    if (d->N * (d->H / WT_TH) * (d->W / WT_TW) >= 65536) return 0;
    const bool xp = d->pro_mode == SISR_PRO_NONE || d->pro_mode == SISR_PRO_ACT || d->pro_mode == SISR_PRO_AFFINE_ACT;
    const bool gp = up || d->gpro_mode == SISR_PRO_BNBWD || d->gpro_mode == SISR_PRO_BNACT_BWD;
    return xp && gp ? 1 : 0;
}

// slabs a launch of this descriptor writes (rows of `slab` at slab_stride): one per workgroup
extern "C" int sisr_wgrad_toimage_eligible(const SisrWgradDesc* d);
int sisr_wgrad_toimage_slabs(const SisrWgradDesc* d);                         // wgrad_toimage.hip
extern "C" int sisr_wgrad_deep_eligible(const SisrWgradDesc* d);               // wgrad_deep.hip

// The persistent kernel writes the gradient part of its slabs as bf16 (231 slabs x 147 KB written and re-read per layer were
// two thirds of the finishing launch; a partial sum rounded to bf16 costs up to ~2e-3 relative on a cancelling total, inside this
// build's error budget -- its tensors are bf16).  SISR_SLAB_BF16=0 keeps fp32 slabs (A/B).
static bool wtrunk_slab_bf16() {
    const char* e = getenv("SISR_SLAB_BF16");
    return !(e && e[0] == '0');
}
// leading elements of every slab row that the launch of `d` stores as bf16 (pass it to sisr_slab_reduce_f32 /
// sisr_bn_bwd_finalize_slab); 0: fp32 slabs
extern "C" int64_t sisr_wgrad_bf16_slab_lead(const SisrWgradDesc* d) {
    if (!d) return 0;
    if (sisr_wgrad_trunk_eligible(d)) return wtrunk_slab_bf16() ? (int64_t)d->slab_elems : 0;
    if (sisr_wgrad_toimage_eligible(d)) return 0;
    return sisr_wgrad_deep_eligible(d) && d->deep.slab_bf16 ? (int64_t)d->slab_elems : 0;
}

extern "C" int sisr_wgrad_bf16_slabs(const SisrWgradDesc* d) {
    if (!d) return SISR_E_BADARG;
    if (sisr_wgrad_trunk_eligible(d)) return wtrunk_grid(d) / (d->Cout == 256 ? 4 : 1);
    if (sisr_wgrad_toimage_eligible(d)) return sisr_wgrad_toimage_slabs(d);
    return sisr_wgrad_deep_eligible(d) ? d->deep.n_pb : d->n_slabs;
}

template <int GPRO>
static int launch_wtrunk(const WTrunkArgs& a, int grid, hipStream_t st) {
    constexpr int lds_bytes = 2 * (WT_XBYTES + WT_DBYTES) + 2 * 64 * 4;
    static SisrLdsCap cap;
    if (int e = sisr_raise_lds_cap(cap, reinterpret_cast<const void*>(&wgrad_trunk_kernel<GPRO>), lds_bytes)) return e;
    hipLaunchKernelGGL((wgrad_trunk_kernel<GPRO>), dim3(grid), dim3(WT_THREADS), lds_bytes, st, a);
    SISR_CHECK_LAUNCH();
    return 0;
}

static WTrunkArgs wtrunk_args(const SisrWgradDesc* d);

// called by sisr_conv2d_wgrad_bf16 for eligible descriptors
int sisr_wgrad_trunk_launch(const SisrWgradDesc* d, hipStream_t st) {
    if (operand_needs_x2(d->gpro_mode) && !d->g2) return SISR_E_BADARG;
    if (d->pro_mode == SISR_PRO_AFFINE_ACT && (!d->pa || !d->pd)) return SISR_E_BADARG;
    if (d->gpro_mode != SISR_PRO_ACT_BWD && (!d->qa || !d->qb || !d->qd || (d->gpro_mode == SISR_PRO_BNACT_BWD && (!d->qs || !d->qt))))
        return SISR_E_BADARG;
    const WTrunkArgs a = wtrunk_args(d);
    const int grid = wtrunk_grid(d);
    if (d->gpro_mode == SISR_PRO_ACT_BWD) return launch_wtrunk<SISR_PRO_ACT_BWD>(a, grid, st);
    if (d->gpro_mode == SISR_PRO_BNBWD) return launch_wtrunk<SISR_PRO_BNBWD>(a, grid, st);
    return launch_wtrunk<SISR_PRO_BNACT_BWD>(a, grid, st);
}

static WTrunkArgs wtrunk_args(const SisrWgradDesc* d) {
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
    a.glog = d->Cout == 256 ? 2 : 0; a.cout_pad = d->Cout == 256 ? 256 : 64; a.gshuffle = d->g_mode == SISR_X_NHWC_UNSHUFFLE2 ? 1 : 0;
    a.slab_bf16 = wtrunk_slab_bf16() ? 1 : 0;
    return a;
}

// ---- a batch of trunk layers (Cout = 64, one gradient-prologue kind): see wgrad_trunk_table_kernel ---------------------------------
extern "C" int sisr_wgrad_trunk_batch_arg_bytes(void) { return (int)sizeof(WTrunkArgs); }

static int wtrunk_batch_check(const SisrWgradDesc* descs, int n) {
    if (!descs || n <= 0 || n > 4096) return SISR_E_BADARG;
    for (int i = 0; i < n; ++i) {
        const SisrWgradDesc* d = descs + i;
        if (!sisr_wgrad_trunk_eligible(d) || d->Cout != 64 || d->gpro_mode != descs[0].gpro_mode) return SISR_E_BADARG;
        if (!d->x1 || !d->g1 || !d->g2 || !d->slab || d->slab_stride < d->slab_elems) return SISR_E_BADARG;
        if (d->pro_mode == SISR_PRO_AFFINE_ACT && (!d->pa || !d->pd)) return SISR_E_BADARG;
        if (!d->qa || !d->qb || !d->qd || (d->gpro_mode == SISR_PRO_BNACT_BWD && (!d->qs || !d->qt))) return SISR_E_BADARG;
    }
    return 0;
}

// fills args_host (n * sisr_wgrad_trunk_batch_arg_bytes() bytes) with the kernel's view of the n descriptors; the caller copies it to
// device memory and passes that copy to sisr_wgrad_trunk_batch (the same staging route as every descriptor table of this library)
extern "C" int sisr_wgrad_trunk_batch_args(const SisrWgradDesc* descs, int32_t n, void* args_host) {
    if (!args_host) return SISR_E_BADARG;
    if (int e = wtrunk_batch_check(descs, n)) return e;
    for (int i = 0; i < n; ++i) {
        const WTrunkArgs a = wtrunk_args(descs + i);
        std::memcpy(static_cast<unsigned char*>(args_host) + (size_t)i * sizeof(WTrunkArgs), &a, sizeof(WTrunkArgs));
    }
    return 0;
}

// wgs_per_layer workgroups (= slabs, rows of each descriptor's `slab`) serve every layer; grid = n * wgs_per_layer
extern "C" int sisr_wgrad_trunk_batch(const SisrWgradDesc* descs, const void* args_dev, int32_t n, int32_t wgs_per_layer, void* stream) {
    if (!args_dev || wgs_per_layer <= 0 || (int64_t)n * wgs_per_layer > 65535) return SISR_E_BADARG;
    if (int e = wtrunk_batch_check(descs, n)) return e;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    constexpr int lds_bytes = 2 * (WT_XBYTES + WT_DBYTES) + 2 * 64 * 4;
    const WTrunkArgs* table = static_cast<const WTrunkArgs*>(args_dev);
    const dim3 grid(n * wgs_per_layer);
    if (descs[0].gpro_mode == SISR_PRO_BNBWD) {
        static SisrLdsCap cap;
        if (int e = sisr_raise_lds_cap(cap, reinterpret_cast<const void*>(&wgrad_trunk_table_kernel<SISR_PRO_BNBWD>), lds_bytes)) return e;
        hipLaunchKernelGGL((wgrad_trunk_table_kernel<SISR_PRO_BNBWD>), grid, dim3(WT_THREADS), lds_bytes, st, table, wgs_per_layer);
    } else {
        static SisrLdsCap cap;
        if (int e = sisr_raise_lds_cap(cap, reinterpret_cast<const void*>(&wgrad_trunk_table_kernel<SISR_PRO_BNACT_BWD>), lds_bytes)) return e;
        hipLaunchKernelGGL((wgrad_trunk_table_kernel<SISR_PRO_BNACT_BWD>), grid, dim3(WT_THREADS), lds_bytes, st, table, wgs_per_layer);
    }
    SISR_CHECK_LAUNCH();
    return 0;
}
