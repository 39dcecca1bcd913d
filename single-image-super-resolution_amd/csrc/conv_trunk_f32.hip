// conv_trunk_f32.hip -- forward and data-gradient conv of the generator's trunk geometry (3x3, 64 -> 64, stride 1,
// pad 1, fp32 NHWC in and out) in the fp32 PARITY build: exact fp32 matrix instructions (v_mfma_f32_32x32x2_f32).
// Reference path: every nn.Conv2d(64, 64, 3, 1, 1) of model_generator.py:29-55 (residual blocks) and :89-93 (trunk end)
// and their data gradients, with the neighbouring BatchNorm-apply / PReLU (prologue), bias, BatchNorm statistics and
// skip-gradient add (epilogue) fused, as the generic kernel (conv_fwd.hip) does.
//
// With fp32 operands this conv is bound by the matrix pipe by a wide margin (10.87 GFLOP per launch = 2.65 M MFMAs of
// 64 cycles; 75 MB of tensors).  The generic kernel reaches ~53 % of the fp32 MFMA peak: its 256-thread workgroups
// alternate staging and MFMA phases and re-stage the weights for every tile.  This kernel keeps the matrix pipe fed:
//   * a workgroup owns ONE output-channel half (32 couts): its share of the weights (32 x 576 floats = 74 KB) stays in
//     LDS for the whole launch ([chunk][tap][co 32][ci 32 + 4 pad]);
//   * it walks 8 x 16 pixel tiles; four consumer waves (one 32-pixel x 32-cout accumulator tile each, 288 MFMAs per tile)
//     and four producer waves that stage the next 32-channel half of a halo tile ([pixel][32 + 4 floats]) into the other
//     LDS buffer behind the MFMAs: one barrier per (tile, channel half);
//   * an fp32 MFMA takes ONE float per lane and operand, and a wave's other instructions do not overlap its own MFMAs (the
//     32x32x2 fp32 form occupies the wave's issue for its 64 cycles: measured 78 cycles per MFMA with two ds_read_b32 each).
//     The K order of the 16 MFMAs of a 32-channel slice is therefore chosen so that a lane's operands are CONTIGUOUS:
//     lane half kk multiplies channels 16 kk + s at step s, so its 16 A values (and its 16 B values) are four
//     ds_read_b128 instead of sixteen ds_read_b32 -- a quarter of the LDS instructions per MFMA;
//   * 1152 pixel tiles x 2 cout halves = exactly 9 units per CU at the benchmark size (no partial round);
//   * role-specific loops with matching barrier counts (see conv_trunk.hip).
// The two workgroups of a cout pair walk the same pixel tiles and share one statistics row: each writes its 32 channels.
#include "sisr_dev.h"
#include "sisr_bf16_stage.h"

typedef unsigned cf_u32x4 __attribute__((ext_vector_type(4)));

#include <algorithm>
#include <cstdlib>
#include <type_traits>

#ifndef CF_XCD_PAIRS
#define CF_XCD_PAIRS 1
#endif
#ifndef CF_CHAINS
#define CF_CHAINS 1
#endif
#ifndef CF_PRODPRIO
#define CF_PRODPRIO 0
#endif
#define CF_TH 8
#define CF_TW 16
#define CF_IH (CF_TH + 2)
#define CF_IW (CF_TW + 2)
#define CF_NPIX (CF_IH * CF_IW)            // 180 halo pixels
#define CF_PSF 36                           // floats per halo pixel in LDS: 32 channels + 4 -- rows stay 16-byte aligned (the operand
                                            // fetch is ds_read_b128) and 144-byte rows put 16 consecutive pixels on 16 different
                                            // 16-byte bank slots (9 p mod 16): conflict-free
#define CF_HALO_BYTES (CF_NPIX * CF_PSF * 4)   // 25920
#define CF_WROW 36                          // floats per (tap, cout) weight row in LDS: its 32 input channels + 4 (as CF_PSF)
#define CF_WCHUNK_BYTES (9 * 32 * CF_WROW * 4) // 41472
#define CF_ITEMS ((CF_NPIX * 8 + 255) / 256)   // 16-byte items (4 channels of a 32-channel half) per producer thread: 6
#define CF_THREADS 512
#define CF_KROWP 100                        // packed fp32 weights: [chunk][r][cout 64][krow = s * 33 + ci], rows of 100
#define CF_PS 33

// phase timeline, developer build only (make trace; tools/trace_trunk.py with ROLE=fwd SISR_PRECISION=fp32)
#ifdef SISR_CONV_TRACE
__device__ unsigned long long sisr_cftrace_buf[512 * 128];
#define CFT(k) do { if (threadIdx.x == 0 && blockIdx.x < 512 && (k) < 64) sisr_cftrace_buf[blockIdx.x * 128 + (k)] = wall_clock64(); } while (0)
#define CFTP(k) do { if (threadIdx.x == 256 && blockIdx.x < 512 && (k) < 64) sisr_cftrace_buf[blockIdx.x * 128 + 64 + (k)] = wall_clock64(); } while (0)
// shader-clock stamps (s_memtime) at kernel start / end: in-kernel clock = delta(s_memtime) / delta(wall) x 100 MHz
__device__ unsigned long long sisr_cfclk_buf[512 * 2];
#define CFTC(k) do { if (threadIdx.x == 0 && blockIdx.x < 512) sisr_cfclk_buf[blockIdx.x * 2 + ((k) - 60)] = clock64(); } while (0)
extern "C" int sisr_cfclk_read(void* dst, int n_u64) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(sisr_cfclk_buf), (size_t)n_u64 * 8, 0, hipMemcpyDeviceToHost);
}
extern "C" int sisr_cftrace_read(void* dst, int n_u64) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(sisr_cftrace_buf), (size_t)n_u64 * 8, 0, hipMemcpyDeviceToHost);
}
#else
#define CFT(k)
#define CFTP(k)
#define CFTC(k)
#endif

struct CTrunkF32Args {
    const float *x1, *x2;
    float* x_out;                         // skip-sum prologue: the materialised operand
    BnFinArgs fin;                        // deferred BatchNorm finalisation (fin.stat != nullptr): pa / pd come from here
    const float *pa, *pb, *pd, *ps, *pt;
    const float* slope_p; float slope;
    const float* wpk;
    const void* wimg;                     // the weights in this kernel's LDS order (SisrWeightDesc.f_ldsimg / d_ldsimg), mode matching SPLIT, or nullptr
    const float* bias;
    const float* res;
    float* y;
    float *stat_part, *cnt_part;          // [streams][2][64], [streams]
    // data-gradient role: the output is the gradient arriving at a BatchNorm whose input is bnb_x -- its backward
    // reductions (SisrConvDesc.bnb_*), one row [2 * 64 + 1] per workgroup (this workgroup's 32 channels, zeros elsewhere)
    const float *bnb_x, *bnb_scale, *bnb_shift, *bnb_mean, *bnb_invstd, *bnb_slope_p;
    float bnb_slope; int bnb_act;
    float* bnb_part;
    int N, H, W;
    int tiles_x, per_img, total, streams;
    uint32_t m_tiles_x, m_per_img;
    // forward role with Cout = 256 stored through PixelShuffle(2) (the generator's upscale conv, model_generator.py:43-48): eight
    // blocks of 32 packed couts per pixel-tile stream (glog = 3) instead of two; packed cout 64 (2 i + j) + c is channel c of
    // output pixel (2 y + i, 2 x + j)
    int glog, cout_pad, shuffle;
    // data gradient of that conv (Cin = 256 read through the un-shuffling view of the [N][2H][2W][64] gradient): one launch per
    // PixelShuffle phase -- the operand is the strided view pixel (2 y + i, 2 x + j) (xsc = 2, xph = 2 i + j), the weights that
    // phase's two 32-channel chunks, and launches 1 .. 3 add onto the output of the one before (res = y)
    int xsc, xph;
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t cf_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}

// two fp32 values -> their bf16 heads (round to nearest even) and the bf16 of what the heads leave: x = hi + lo up to 2^-17 |x|
__device__ __forceinline__ void cf_split2(float v0, float v1, unsigned& hw, unsigned& lw) {
    const f32x2 v = {v0, v1};
    hw = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
    const f32x2 r = {v0 - __uint_as_float(hw << 16), v1 - __uint_as_float(hw & 0xFFFF0000u)};
    lw = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2));
}

// SPLIT: the same kernel with every fp32 operand held in LDS as a (hi, lo) pair of bf16 -- 32 channels of a pixel / of a weight
// row are [32 hi][32 lo], the 128 bytes the 32 floats took -- and the contraction done by the bf16 matrix instruction
// (v_mfma_f32_32x32x16_bf16) as hi*hi + hi*lo + lo*hi into the fp32 accumulators: 54 MFMAs of 32 cycles per stage instead of
// 144 of 64, products exact, operands good to 2^-17 relative (the lo*lo term, 2^-18, is dropped).  Tensors in HBM, prologue
// arithmetic, accumulation, statistics and epilogue are the fp32 kernel's, unchanged.
template <int PRO, bool SPLIT>
__global__ void __launch_bounds__(CF_THREADS, 2) conv_trunk_f32_kernel(const CTrunkF32Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    // [weights: 2 chunks][halo buffer 0][halo buffer 1][reduction scratch]
    unsigned char* halo0 = lds + 2 * CF_WCHUNK_BYTES;
    float* red = reinterpret_cast<float*>(halo0 + 2 * CF_HALO_BYTES);          // [4 waves][32][3]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool consumer = wave < 4;
    const int l31 = lane & 31, kk = lane >> 5;
    // block of 32 output channels, pixel-tile stream.  The workgroups of one stream stage the SAME halo tiles in the same order: they
    // sit 8 apart in the grid, i.e. (workgroups being dealt round-robin over the 8 XCDs) on ONE XCD, so that the second reader of
    // a tile finds it in that XCD's L2 instead of fetching it again over the fabric
#if CF_XCD_PAIRS
    const bool xcd = (a.streams & 7) == 0;
    const int hc = xcd ? (blockIdx.x >> 3) & ((1 << a.glog) - 1) : blockIdx.x & ((1 << a.glog) - 1);
    const int stream = xcd ? (blockIdx.x & 7) | ((blockIdx.x >> (3 + a.glog)) << 3) : blockIdx.x >> a.glog;
#else
    const int hc = blockIdx.x & ((1 << a.glog) - 1), stream = blockIdx.x >> a.glog;
#endif
    const unsigned tbytes = (unsigned)a.N * (unsigned)a.H * (unsigned)a.W * 256u;
    constexpr bool TWO = PRO == SISR_PRO_BNBWD || PRO == SISR_PRO_BNACT_BWD || PRO == SISR_PRO_RES_AFFINE || PRO == SISR_PRO_ACT_BWD;
    constexpr bool SUM = PRO == SISR_PRO_RES_AFFINE;          // skip-sum prologue: lrelu(x1) + (a x2 + d), stored back once
    auto tile_coords = [&](int T, int& n, int& ty, int& tx) {
        n = fdiv(T, a.m_per_img);
        const int rem = T - n * a.per_img;
        ty = fdiv(rem, a.m_tiles_x);
        tx = rem - ty * a.tiles_x;
    };

    CFT(0);
    CFTC(60);
    // deferred BatchNorm finalisation: scale / shift of the prologue's BatchNorm from its statistics rows, in LDS (the halo
    // weights region is free yet: 48 KB of scratch; the constants behind the reduction scratch)
    float* kfin = red + 4 * 32 * 3;
    const bool fin = (PRO == SISR_PRO_AFFINE_ACT || PRO == SISR_PRO_RES_AFFINE) && a.fin.stat != nullptr;
    if (fin) bn_finalize_in_kernel(a.fin, reinterpret_cast<double*>(lds), kfin, blockIdx.x == 0);      // (scratch: the weights region, filled later)
    // ---- this workgroup's weights into LDS: global rows are contiguous over (s, ci) for one cout, LDS rows over couts.
    // Every wave takes part, from inside its role branch: the producers put the loads of their first two stages in flight
    // before it ---
    auto fill_weights = [&]() {
        float* wl = reinterpret_cast<float*>(lds);
        if (a.wimg != nullptr) {
            // the image is this workgroup's weights region word for word ([chunk][tap][cout 32][32 + 4]): a 16-byte copy,
            // all loads in flight, then the LDS writes
            constexpr int UNITS = 2 * 9 * 32 * CF_WROW / 4;                     // 5184 16-byte units
            constexpr int W_IT16 = (UNITS + CF_THREADS - 1) / CF_THREADS;         // 11
            const cf_u32x4* src = reinterpret_cast<const cf_u32x4*>(a.wimg) + hc * UNITS;
            cf_u32x4 wv16[W_IT16];
#pragma unroll
            for (int it = 0; it < W_IT16; ++it) {
                const int u = tid + it * CF_THREADS;
                wv16[it] = u < UNITS ? src[u] : cf_u32x4{0u, 0u, 0u, 0u};
            }
#pragma unroll
            for (int it = 0; it < W_IT16; ++it) {
                const int u = tid + it * CF_THREADS;
                if (u < UNITS) reinterpret_cast<cf_u32x4*>(wl)[u] = wv16[it];
            }
            return;
        }
        constexpr int W_IT = 2 * 9 * 32 * 32 / CF_THREADS;                     // 36 elements per thread
        float wv[W_IT];
#pragma unroll
        for (int it = 0; it < W_IT; ++it) {                                     // all loads in flight, then the LDS writes
            const int e = tid + it * CF_THREADS;
            const int ci = e & 31, s = (e >> 5) % 3, t2 = (e >> 5) / 3;        // t2 = (q * 3 + r) * 32 + co
            const int co = t2 & 31, qr = t2 >> 5;                               // qr = q * 3 + r
            wv[it] = a.wpk[((int64_t)qr * a.cout_pad + 32 * hc + co) * CF_KROWP + s * CF_PS + ci];
        }
#pragma unroll
        for (int it = 0; it < W_IT; ++it) {
            const int e = tid + it * CF_THREADS;
            const int ci = e & 31, s = (e >> 5) % 3, t2 = (e >> 5) / 3;
            const int co = t2 & 31, qr = t2 >> 5;
            const int q = qr / 3, r = qr - 3 * q;
            if constexpr (SPLIT) {
                // lanes 2m / 2m + 1 hold input channels 2m / 2m + 1 of one row: the even lane writes the hi pair, the odd one the lo pair
                const float vo = __shfl_xor(wv[it], 1);
                const bool odd = ci & 1;
                unsigned hw, lw;
                cf_split2(odd ? vo : wv[it], odd ? wv[it] : vo, hw, lw);
                reinterpret_cast<unsigned*>(wl)[((q * 9 + r * 3 + s) * 32 + co) * CF_WROW + (odd ? 16 : 0) + (ci >> 1)] = odd ? lw : hw;
            } else {
                wl[((q * 9 + r * 3 + s) * 32 + co) * CF_WROW + ci] = wv[it];
            }
        }
    };

    const int n_mine = stream < a.total ? (a.total - stream + a.streams - 1) / a.streams : 0;   // pixel tiles of this workgroup
    const int n_stages = 2 * n_mine;                                                               // (tile, channel half)

    if (!consumer) {
#if CF_PRODPRIO
        __builtin_amdgcn_s_setprio(3);
#endif
        // ---- producers: stage j + 1 = (tile (j + 1) / 2, channel half (j + 1) % 2) while the consumers multiply stage j -----
        // item k of thread pt: halo pixel pt / 8 + 32 k, channels 4 (pt % 8) .. + 3 of the 32-channel half
        const int pt = tid & 255, quad = tid & 7, m0 = pt >> 3;
        const int tiles_y = a.per_img / a.tiles_x;
        const float slope = PRO != SISR_PRO_NONE ? (a.slope_p ? a.slope_p[0] : a.slope) : 1.f;
        f32x4 ka[2], kb[2], kd[2], ks[2], kt[2];
        int rel[CF_ITEMS];
        // 5 bits per item: halo row 0 / last row / column 0 / last column / beyond the 180 halo pixels.  A tile's edge pattern (the
        // same 5 bits: image missing above / below / left / right, and 1) replicated over the items and ANDed with this is non-zero
        // exactly for the items whose LDS slot must be zero (conv_trunk.hip's scheme).  The producers' instructions compete with the
        // consumer wave's MFMA stream for their SIMD's issue slots (profiles/r03_trace_trunk_f32_fwd.txt: 5.4 us of producer work
        // per 4.7 us stage, the consumers 0.55 us at every second barrier), so their instruction count is what is tuned here.
        unsigned flags = 0;
#pragma unroll
        for (int k = 0; k < CF_ITEMS; ++k) {
            const int px = m0 + 32 * k;
            const int py = px / CF_IW, pxx = px - py * CF_IW;
            rel[k] = ((py - 1) * a.xsc * a.xsc * a.W + (pxx - 1) * a.xsc) * 256 + quad * 16;
            const unsigned f = (py == 0 ? 1u : 0u) | (py == CF_IH - 1 ? 2u : 0u) | (pxx == 0 ? 4u : 0u) | (pxx == CF_IW - 1 ? 8u : 0u) |
                               (px >= CF_NPIX ? 16u : 0u);
            flags |= f << (5 * k);
            asm volatile("" : "+v"(rel[k]));                  // (opaque: kept in a register, not re-derived from m0 by every issue())
        }
        const bool last_beyond = m0 + 32 * (CF_ITEMS - 1) >= CF_NPIX;
        const int ldso = SPLIT ? m0 * CF_PSF * 4 + quad * 8 : (m0 * CF_PSF + quad * 4) * 4;
        const bool easy_slope = slope >= 0.f && slope <= 1.f;

        // two staging register sets: the loads of stage j + 2 fly while stage j + 1 is transformed and written to LDS (a
        // stage lasts ~4 us of MFMAs; a cold load round trip under a chip-wide load burst is not much shorter).  issue() is
        // always executed, so that the loop has no control flow around loads and the compiler's wait counts stay exact.  Every
        // item is loaded from where it would sit in the tensor, inside the image or not (a buffer load past either end of the
        // tensor returns zeros; one that lands on a neighbouring row's pixels returns values commit() replaces by zeros): no
        // per-item address select.
        struct Stage { f32x4 s1[CF_ITEMS], s2[CF_ITEMS]; unsigned bad, origin; bool edge; };
        Stage stA, stB;
        auto issue = [&](int j, Stage& st) {
            const int T = stream + (j >> 1) * a.streams, q = j & 1;
            const unsigned xbytes = (unsigned)(a.xsc * a.xsc) * tbytes;
            const __amdgpu_buffer_rsrc_t r1 = cf_rsrc(a.x1, xbytes), r2 = cf_rsrc(TWO ? a.x2 : a.x1, xbytes);
            int n, ty, tx;
            tile_coords(T, n, ty, tx);
            const unsigned origin = (unsigned)(((n * a.xsc * a.H + a.xsc * ty * CF_TH + (a.xph >> 1)) * a.xsc * a.W + a.xsc * tx * CF_TW + (a.xph & 1)) * 256 + q * 128);
            const unsigned e = (ty == 0 ? 1u : 0u) | (ty == tiles_y - 1 ? 2u : 0u) | (tx == 0 ? 4u : 0u) | (tx == a.tiles_x - 1 ? 8u : 0u);
            st.edge = e != 0u;                                               // wave-uniform: this tile has items outside the image
            st.bad = flags & ((e | 16u) * 0x02108421u);                      // (stages past the end are never committed)
            st.origin = origin;
#pragma unroll
            for (int k = 0; k < CF_ITEMS; ++k) {
                const unsigned voff = origin + (unsigned)rel[k];
                st.s1[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r1, voff, 0, 0));
                if (TWO) st.s2[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r2, voff, 0, 0));
            }
        };
        // EASY: 0 <= slope <= 1 (every slope this model family uses): leaky ReLU = max(v, slope v), no compare + select.
        // EDGE: the tile has items outside the image (their LDS slots are zero AFTER the transform); an interior tile skips the selects.
        auto commit_t = [&](auto easy_, auto edge_, int j, const Stage& st) {
            constexpr int EASY = decltype(easy_)::value;
            constexpr bool EDGE = decltype(edge_)::value;
            const int q = j & 1;
            float* img = reinterpret_cast<float*>(halo0 + (j & 1) * CF_HALO_BYTES + ldso);
            // (selects, not dynamically indexed arrays: those would live in scratch)
            const f32x4 qa = q ? ka[1] : ka[0], qb = q ? kb[1] : kb[0], qd = q ? kd[1] : kd[0], qs = q ? ks[1] : ks[0], qt = q ? kt[1] : kt[0];
            const __amdgpu_buffer_rsrc_t ro = cf_rsrc(SUM ? a.x_out : a.x1, tbytes);
#pragma unroll
            for (int k = 0; k < CF_ITEMS; ++k) {
                if (k == CF_ITEMS - 1 && last_beyond) break;
                const bool ok = !EDGE || ((st.bad >> (5 * k)) & 31u) == 0u;
                auto value = [&](int c) {
                    const float v = st.s1[k][c];
                    if (PRO == SISR_PRO_NONE) return v;
                    if (PRO == SISR_PRO_ACT) return lrelu_t<EASY>(v, slope);
                    if (PRO == SISR_PRO_AFFINE_ACT) return lrelu_t<EASY>(qa[c] * v + qd[c], slope);
                    if (SUM) return lrelu_t<EASY>(v, slope) + (qa[c] * st.s2[k][c] + qd[c]);      // as sisr_eltwise_res_affine
                    if (PRO == SISR_PRO_ACT_BWD) return st.s2[k][c] > 0.f ? v : slope * v;    // act'(pre-activation) * gradient
                    const float bx = st.s2[k][c];
                    float g = v;
                    if (PRO == SISR_PRO_BNACT_BWD) g = qs[c] * bx + qt[c] > 0.f ? v : slope * v;
                    return qa[c] * g + qb[c] * bx + qd[c];
                };
                const f32x4 osum = {value(0), value(1), value(2), value(3)};
                const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
                if constexpr (SPLIT) {
                    typedef unsigned cf_u32x2 __attribute__((ext_vector_type(2)));
                    unsigned h0, l0, h1, l1;
                    cf_split2(osum[0], osum[1], h0, l0);
                    cf_split2(osum[2], osum[3], h1, l1);
                    const cf_u32x2 hw = {h0, h1}, lw = {l0, l1};
                    const cf_u32x2 z2 = {0u, 0u};
                    *reinterpret_cast<cf_u32x2*>(img + k * 32 * CF_PSF) = ok ? hw : z2;
                    *reinterpret_cast<cf_u32x2*>(img + k * 32 * CF_PSF + 16) = ok ? lw : z2;
                } else
                *reinterpret_cast<f32x4*>(img + k * 32 * CF_PSF) = ok ? osum : zero4;        // the halo is zero AFTER the transform
                // skip-sum prologue: the tile's own 8 x 16 pixels (no halo flag) store the materialised sum, once per pixel
                // and channel half -- by the workgroup of cout half 0 (its partner stages the same tiles)
                if (SUM && hc == 0 && ((flags >> (5 * k)) & 31u) == 0u)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(cf_u32x4, osum), ro, st.origin + (unsigned)rel[k], 0, 0);
            }
        };
        auto commit = [&](int j, const Stage& st) {
            using T1 = std::integral_constant<int, 1>; using T0 = std::integral_constant<int, 0>;
            // (prologues without a leaky ReLU have nothing that depends on EASY: one instantiation)
            constexpr bool HAS_ACT = PRO == SISR_PRO_ACT || PRO == SISR_PRO_AFFINE_ACT || SUM;
            if (!HAS_ACT || easy_slope) {
                if (st.edge) commit_t(T1{}, std::true_type{}, j, st); else commit_t(T1{}, std::false_type{}, j, st);
            } else {
                if (st.edge) commit_t(T0{}, std::true_type{}, j, st); else commit_t(T0{}, std::false_type{}, j, st);
            }
        };

        issue(0, stA);
        issue(1, stB);
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = q * 32 + quad * 4 + j;
                constexpr bool AFF = PRO == SISR_PRO_AFFINE_ACT || (TWO && PRO != SISR_PRO_ACT_BWD);    // prologues with a / d constants
                ka[q][j] = fin ? kfin[c] : AFF ? a.pa[c] : 1.f;
                kd[q][j] = fin ? kfin[64 + c] : AFF ? a.pd[c] : 0.f;
                kb[q][j] = (TWO && !SUM && PRO != SISR_PRO_ACT_BWD) ? a.pb[c] : 0.f;
                ks[q][j] = PRO == SISR_PRO_BNACT_BWD ? a.ps[c] : 0.f;
                kt[q][j] = PRO == SISR_PRO_BNACT_BWD ? a.pt[c] : 0.f;
            }
        fill_weights();
        __syncthreads();
        if (n_stages > 0) commit(0, stA);
        __syncthreads();
        // unrolled by two: each set has a fixed name in each half (stB holds stage j + 1 in the first)
        int j = 0;
        while (j < n_stages) {
            CFTP(4 + 6 * j);
            issue(j + 2, stA);
            CFTP(5 + 6 * j);
            if (j + 1 < n_stages) commit(j + 1, stB);
            CFTP(8 + 6 * j);
            __syncthreads();      // stage j + 1 is complete; the consumers have finished reading stage j
            CFTP(9 + 6 * j);
            if (++j >= n_stages) break;
            CFTP(4 + 6 * j);
            issue(j + 2, stB);
            CFTP(5 + 6 * j);
            if (j + 1 < n_stages) commit(j + 1, stA);
            CFTP(8 + 6 * j);
            __syncthreads();
            CFTP(9 + 6 * j);
            ++j;
        }
    } else {
        // ---- consumers: wave w = tile rows 2 w, 2 w + 1 (32 pixels) x this workgroup's 32 couts -------------------------------
        // operand lane roles (sisr_dev.h; K order chosen here): A = x[pixel l31][ci = 16 kk + s], B = W[ci = 16 kk + s][co = l31];
        // accumulator register i of a lane = pixel mfma_row(i, lane) of the sub-tile, cout l31
        const int co = a.shuffle ? 32 * (hc & 1) + l31 : 32 * hc + l31;           // channel of the (stored) output tensor
        const int ph = hc >> 1;                                                     // PixelShuffle phase (i, j) of this block
        // step s of a 32-channel slice: A = x[pixel l31][ci = 16 kk + s], B = W[ci = 16 kk + s][co = l31]
        // (SPLIT: step ks of a 32-channel slice is 16 channels, lane half kk multiplies channels 16 ks + 8 kk .. + 7: 16 bytes of the
        // hi half of the row, and the same 16 bytes of the lo half 64 bytes on)
        const int abase = SPLIT ? ((2 * wave + (l31 >> 4)) * CF_IW + (l31 & 15)) * CF_PSF * 4 + 16 * kk
                                : (((2 * wave + (l31 >> 4)) * CF_IW + (l31 & 15)) * CF_PSF + 16 * kk) * 4;
        const int bbase = SPLIT ? l31 * CF_WROW * 4 + 16 * kk : (l31 * CF_WROW + 16 * kk) * 4;
        // (bias is in ORIGINAL channel order: packed cout (phase, channel c) of a shuffled layer = original c * 4 + phase)
        const float bv = a.bias != nullptr ? a.bias[a.shuffle ? co * 4 + ph : co] : 0.f;
        float st_shift = 0.f, st_s1 = 0.f, st_s2 = 0.f;     // running statistics of this lane's values, shifted sums
        int st_n = 0;
        const __amdgpu_buffer_rsrc_t ry = cf_rsrc(a.y, a.shuffle ? 4u * tbytes : tbytes), rr = cf_rsrc(a.res != nullptr ? a.res : a.y, tbytes);
        f32x16 acc;
#if CF_CHAINS == 2
        f32x16 acc2;                                          // second accumulation chain (odd K steps), folded into acc before the epilogue
#endif
        f32x16 rv, xv;                                        // residual / BatchNorm-input values of the tile (requested a stage early)
        const bool has_x = a.bnb_part != nullptr;
        const __amdgpu_buffer_rsrc_t rxb = cf_rsrc(has_x ? a.bnb_x : a.y, tbytes);
        float b_sc = 0.f, b_sf = 0.f, b_mu = 0.f, b_is = 0.f, b_slope = 1.f;
        float rs1 = 0.f, rs2 = 0.f, rsl = 0.f;                // running sums of this lane's channel: g, g * xhat, slope term
        if (has_x) {
            b_sc = a.bnb_scale[co]; b_sf = a.bnb_shift[co]; b_mu = a.bnb_mean[co]; b_is = a.bnb_invstd[co];
            b_slope = a.bnb_slope_p ? a.bnb_slope_p[0] : a.bnb_slope;
        }
        fill_weights();
        __syncthreads();
        __syncthreads();
        for (int j = 0; j < n_stages; ++j) {
            const int q = j & 1;
            CFT(4 + 6 * j);
            int n, ty, tx;
            tile_coords(stream + (j >> 1) * a.streams, n, ty, tx);
            // (shuffled store: pixel (y, x) of phase (i, j) lands on (2 y + i, 2 x + j) of the [N][2H][2W][64] tensor)
            const unsigned obase = a.shuffle ? (unsigned)((((n * 2 * a.H + 2 * (ty * CF_TH + 2 * wave) + (ph >> 1)) * 2 * a.W + 2 * tx * CF_TW + (ph & 1)) * 64 + co) * 4)
                                             : (unsigned)((((n * a.H + ty * CF_TH + 2 * wave) * a.W + tx * CF_TW) * 64 + co) * 4);
            const int orow = a.shuffle ? 4 * a.W * 256 : a.W * 256, ocol = a.shuffle ? 512 : 256;       // bytes per tile row / column of the store
            if (q == 0) {
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] = bv;
#if CF_CHAINS == 2
#pragma unroll
                for (int i = 0; i < 16; ++i) acc2[i] = 0.f;
#endif
            } else {
                // the skip gradient and the BatchNorm input of this tile, in accumulator layout: in flight behind the second
                // half's MFMAs
                if (a.res != nullptr) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int p = mfma_row(i, lane);
                        rv[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rr, obase + (unsigned)(((p >> 4) * a.W + (p & 15)) * 256), 0, 0));
                    }
                }
                if (has_x) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int p = mfma_row(i, lane);
                        xv[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rxb, obase + (unsigned)(((p >> 4) * a.W + (p & 15)) * 256), 0, 0));
                    }
                }
            }
            const unsigned char* ab = halo0 + (j & 1) * CF_HALO_BYTES + abase;
            const unsigned char* bb = lds + q * CF_WCHUNK_BYTES + bbase;
            // software pipeline over the 18 half-taps (8 MFMAs each): the operands of half-tap u + 2 are requested before the
            // MFMAs of half-tap u, so a whole half-tap (512 cycles) of reads is always in flight (left alone the compiler
            // requests a pair of operands right before the MFMAs that need them; groups of 8 reads keep the wait expressible in
            // the 4-bit lgkmcnt)
            if constexpr (SPLIT) {
                // tap t: two 16-channel steps x (hi, lo) per operand = eight 16-byte reads, six MFMAs; the reads run two taps ahead
                bf16x8 ah[9][2], al[9][2], bh[9][2], bl[9][2];
                auto fetch_s = [&](int t) {
                    const unsigned char* pa = ab + ((t / 3) * CF_IW + (t % 3)) * CF_PSF * 4;
                    const unsigned char* pb = bb + t * 32 * CF_WROW * 4;
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        ah[t][ks] = *reinterpret_cast<const bf16x8*>(pa + 32 * ks);
#ifdef CF_ABLATE_BREADS        // timing-only (wrong results): the weight fragments are not read from LDS (what registers would give)
                        al[t][ks] = *reinterpret_cast<const bf16x8*>(pa + 64 + 32 * ks);
                        bh[t][ks] = al[t][ks]; bl[t][ks] = ah[t][ks];
                        asm volatile("" :: "v"(pb));
#else
                        bh[t][ks] = *reinterpret_cast<const bf16x8*>(pb + 32 * ks);
                        al[t][ks] = *reinterpret_cast<const bf16x8*>(pa + 64 + 32 * ks);
                        bl[t][ks] = *reinterpret_cast<const bf16x8*>(pb + 64 + 32 * ks);
#endif
                    }
                };
                fetch_s(0);
                fetch_s(1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    if (t + 2 < 9) fetch_s(t + 2);
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[t][ks], bh[t][ks], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[t][ks], bl[t][ks], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[t][ks], bh[t][ks], acc, 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
            f32x4 af[18][2], bf[18][2];                     // half-tap u: steps s0 .. s0 + 7 = two 16-byte reads per operand
            auto fetch = [&](int u) {
                const int t = u >> 1, s0 = 8 * (u & 1);
#pragma unroll
                for (int v = 0; v < 2; ++v) {
                    af[u][v] = *reinterpret_cast<const f32x4*>(ab + (((t / 3) * CF_IW + (t % 3)) * CF_PSF + s0 + 4 * v) * 4);
                    bf[u][v] = *reinterpret_cast<const f32x4*>(bb + ((t * 32) * CF_WROW + s0 + 4 * v) * 4);
                }
            };
            fetch(0);
            fetch(1);
            __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
#pragma unroll
            for (int u = 0; u < 18; ++u) {
                if (u + 2 < 18) fetch(u + 2);
#pragma unroll
#if CF_CHAINS == 2
                for (int s = 0; s < 8; s += 2) {
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[u][s >> 2][s & 3], bf[u][s >> 2][s & 3], acc, 0, 0, 0);
                    acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(af[u][s >> 2][(s & 3) + 1], bf[u][s >> 2][(s & 3) + 1], acc2, 0, 0, 0);
                }
#else
                for (int s = 0; s < 8; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[u][s >> 2][s & 3], bf[u][s >> 2][s & 3], acc, 0, 0, 0);
#endif
                if (u + 2 < 18) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
            }
            }
            CFT(6 + 6 * j);
#ifdef CF_ABLATE_EPI            // timing-only (no output): what the consumers' epilogue costs
            if (q == 1 && a.N < 0) {
#else
            if (q == 1) {
#endif
#if CF_CHAINS == 2
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] += acc2[i];
#endif
                // ---- epilogue of the tile: skip gradient, statistics, stores (128 contiguous bytes per pixel and half wave) ---
                if (a.res != nullptr) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[i] += rv[i];
                }
                if (has_x) {                                           // (two straight-line versions, not a branch per element)
                    if (a.bnb_act) {
#pragma unroll
                        for (int i = 0; i < 16; ++i) {
                            float gv = acc[i];
                            const float z = b_sc * xv[i] + b_sf;
                            const bool neg = !(z > 0.f);
                            rsl += neg ? gv * z : 0.f;
                            gv = neg ? gv * b_slope : gv;
                            rs1 += gv;
                            rs2 += gv * ((xv[i] - b_mu) * b_is);
                        }
                    } else {
#pragma unroll
                        for (int i = 0; i < 16; ++i) {
                            rs1 += acc[i];
                            rs2 += acc[i] * ((xv[i] - b_mu) * b_is);
                        }
                    }
                }
                if (a.stat_part != nullptr) {
                    if (st_n == 0) {                                    // shift = mean of the first tile's values of this lane
                        float s = 0.f;
#pragma unroll
                        for (int i = 0; i < 16; ++i) s += acc[i];
                        st_shift = s * (1.f / 16.f);
                    }
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const float dv = acc[i] - st_shift;
                        st_s1 += dv;
                        st_s2 += dv * dv;
                    }
                    st_n += 16;
                }
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int p = mfma_row(i, lane);
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, (float)acc[i]), ry,
                                                          obase + (unsigned)((p >> 4) * orow + (p & 15) * ocol), 0, 0);
                }
            }
#ifdef CF_ABLATE_EPI
            asm volatile("" :: "v"(acc));
#endif
            CFT(8 + 6 * j);
            __syncthreads();
            CFT(9 + 6 * j);
        }
        CFT(3);

        // ---- this workgroup's 32 channels of the statistics row it shares with its cout partner ------------------------------
        if (a.stat_part != nullptr) {
            float nn = (float)st_n, mu = 0.f, m2 = 0.f;
            if (st_n > 0) {
                const float m1 = st_s1 / nn;
                mu = st_shift + m1;
                m2 = st_s2 - st_s1 * m1;
            }
            const float nb = __shfl_xor(nn, 32), mub = __shfl_xor(mu, 32), m2b = __shfl_xor(m2, 32);
            const float nt = nn + nb;
            if (nt > 0.f) { const float dl = mub - mu, f = nb / nt; mu += dl * f; m2 += m2b + dl * dl * nn * f; }
            nn = nt;
            if (kk == 0) { float* r = red + (wave * 32 + l31) * 3; r[0] = nn; r[1] = mu; r[2] = m2; }
        }
        if (has_x) {
            rs1 += __shfl_xor(rs1, 32);
            rs2 += __shfl_xor(rs2, 32);
            rsl += __shfl_xor(rsl, 32);
            if (kk == 0) { float* r = red + (wave * 32 + l31) * 3; r[0] = rs1; r[1] = rs2; r[2] = rsl; }
        }
    }
    if (a.bnb_part != nullptr) {
        // one row per workgroup: its 32 channels (the four row-group waves summed in a fixed order), zeros for the other half
        __syncthreads();
        float* wk = a.bnb_part + (int64_t)blockIdx.x * 129;
        if (tid < 64) {
            const int c = tid & 31;
            const bool mine = (tid >> 5) == hc;
            float s1 = 0.f, s2 = 0.f;
            if (mine) {
                s1 = (red[c * 3] + red[(32 + c) * 3]) + (red[(64 + c) * 3] + red[(96 + c) * 3]);
                s2 = (red[c * 3 + 1] + red[(32 + c) * 3 + 1]) + (red[(64 + c) * 3 + 1] + red[(96 + c) * 3 + 1]);
            }
            wk[tid] = s1;
            wk[64 + tid] = s2;
        }
        if (tid == 64) {
            float sl = 0.f;
            for (int i = 0; i < 128; ++i) sl += red[i * 3 + 2];                          // fixed order: deterministic
            wk[128] = sl;
        }
    }
    if (a.stat_part != nullptr) {
        __syncthreads();
        if (tid < 32) {
            // the four row-group waves of channel tid, merged in a fixed order with Chan's formula
            float nn = red[tid * 3], mm = red[tid * 3 + 1], qq = red[tid * 3 + 2];
#pragma unroll
            for (int w = 1; w < 4; ++w) {
                const float* r1 = red + (w * 32 + tid) * 3;
                const float nb = r1[0], nt = nn + nb;
                if (nt > 0.f) { const float dl = r1[1] - mm, f = nb / nt; mm += dl * f; qq += r1[2] + dl * dl * nn * f; }
                nn = nt;
            }
            float* sp = a.stat_part + (int64_t)stream * 128 + 32 * hc + tid;
            sp[0] = mm;
            sp[64] = qq;
            if (tid == 0 && hc == 0) a.cnt_part[stream] = nn;
        }
    }
    CFT(63);
    CFTC(61);
}

// ---- host ----------------------------------------------------------------------------------------------------------
static int cf_streams(const SisrConvDesc* d) {
    const int total = d->N * (d->H / CF_TH) * (d->W / CF_TW);
    const int cus = sisr_cu_slots();
    const int slots = std::max(1, cus / (d->Cout == 256 ? 8 : 2));     // one workgroup per block of 32 couts and pixel-tile stream
    const int rounds = (total + slots - 1) / slots;
    return (total + rounds - 1) / rounds;       // equal shares
}

// 1: forward role, 2: data-gradient role, 0: not this kernel's geometry / fusions
extern "C" int sisr_conv2d_trunk_f32_eligible(const SisrConvDesc* d) {
    const char* sw = getenv("SISR_TRUNK");                      // A/B switch: SISR_TRUNK=0 keeps the generic kernel
    if (!d || (sw && sw[0] == '0')) return 0;
    const char* sw2 = getenv("SISR_TRUNK_F32CONV");
    if (sw2 && sw2[0] == '0') return 0;
    const char* swu = getenv("SISR_TRUNK_UP");                 // A/B switch for the upscale conv alone
    // the upscale conv's data gradient: 256 -> 64 over the un-shuffling view of the gradient, activation-backward prologue: four
    // launches of the data-gradient role, one per PixelShuffle phase (see CTrunkF32Args.xsc)
    if (!(swu && swu[0] == '0') && d->Cin == 256 && d->Cout == 64 && d->KH == 3 && d->KW == 3 && d->stride == 1 && d->pad_y == 1 && d->pad_x == 1 &&
        d->x_mode == SISR_X_NHWC_UNSHUFFLE2 && d->pro_mode == SISR_PRO_ACT_BWD && d->x2 && d->y_mode == SISR_Y_NHWC && !d->x_bf16 && !d->y_bf16 &&
        !d->res_bf16 && d->Ho == d->H && d->Wo == d->W && !(d->H % CF_TH) && !(d->W % CF_TW) && d->y_sy == 1 && d->y_sx == 1 && !d->y_oy && !d->y_ox &&
        d->y_H == d->Ho && d->y_W == d->Wo && d->epi_act == SISR_EPI_NONE && d->plan.CK == 32 && d->plan.PS == CF_PS && d->plan.KROWP == CF_KROWP &&
        d->plan.CoutPad == 64 && d->plan.n_chunk == 8 && (int64_t)d->N * d->H * d->W * 1024 < (1ll << 31) &&
        d->N * (d->H / CF_TH) * (d->W / CF_TW) < 65536 && !d->stat_part && !d->bias && !d->bnb_part && !d->fin_stat)
        return 2;
    if (d->Cin != 64 || d->KH != 3 || d->KW != 3 || d->stride != 1 || d->pad_y != 1 || d->pad_x != 1) return 0;
    // Cout = 64 (trunk), or 256 stored through PixelShuffle(2) -- the upscale conv, forward role without statistics
    const bool up = !(swu && swu[0] == '0') && d->Cout == 256 && d->y_mode == SISR_Y_NHWC_SHUFFLE2 && d->plan.CoutPad == 256 && !d->stat_part &&
                    !d->res && !d->bnb_part && !d->fin_stat &&
                    (d->pro_mode == SISR_PRO_NONE || d->pro_mode == SISR_PRO_ACT || d->pro_mode == SISR_PRO_AFFINE_ACT);
    if (!up && (d->Cout != 64 || d->y_mode != SISR_Y_NHWC || d->plan.CoutPad != 64)) return 0;
    if (d->x_mode != SISR_X_NHWC || d->x_bf16 || d->y_bf16 || d->res_bf16) return 0;
    if (d->Ho != d->H || d->Wo != d->W || (d->H % CF_TH) || (d->W % CF_TW)) return 0;
    if (!up && (d->y_sy != 1 || d->y_sx != 1 || d->y_oy || d->y_ox || d->y_H != d->Ho || d->y_W != d->Wo)) return 0;
    if (d->epi_act != SISR_EPI_NONE) return 0;
    if (d->plan.CK != 32 || d->plan.PS != CF_PS || d->plan.KROWP != CF_KROWP || d->plan.n_chunk != 2) return 0;
    if ((int64_t)d->N * d->H * d->W * 256 * (up ? 4 : 1) >= (1ll << 31)) return 0;
    if (d->N * (d->H / CF_TH) * (d->W / CF_TW) >= 65536) return 0;
    const bool fwd_pro = d->pro_mode == SISR_PRO_NONE || d->pro_mode == SISR_PRO_ACT || d->pro_mode == SISR_PRO_AFFINE_ACT ||
                         (d->pro_mode == SISR_PRO_RES_AFFINE && d->x2 && d->x_out && ((d->pa && d->pd) || d->fin_stat));
    if (d->fin_stat && !((d->pro_mode == SISR_PRO_AFFINE_ACT || d->pro_mode == SISR_PRO_RES_AFFINE) && d->fin_cnt && d->fin_gamma &&
                         d->fin_beta && d->fin_rm && d->fin_rv && d->fin_k && d->fin_rows > 0))
        return 0;
    if (fwd_pro && !d->res && !d->bnb_part) return 1;
    const bool bwd_pro = d->pro_mode == SISR_PRO_BNBWD || d->pro_mode == SISR_PRO_BNACT_BWD;
    if (bwd_pro && !d->stat_part && !d->bias && (!d->bnb_part || (d->bnb_x && !d->bnbx_bf16))) return 2;
    return 0;
}

// rows of stat_part / cnt_part a launch of this descriptor writes
extern "C" int sisr_conv2d_f32_parts(const SisrConvDesc* d) {
    if (!d) return SISR_E_BADARG;
    return sisr_conv2d_trunk_f32_eligible(d) ? cf_streams(d) : d->plan.n_tiles;
}

// rows of bnb_part (one per workgroup) a data-gradient launch of this descriptor writes; 0: this kernel does not take it
extern "C" int sisr_conv2d_f32_bnb_parts(const SisrConvDesc* d) {
    if (!d) return SISR_E_BADARG;
    return sisr_conv2d_trunk_f32_eligible(d) == 2 ? 2 * cf_streams(d) : 0;
}

template <int PRO, bool SPLIT>
static int launch_cf_t(const CTrunkF32Args& a, hipStream_t st) {
    constexpr int lds_bytes = 2 * CF_WCHUNK_BYTES + 2 * CF_HALO_BYTES + 4 * 32 * 3 * 4 + 128 * 4;
    static SisrLdsCap cap;
    if (int e = sisr_raise_lds_cap(cap, reinterpret_cast<const void*>(&conv_trunk_f32_kernel<PRO, SPLIT>), lds_bytes)) return e;
    hipLaunchKernelGGL((conv_trunk_f32_kernel<PRO, SPLIT>), dim3(a.streams << a.glog), dim3(CF_THREADS), lds_bytes, st, a);
    SISR_CHECK_LAUNCH();
    return 0;
}
template <int PRO>
static int launch_cf(const CTrunkF32Args& a, bool split, hipStream_t st) {
    return split ? launch_cf_t<PRO, true>(a, st) : launch_cf_t<PRO, false>(a, st);
}

// called by sisr_conv2d_f32 for eligible descriptors
int sisr_conv2d_trunk_f32_launch(const SisrConvDesc* d, hipStream_t st) {
    if (operand_needs_x2(d->pro_mode) && !d->x2) return SISR_E_BADARG;
    CTrunkF32Args a{};
    a.fin.stat = d->fin_stat; a.fin.cnt = d->fin_cnt; a.fin.gamma = d->fin_gamma; a.fin.beta = d->fin_beta;
    a.fin.rm = d->fin_rm; a.fin.rv = d->fin_rv; a.fin.k = d->fin_k; a.fin.rows = d->fin_rows; a.fin.momentum = d->fin_momentum; a.fin.eps = d->fin_eps;
    a.x1 = d->x1; a.x2 = d->x2; a.x_out = d->x_out; a.pa = d->pa; a.pb = d->pb; a.pd = d->pd; a.ps = d->ps; a.pt = d->pt;
    a.slope_p = d->pro_slope_p; a.slope = d->pro_slope;
    a.wpk = d->wpk; a.bias = d->bias; a.res = d->res; a.y = d->y; a.stat_part = d->stat_part; a.cnt_part = d->cnt_part;
    // the LDS-order image behind the standard one, when the caller packed it in the mode this launch computes in (trunk layers only)
    const int img_mode = (d->plan.variant >> 1) & 3;
    a.wimg = (d->Cin == 64 && d->Cout == 64 && img_mode == (d->mfma_split ? 2 : 1)) ? static_cast<const void*>(d->wpk + d->plan.wpk_elems) : nullptr;
    a.N = d->N; a.H = d->H; a.W = d->W;
    a.tiles_x = d->W / CF_TW; a.per_img = (d->H / CF_TH) * a.tiles_x; a.total = d->N * a.per_img;
    a.streams = cf_streams(d);
    a.glog = d->Cout == 256 ? 3 : 1; a.cout_pad = d->Cout == 256 ? 256 : 64; a.shuffle = d->y_mode == SISR_Y_NHWC_SHUFFLE2 ? 1 : 0;
    a.bnb_x = d->bnb_x; a.bnb_scale = d->bnb_scale; a.bnb_shift = d->bnb_shift; a.bnb_mean = d->bnb_mean; a.bnb_invstd = d->bnb_invstd;
    a.bnb_slope_p = d->bnb_slope_p; a.bnb_slope = d->bnb_slope; a.bnb_act = d->bnb_act; a.bnb_part = d->bnb_part;
    a.m_tiles_x = fdiv_magic(a.tiles_x); a.m_per_img = fdiv_magic(a.per_img);
    a.xsc = 1; a.xph = 0;
    if (d->Cin == 256) {
        // [8 chunks][3 filter rows][64 couts][KROWP]: phase ph owns chunks 2 ph, 2 ph + 1
        a.xsc = 2;
        for (int ph = 0; ph < 4; ++ph) {
            a.xph = ph;
            a.wpk = d->wpk + (size_t)ph * 2 * 3 * 64 * CF_KROWP;
            a.res = ph == 0 ? d->res : d->y;
            if (int e = launch_cf<SISR_PRO_ACT_BWD>(a, d->mfma_split != 0, st)) return e;
        }
        return 0;
    }
    switch (d->pro_mode) {
        case SISR_PRO_NONE: return launch_cf<SISR_PRO_NONE>(a, d->mfma_split != 0, st);
        case SISR_PRO_ACT: return launch_cf<SISR_PRO_ACT>(a, d->mfma_split != 0, st);
        case SISR_PRO_AFFINE_ACT: return launch_cf<SISR_PRO_AFFINE_ACT>(a, d->mfma_split != 0, st);
        case SISR_PRO_BNBWD: return launch_cf<SISR_PRO_BNBWD>(a, d->mfma_split != 0, st);
        case SISR_PRO_BNACT_BWD: return launch_cf<SISR_PRO_BNACT_BWD>(a, d->mfma_split != 0, st);
        case SISR_PRO_RES_AFFINE: return launch_cf<SISR_PRO_RES_AFFINE>(a, d->mfma_split != 0, st);
    }
    return SISR_E_UNSUPPORTED;
}
