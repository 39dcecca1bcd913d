// wgrad_trunk_f32.hip -- weight gradient of the generator's trunk geometry (3x3, 64 -> 64, stride 1, pad 1) in the
// fp32 PARITY build: fp32 NHWC tensors, exact fp32 matrix instructions (v_mfma_f32_32x32x2_f32).  Reference path: the
// autograd weight gradient of every nn.Conv2d(64, 64, 3, 1, 1) of model_generator.py:29-55 and :89-93.
//
//   dW[co][tap][ci] = sum over pixels p of dy[p][co] * x[p + tap][ci],      db[co] = sum_p dy[p][co]
//
// Same structure as the bf16 kernel of wgrad_trunk.hip (persistent workgroups, the whole 64 x 576 gradient in the
// accumulators of four consumer waves across all of a workgroup's tiles, four producer waves staging the next tile
// behind the MFMAs, role-specific tile loops), with what fp32 changes:
//   * the matrix pipe is the binding resource by a wide margin (a 32x32x2 fp32 MFMA takes 64 cycles and moves 1/8 of a
//     bf16 one's K): 288 MFMAs = 7.7 us per 4 x 16 pixel tile and wave against ~1 us of staging, so BOTH operands are
//     staged by the producers and nothing about staging needs tuning;
//   * tiles of 4 x 16 pixels: 2304 tiles = exactly 9 per CU at the benchmark size -- an MFMA-bound kernel pays for every
//     idle CU of a partial round (8 x 16 tiles: 4.5 per CU);
//   * an fp32 MFMA operand is ONE float per lane (lane = channel, lane half = pixel of the K pair), so plain pixel-major
//     LDS images [pixel][64 floats] are read conflict-free with ds_read_b32 -- no transposing reads, no padding.
// Slab layout as the generic fp32 kernel's (conv_wgrad.hip): [chunk][filter row][s * 33 + ci][64 co] + bias row, so the
// reduction and un-packing kernels are shared.  One slab per workgroup.
#include "sisr_dev.h"
#include "sisr_bf16_stage.h"

#include <algorithm>
#include <cstring>
#include <cstdlib>
#include <type_traits>

#define WF_TH 4
#define WF_TW 16
#define WF_IH (WF_TH + 2)
#define WF_IW (WF_TW + 2)
#define WF_NPIX (WF_IH * WF_IW)            // 108 halo pixels
#define WF_PB 256                           // LDS bytes per pixel: 64 floats
#define WF_XBYTES (WF_NPIX * WF_PB)         // 27648
#define WF_DBYTES (WF_TH * WF_TW * WF_PB)   // 16384
#define WF_XITEMS ((WF_NPIX * 16 + 255) / 256)   // 16-byte items (4 channels) per producer thread: 7
#define WF_THREADS 512
// split build (SisrWgradDesc.mfma_split): every fp32 value in LDS as a (hi, lo) pair of bf16 -- a pixel is [64 hi][64 lo] plus
// 32 bytes, so that the 4 pixel rows of a transposing read sit 8 banks apart
#define WS_PB 288
#define WS_XBYTES (WF_NPIX * WS_PB)         // 31104
#define WS_DBYTES (WF_TH * WF_TW * WS_PB)   // 18432
#define WF_PS 33                            // generic fp32 plan: krow = s * PS + ci
#define WF_KROWP 100

// barrier-wait accounting, developer build only (-DSISR_BARRIER_ACCT; tools/barrier_acct.py): see conv_trunk.hip
#ifdef SISR_BARRIER_ACCT
__device__ unsigned long long sisr_wfacct_buf[512 * 8];
extern "C" int sisr_wfacct_read(void* dst, int n_u64) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(sisr_wfacct_buf), (size_t)n_u64 * 8, 0, hipMemcpyDeviceToHost);
}
#define WFA_DECL unsigned long long ba_wait = 0, ba_t0 = clock64()
#define WFA_SYNC() do { const unsigned long long b0_ = clock64(); __syncthreads(); ba_wait += clock64() - b0_; } while (0)
#define WFA_STORE(slot) do { if ((threadIdx.x & 63) == 0 && blockIdx.x < 512) { sisr_wfacct_buf[blockIdx.x * 8 + (slot)] = clock64() - ba_t0; sisr_wfacct_buf[blockIdx.x * 8 + (slot) + 1] = ba_wait; } } while (0)
#define WFA_MARK(slot) do { if (threadIdx.x == 0 && blockIdx.x < 512) sisr_wfacct_buf[blockIdx.x * 8 + (slot)] = clock64(); } while (0)
#else
#define WFA_DECL
#define WFA_SYNC() __syncthreads()
#define WFA_STORE(slot)
#define WFA_MARK(slot)
#endif

struct WTrunkF32Args {
    const float *x1, *g1, *g2;
    const float *pa, *pd;                   // x prologue (AFFINE_ACT)
    const float* xslope_p; float xslope;
    const float *qa, *qb, *qd, *qs, *qt;    // gradient prologue
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
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t wf_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}

// two fp32 values -> their bf16 heads (round to nearest even) and the bf16 of what the heads leave: x = hi + lo up to 2^-17 |x|
__device__ __forceinline__ void wf_split2(float v0, float v1, unsigned& hw, unsigned& lw) {
    const f32x2 v = {v0, v1};
    hw = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
    const f32x2 r = {v0 - __uint_as_float(hw << 16), v1 - __uint_as_float(hw & 0xFFFF0000u)};
    lw = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2));
}
__device__ __forceinline__ void wf_store_split(unsigned char* dst, const f32x4& o) {
    unsigned h0, l0, h1, l1;
    wf_split2(o[0], o[1], h0, l0);
    wf_split2(o[2], o[3], h1, l1);
    const u32x2 hw = {h0, h1}, lw = {l0, l1};
    *reinterpret_cast<u32x2*>(dst) = hw;
    *reinterpret_cast<u32x2*>(dst + 128) = lw;
}
// K = 16 pixels of one channel per lane half: two transposing reads (4 pixel rows each) from a pixel-major split image
__device__ __forceinline__ bf16x8 ws_frag(const unsigned char* p) {
    const s16x4 lo = lds_tr16(reinterpret_cast<const __bf16*>(p)), hi = lds_tr16(reinterpret_cast<const __bf16*>(p + 4 * WS_PB));
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

// SPLIT: the same kernel with the contraction on the bf16 matrix instruction (v_mfma_f32_32x32x16_bf16: one K step = the 16
// pixels of a tile row) as hi*hi + hi*lo + lo*hi over (hi, lo) bf16 pairs of both fp32 operands: 108 MFMAs of 32 cycles per
// tile and wave instead of 288 of 64.  The producers split while they stage; the operands are fetched with transposing reads
// as in wgrad_trunk.hip.  Tensors in HBM, prologue arithmetic, bias sums, accumulation and the slab are the fp32 kernel's.
// wg_index / wg_count: this workgroup's place among the workgroups that serve `a` (see wgrad_trunk_f32_table_kernel); lds: [2
// buffers][x halo image | dy image]
template <int GPRO, bool SPLIT>
__device__ __forceinline__ void wgrad_trunk_f32_body(const WTrunkF32Args& a, const int wg_index, const int wg_count, unsigned char* lds) {

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool consumer = wave < 4;
    const int l31 = lane & 31, kk = lane >> 5;
    const int h = wave & 1, gq = (wave >> 1) & 1;             // consumer: output-channel half, input-channel chunk
    const unsigned tbytes = (unsigned)a.N * (unsigned)a.H * (unsigned)a.W * 256u;
    auto tile_origin = [&](int T, int& ty, int& tx) {
        const int n = fdiv(T, a.m_per_img);
        const int rem = T - n * a.per_img;
        ty = fdiv(rem, a.m_tiles_x);
        tx = rem - ty * a.tiles_x;
        return (unsigned)(((n * a.H + ty * WF_TH) * a.W + tx * WF_TW) * 256);
    };
    const int cg = wg_index & ((1 << a.glog) - 1);                      // cout group (shuffle phase) of this workgroup
    const int t_first = wg_index >> a.glog, t_step = wg_count >> a.glog;
    float* sl = a.slab + (int64_t)t_first * a.slab_stride;
    constexpr int XB = SPLIT ? WS_XBYTES : WF_XBYTES, BUF = SPLIT ? WS_XBYTES + WS_DBYTES : WF_XBYTES + WF_DBYTES, PB = SPLIT ? WS_PB : WF_PB;
    // (accounting marks of thread 0, a consumer: 4 = role state ready, 5 = first barrier passed, 6 = tile loop done, 7 = end)
    if (!consumer) {
        // ---- producers: both operands of tile T + 1 while the consumers multiply tile T ------------------------------------
        // item k of thread pt: pixel pt / 16 + 16 k (of the halo for x, of the tile for the gradient), channels 4 (pt % 16) ..
        const int pt = tid & 255, quad = tid & 15, m0 = pt >> 4;
        const int tiles_y = a.per_img / a.tiles_x;
        const float xslope = a.xpro != SISR_PRO_NONE ? (a.xslope_p ? a.xslope_p[0] : a.xslope) : 1.f;
        const float gslope = a.gslope_p ? a.gslope_p[0] : a.gslope;
        const bool aff = a.xpro == SISR_PRO_AFFINE_ACT;
        f32x4 ka, kd, qa, qb, qd, qs, qt;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            ka[j] = aff ? a.pa[quad * 4 + j] : 1.f; kd[j] = aff ? a.pd[quad * 4 + j] : 0.f;
            qa[j] = GPRO != SISR_PRO_ACT_BWD ? a.qa[quad * 4 + j] : 0.f; qb[j] = GPRO != SISR_PRO_ACT_BWD ? a.qb[quad * 4 + j] : 0.f;
            qd[j] = GPRO != SISR_PRO_ACT_BWD ? a.qd[quad * 4 + j] : 0.f;
            qs[j] = GPRO == SISR_PRO_BNACT_BWD ? a.qs[quad * 4 + j] : 0.f;
            qt[j] = GPRO == SISR_PRO_BNACT_BWD ? a.qt[quad * 4 + j] : 0.f;
        }
        f32x4 bsum = {0.f, 0.f, 0.f, 0.f};                    // bias gradient: this thread's 4 channels, its pixels
        // tile-independent part of the x items: byte offset from the tile's first pixel, halo edge flags (4 bits per item:
        // halo row 0, last row, column 0, last column; "beyond the halo" is folded into the validity of the last item)
        int xrel[WF_XITEMS];
        unsigned xflags = 0;
#pragma unroll
        for (int k = 0; k < WF_XITEMS; ++k) {
            const int px = m0 + 16 * k;
            const int py = px / WF_IW, pxx = px - py * WF_IW;
            xrel[k] = ((py - 1) * a.W + (pxx - 1)) * 256 + quad * 16;
            const unsigned f = px >= WF_NPIX ? 15u
                               : (py == 0 ? 1u : 0u) | (py == WF_IH - 1 ? 2u : 0u) | (pxx == 0 ? 4u : 0u) | (pxx == WF_IW - 1 ? 8u : 0u);
            xflags |= f << (4 * k);
        }
        const bool last_beyond = m0 + 16 * (WF_XITEMS - 1) >= WF_NPIX;
        // tile pixel m0 + 16 k: one row further down per item; a shuffled gradient is read through the strided view of this
        // workgroup's phase (pixel pitch 2, row pitch 2 * 2W)
        const int gsc = a.gshuffle ? 2 : 1, gW = gsc * a.W;
        const int grel0 = ((m0 >> 4) * gsc * gW + (m0 & 15) * gsc) * 256 + quad * 16;
        const int gstep = gsc * gW * 256;
        const unsigned gbytes = (unsigned)(gsc * gsc) * tbytes;
        const int xlds0 = m0 * PB + quad * (SPLIT ? 8 : 16), glds0 = XB + m0 * PB + quad * (SPLIT ? 8 : 16);

        // two staging register sets: the loads of tile T + 2 are in flight while tile T + 1 is transformed and written to LDS
        // (a tile lasts ~8 us of MFMAs; loading synchronously inside that window left the consumers waiting at 12 % of their
        // barriers' time for a late tile: tools/barrier_acct.py).  issue() is always executed -- past the last tile every offset
        // is out of range: an instruction, no traffic -- so that the loop has no control flow around loads.
        // (instruction count of the producers is what is tuned here -- their instructions compete with the consumer wave's MFMA stream
        // for the SIMD's issue slots, and the consumers of the exact kernel waited 10 % of the tile loop for them: no per-item address
        // selects (a buffer load past either end of the tensor returns zeros, one that lands on a neighbouring row's pixels
        // returns values commit() replaces by zeros), the edge test as one AND with the tile's replicated edge pattern, leaky
        // ReLU as max(v, slope v) for slopes in [0, 1], and interior tiles skip the zero selects)
        struct Stage { f32x4 sx[WF_XITEMS], s1[4], s2[4]; unsigned bad; bool edge; };
        Stage stA, stB;
        const bool easy_slope = xslope >= 0.f && xslope <= 1.f;
        auto issue = [&](int T, Stage& st) {
            const __amdgpu_buffer_rsrc_t rx = wf_rsrc(a.x1, tbytes), r1 = wf_rsrc(a.g1, gbytes), r2 = wf_rsrc(a.g2, gbytes);
            int ty, tx;
            const unsigned origin = tile_origin(T, ty, tx);
            const int n_img = fdiv(T, a.m_per_img);
            const unsigned gorigin = a.gshuffle ? (unsigned)(((n_img * 2 * a.H + 2 * ty * WF_TH + (cg >> 1)) * gW + 2 * tx * WF_TW + (cg & 1)) * 256) : origin;
            // (tiles past the end are never committed; their offsets lie past the tensor or on its last rows: zeros / unused values)
            const unsigned e = (ty == 0 ? 1u : 0u) | (ty == tiles_y - 1 ? 2u : 0u) | (tx == 0 ? 4u : 0u) | (tx == a.tiles_x - 1 ? 8u : 0u);
            st.edge = e != 0u;                                               // wave-uniform
            st.bad = xflags & (e * 0x01111111u);                             // non-zero nibble k: item k lies outside the image
#pragma unroll
            for (int k = 0; k < WF_XITEMS; ++k)
                st.sx[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, origin + (unsigned)xrel[k], 0, 0));
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const unsigned voff = gorigin + (unsigned)(grel0 + k * gstep);
                st.s1[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r1, voff, 0, 0));
                st.s2[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r2, voff, 0, 0));
            }
        };
        auto commit_x = [&](auto easy_, auto edge_, const Stage& st, unsigned char* img) {
            constexpr int EASY = decltype(easy_)::value;
            constexpr bool EDGE = decltype(edge_)::value;
            // x: lrelu(a x + d) (a = 1, d = 0, slope = 1 degenerate to ACT / NONE), zero outside the image
#pragma unroll
            for (int k = 0; k < WF_XITEMS; ++k) {
                const bool ok = !EDGE || ((st.bad >> (4 * k)) & 15u) == 0u;
                f32x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float v = lrelu_t<EASY>(ka[j] * st.sx[k][j] + kd[j], xslope);
                    o[j] = ok ? v : 0.f;
                }
                if (k < WF_XITEMS - 1 || !last_beyond) {
                    if constexpr (SPLIT) wf_store_split(img + xlds0 + k * 16 * PB, o);
                    else *reinterpret_cast<f32x4*>(img + xlds0 + k * 16 * PB) = o;
                }
            }
        };
        auto commit = [&](const Stage& st, int b) {
            unsigned char* img = lds + b * BUF;
            using T1 = std::integral_constant<int, 1>; using T0 = std::integral_constant<int, 0>;
            if (easy_slope) { if (st.edge) commit_x(T1{}, std::true_type{}, st, img); else commit_x(T1{}, std::false_type{}, st, img); }
            else { if (st.edge) commit_x(T0{}, std::true_type{}, st, img); else commit_x(T0{}, std::false_type{}, st, img); }
            // gradient: BatchNorm backward (through the activation for BNACT_BWD); the bias gradient is summed on the way
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                f32x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float g = st.s1[k][j];
                    const float bx = st.s2[k][j];
                    if (GPRO == SISR_PRO_BNACT_BWD) g = qs[j] * bx + qt[j] > 0.f ? g : gslope * g;
                    if (GPRO == SISR_PRO_ACT_BWD) o[j] = bx > 0.f ? g : gslope * g;          // act'(pre-activation) * gradient
                    else o[j] = qa[j] * g + qb[j] * bx + qd[j];
                }
                bsum += o;
                if constexpr (SPLIT) wf_store_split(img + glds0 + k * 16 * PB, o);
                else *reinterpret_cast<f32x4*>(img + glds0 + k * 16 * PB) = o;
            }
        };

        int T = t_first;
        issue(T, stA);
        issue(T + t_step, stB);
        if (T < a.total) commit(stA, 0);
        __syncthreads();
        WFA_DECL;
        // unrolled by two: each staging set has a fixed name in each half (stB holds tile T + grid in the first)
        int cur = 0;
        while (T < a.total) {
            issue(T + 2 * t_step, stA);
            if (T + t_step < a.total) commit(stB, cur ^ 1);
            WFA_SYNC();           // the next tile's images are complete; the consumers have finished reading this one
            T += t_step; cur ^= 1;
            if (T >= a.total) break;
            issue(T + 2 * t_step, stB);
            if (T + t_step < a.total) commit(stA, cur ^ 1);
            WFA_SYNC();
            T += t_step; cur ^= 1;
        }
        if (wave == 4) WFA_STORE(2);
        __builtin_amdgcn_s_waitcnt(0x0F70);                  // vmcnt(0): the prefetch loads past the last tile
        if (a.bias_slab != nullptr) *reinterpret_cast<f32x4*>(lds + pt * 16) = bsum;      // the images are free by now
    } else {
        // ---- consumers: (32 output channels) x (9 taps x 32 input channels) in accumulators, across all tiles ---------------
        f32x16 acc[9];
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
        // operand lane roles (sisr_dev.h): A = x[pixel 2 s + kk][ci = l31], B = dy[pixel 2 s + kk][co = l31]
        const int xoff = kk * WF_PB + (32 * gq + l31) * 4, doff = WF_XBYTES + kk * WF_PB + (32 * h + l31) * 4;
        WFA_MARK(4);
        __syncthreads();
        WFA_MARK(5);
        WFA_DECL;
        int cur = 0;
        // (SPLIT) transposing-read lane roles: 16-lane group grp -> (channel half grp & 1, pixel half grp >> 1 of the 16-pixel K
        // step); inside the group lane 4q + p addresses (pixel row q, channels 4p .. 4p + 3)
        const int grp = lane >> 4, tq = (lane & 15) >> 2, tp = lane & 3;
        const int rd_pix = (8 * (grp >> 1) + tq) * WS_PB + (16 * (grp & 1) + 4 * tp) * 2;
        for (int T = t_first; T < a.total; T += t_step, cur ^= 1) {
            if constexpr (SPLIT) {
                // halo rows R = 0 .. 5: the three column shifts of row R against the gradient rows R, R - 1, R - 2; small terms first
                const unsigned char* xs = lds + cur * BUF + rd_pix + 64 * gq;
                const unsigned char* ds = lds + cur * BUF + XB + rd_pix + 64 * h;
                bf16x8 dyh[WF_TH], dyl[WF_TH];
#pragma unroll
                for (int R = 0; R < WF_IH; ++R) {
                    if (R < WF_TH) { dyh[R] = ws_frag(ds + R * 16 * WS_PB); dyl[R] = ws_frag(ds + R * 16 * WS_PB + 128); }
                    bf16x8 xh[3], xl[3];
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) { xh[kx] = ws_frag(xs + (R * WF_IW + kx) * WS_PB); xl[kx] = ws_frag(xs + (R * WF_IW + kx) * WS_PB + 128); }
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky) {
                        const int r = R - ky;
                        if (r < 0 || r >= WF_TH) continue;
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx) {
                            acc[ky * 3 + kx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl[kx], dyh[r], acc[ky * 3 + kx], 0, 0, 0);
                            acc[ky * 3 + kx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh[kx], dyl[r], acc[ky * 3 + kx], 0, 0, 0);
                            acc[ky * 3 + kx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh[kx], dyh[r], acc[ky * 3 + kx], 0, 0, 0);
                        }
                    }
                }
                WFA_SYNC();
                continue;
            }
            const unsigned char* xb = lds + cur * (WF_XBYTES + WF_DBYTES) + xoff;
            const unsigned char* db = lds + cur * (WF_XBYTES + WF_DBYTES) + doff;
            // halo row R, K step s (pixels 2 s, 2 s + 1 of the row): the three column shifts of the x row against the
            // gradient rows R, R - 1, R - 2 -- every address is base + immediate
#pragma unroll
            for (int R = 0; R < WF_IH; ++R) {
#pragma unroll
                for (int s = 0; s < WF_TW / 2; ++s) {
                    float xf[3], dyf[3];
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) xf[kx] = *reinterpret_cast<const float*>(xb + (R * WF_IW + 2 * s + kx) * WF_PB);
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky) {
                        const int r = R - ky;
                        if (r < 0 || r >= WF_TH) continue;
                        dyf[ky] = *reinterpret_cast<const float*>(db + (r * WF_TW + 2 * s) * WF_PB);
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx)
                            acc[ky * 3 + kx] = __builtin_amdgcn_mfma_f32_32x32x2f32(xf[kx], dyf[ky], acc[ky * 3 + kx], 0, 0, 0);
                    }
                }
            }
            WFA_SYNC();
        }
        if (wave == 0) WFA_STORE(0);
        WFA_MARK(6);
        // one slab per workgroup: [chunk gq][filter row ky][kx * 33 + ci][64 co]
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i)
                sl[((gq * 3 + t / 3) * WF_KROWP + (t % 3) * WF_PS + mfma_row(i, lane)) * a.cout_pad + 64 * cg + 32 * h + l31] = acc[t][i];
    }
    if (a.bias_slab != nullptr) {                       // ... and its bias row
        __syncthreads();
        if (tid < 64) {
            const float* bs = reinterpret_cast<const float*>(lds);
            const int q4 = tid >> 2, j = tid & 3;
            float s = 0.f;
            for (int i = 0; i < 16; ++i) s += bs[(q4 + 16 * i) * 4 + j];                  // fixed order: deterministic
            a.bias_slab[(int64_t)t_first * a.slab_stride + 64 * cg + tid] = s;
        }
    }
    WFA_MARK(7);
}

template <int GPRO, bool SPLIT>
__global__ void __launch_bounds__(WF_THREADS, 2) wgrad_trunk_f32_kernel(const WTrunkF32Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    wgrad_trunk_f32_body<GPRO, SPLIT>(a, blockIdx.x, gridDim.x, lds);
}

// several layers of one gradient-prologue kind in ONE launch (as wgrad_trunk.hip's table kernel): workgroups [z * wpl, (z + 1) * wpl)
// serve table[z] and write wpl slabs per layer instead of one per CU (256 x 147 KB written and re-read per layer)
template <int GPRO, bool SPLIT>
__global__ void __launch_bounds__(WF_THREADS, 2) wgrad_trunk_f32_table_kernel(const WTrunkF32Args* __restrict__ table, const int wpl) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int z = (int)blockIdx.x / wpl;
    const WTrunkF32Args a = table[z];
    wgrad_trunk_f32_body<GPRO, SPLIT>(a, (int)blockIdx.x - z * wpl, wpl, lds);
}

// ---- host ----------------------------------------------------------------------------------------------------------
static int wf_grid(const SisrWgradDesc* d) {
    const int total = d->N * (d->H / WF_TH) * (d->W / WF_TW);
    const int cus = sisr_cu_slots();
    const int G = d->Cout == 256 ? 4 : 1;       // cout groups: each tile stream is served by G workgroups
    const int slots = std::max(1, cus / G);
    const int rounds = (total + slots - 1) / slots;
    return G * ((total + rounds - 1) / rounds);  // equal shares
}

extern "C" int sisr_wgrad_trunk_f32_eligible(const SisrWgradDesc* d) {
    const char* sw = getenv("SISR_TRUNK");                      // A/B switch: SISR_TRUNK=0 keeps the generic kernel
    if (!d || (sw && sw[0] == '0')) return 0;
    const char* sw2 = getenv("SISR_TRUNK_WGRAD");
    if (sw2 && sw2[0] == '0') return 0;
    if (d->Cin != 64 || d->KH != 3 || d->KW != 3 || d->stride != 1 || d->pad_y != 1 || d->pad_x != 1) return 0;
    // Cout = 64 (trunk: BatchNorm-backward gradient prologues), or 256 with the gradient stored shuffled and an
    // activation-backward prologue -- the upscale conv
    const char* swu = getenv("SISR_TRUNK_UP");                 // A/B switch for the upscale conv alone
    const bool up = !(swu && swu[0] == '0') && d->Cout == 256 && d->g_mode == SISR_X_NHWC_UNSHUFFLE2 && d->CoutPad == 256 &&
                    d->gpro_mode == SISR_PRO_ACT_BWD && (int64_t)d->N * d->H * d->W * 1024 < (1ll << 31);
    if (!up && (d->Cout != 64 || d->g_mode != SISR_X_NHWC || d->CoutPad != 64)) return 0;
    if (d->x_mode != SISR_X_NHWC || d->x_bf16 || d->g_bf16) return 0;
    if (d->Ho != d->H || d->Wo != d->W || (d->H % WF_TH) || (d->W % WF_TW)) return 0;
    if (d->CK != 32 || d->PS != WF_PS || d->KROWP != WF_KROWP || d->n_chunk != 2) return 0;
    if ((int64_t)d->N * d->H * d->W * 256 >= (1ll << 31)) return 0;
    if (d->N * (d->H / WF_TH) * (d->W / WF_TW) >= 65536) return 0;
    const bool xp = d->pro_mode == SISR_PRO_NONE || d->pro_mode == SISR_PRO_ACT || d->pro_mode == SISR_PRO_AFFINE_ACT;
    const bool gp = up || d->gpro_mode == SISR_PRO_BNBWD || d->gpro_mode == SISR_PRO_BNACT_BWD;
    return xp && gp ? 1 : 0;
}

// slabs a launch of this descriptor writes (rows of `slab` at slab_stride)
extern "C" int sisr_wgrad_thin_eligible(const SisrWgradDesc* d);
int sisr_wgrad_thin_slabs(const SisrWgradDesc* d);                            // wgrad_thin.hip
extern "C" int sisr_wgrad_toimage_f32_eligible(const SisrWgradDesc* d);
int sisr_wgrad_toimage_slabs(const SisrWgradDesc* d);                         // wgrad_toimage.hip

extern "C" int sisr_wgrad_f32_slabs(const SisrWgradDesc* d) {
    if (!d) return SISR_E_BADARG;
    if (sisr_wgrad_trunk_f32_eligible(d)) return wf_grid(d) / (d->Cout == 256 ? 4 : 1);
    if (sisr_wgrad_thin_eligible(d)) return sisr_wgrad_thin_slabs(d);
    return sisr_wgrad_toimage_f32_eligible(d) ? sisr_wgrad_toimage_slabs(d) : d->n_slabs;
}

template <int GPRO, bool SPLIT>
static int launch_wf_t(const WTrunkF32Args& a, int grid, hipStream_t st) {
    constexpr int lds_bytes = SPLIT ? 2 * (WS_XBYTES + WS_DBYTES) : 2 * (WF_XBYTES + WF_DBYTES);
    static SisrLdsCap cap;
    if (int e = sisr_raise_lds_cap(cap, reinterpret_cast<const void*>(&wgrad_trunk_f32_kernel<GPRO, SPLIT>), lds_bytes)) return e;
    hipLaunchKernelGGL((wgrad_trunk_f32_kernel<GPRO, SPLIT>), dim3(grid), dim3(WF_THREADS), lds_bytes, st, a);
    SISR_CHECK_LAUNCH();
    return 0;
}
template <int GPRO>
static int launch_wf(const WTrunkF32Args& a, bool split, int grid, hipStream_t st) {
    return split ? launch_wf_t<GPRO, true>(a, grid, st) : launch_wf_t<GPRO, false>(a, grid, st);
}

static WTrunkF32Args wf_args(const SisrWgradDesc* d);

// called by sisr_conv2d_wgrad_f32 for eligible descriptors
int sisr_wgrad_trunk_f32_launch(const SisrWgradDesc* d, hipStream_t st) {
    if (operand_needs_x2(d->gpro_mode) && !d->g2) return SISR_E_BADARG;
    if (d->pro_mode == SISR_PRO_AFFINE_ACT && (!d->pa || !d->pd)) return SISR_E_BADARG;
    if (d->gpro_mode != SISR_PRO_ACT_BWD && (!d->qa || !d->qb || !d->qd || (d->gpro_mode == SISR_PRO_BNACT_BWD && (!d->qs || !d->qt))))
        return SISR_E_BADARG;
    const WTrunkF32Args a = wf_args(d);
    const int grid = wf_grid(d);
    if (d->gpro_mode == SISR_PRO_ACT_BWD) return launch_wf<SISR_PRO_ACT_BWD>(a, d->mfma_split != 0, grid, st);
    if (d->gpro_mode == SISR_PRO_BNBWD) return launch_wf<SISR_PRO_BNBWD>(a, d->mfma_split != 0, grid, st);
    return launch_wf<SISR_PRO_BNACT_BWD>(a, d->mfma_split != 0, grid, st);
}

static WTrunkF32Args wf_args(const SisrWgradDesc* d) {
    WTrunkF32Args a{};
    a.x1 = d->x1; a.g1 = d->g1; a.g2 = d->g2;
    a.pa = d->pa; a.pd = d->pd; a.xslope_p = d->pro_slope_p; a.xslope = d->pro_slope;
    a.qa = d->qa; a.qb = d->qb; a.qd = d->qd; a.qs = d->qs; a.qt = d->qt;
    a.gslope_p = d->gpro_slope_p; a.gslope = d->gpro_slope;
    a.slab = d->slab; a.bias_slab = d->bias_slab; a.slab_stride = d->slab_stride;
    a.N = d->N; a.H = d->H; a.W = d->W;
    a.tiles_x = d->W / WF_TW; a.per_img = (d->H / WF_TH) * a.tiles_x; a.total = d->N * a.per_img;
    a.m_tiles_x = fdiv_magic(a.tiles_x); a.m_per_img = fdiv_magic(a.per_img);
    a.xpro = d->pro_mode;
    a.glog = d->Cout == 256 ? 2 : 0; a.cout_pad = d->Cout == 256 ? 256 : 64; a.gshuffle = d->g_mode == SISR_X_NHWC_UNSHUFFLE2 ? 1 : 0;
    return a;
}

// ---- a batch of trunk layers (Cout = 64, one gradient-prologue kind, one mfma_split setting) --------------------------------------
extern "C" int sisr_wgrad_trunk_f32_batch_arg_bytes(void) { return (int)sizeof(WTrunkF32Args); }

static int wf_batch_check(const SisrWgradDesc* descs, int n) {
    if (!descs || n <= 0 || n > 4096) return SISR_E_BADARG;
    for (int i = 0; i < n; ++i) {
        const SisrWgradDesc* d = descs + i;
        if (!sisr_wgrad_trunk_f32_eligible(d) || d->Cout != 64 || d->gpro_mode != descs[0].gpro_mode) return SISR_E_BADARG;
        if ((d->mfma_split != 0) != (descs[0].mfma_split != 0)) return SISR_E_BADARG;
        if (!d->x1 || !d->g1 || !d->g2 || !d->slab || d->slab_stride < d->slab_elems) return SISR_E_BADARG;
        if (d->pro_mode == SISR_PRO_AFFINE_ACT && (!d->pa || !d->pd)) return SISR_E_BADARG;
        if (!d->qa || !d->qb || !d->qd || (d->gpro_mode == SISR_PRO_BNACT_BWD && (!d->qs || !d->qt))) return SISR_E_BADARG;
    }
    return 0;
}

extern "C" int sisr_wgrad_trunk_f32_batch_args(const SisrWgradDesc* descs, int32_t n, void* args_host) {
    if (!args_host) return SISR_E_BADARG;
    if (int e = wf_batch_check(descs, n)) return e;
    for (int i = 0; i < n; ++i) {
        const WTrunkF32Args a = wf_args(descs + i);
        std::memcpy(static_cast<unsigned char*>(args_host) + (size_t)i * sizeof(WTrunkF32Args), &a, sizeof(WTrunkF32Args));
    }
    return 0;
}

template <int GPRO, bool SPLIT>
static int launch_wf_table(const WTrunkF32Args* table, int n, int wpl, hipStream_t st) {
    constexpr int lds_bytes = SPLIT ? 2 * (WS_XBYTES + WS_DBYTES) : 2 * (WF_XBYTES + WF_DBYTES);
    static SisrLdsCap cap;
    if (int e = sisr_raise_lds_cap(cap, reinterpret_cast<const void*>(&wgrad_trunk_f32_table_kernel<GPRO, SPLIT>), lds_bytes)) return e;
    hipLaunchKernelGGL((wgrad_trunk_f32_table_kernel<GPRO, SPLIT>), dim3(n * wpl), dim3(WF_THREADS), lds_bytes, st, table, wpl);
    SISR_CHECK_LAUNCH();
    return 0;
}

extern "C" int sisr_wgrad_trunk_f32_batch(const SisrWgradDesc* descs, const void* args_dev, int32_t n, int32_t wgs_per_layer, void* stream) {
    if (!args_dev || wgs_per_layer <= 0 || (int64_t)n * wgs_per_layer > 65535) return SISR_E_BADARG;
    if (int e = wf_batch_check(descs, n)) return e;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const WTrunkF32Args* table = static_cast<const WTrunkF32Args*>(args_dev);
    const bool split = descs[0].mfma_split != 0;
    if (descs[0].gpro_mode == SISR_PRO_BNBWD)
        return split ? launch_wf_table<SISR_PRO_BNBWD, true>(table, n, wgs_per_layer, st) : launch_wf_table<SISR_PRO_BNBWD, false>(table, n, wgs_per_layer, st);
    return split ? launch_wf_table<SISR_PRO_BNACT_BWD, true>(table, n, wgs_per_layer, st)
                 : launch_wf_table<SISR_PRO_BNACT_BWD, false>(table, n, wgs_per_layer, st);
}
