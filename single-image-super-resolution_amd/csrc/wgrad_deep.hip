// wgrad_deep.hip -- weight gradient of the 3x3 layers with channels in 64s on the bf16 matrix cores
// (v_mfma_f32_32x32x16_bf16, fp32 accumulate): the discriminator's conv stack (model_discriminator.py:10,39-44) and the
// generator's trunk at the sizes the persistent kernel (wgrad_trunk.hip) does not take.
//
//   dW[tap][ci][co] = sum over output pixels p of  x(p * stride + tap - 1)[ci] * dy(p)[co]
//
// Why a second family next to wgrad_bf16.hip: there a workgroup owns 32 input channels x 64 output channels, restages the dy
// tile once per 32-channel chunk and does stage -> barrier -> MFMAs -> barrier with nothing in flight across a barrier; the
// seven layers of the discriminator measured 26-54 us each at 96 x 96 (2.7 GFLOP: 1 us of MFMA time).  Here
//   * a workgroup owns 64 ci x 64 co x all 9 taps: wave (h, g) of the four CONSUMER waves keeps nine 32 x 32 accumulators
//     (ci half h, co half g) over ALL of the workgroup's pixel tiles -- a staged x pixel is used by 2 x 9 MFMAs, a dy pixel by
//     2 x 9 -- and the dy tile is staged once per 64 input channels;
//   * the contraction runs over the tile's positions in HALO COORDINATES: dy is laid out in LDS on the pitch of the x halo
//     (zeros in the padding columns / rows), so position j of tap (ky, kx) pairs dy[j] with x[j + ky * IW + kx] -- one constant
//     per tap, no per-row bookkeeping, no padding of 6 / 12 / 24 wide rows to 16.  Stride 2: the x halo is stored as four
//     (row parity, column parity) planes, position j of tap (ky, kx) reads plane (ky & 1, kx & 1) at j + (ky >> 1) * IWd + (kx >> 1);
//   * tiles are TH rows of the flattened (image, row) space x TW columns (as conv_deep.hip), so small maps straddle images;
//   * four PRODUCER waves stage the next tile (global -> registers two tiles ahead -> lazy-operand transform -> LDS, the other
//     buffer) while the consumer waves run the MFMAs of this one: one workgroup barrier per tile, and the producers' vector
//     work shares the SIMDs with the consumers' matrix work;
//   * the bias gradient is summed by the producers from the dy values passing through their registers.
// Output: one slab per pixel block in the layout of wgrad_bf16.hip ([32-channel chunk][tap][ci 32][CoutPad], bias partials
// behind it), reduced in fixed order by sisr_slab_reduce_f32 -- deterministic.  The gradient part of a slab is bf16 unless
// SISR_SLAB_BF16=0 (as the persistent kernel's slabs: half the bytes written and re-read).
#include "sisr_dev.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "sisr_bf16_stage.h"

#define WD_THREADS 512
#define WD_PROD 256                     // producer threads (waves 4 .. 7)
#define WD_PSB 64                       // bytes of a pixel in one 32-channel half image (16 banks: conflict-free transposing reads)
#define WD_CONST_BYTES (7 * 64 * 4)     // prologue constants of the workgroup's channels: x: a, d; dy: a, b, d, s, t

__device__ __forceinline__ bf16x8 wd_frag8(const unsigned char* p) {
    const s16x4 lo = lds_tr16(reinterpret_cast<const __bf16*>(p)), hi = lds_tr16(reinterpret_cast<const __bf16*>(p + 4 * WD_PSB));
    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}

// 8 channels of one pixel through the operand's prologue; bf16 in, bf16 out; SUM: add the transformed fp32 values to sum
template <int PRO, bool SUM>
__device__ __forceinline__ u32x4 wd_apply8(u32x4 a, u32x4 b, const f32x8& ka, const f32x8& kb, const f32x8& kd, const f32x8& ks,
                                           const f32x8& kt, float slope, bool ok, f32x8& sum) {
    u32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float a0 = __uint_as_float(a[j] << 16), a1 = __uint_as_float(a[j] & 0xFFFF0000u);
        float b0 = 0.f, b1 = 0.f;
        if (PRO == SISR_PRO_BNBWD || PRO == SISR_PRO_BNACT_BWD || PRO == SISR_PRO_ACT_BWD) {
            b0 = __uint_as_float(b[j] << 16); b1 = __uint_as_float(b[j] & 0xFFFF0000u);
        }
        float r0, r1;
        if (PRO == SISR_PRO_NONE) { r0 = a0; r1 = a1; }
        else if (PRO == SISR_PRO_ACT) { r0 = lrelu(a0, slope); r1 = lrelu(a1, slope); }
        else if (PRO == SISR_PRO_AFFINE_ACT) {
            r0 = lrelu(ka[2 * j] * a0 + kd[2 * j], slope); r1 = lrelu(ka[2 * j + 1] * a1 + kd[2 * j + 1], slope);
        } else if (PRO == SISR_PRO_BNBWD) {
            r0 = ka[2 * j] * a0 + kb[2 * j] * b0 + kd[2 * j];
            r1 = ka[2 * j + 1] * a1 + kb[2 * j + 1] * b1 + kd[2 * j + 1];
        } else if (PRO == SISR_PRO_BNACT_BWD) {
            const float z0 = ks[2 * j] * b0 + kt[2 * j], z1 = ks[2 * j + 1] * b1 + kt[2 * j + 1];
            const float g0 = z0 > 0.f ? a0 : slope * a0, g1 = z1 > 0.f ? a1 : slope * a1;
            r0 = ka[2 * j] * g0 + kb[2 * j] * b0 + kd[2 * j];
            r1 = ka[2 * j + 1] * g1 + kb[2 * j + 1] * b1 + kd[2 * j + 1];
        } else {                                               // ACT_BWD
            r0 = b0 > 0.f ? a0 : slope * a0; r1 = b1 > 0.f ? a1 : slope * a1;
        }
        // prologues with f(0) != 0: positions outside the image / tile must be zero AFTER the transform
        if (!(PRO == SISR_PRO_NONE || PRO == SISR_PRO_ACT || PRO == SISR_PRO_ACT_BWD)) { r0 = ok ? r0 : 0.f; r1 = ok ? r1 : 0.f; }
        if (SUM) { sum[2 * j] += r0; sum[2 * j + 1] += r1; }
        o[j] = pack_bf16x2(r0, r1);
    }
    return o;
}

// all but lgkmcnt: the producers' prefetch loads stay in flight across the workgroup barrier (NOT __syncthreads(): its fence
// waits for vmcnt(0)); every LDS operation of the wave is complete before it arrives
#define WD_BARRIER()                                                                                                               \
    do {                                                                                                                           \
        asm volatile("" ::: "memory");                                                                                             \
        __builtin_amdgcn_s_waitcnt(0xC07F);                                                                                        \
        __builtin_amdgcn_s_barrier();                                                                                              \
        asm volatile("" ::: "memory");                                                                                             \
    } while (0)

struct WdTile {
    int q0, ox0, pb0;               // first flattened output row, first output column, padded row of the halo's first row
    int npix, npos;                 // x halo pixels; positions of the contraction (padded to 16)
    int qend;                       // one past the last flattened output row
};

// S: stride.  NITX / NITD: 16-byte staging items per producer thread (x halo, dy positions).  TWO: two-tensor dy prologue.
template <int S, int NITX, int NITD, bool TWO>
__device__ __forceinline__ void wgrad_deep_body(const SisrWgradDesc& d, unsigned char* lds, int blk, int pblk) {
    const SisrWgradDeepPlan& p = d.deep;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cib = blk / p.n_cob, cob = blk - cib * p.n_cob;
    const int t_begin = pblk * p.tiles_per_pb, ntile = min(p.n_tiles, t_begin + p.tiles_per_pb) - t_begin;
    const int XH = p.XP_max * WD_PSB, DH = p.NPOS_max * WD_PSB;      // bytes of one 32-channel half image
    const int BUF = 2 * (XH + DH);
    unsigned char* cst = lds + 2 * BUF;
    const int NQ = d.N * d.Ho, IW = p.IW, IWd = p.IWd;
    const int XPL = S == 2 ? p.XP_max >> 2 : p.XP_max;               // pixels of one parity plane

    auto rbase = [&](int q) { const int n = fdiv(q, p.m_ho); return n * p.PR + (q - n * d.Ho) * S; };
    auto tile_geom = [&](int t) {
        WdTile g;
        const int tq = fdiv(t, p.m_tiles_x), tx = t - tq * p.tiles_x;
        g.q0 = tq * p.TH; g.ox0 = tx * p.TW;
        g.qend = min(g.q0 + p.TH, NQ);
        g.pb0 = rbase(g.q0);
        const int span = rbase(g.qend - 1) - g.pb0;                  // a multiple of S (PR is)
        g.npix = (span + 3) * IW;
        g.npos = ((span / S + 1) * IWd + 15) & ~15;
        return g;
    };

    // stale LDS must never hold a NaN pattern (0 x NaN): every byte zero once; then the prologue constants of this workgroup's channels
    for (int o = tid * 16; o < 2 * BUF; o += WD_THREADS * 16) *reinterpret_cast<u32x4*>(lds + o) = u32x4{0u, 0u, 0u, 0u};
    if (tid < 7 * 64) {
        const int k = tid >> 6, c = tid & 63;
        const float* src = k == 0 ? d.pa : k == 1 ? d.pd : k == 2 ? d.qa : k == 3 ? d.qb : k == 4 ? d.qd : k == 5 ? d.qs : d.qt;
        const int ch = (k < 2 ? cib : cob) * 64 + c;
        reinterpret_cast<float*>(cst)[tid] = src ? src[ch] : 0.f;
    }
    __syncthreads();

    if (wave >= 4) {
        // ================================================= producers ==========================================================
        const int ptid = tid - WD_PROD;
        const int oct = ptid & 7, pq = ptid >> 3;                     // channel octet of every item of this thread; first pixel
        const unsigned lbase_x = (unsigned)((oct >> 2) * XH + (oct & 3) * 16);
        const unsigned lbase_d = (unsigned)(2 * XH + (oct >> 2) * DH + (oct & 3) * 16 + pq * WD_PSB);
        unsigned lofx[NITX];
#pragma unroll
        for (int u = 0; u < NITX; ++u) {
            const int pix = pq + 32 * u;
            if (S == 1) lofx[u] = lbase_x + (unsigned)(pix * WD_PSB);
            else {
                const int hr = fdiv(pix, p.m_iw), hc = pix - hr * IW;
                lofx[u] = lbase_x + (unsigned)(((((hr & 1) * 2 + (hc & 1)) * XPL) + (hr >> 1) * IWd + (hc >> 1)) * WD_PSB);
            }
        }
        const unsigned xbytes = (unsigned)d.N * (unsigned)d.H * (unsigned)d.W * (unsigned)d.Cin * 2u;
        const unsigned gbytes = (unsigned)d.N * (unsigned)d.Ho * (unsigned)d.Wo * (unsigned)d.Cout * 2u;
        const __amdgpu_buffer_rsrc_t rx = bf_rsrc(d.x1, xbytes);
        const __amdgpu_buffer_rsrc_t rg1 = bf_rsrc(d.g1, gbytes), rg2 = bf_rsrc(TWO ? d.g2 : d.g1, gbytes);
        const unsigned cx = (unsigned)(cib * 64 + oct * 8) * 2u, cg = (unsigned)(cob * 64 + oct * 8) * 2u;
        const float xslope = d.pro_slope_p ? d.pro_slope_p[0] : d.pro_slope;
        const float gslope = d.gpro_slope_p ? d.gpro_slope_p[0] : d.gpro_slope;
#ifdef WD_DBG_NOXFORM                   // developer build: every prologue is a plain copy -- what the transforms' vector work costs
        const int xpro = SISR_PRO_NONE, gpro = TWO ? SISR_PRO_ACT_BWD : SISR_PRO_NONE;
#else
        const int xpro = d.pro_mode, gpro = d.gpro_mode;
#endif
        f32x8 bsum = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

        struct Set {
            u32x4 x[NITX], ga[NITD], gb[TWO ? NITD : 1];
            unsigned okx, okg;
            int npix, npos;
        };
        auto issue = [&](Set& s, int t) {
            const WdTile g = tile_geom(t);
            s.npix = g.npix; s.npos = g.npos;
            s.okx = 0; s.okg = 0;
            const int ix0 = g.ox0 * S - 1;
#pragma unroll
            for (int u = 0; u < NITX; ++u) {
                const int pix = pq + 32 * u;
                const int hr = fdiv(pix, p.m_iw), hc = pix - hr * IW;
                const int P = g.pb0 + hr;
                const int n = fdiv(P, p.m_pr), iy = P - n * p.PR - 1, ix = ix0 + hc;
                const bool ok = pix < g.npix && n < d.N && (unsigned)iy < (unsigned)d.H && (unsigned)ix < (unsigned)d.W;
                const unsigned off = ok ? (unsigned)(((n * d.H + iy) * d.W + ix) * d.Cin) * 2u + cx : 0x80000000u;
                s.x[u] = __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0);
                s.okx |= ok ? (1u << u) : 0u;
            }
#pragma unroll
            for (int u = 0; u < NITD; ++u) {
                const int pos = pq + 32 * u;
                const int hrow = fdiv(pos, p.m_iwd), c = pos - hrow * IWd;
                const int P = g.pb0 + hrow * S;
                const int n = fdiv(P, p.m_pr), oy = (P - n * p.PR) / S;
                const int q = n * d.Ho + oy, ox = g.ox0 + c;
                const bool ok = pos < g.npos && oy < d.Ho && q < g.qend && c < p.TW && ox < d.Wo;
                const unsigned off = ok ? (unsigned)((q * d.Wo + ox) * d.Cout) * 2u + cg : 0x80000000u;
                s.ga[u] = __builtin_amdgcn_raw_buffer_load_b128(rg1, off, 0, 0);
                if constexpr (TWO) s.gb[u] = __builtin_amdgcn_raw_buffer_load_b128(rg2, off, 0, 0);
                s.okg |= ok ? (1u << u) : 0u;
            }
        };
        auto ldc = [&](int k) {                                   // constants row k, this thread's octet
            const float* r = reinterpret_cast<const float*>(cst) + k * 64 + oct * 8;
            f32x8 v;
            const f32x4 lo = *reinterpret_cast<const f32x4*>(r), hi = *reinterpret_cast<const f32x4*>(r + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) { v[j] = lo[j]; v[4 + j] = hi[j]; }
            return v;
        };
        const f32x8 zero8 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        auto commit_x = [&](const Set& s, unsigned char* buf, auto pro_c) {
            constexpr int PRO = decltype(pro_c)::value;
            f32x8 ka = zero8, kd = zero8, dummy = zero8;
            if (PRO == SISR_PRO_AFFINE_ACT) { ka = ldc(0); kd = ldc(1); }
#pragma unroll
            for (int u = 0; u < NITX; ++u) {
                const u32x4 v = wd_apply8<PRO, false>(s.x[u], s.x[u], ka, zero8, kd, zero8, zero8, xslope, (s.okx >> u) & 1u, dummy);
                if (pq + 32 * u < s.npix) *reinterpret_cast<u32x4*>(buf + lofx[u]) = v;
            }
        };
        auto commit_g = [&](const Set& s, unsigned char* buf, auto pro_c) {
            constexpr int PRO = decltype(pro_c)::value;
            f32x8 ka = zero8, kb = zero8, kd = zero8, ks = zero8, kt = zero8;
            if (PRO == SISR_PRO_BNBWD || PRO == SISR_PRO_BNACT_BWD) { ka = ldc(2); kb = ldc(3); kd = ldc(4); }
            if (PRO == SISR_PRO_BNACT_BWD) { ks = ldc(5); kt = ldc(6); }
#pragma unroll
            for (int u = 0; u < NITD; ++u) {
                const u32x4 v = wd_apply8<PRO, true>(s.ga[u], TWO ? s.gb[TWO ? u : 0] : s.ga[u], ka, kb, kd, ks, kt, gslope, (s.okg >> u) & 1u, bsum);
                if (pq + 32 * u < s.npos) *reinterpret_cast<u32x4*>(buf + lbase_d + u * (32 * WD_PSB)) = v;
            }
        };
        auto commit = [&](const Set& s, unsigned char* buf) {
            if (xpro == SISR_PRO_AFFINE_ACT) commit_x(s, buf, std::integral_constant<int, SISR_PRO_AFFINE_ACT>{});
            else if (xpro == SISR_PRO_ACT) commit_x(s, buf, std::integral_constant<int, SISR_PRO_ACT>{});
            else commit_x(s, buf, std::integral_constant<int, SISR_PRO_NONE>{});
            if constexpr (TWO) {
                if (gpro == SISR_PRO_BNACT_BWD) commit_g(s, buf, std::integral_constant<int, SISR_PRO_BNACT_BWD>{});
                else if (gpro == SISR_PRO_BNBWD) commit_g(s, buf, std::integral_constant<int, SISR_PRO_BNBWD>{});
                else commit_g(s, buf, std::integral_constant<int, SISR_PRO_ACT_BWD>{});
            } else {
                commit_g(s, buf, std::integral_constant<int, SISR_PRO_NONE>{});
            }
        };

        // two register sets: the loads of tiles i + 1 and i + 2 are in flight while tile i is in the matrix cores
        Set sA, sB;
        issue(sA, t_begin);
        if (ntile > 1) issue(sB, t_begin + 1);
        commit(sA, lds);
        if (ntile > 2) issue(sA, t_begin + 2);
        WD_BARRIER();
        for (int i = 0; i < ntile; i += 2) {
            // consumers: tile i in buffer 0
            if (i + 1 < ntile) { commit(sB, lds + BUF); if (i + 3 < ntile) issue(sB, t_begin + i + 3); }
            WD_BARRIER();
            if (i + 1 >= ntile) break;
            // consumers: tile i + 1 in buffer 1
            if (i + 2 < ntile) { commit(sA, lds); if (i + 4 < ntile) issue(sA, t_begin + i + 4); }
            WD_BARRIER();
        }
        // bias partial of this pixel block: the 32 producer threads of an octet, summed in thread order
        float* bsh = reinterpret_cast<float*>(lds);
        *reinterpret_cast<f32x4*>(bsh + ptid * 8) = f32x4{bsum[0], bsum[1], bsum[2], bsum[3]};
        *reinterpret_cast<f32x4*>(bsh + ptid * 8 + 4) = f32x4{bsum[4], bsum[5], bsum[6], bsum[7]};
        WD_BARRIER();
        if (wave == 4 && cib == 0 && d.bias_slab != nullptr) {
            const int o8 = lane >> 3, j = lane & 7;
            float s = 0.f;
            for (int k = 0; k < 32; ++k) s += bsh[(k * 8 + o8) * 8 + j];
            d.bias_slab[(int64_t)pblk * d.slab_stride + cob * 64 + lane] = s;
        }
        return;
    }

    // =================================================== consumers ============================================================
    const int h = wave >> 1, g = wave & 1;
    const int grp = lane >> 4, tq = (lane & 15) >> 2, tp = lane & 3;
    const int pix_l = 8 * (grp >> 1) + tq, ch_l = 16 * (grp & 1) + 4 * tp;      // transposing-read lane roles (wgrad_bf16.hip)
    const int a0 = h * XH + pix_l * WD_PSB + ch_l * 2;
    const int b0 = 2 * XH + g * DH + pix_l * WD_PSB + ch_l * 2;
    int tapoff[9];
#pragma unroll
    for (int a = 0; a < 9; ++a) {
        const int ky = a / 3, kx = a - 3 * ky;
        tapoff[a] = (S == 1 ? ky * IW + kx : ((ky & 1) * 2 + (kx & 1)) * XPL + (ky >> 1) * IWd + (kx >> 1)) * WD_PSB + a0;
    }
    f32x16 acc[9];
#pragma unroll
    for (int a = 0; a < 9; ++a)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[a][i] = 0.f;

    WD_BARRIER();                                                  // tile 0 is in buffer 0
    for (int i = 0; i < ntile; ++i) {
        const unsigned char* buf = lds + (i & 1) * BUF;
#ifdef WD_DBG_NOK                       // developer build (tools/wd_dbg.sh): the consumers only keep the barriers -- what the producers cost alone
        const int nks = 0;
#else
        const int nks = tile_geom(t_begin + i).npos >> 4;
#endif
        // K steps of 16 positions; taps in two groups (5 + 4): the fragments of one group are requested while the MFMAs of the
        // other run (pinned order -- left alone the scheduler sinks every read to just before its use)
        bf16x8 bcur, bnext, fa[5], fb[4];
        const unsigned char* xp = buf;
        const unsigned char* dp = buf + b0;
        bcur = wd_frag8(dp);
#pragma unroll
        for (int a = 0; a < 5; ++a) fa[a] = wd_frag8(xp + tapoff[a]);
        for (int k = 0; k < nks; ++k) {
#pragma unroll
            for (int a = 0; a < 4; ++a) fb[a] = wd_frag8(xp + tapoff[5 + a]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int a = 0; a < 5; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a], bcur, acc[a], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            xp += 16 * WD_PSB; dp += 16 * WD_PSB;
            if (k + 1 < nks) {
                bnext = wd_frag8(dp);
#pragma unroll
                for (int a = 0; a < 5; ++a) fa[a] = wd_frag8(xp + tapoff[a]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int a = 0; a < 4; ++a) acc[5 + a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[a], bcur, acc[5 + a], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (k + 1 < nks) bcur = bnext;
        }
        WD_BARRIER();                                              // this buffer is free; the next tile is in the other one
    }
    WD_BARRIER();                                                  // (the producers' bias partials)

    // slab of this pixel block: [chunk][tap][ci 32][CoutPad]; wave (h, g) owns chunk 2 cib + h, couts cob * 64 + 32 g ..
    {
        float* sl = d.slab + (int64_t)pblk * d.slab_stride;
        const bool as_bf16 = p.slab_bf16 != 0;
        const int chunk = 2 * cib + h, co = cob * 64 + g * 32 + (lane & 31);
#pragma unroll
        for (int a = 0; a < 9; ++a)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int64_t idx = ((int64_t)(chunk * 9 + a) * 32 + mfma_row(i, lane)) * d.CoutPad + co;
                if (as_bf16) reinterpret_cast<__bf16*>(sl)[idx] = (__bf16)acc[a][i];
                else sl[idx] = acc[a][i];
            }
    }
}

template <int S, int NITX, int NITD, bool TWO>
__global__ void __launch_bounds__(WD_THREADS, 1) wgrad_deep_kernel(const SisrWgradDesc d) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    wgrad_deep_body<S, NITX, NITD, TWO>(d, lds, blockIdx.x, blockIdx.y);
}

// several layers in ONE launch (a flat grid over the members' workgroups): at 96 x 96 a layer alone gives each of its ~150-250 workgroups four tiles -- 15 us of
// prologue and slab stores around 6 us of work.  A batch plans every member for its SHARE of the chip (sisr_wgrad_deep_plan's
// target_wg), so a workgroup walks 3-4 times as many tiles behind the same fixed costs and the batch writes a third of the slabs.
// The weight gradients of a backward pass have no consumer before the optimizer step, so the caller is free to collect them.
template <int S, int NITX, int NITD, bool TWO>
__global__ void __launch_bounds__(WD_THREADS, 1) wgrad_deep_table_kernel(const SisrWgradDesc* __restrict__ table, int n) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    // member of this workgroup: the last one whose first flat index is not beyond it (a 3-D grid sized for the largest member would
    // launch ~60 empty 512-thread workgroups per real one: measured 530 us for 256 workgroups of work)
    int z = 0;
    for (int i = 1; i < n; ++i) z = (int)blockIdx.x >= table[i].deep.batch_first_wg ? i : z;
    const SisrWgradDesc& d = table[z];
    const int local = (int)blockIdx.x - d.deep.batch_first_wg, blocks = d.deep.n_cib * d.deep.n_cob;
    // (tried: members aligned to 8 and eight pixel blocks interleaved so that all channel blocks of a pixel block share an XCD's L2 --
    // no change of the iteration times, cfg4 12.47 vs 12.46-12.50 ms: the simple order stays)
    const int pblk = local / blocks;
    wgrad_deep_body<S, NITX, NITD, TWO>(d, lds, local - pblk * blocks, pblk);
}

// ---- host -----------------------------------------------------------------------------------------------------------------------
static bool wd_slab_bf16() {
    const char* e = getenv("SISR_SLAB_BF16");
    return !(e && e[0] == '0');
}

extern "C" int sisr_wgrad_deep_plan(SisrWgradDesc* d, int32_t target_wg) {
    if (!d) return SISR_E_BADARG;
    SisrWgradDeepPlan& p = d->deep;
    std::memset(&p, 0, sizeof(p));
    if (const char* e = getenv("SISR_WGRAD_DEEP")) if (e[0] == '0') return SISR_E_UNSUPPORTED;   // A/B switch: keep the generic kernel
    if (d->KH != 3 || d->KW != 3 || d->pad_y != 1 || d->pad_x != 1) return SISR_E_UNSUPPORTED;
    if (d->stride != 1 && d->stride != 2) return SISR_E_UNSUPPORTED;
    if ((d->Cin % 64) || (d->Cout % 64) || d->CoutPad != d->Cout || d->N <= 0) return SISR_E_UNSUPPORTED;
    const int S = d->stride;
    if (d->Ho != (d->H + 2 - 3) / S + 1 || d->Wo != (d->W + 2 - 3) / S + 1) return SISR_E_BADARG;
    // 32-bit byte offsets with 2^31 as the out-of-range marker
    if ((int64_t)d->N * d->H * d->W * d->Cin * 2 >= (1ll << 31) || (int64_t)d->N * d->Ho * d->Wo * d->Cout * 2 >= (1ll << 31)) return SISR_E_TOOBIG;
    const int NQ = d->N * d->Ho;
    p.PR = (d->H + 2 + S - 1) / S * S;                      // top padding row + image + bottom padding, a multiple of the stride
    if (p.PR < d->Ho * S || (int64_t)d->N * p.PR + 512 >= 65536 || NQ >= 65536) return SISR_E_TOOBIG;
    const int NITX_CAP = S == 1 ? 6 : 10, NITD_CAP = S == 1 ? 4 : 3;
    auto rbase = [&](int q) { const int n = q / d->Ho; return n * p.PR + (q - n * d->Ho) * S; };
    // tile search: fewest K steps (16 positions each) over the whole layer, a tile charged 2 steps for its barrier and maps
    double best = -1.0;
    int bTH = 0, bTW = 0;
    const int tw_cand[5] = {d->Wo <= 64 ? d->Wo : 0, 32, 16, 12, 8};
    for (int ci = 0; ci < 5; ++ci) {
        const int TW = tw_cand[ci];
        if (TW <= 0 || TW > d->Wo) continue;
        const int IW = S == 1 ? TW + 2 : 2 * TW + 2, IWd = S == 1 ? IW : TW + 1;
        const int tiles_x = (d->Wo + TW - 1) / TW;
        for (int TH = std::min(NQ, 64); TH >= 1; --TH) {
            int span_max = 0;
            int64_t steps = 0;
            const int tiles_q = (NQ + TH - 1) / TH;
            for (int tq = 0; tq < tiles_q; ++tq) {
                const int q0 = tq * TH, qe = std::min(q0 + TH, NQ);
                const int span = rbase(qe - 1) - rbase(q0);
                span_max = std::max(span_max, span);
                steps += (((span / S + 1) * IWd + 15) >> 4) + 2;
            }
            const int IH = span_max + 3, npos = ((span_max / S + 1) * IWd + 15) & ~15;
            if ((IH * IW * 8 + WD_PROD - 1) / WD_PROD > NITX_CAP || (npos * 8 + WD_PROD - 1) / WD_PROD > NITD_CAP) continue;
            const double cost = (double)steps * tiles_x;
            if (best < 0 || cost < best) { best = cost; bTH = TH; bTW = TW; }
        }
    }
    if (best < 0) return SISR_E_UNSUPPORTED;
    p.TH = bTH; p.TW = bTW;
    p.tiles_x = (d->Wo + p.TW - 1) / p.TW; p.tiles_q = (NQ + p.TH - 1) / p.TH;
    p.n_tiles = p.tiles_x * p.tiles_q;
    if (p.n_tiles >= 65536) return SISR_E_TOOBIG;
    p.IW = S == 1 ? p.TW + 2 : 2 * p.TW + 2;
    p.IWd = S == 1 ? p.IW : p.TW + 1;
    int span_max = 0;
    for (int tq = 0; tq < p.tiles_q; ++tq) {
        const int q0 = tq * p.TH, qe = std::min(q0 + p.TH, NQ);
        span_max = std::max(span_max, rbase(qe - 1) - rbase(q0));
    }
    p.IH_max = span_max + 3;
    p.IHd_max = span_max / S + 1;
    p.NPOS_max = (p.IHd_max * p.IWd + 15) & ~15;
    if (S == 1) p.XP_max = std::max(p.IH_max * p.IW, p.NPOS_max + 2 * p.IW + 2);
    else p.XP_max = 4 * std::max(((p.IH_max + 1) / 2) * p.IWd, p.NPOS_max + p.IWd + 1);
    p.NITX = (p.IH_max * p.IW * 8 + WD_PROD - 1) / WD_PROD;
    p.NITD = (p.NPOS_max * 8 + WD_PROD - 1) / WD_PROD;
    p.lds_bytes = 4 * (p.XP_max + p.NPOS_max) * WD_PSB + WD_CONST_BYTES;
    if (p.lds_bytes > 160 * 1024 || p.lds_bytes < WD_PROD * 8 * 4) return SISR_E_UNSUPPORTED;
    if (p.IH_max * p.IW >= 65536 || p.NPOS_max >= 65536) return SISR_E_TOOBIG;
    p.n_cib = d->Cin / 64; p.n_cob = d->Cout / 64;
    // pixel blocks (= slabs): each workgroup walks ceil(n_tiles / n_pb) tiles (measured ~3.5 us each with the chip full: the staging
    // loads of 256 workgroups run at ~4 TB/s; + ~8 us of prologue and slab stores) in ceil(workgroups / 256) rounds; every slab is
    // written once and re-read once by the reduction
    p.slab_bf16 = wd_slab_bf16() ? 1 : 0;
    {
        const int blocks = p.n_cib * p.n_cob;
        const double slab_us = (double)d->slab_elems * (p.slab_bf16 ? 2 : 4) * 2.0 / 3.5e6;     // write + re-read at ~3.5 TB/s
        double bt = -1.0;
        int bpb = 1;
        const int cap = target_wg > 0 ? std::max(1, target_wg / blocks) : p.n_tiles;
        for (int npb = 1; npb <= std::min(p.n_tiles, cap); ++npb) {
            const int tp = (p.n_tiles + npb - 1) / npb;
            const int real = (p.n_tiles + tp - 1) / tp;
            if (real != npb) continue;
            const int rounds = (blocks * npb + 255) / 256;
            const double t = rounds * (tp * 3.5 + 8.0) + npb * slab_us;
            if (bt < 0 || t < bt) { bt = t; bpb = npb; }
        }
        if (const char* e = getenv("SISR_WGRAD_DEEP_PB")) {
            const int v = std::max(1, std::min(p.n_tiles, atoi(e)));
            const int tp = (p.n_tiles + v - 1) / v;
            bpb = (p.n_tiles + tp - 1) / tp;
        }
        p.n_pb = bpb;
        p.tiles_per_pb = (p.n_tiles + bpb - 1) / bpb;
    }
    p.m_tiles_x = fdiv_magic(p.tiles_x); p.m_tw = fdiv_magic(p.TW); p.m_ho = fdiv_magic(d->Ho);
    p.m_pr = fdiv_magic(p.PR); p.m_iw = fdiv_magic(p.IW); p.m_iwd = fdiv_magic(p.IWd);
    p.enabled = 1;
    return 0;
}

// a fully filled descriptor (operands, modes, storage flags) will run here
extern "C" int sisr_wgrad_deep_eligible(const SisrWgradDesc* d) {
    if (!d || !d->deep.enabled) return 0;
    if (!d->x_bf16 || !d->g_bf16 || d->x_mode != SISR_X_NHWC || d->g_mode != SISR_X_NHWC) return 0;
    if (d->pro_mode != SISR_PRO_NONE && d->pro_mode != SISR_PRO_ACT && d->pro_mode != SISR_PRO_AFFINE_ACT) return 0;
    const int gp = d->gpro_mode;
    if (gp != SISR_PRO_NONE && gp != SISR_PRO_BNBWD && gp != SISR_PRO_BNACT_BWD && gp != SISR_PRO_ACT_BWD) return 0;
    return 1;
}

template <int S, int NITX, int NITD>
static int launch_wd(const SisrWgradDesc* d, hipStream_t st) {
    const SisrWgradDeepPlan& p = d->deep;
    const bool two = operand_needs_x2(d->gpro_mode);
    const dim3 grid(p.n_cib * p.n_cob, p.n_pb);
    if (two) {
        static SisrLdsCap cap;
        if (int e = sisr_raise_lds_cap(cap, reinterpret_cast<const void*>(&wgrad_deep_kernel<S, NITX, NITD, true>), p.lds_bytes, 0)) return e;
        hipLaunchKernelGGL((wgrad_deep_kernel<S, NITX, NITD, true>), grid, dim3(WD_THREADS), p.lds_bytes, st, *d);
    } else {
        static SisrLdsCap cap;
        if (int e = sisr_raise_lds_cap(cap, reinterpret_cast<const void*>(&wgrad_deep_kernel<S, NITX, NITD, false>), p.lds_bytes, 0)) return e;
        hipLaunchKernelGGL((wgrad_deep_kernel<S, NITX, NITD, false>), grid, dim3(WD_THREADS), p.lds_bytes, st, *d);
    }
    SISR_CHECK_LAUNCH();
    return 0;
}

int sisr_wgrad_deep_launch(const SisrWgradDesc* d, hipStream_t st) {
    const SisrWgradDeepPlan& p = d->deep;
    if (d->pro_mode == SISR_PRO_AFFINE_ACT && (!d->pa || !d->pd)) return SISR_E_BADARG;
    const int gp = d->gpro_mode;
    if ((gp == SISR_PRO_BNBWD || gp == SISR_PRO_BNACT_BWD) && (!d->qa || !d->qb || !d->qd)) return SISR_E_BADARG;
    if (gp == SISR_PRO_BNACT_BWD && (!d->qs || !d->qt)) return SISR_E_BADARG;
    if (p.n_pb <= 0 || p.tiles_per_pb <= 0 || (p.n_pb - 1) * p.tiles_per_pb >= p.n_tiles) return SISR_E_BADARG;
    if (d->stride == 1) {
        if (p.NITX > 6 || p.NITD > 4) return SISR_E_BADARG;
        return launch_wd<1, 6, 4>(d, st);
    }
    if (p.NITX > 10 || p.NITD > 3) return SISR_E_BADARG;
    return launch_wd<2, 10, 3>(d, st);
}

template <int S, int NITX, int NITD>
static int launch_wd_table(const SisrWgradDesc* table_dev, int n, dim3 grid, int lds_bytes, bool two, hipStream_t st) {
    if (two) {
        static SisrLdsCap cap;
        if (int e = sisr_raise_lds_cap(cap, reinterpret_cast<const void*>(&wgrad_deep_table_kernel<S, NITX, NITD, true>), lds_bytes, 0)) return e;
        hipLaunchKernelGGL((wgrad_deep_table_kernel<S, NITX, NITD, true>), grid, dim3(WD_THREADS), lds_bytes, st, table_dev, n);
    } else {
        static SisrLdsCap cap;
        if (int e = sisr_raise_lds_cap(cap, reinterpret_cast<const void*>(&wgrad_deep_table_kernel<S, NITX, NITD, false>), lds_bytes, 0)) return e;
        hipLaunchKernelGGL((wgrad_deep_table_kernel<S, NITX, NITD, false>), grid, dim3(WD_THREADS), lds_bytes, st, table_dev, n);
    }
    SISR_CHECK_LAUNCH();
    return 0;
}

// table_host: the n fully filled descriptors (all wgrad_deep-eligible, same stride, all with or all without a two-tensor gradient
// prologue); table_dev: the same bytes in device memory (the kernel reads its descriptor from there)
extern "C" int sisr_wgrad_deep_batch(const SisrWgradDesc* table_host, const SisrWgradDesc* table_dev, int32_t n, void* stream) {
    if (!table_host || !table_dev || n <= 0 || n > 4096) return SISR_E_BADARG;
    const int S = table_host[0].stride;
    const bool two = operand_needs_x2(table_host[0].gpro_mode);
    int total = 0, lds = 0;
    for (int i = 0; i < n; ++i) {
        const SisrWgradDesc* d = table_host + i;
        const SisrWgradDeepPlan& p = d->deep;
        if (!sisr_wgrad_deep_eligible(d) || d->stride != S || operand_needs_x2(d->gpro_mode) != two) return SISR_E_BADARG;
        if (!d->x1 || !d->g1 || !d->slab || (two && !d->g2) || d->slab_stride < d->slab_elems) return SISR_E_BADARG;
        if (d->pro_mode == SISR_PRO_AFFINE_ACT && (!d->pa || !d->pd)) return SISR_E_BADARG;
        const int gp = d->gpro_mode;
        if ((gp == SISR_PRO_BNBWD || gp == SISR_PRO_BNACT_BWD) && (!d->qa || !d->qb || !d->qd)) return SISR_E_BADARG;
        if (gp == SISR_PRO_BNACT_BWD && (!d->qs || !d->qt)) return SISR_E_BADARG;
        if (p.n_pb <= 0 || p.tiles_per_pb <= 0 || (p.n_pb - 1) * p.tiles_per_pb >= p.n_tiles) return SISR_E_BADARG;
        if (S == 1 ? (p.NITX > 6 || p.NITD > 4) : (p.NITX > 10 || p.NITD > 3)) return SISR_E_BADARG;
        if (p.batch_first_wg != total) return SISR_E_BADARG;       // the caller numbers the members' workgroups consecutively
        total += p.n_cib * p.n_cob * p.n_pb;
        lds = std::max(lds, p.lds_bytes);
    }
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const dim3 grid(total);
    return S == 1 ? launch_wd_table<1, 6, 4>(table_dev, n, grid, lds, two, st) : launch_wd_table<2, 10, 3>(table_dev, n, grid, lds, two, st);
}
