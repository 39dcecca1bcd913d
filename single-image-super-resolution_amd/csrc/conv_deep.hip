// conv_deep.hip -- split-K implicit-GEMM convolution on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16, fp32 accumulate)
// for the layers whose contraction is deep and whose images are small: the discriminator's strided-conv stack
// (model_discriminator.py:10,39-44), the VGG19 feature convs (model_content_extractor.py:43) and the generator's trunk at the
// sizes the persistent trunk kernels do not take.  Forward and data-gradient roles (a data gradient is a convolution over dy
// with the flipped / transposed weight image; the four output-parity classes of a stride-2 layer's data gradient are four
// descriptors with 1 / 2 / 2 / 4 taps and a strided output scatter).
//
// Why a new family: at B16 these layers are 3-11 GFLOP with 576 .. 36,864 output pixels -- the generic kernel
// (conv_bf16.hip) gave them 48-160 workgroups that each walked ALL 8-16 input-channel chunks with a stage -> barrier -> 36
// MFMAs -> barrier loop and nothing in flight across a barrier (18-66 us per launch, 0.03 of the bf16 MFMA peak).  Here:
//   * implicit GEMM  [128 pixels] x [BN = 64 | 128 couts] x [K = 32-channel chunks x taps] per workgroup, 2 x 2 waves of
//     64 x BN/2 (4 or 2 accumulators, every A / B fragment used twice: 1 .. 1.5 LDS reads per MFMA instead of 1.5 .. 3);
//   * K SPLIT over workgroups (SisrDeepPlan.split) so that a launch fills the chip: fp32 partial tiles in accumulator order
//     (1 KB per store instruction), summed in a fixed order by conv_deep_finish_kernel, which owns the epilogue;
//   * an output tile is TH rows of the FLATTENED (image, row) space x TW columns: a band of full-width rows may straddle
//     images, so 12 x 12 and 6 x 6 maps fill 120 / 126 of the 128 MFMA rows (the generic kernel: 72 / 108); the LDS halo image
//     lives in padded-row space (PR rows per image: top / bottom padding rows are real, zero-filled rows of the image);
//   * weights streamed LDS-DIRECT (buffer_load ... lds, no registers, no VALU) one (chunk, tap row) stage ahead of the MFMAs
//     from an image that sisr_weights_prepare writes in LDS order, row padding included: [chunk][tap row][cout][KW * 32 + 8];
//   * the input halo of the next chunk is fetched into registers at the first tap row of a chunk and transformed (lazy
//     operand: BatchNorm apply / activation / their backward forms) and written to the other LDS buffer at the last one;
//     a thread's global offsets are computed once per launch (they do not depend on the chunk).
// Epilogue (shared by the direct and the finishing path): 1 / sigma, bias, BatchNorm statistics (count, mean, M2 per tile),
// residual add and the fused BatchNorm-backward reductions of the data-gradient role, 16-byte NHWC bf16 stores through an
// LDS transpose (ds_read_b64_tr_b16).
#include "sisr_dev.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "sisr_bf16_stage.h"

#define DP_BM 128
#define DP_PSB 80                       // bytes per halo pixel in LDS: 32 bf16 + 8 pad (odd number of 16-byte slots)
#define DP_THREADS 256

__device__ __forceinline__ f32x16 dp_mfma(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ void dp_stat_merge(float& n, float& mu, float& m2, float nb, float mub, float m2b) {
    const float nt = n + nb;
    if (nt > 0.f) {
        const float dl = mub - mu, f = nb / nt;
        mu += dl * f;
        m2 += m2b + dl * dl * n * f;
    }
    n = nt;
}

// 8 channels of one pixel through the operand's prologue (see SISR_PRO_* in sisr_hip.h); bf16 in, bf16 out
template <int PRO>
__device__ __forceinline__ u32x4 deep_apply8(u32x4 a, u32x4 b, const f32x8& ka, const f32x8& kb, const f32x8& kd,
                                             const f32x8& ks, const f32x8& kt, float slope, bool ok) {
    if (PRO == SISR_PRO_NONE) return a;                       // zeros outside the image already (hardware range check)
    u32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float a0 = __uint_as_float(a[j] << 16), a1 = __uint_as_float(a[j] & 0xFFFF0000u);
        float b0 = 0.f, b1 = 0.f;
        if (PRO == SISR_PRO_BNBWD || PRO == SISR_PRO_BNACT_BWD || PRO == SISR_PRO_ACT_BWD) {
            b0 = __uint_as_float(b[j] << 16); b1 = __uint_as_float(b[j] & 0xFFFF0000u);
        }
        float r0, r1;
        if (PRO == SISR_PRO_ACT) { r0 = lrelu(a0, slope); r1 = lrelu(a1, slope); }
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
        // prologues with f(0) != 0: the halo must be zero AFTER the transform
        const unsigned pk = pack_bf16x2(r0, r1);
        o[j] = (PRO == SISR_PRO_ACT || PRO == SISR_PRO_ACT_BWD || ok) ? pk : 0u;
    }
    return o;
}

// the staged halo items of a thread: loads of one chunk (always executed -- items beyond the tile carry the out-of-range
// offset and come back as zeros -- so that the wait counts stay exact), then transform + 16-byte LDS writes
template <int NITM, bool TWO>
__device__ __forceinline__ void deep_issue_in(const __amdgpu_buffer_rsrc_t r1, const __amdgpu_buffer_rsrc_t r2, const unsigned (&goff)[NITM],
                                              int chunk, u32x4 (&sa)[NITM], u32x4 (&sb)[TWO ? NITM : 1]) {
    const unsigned so = (unsigned)chunk * 64u;
#pragma unroll
    for (int u = 0; u < NITM; ++u) {
        sa[u] = __builtin_amdgcn_raw_buffer_load_b128(r1, goff[u], so, 0);
        if constexpr (TWO) sb[u] = __builtin_amdgcn_raw_buffer_load_b128(r2, goff[u], so, 0);
    }
}
template <int PRO, int NITM, bool TWO>
__device__ __forceinline__ void deep_commit_t(const SisrConvDesc& d, int chunk, unsigned char* buf, unsigned loff0, unsigned okm, int npix,
                                              float slope, const u32x4 (&sa)[NITM], const u32x4 (&sb)[TWO ? NITM : 1]) {
    const int tid = threadIdx.x;
    const f32x8 zero8 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    f32x8 ka = zero8, kb = zero8, kd = zero8, ks = zero8, kt = zero8;
    const int c0 = chunk * 32 + (tid & 3) * 8;
    if (PRO == SISR_PRO_AFFINE_ACT || PRO == SISR_PRO_BNBWD || PRO == SISR_PRO_BNACT_BWD) {
        ka = *reinterpret_cast<const f32x8*>(d.pa + c0);
        kd = *reinterpret_cast<const f32x8*>(d.pd + c0);
    }
    if (PRO == SISR_PRO_BNBWD || PRO == SISR_PRO_BNACT_BWD) kb = *reinterpret_cast<const f32x8*>(d.pb + c0);
    if (PRO == SISR_PRO_BNACT_BWD) {
        ks = *reinterpret_cast<const f32x8*>(d.ps + c0);
        kt = *reinterpret_cast<const f32x8*>(d.pt + c0);
    }
#pragma unroll
    for (int u = 0; u < NITM; ++u) {
        const bool ok = (okm >> u) & 1u;
        const u32x4 v = deep_apply8<PRO>(sa[u], sb[TWO ? u : 0], ka, kb, kd, ks, kt, slope, ok);
        if ((tid >> 2) + u * (DP_THREADS / 4) < npix) *reinterpret_cast<u32x4*>(buf + loff0 + u * (DP_THREADS / 4 * DP_PSB)) = v;
    }
}
template <int NITM, bool TWO>
__device__ __forceinline__ void deep_commit(const SisrConvDesc& d, int pro, int chunk, unsigned char* buf, unsigned loff0, unsigned okm,
                                            int npix, float slope, const u32x4 (&sa)[NITM], const u32x4 (&sb)[TWO ? NITM : 1]) {
    if constexpr (TWO) {
        if (pro == SISR_PRO_BNACT_BWD) deep_commit_t<SISR_PRO_BNACT_BWD, NITM, TWO>(d, chunk, buf, loff0, okm, npix, slope, sa, sb);
        else if (pro == SISR_PRO_BNBWD) deep_commit_t<SISR_PRO_BNBWD, NITM, TWO>(d, chunk, buf, loff0, okm, npix, slope, sa, sb);
        else deep_commit_t<SISR_PRO_ACT_BWD, NITM, TWO>(d, chunk, buf, loff0, okm, npix, slope, sa, sb);
    } else {
        if (pro == SISR_PRO_AFFINE_ACT) deep_commit_t<SISR_PRO_AFFINE_ACT, NITM, TWO>(d, chunk, buf, loff0, okm, npix, slope, sa, sb);
        else if (pro == SISR_PRO_ACT) deep_commit_t<SISR_PRO_ACT, NITM, TWO>(d, chunk, buf, loff0, okm, npix, slope, sa, sb);
        else deep_commit_t<SISR_PRO_NONE, NITM, TWO>(d, chunk, buf, loff0, okm, npix, slope, sa, sb);
    }
}

// ---- the tile -> memory maps shared by the main and the finishing kernel -------------------------------------------------
struct DeepTile {
    int q0, ox0;                    // first flattened output row, first output column
    int NQ;                         // N * Ho
};
__device__ __forceinline__ DeepTile deep_tile(const SisrConvDesc& d, int mt) {
    const SisrDeepPlan& p = d.deep;
    DeepTile t;
    const int tq = fdiv(mt, p.m_tiles_x), tx = mt - tq * p.tiles_x;
    t.q0 = tq * p.TH; t.ox0 = tx * p.TW; t.NQ = d.N * d.Ho;
    return t;
}

// output-parity classes (SisrConvDesc.wdeep_c): class c of a 4-class launch writes pixel (2 a + (c >> 1), 2 b + (c & 1))
struct DeepClass {
    int y_oy, y_ox;
    int row_base;                   // first partial row / workspace tile of this class: c * (pixel tiles)
};
__device__ __forceinline__ DeepClass deep_class(const SisrConvDesc& d, int cls) {
    DeepClass c;
    const bool multi = d.deep.classes > 1;
    c.y_oy = multi ? (cls >> 1) : d.y_oy;
    c.y_ox = multi ? (cls & 1) : d.y_ox;
    c.row_base = cls * d.deep.tiles_x * d.deep.tiles_q;
    return c;
}

// LDS of the epilogue: [row_off 128 ints][red 12 * BN floats][img_r][img_x (each 128 x (BN + 8) bf16, only with res / bnb)]
// [img_y BN x 132 bf16]
__host__ __device__ static inline int deep_epi_lds(int BN, bool images) {
    return 512 + 48 * BN + (images ? 2 * DP_BM * (BN + 8) * 2 : 0) + BN * (DP_BM + 4) * 2;
}

// Epilogue of one 128 x BN tile held in accumulator layout: wave (wm, wn) = (wave >> 1, wave & 1) owns rows 64 wm .. and
// columns 32 NSUB wn ..; acc[ms][ns][i]: row 64 wm + 32 ms + mfma_row(i, lane), column 32 NSUB wn + 32 ns + (lane & 31).
// Every LDS buffer of the caller is free (a barrier has been passed).
template <int NSUB>
__device__ __forceinline__ void deep_epilogue(const SisrConvDesc& d, f32x16 (&acc)[2][NSUB], unsigned char* lds, int mt, int nt, int cls) {
    const SisrDeepPlan& p = d.deep;
    constexpr int BN = NSUB * 64, WN = NSUB * 32;
    constexpr int RS = BN + 8, YS = DP_BM + 4;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, kk = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const DeepTile t = deep_tile(d, mt);
    const DeepClass dc = deep_class(d, cls);
    const bool has_r = d.res != nullptr, has_x = d.bnb_part != nullptr;
    int* row_off = reinterpret_cast<int*>(lds);
    float* red = reinterpret_cast<float*>(lds + 512);
    __bf16* img_r = reinterpret_cast<__bf16*>(lds + 512 + 48 * BN);
    __bf16* img_x = img_r + DP_BM * RS;
    __bf16* img_y = (has_r || has_x) ? img_x + DP_BM * RS : img_r;
    const int cout_base = nt * BN;

    if (tid < DP_BM) {
        const int r = fdiv(tid, p.m_tw), c = tid - r * p.TW;
        const int q = t.q0 + r, ox = t.ox0 + c;
        int off = -1;
        if (r < p.TH && q < t.NQ && ox < d.Wo) {
            const int n = fdiv(q, p.m_ho), oy = q - n * d.Ho;
            const int py = oy * d.y_sy + dc.y_oy, px = ox * d.y_sx + dc.y_ox;
            off = ((n * d.y_H + py) * d.y_W + px) * d.Cout;
        }
        row_off[tid] = off;
    }
    __syncthreads();

    const float scale = d.epi_scale_p ? d.epi_scale_p[0] : 1.f;
#pragma unroll
    for (int ns = 0; ns < NSUB; ++ns) {
        const int cp = cout_base + wn * WN + ns * 32 + l31;
        const float bv = d.bias != nullptr ? d.bias[cp] : 0.f;
#pragma unroll
        for (int ms = 0; ms < 2; ++ms)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[ms][ns][i] = acc[ms][ns][i] * scale + bv;
    }
    bool rv[2][16];
#pragma unroll
    for (int ms = 0; ms < 2; ++ms)
#pragma unroll
        for (int i = 0; i < 16; ++i) rv[ms][i] = row_off[wm * 64 + ms * 32 + mfma_row(i, lane)] >= 0;

    if (d.stat_part != nullptr) {
        // BatchNorm statistics of the tile: every lane reduces its rows to (count, mean, M2), lane halves merge by a
        // shuffle, the two waves of a column half through LDS in a fixed order (Chan et al.; deterministic)
        float n = 0.f;
#pragma unroll
        for (int ms = 0; ms < 2; ++ms)
#pragma unroll
            for (int i = 0; i < 16; ++i) n += rv[ms][i] ? 1.f : 0.f;
        const float inv = n > 0.f ? 1.f / n : 0.f;
        const float n_o = __shfl_xor(n, 32);
#pragma unroll
        for (int ns = 0; ns < NSUB; ++ns) {
            float sm = 0.f;
#pragma unroll
            for (int ms = 0; ms < 2; ++ms)
#pragma unroll
                for (int i = 0; i < 16; ++i) sm += rv[ms][i] ? acc[ms][ns][i] : 0.f;
            float mu = sm * inv, m2 = 0.f;
#pragma unroll
            for (int ms = 0; ms < 2; ++ms)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float dv = acc[ms][ns][i] - mu;
                    m2 += rv[ms][i] ? dv * dv : 0.f;
                }
            float nn = n;
            dp_stat_merge(nn, mu, m2, n_o, __shfl_xor(mu, 32), __shfl_xor(m2, 32));
            if (kk == 0) {
                float* r = red + (wm * BN + wn * WN + ns * 32 + l31) * 3;
                r[0] = nn; r[1] = mu; r[2] = m2;
            }
        }
        __syncthreads();
        if (tid < BN) {
            float nn = red[tid * 3], mu = red[tid * 3 + 1], m2 = red[tid * 3 + 2];
            const float* r = red + (BN + tid) * 3;
            dp_stat_merge(nn, mu, m2, r[0], r[1], r[2]);
            float* sp = d.stat_part + (int64_t)(dc.row_base + mt) * 2 * d.Cout + cout_base + tid;
            sp[0] = mu;
            sp[d.Cout] = m2;
            if (tid == 0 && nt == 0) d.cnt_part[dc.row_base + mt] = nn;
        }
    }

    const unsigned ypix = (unsigned)max(d.N * d.y_H * d.y_W, d.N * d.Ho * d.Wo);
    const unsigned ybytes2 = ypix * (unsigned)d.Cout * 2u;
    const int grp = lane >> 4, tq = (lane & 15) >> 2, tp = lane & 3;
    s16x4 pr[2][NSUB][4], px[2][NSUB][4];
    if (has_r || has_x) {
        // residual / BatchNorm input of the tile: global (16 bytes = 8 channels of a pixel) -> [pixel][RS] images ->
        // transposing reads into accumulator layout
        const __amdgpu_buffer_rsrc_t rr = bf_rsrc(has_r ? d.res : d.bnb_x, ybytes2);
        const __amdgpu_buffer_rsrc_t rx = bf_rsrc(has_x ? d.bnb_x : d.res, ybytes2);
        constexpr int OCT = BN / 8;
        constexpr int ITEMS = DP_BM * OCT / DP_THREADS;           // 16-byte items per thread and tensor
        u32x4 vr[ITEMS], vx[ITEMS];
#pragma unroll
        for (int k = 0; k < ITEMS; ++k) {
            const int idx = tid + k * DP_THREADS, m = idx / OCT, oc = idx - m * OCT;
            const int ro = row_off[m], ch = cout_base + oc * 8;
            const unsigned vo = ro >= 0 ? (unsigned)(ro + ch) * 2u : 0x80000000u;
            if (has_r) vr[k] = __builtin_amdgcn_raw_buffer_load_b128(rr, vo, 0, 0);
            if (has_x) vx[k] = __builtin_amdgcn_raw_buffer_load_b128(rx, vo, 0, 0);
        }
#pragma unroll
        for (int k = 0; k < ITEMS; ++k) {
            const int idx = tid + k * DP_THREADS, m = idx / OCT, oc = idx - m * OCT;
            if (has_r) *reinterpret_cast<u32x4*>(img_r + m * RS + oc * 8) = vr[k];
            if (has_x) *reinterpret_cast<u32x4*>(img_x + m * RS + oc * 8) = vx[k];
        }
        __syncthreads();
#pragma unroll
        for (int ms = 0; ms < 2; ++ms)
#pragma unroll
            for (int ns = 0; ns < NSUB; ++ns)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    // this 16-lane group's block: pixels 8g + 4 (grp >> 1) .. + 3, channels 16 (grp & 1) .. + 15 of the sub-tile
                    const int off = (wm * 64 + ms * 32 + 8 * g + 4 * (grp >> 1) + tq) * RS + wn * WN + ns * 32 + 16 * (grp & 1) + 4 * tp;
                    if (has_r) pr[ms][ns][g] = lds_tr16(img_r + off);
                    if (has_x) px[ms][ns][g] = lds_tr16(img_x + off);
                }
    }
    if (has_r) {
#pragma unroll
        for (int ms = 0; ms < 2; ++ms)
#pragma unroll
            for (int ns = 0; ns < NSUB; ++ns)
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    acc[ms][ns][i] += bf16_bits_to_f32((unsigned short)pr[ms][ns][i >> 2][i & 3]);
    }
    if (has_x) {
        // BatchNorm-backward reductions of the gradient tile just formed (SisrConvDesc.bnb_*): one row per (tile, cout tile);
        // the columns of the other cout tiles are written as zeros so that the finishing kernel can sum rows blindly
        const float bslope = d.bnb_slope_p ? d.bnb_slope_p[0] : d.bnb_slope;
        float ssl = 0.f;
        __syncthreads();                                   // `red` may still hold the forward statistics
#pragma unroll
        for (int ns = 0; ns < NSUB; ++ns) {
            const int cp = cout_base + wn * WN + ns * 32 + l31;
            const float sc = d.bnb_scale[cp], sf = d.bnb_shift[cp], mu = d.bnb_mean[cp], is = d.bnb_invstd[cp];
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int ms = 0; ms < 2; ++ms)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float xv = bf16_bits_to_f32((unsigned short)px[ms][ns][i >> 2][i & 3]);
                    float g = rv[ms][i] ? (float)acc[ms][ns][i] : 0.f;
                    if (d.bnb_act) {
                        const float z = sc * xv + sf;
                        if (!(z > 0.f)) { ssl += g * z; g *= bslope; }
                    }
                    s1 += g;
                    s2 += g * ((xv - mu) * is);
                }
            s1 += __shfl_xor(s1, 32);
            s2 += __shfl_xor(s2, 32);
            if (kk == 0) { red[(wm * BN + wn * WN + ns * 32 + l31) * 2] = s1; red[(wm * BN + wn * WN + ns * 32 + l31) * 2 + 1] = s2; }
        }
        ssl = wave_sum(ssl);
        if (lane == 0) red[4 * BN + wave] = ssl;
        __syncthreads();
        float* wk = d.bnb_part + (int64_t)((dc.row_base + mt) * p.n_ntiles + nt) * (2 * d.Cout + 1);
        for (int c = tid; c < d.Cout; c += DP_THREADS) {
            const int cl = c - cout_base;
            float s1 = 0.f, s2 = 0.f;
            if (cl >= 0 && cl < BN) { s1 = red[cl * 2] + red[(BN + cl) * 2]; s2 = red[cl * 2 + 1] + red[(BN + cl) * 2 + 1]; }
            wk[c] = s1;
            wk[d.Cout + c] = s2;
        }
        if (tid == 0) wk[2 * d.Cout] = red[4 * BN] + red[4 * BN + 1] + red[4 * BN + 2] + red[4 * BN + 3];
    }
    // output: accumulators -> bf16 -> [channel][YS] image (a wave writes and re-reads only its own 64 pixels x 32 NSUB
    // channels: no barrier), then 16-byte stores of 8 consecutive channels of one pixel
#pragma unroll
    for (int ms = 0; ms < 2; ++ms)
#pragma unroll
        for (int ns = 0; ns < NSUB; ++ns)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                bf16x4 h;
#pragma unroll
                for (int j = 0; j < 4; ++j) { const float v = acc[ms][ns][4 * g + j]; h[j] = (__bf16)v; }
                *reinterpret_cast<bf16x4*>(img_y + (wn * WN + ns * 32 + l31) * YS + wm * 64 + ms * 32 + 8 * g + 4 * kk) = h;
            }
    const __amdgpu_buffer_rsrc_t ry = bf_rsrc(d.y, ybytes2);
    constexpr int OCTW = NSUB * 4;                                  // channel octets of a wave
    // a wave owns 4 pixel blocks x OCTW octets; one pass serves 4 of those pairs (one per 16-lane group): OCTW passes
#pragma unroll
    for (int ps = 0; ps < OCTW; ++ps) {
        const int pb = 2 * (ps / (OCTW / 2)) + (grp & 1);          // 16-pixel block of this wave's 64 rows
        const int oc = 2 * (ps % (OCTW / 2)) + (grp >> 1);          // channel octet of this wave's columns
        const int m0 = wm * 64 + 16 * pb;
        const __bf16* src = img_y + (wn * WN + oc * 8 + tq) * YS + m0 + 4 * tp;
        const s16x4 lo = lds_tr16(src), hi = lds_tr16(src + 4 * YS);
        const int ro = row_off[m0 + (lane & 15)];
        const int cp = cout_base + wn * WN + oc * 8;
        const unsigned vo = ro >= 0 ? (unsigned)(ro + cp) * 2u : 0x80000000u;
        const s16x8 v8 = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v8), ry, vo, 0, 0);
    }
}

// partial tiles of the K split: [tile = mt * n_ntiles + nt][slice][wave][ms][ns][g 4][lane 64][4] fp32
template <int NSUB>
__device__ __forceinline__ int64_t deep_ws_index(const SisrDeepPlan& p, int cls, int mt, int nt, int slice) {
    return ((int64_t)((cls * p.tiles_x * p.tiles_q + mt) * p.n_ntiles + nt) * p.split + slice) * (DP_BM * NSUB * 64);
}

// ---- the main kernel -------------------------------------------------------------------------------------------------------
// NSUB: 32-cout accumulator columns per wave (BN = 64 NSUB).  KW: taps per tap row (compile time: the MFMA loop of a stage is
// unrolled).  NITM: staging items per thread provided for (4: stride 1, 10: stride 2).  TWO: two-tensor prologue.
template <int NSUB, int KW, int NITM, bool TWO>
__global__ void __launch_bounds__(DP_THREADS, NSUB == 2 ? 1 : 2) conv_deep_kernel(const SisrConvDesc d) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const SisrDeepPlan& p = d.deep;
    constexpr int BN = NSUB * 64, WN = NSUB * 32;
    constexpr int WRB = KW * 64 + 16;                              // bytes of a weight row: KW * 32 bf16 + 8 pad
    constexpr int WSTAGE = BN * WRB;                               // bytes of one (chunk, tap row) stage
    constexpr int WPIECES = WSTAGE / 1024;                         // 1 KB pieces (one LDS-direct wave instruction each)
    static_assert(WSTAGE % 1024 == 0, "stage = whole 1 KB pieces");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, kk = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int nt = blockIdx.x, mt = blockIdx.y;
    const int cls = p.classes > 1 ? (int)blockIdx.z / p.split : 0, slice = (int)blockIdx.z - cls * p.split;
    const int S = d.stride, KH = p.classes > 1 ? d.deep_ckh[cls] : d.KH;
    const DeepTile t = deep_tile(d, mt);
    const int IW = p.IW;
    auto rbase = [&](int q) { const int n = fdiv(q, p.m_ho); return n * p.PR + (q - n * d.Ho) * S; };
    const int qlast = min(t.q0 + p.TH, t.NQ) - 1;
    const int pb0 = rbase(t.q0);
    const int IH = rbase(qlast) - pb0 + KH;
    const int npix = IH * IW;
    const int in_bytes = (p.IH_max * IW * DP_PSB + 15) & ~15;
    unsigned char* in_buf = lds;                                   // [2][in_bytes]
    unsigned char* w_buf = lds + 2 * in_bytes;                     // [3][WSTAGE]: ring of weight stages

    // ---- staging map of this thread: item e = tid + 256 u = (halo pixel e >> 2, channel octet e & 3) -------------------------
    unsigned goff[NITM];
    unsigned okm = 0;
    {
        const int ix0 = t.ox0 * S - d.pad_x;
#pragma unroll
        for (int u = 0; u < NITM; ++u) {
            const int e = tid + u * DP_THREADS;
            const int pix = e >> 2, oct = e & 3;
            const int hr = fdiv(pix, p.m_iw), hc = pix - hr * IW;
            const int prow = pb0 + hr;
            const int n = fdiv(prow, p.m_pr), iy = prow - n * p.PR - d.pad_y;
            const int ix = ix0 + hc;
            const bool ok = pix < npix && n < d.N && (unsigned)iy < (unsigned)d.H && (unsigned)ix < (unsigned)d.W;
            goff[u] = ok ? (unsigned)(((n * d.H + iy) * d.W + ix) * d.Cin * 2 + oct * 16) : 0x80000000u;
            okm |= ok ? (1u << u) : 0u;
        }
    }
    const unsigned loff0 = (unsigned)((tid >> 2) * DP_PSB + (tid & 3) * 16);
    const unsigned xbytes = (unsigned)d.N * (unsigned)d.H * (unsigned)d.W * (unsigned)d.Cin * 2u;
    const __amdgpu_buffer_rsrc_t r1 = bf_rsrc(d.x1, xbytes), r2 = bf_rsrc(TWO ? d.x2 : d.x1, xbytes);
    const __amdgpu_buffer_rsrc_t rw = bf_rsrc(p.classes > 1 ? d.wdeep_c[cls] : d.wdeep, (unsigned)(p.n_chunk * KH * d.Cout * (KW * 32 + 8)) * 2u);
    const float slope = d.pro_slope_p ? d.pro_slope_p[0] : d.pro_slope;
    const int pro = d.pro_mode;

    u32x4 sa[NITM], sb[TWO ? NITM : 1];
    auto issue_in = [&](int chunk) { deep_issue_in<NITM, TWO>(r1, r2, goff, chunk, sa, sb); };
    auto commit = [&](int chunk, unsigned char* buf) { deep_commit<NITM, TWO>(d, pro, chunk, buf, loff0, okm, npix, slope, sa, sb); };
    // one weight stage, LDS-direct: the stage is a contiguous WSTAGE-byte block of the image; wave w lays down pieces w, w + 4, ..
    // -- PW pieces per wave, no branch around a load (a wave whose last piece index runs past the stage lays the LAST piece down
    // again: the same bytes to the same place as the wave that owns it), so every wave's vmcnt arithmetic is the same
    constexpr int PW = (WPIECES + 3) / 4;
    const unsigned wlane = (unsigned)lane * 16u;
    auto stage_so = [&](int chunk, int ky) { return (unsigned)(((chunk * KH + ky) * d.Cout + nt * BN) * WRB); };
    auto issue_piece = [&](int k, unsigned so, unsigned char* buf) {
        const int piece = min(wave + 4 * k, WPIECES - 1);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void*)(buf + piece * 1024), 16,
                                                 wlane + (unsigned)(piece * 1024), so, 0, 0);
    };
    auto issue_w = [&](unsigned so, unsigned char* buf) {
#pragma unroll
        for (int k = 0; k < PW; ++k) issue_piece(k, so, buf);
    };

    // ---- fragment addresses ---------------------------------------------------------------------------------------------------
    int a_base[2];
#pragma unroll
    for (int ms = 0; ms < 2; ++ms) {
        const int m = wm * 64 + ms * 32 + l31;
        const int r = fdiv(m, p.m_tw), c = m - r * p.TW;
        int base = 0;
        if (r < p.TH && t.q0 + r < t.NQ) base = ((rbase(t.q0 + r) - pb0) * IW + c * S) * DP_PSB;
        a_base[ms] = base + kk * 16;
    }
    int b_base[NSUB];
#pragma unroll
    for (int ns = 0; ns < NSUB; ++ns) b_base[ns] = (wn * WN + ns * 32 + l31) * WRB + kk * 16;

    f32x16 acc[2][NSUB];
#pragma unroll
    for (int ms = 0; ms < 2; ++ms)
#pragma unroll
        for (int ns = 0; ns < NSUB; ++ns)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[ms][ns][i] = 0.f;

    // the MFMAs of one stage: KW taps x two K = 16 steps; the fragments of step s + 1 are requested before the MFMAs of step s.
    // DMA: the LDS-direct pieces of the stage after next are issued BETWEEN the steps (an LDS-direct load costs the wave ~60-100
    // issue cycles: in front of the MFMAs, as a block, those cycles were half of a stage; behind an MFMA group they are hidden)
    auto mma_stage = [&](const unsigned char* ain, const unsigned char* wb, auto dma_c, unsigned so_next, unsigned char* buf_next) {
        constexpr bool DMA = decltype(dma_c)::value;
        constexpr int STEPS = 2 * KW;
        bf16x8 fa[2][2], fb[2][NSUB];
        auto fetch = [&](int step, int buf) {
            const int kx = step >> 1, k2 = step & 1;
#pragma unroll
            for (int ms = 0; ms < 2; ++ms) fa[buf][ms] = *reinterpret_cast<const bf16x8*>(ain + a_base[ms] + kx * DP_PSB + k2 * 32);
#pragma unroll
            for (int ns = 0; ns < NSUB; ++ns) fb[buf][ns] = *reinterpret_cast<const bf16x8*>(wb + b_base[ns] + kx * 64 + k2 * 32);
        };
        fetch(0, 0);
#pragma unroll
        for (int step = 0; step < STEPS; ++step) {
            // pinned order: the NEXT step's fragment reads (+ this step's share of the DMA pieces), then THIS step's MFMAs
            // (left to itself the scheduler sinks every read to just before its use: LDS latency once per pair of MFMAs)
            if (step + 1 < STEPS) fetch(step + 1, (step + 1) & 1);
            if constexpr (DMA) {
#pragma unroll
                for (int k = 0; k < PW; ++k)
                    if (k * STEPS / PW == step) issue_piece(k, so_next, buf_next);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ms = 0; ms < 2; ++ms)
#pragma unroll
                for (int ns = 0; ns < NSUB; ++ns) acc[ms][ns] = dp_mfma(fa[step & 1][ms], fb[step & 1][ns], acc[ms][ns]);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // wait until all but the wave's `N` youngest vector-memory operations are done, and every LDS operation; then the workgroup
    // barrier.  NOT __syncthreads(): its fence would wait for vmcnt(0) and drain the LDS-direct loads that are meant to stay in flight
#define DP_WAIT_BARRIER(N)                                                                                                        \
    do {                                                                                                                           \
        asm volatile("" ::: "memory");                                                                                             \
        __builtin_amdgcn_s_waitcnt((((N) & 15) | 0x70 | (((N) >> 4) << 14)));                                                      \
        __builtin_amdgcn_s_barrier();                                                                                              \
        asm volatile("" ::: "memory");                                                                                             \
    } while (0)

    // ---- K loop over this slice's chunks: stage s = (chunk, tap row); weight ring of 3 stages: the pieces of stage s + 2 are issued
    // during the MFMAs of stage s and have all of stage s + 1 to land; the halo of chunk c + 1 is requested at the first tap row of
    // chunk c and committed at its last one.  A wave's vector-memory operations retire in issue order, so "stage s + 1 has landed"
    // is "all but the operations issued after its last piece are done": the pieces of stage s + 2 and this stage's halo loads.
    const int c_begin = slice * p.cps, c_end = min(p.n_chunk, c_begin + p.cps);
    const int nst = (c_end - c_begin) * KH;
    constexpr int NIN = NITM * (TWO ? 2 : 1);
    auto so_of = [&](int st) { const int c = st / KH; return stage_so(c_begin + c, st - c * KH); };
    issue_in(c_begin);
    issue_w(so_of(0), w_buf);
    if (nst > 1) issue_w(so_of(1), w_buf + WSTAGE);
    commit(c_begin, in_buf);
    if (nst > 1) DP_WAIT_BARRIER(PW); else DP_WAIT_BARRIER(0);      // stage 0 landed (stage 1 may still be in flight)
    int sidx = 0;
    for (int chunk = c_begin; chunk < c_end; ++chunk) {
        const int ib = (chunk - c_begin) & 1;
        const bool more = chunk + 1 < c_end;
        for (int ky = 0; ky < KH; ++ky, ++sidx) {
            const bool last_row = ky == KH - 1;
            const bool in_now = ky == 0 && more, dma_now = sidx + 2 < nst;
            if (in_now) issue_in(chunk + 1);
            const unsigned char* ain = in_buf + ib * in_bytes + ky * IW * DP_PSB;
            const unsigned char* wb = w_buf + (sidx % 3) * WSTAGE;
            if (dma_now) mma_stage(ain, wb, std::true_type{}, so_of(sidx + 2), w_buf + ((sidx + 2) % 3) * WSTAGE);
            else mma_stage(ain, wb, std::false_type{}, 0u, w_buf);
            if (last_row && more) commit(chunk + 1, in_buf + (ib ^ 1) * in_bytes);
            if (sidx + 1 >= nst) DP_WAIT_BARRIER(0);                // last stage: nothing may be in flight when the epilogue reuses LDS
            else if (dma_now && in_now) DP_WAIT_BARRIER(PW + NIN);
            else if (dma_now) DP_WAIT_BARRIER(PW);
            else if (in_now) DP_WAIT_BARRIER(NIN);
            else DP_WAIT_BARRIER(0);
        }
    }
#undef DP_WAIT_BARRIER

    if (p.split > 1) {
        float* ws = d.deep_ws + deep_ws_index<NSUB>(p, cls, mt, nt, slice);
#pragma unroll
        for (int ms = 0; ms < 2; ++ms)
#pragma unroll
            for (int ns = 0; ns < NSUB; ++ns)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 v;
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = acc[ms][ns][4 * g + j];
                    *reinterpret_cast<f32x4*>(ws + ((((wave * 2 + ms) * NSUB + ns) * 4 + g) * 64 + lane) * 4) = v;
                }
        return;
    }
    deep_epilogue<NSUB>(d, acc, lds, mt, nt, cls);
}

// sums the K slices of one tile in slice order and runs the epilogue
template <int NSUB>
__global__ void __launch_bounds__(DP_THREADS, NSUB == 2 ? 1 : 2) conv_deep_finish_kernel(const SisrConvDesc d) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const SisrDeepPlan& p = d.deep;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nt = blockIdx.x, mt = blockIdx.y, cls = blockIdx.z;
    f32x16 acc[2][NSUB];
#pragma unroll
    for (int ms = 0; ms < 2; ++ms)
#pragma unroll
        for (int ns = 0; ns < NSUB; ++ns)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[ms][ns][i] = 0.f;
    const float* ws = d.deep_ws + deep_ws_index<NSUB>(p, cls, mt, nt, 0);
    constexpr int SLICE = DP_BM * NSUB * 64;
    // slices in groups of four: every load of a group is in flight before the first add (the sum keeps slice order)
    for (int z0 = 0; z0 < p.split; z0 += 4) {
#pragma unroll
        for (int ms = 0; ms < 2; ++ms)
#pragma unroll
            for (int ns = 0; ns < NSUB; ++ns)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int off = ((((wave * 2 + ms) * NSUB + ns) * 4 + g) * 64 + lane) * 4;
                    f32x4 v[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        // (slices past the last repeat it with weight 0: no branch around a load)
                        const int z = min(z0 + u, p.split - 1);
                        v[u] = *reinterpret_cast<const f32x4*>(ws + (int64_t)z * SLICE + off);
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const float wgt = z0 + u < p.split ? 1.f : 0.f;
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[ms][ns][4 * g + j] += wgt * v[u][j];
                    }
                }
    }
    deep_epilogue<NSUB>(d, acc, lds, mt, nt, cls);
}

// ---- host ---------------------------------------------------------------------------------------------------------------------
static inline int dp_round_up(int v, int m) { return (v + m - 1) / m * m; }

extern "C" int sisr_conv2d_deep_plan(SisrConvDesc* d, int32_t target_wg, int32_t prefer_bn, int32_t classes) {
    if (!d) return SISR_E_BADARG;
    SisrDeepPlan& p = d->deep;
    std::memset(&p, 0, sizeof(p));
    if (const char* e = getenv("SISR_DEEP")) if (e[0] == '0') return SISR_E_UNSUPPORTED;        // A/B switch: keep the generic kernel
    if (d->N <= 0 || d->Cin <= 0 || d->Cout <= 0 || d->KH <= 0 || d->KW <= 0) return SISR_E_BADARG;
    if ((d->Cin % 32) || (d->Cout % 64) || d->KH > 3 || d->KW > 3) return SISR_E_UNSUPPORTED;
    if (d->stride != 1 && d->stride != 2) return SISR_E_UNSUPPORTED;
    if (d->x_mode != SISR_X_NHWC) return SISR_E_UNSUPPORTED;
    if (d->pad_y < 0 || d->pad_x < 0) return SISR_E_UNSUPPORTED;
    const int S = d->stride;
    const int64_t ypix = std::max((int64_t)d->N * d->y_H * d->y_W, (int64_t)d->N * d->Ho * d->Wo);
    // 32-bit byte offsets with 2^31 as the out-of-range marker: bf16 tensors below 2 GB
    if (ypix * d->Cout * 2 >= (1ll << 31) || (int64_t)d->N * d->H * d->W * d->Cin * 2 >= (1ll << 31)) return SISR_E_TOOBIG;
    if ((int64_t)d->N * d->Ho >= 65536) return SISR_E_TOOBIG;
    if (classes != 1 && classes != 4) return SISR_E_BADARG;
    if (classes == 4 && (d->KH != 2 || d->KW != 2 || d->stride != 1 || d->pad_y || d->pad_x || d->y_sy != 2 || d->y_sx != 2 ||
                         d->y_H != 2 * d->Ho || d->y_W != 2 * d->Wo || d->H != d->Ho || d->W != d->Wo))
        return SISR_E_UNSUPPORTED;
    p.classes = classes;
    p.n_chunk = d->Cin / 32;
    const int NQ = d->N * d->Ho;
    if (d->Ho % 8 == 0 && d->Wo % 16 == 0) {
        p.TH = 8; p.TW = 16; p.tiles_x = d->Wo / 16; p.tiles_q = NQ / 8;
    } else if (d->Wo <= 42) {
        p.TW = d->Wo; p.TH = std::min(DP_BM / d->Wo, NQ); p.tiles_x = 1; p.tiles_q = (NQ + p.TH - 1) / p.TH;
    } else {
        return SISR_E_UNSUPPORTED;
    }
    // cout tile: with plenty of pixel tiles 64-cout workgroups (two per CU, each other's stalls covered) measure 20-30 % faster in the
    // forward role; few pixel tiles, or a caller that announces a heavy prologue, take 128 (half the staging work per MFMA)
    const int n_mtiles0 = p.tiles_x * p.tiles_q * classes;
    p.BN = (d->Cout % 128 == 0) ? 128 : 64;
    if (p.BN == 128 && prefer_bn != 128 && (prefer_bn == 64 || n_mtiles0 >= 128)) p.BN = 64;
    // a 128-cout grid whose LAST round is less than half full (one workgroup per CU: 77 tiles x 4 cout tiles = 308 workgroups on 256 CUs
    // run as two full-length rounds) takes 64-cout tiles whatever the prologue: two workgroups per CU start the tail as slots free up
    // (measured at 24 x 24, B16: 512 -> 512 forward 114 -> 88 us, data gradient 153 -> 127 us; 256 -> 512 forward 63 -> 49 us)
    if (p.BN == 128) {
        const int base128 = n_mtiles0 * (d->Cout / 128), rem = base128 % 256;
        if (base128 > 256 && base128 < 768 && rem > 0 && rem <= 128) p.BN = 64;
    }
    if (const char* e = getenv("SISR_DEEP_BN")) { const int v = atoi(e); if (v == 64 || (v == 128 && d->Cout % 128 == 0)) p.BN = v; }   // A/B knob
    p.n_ntiles = d->Cout / p.BN;
    const int pad_bot = std::max(0, (d->Ho - 1) * S + d->KH - 1 - d->pad_y - (d->H - 1));
    p.PR = d->pad_y + d->H + pad_bot;
    if (p.PR < d->Ho * S) return SISR_E_UNSUPPORTED;
    const bool band = !(d->Ho % 8 == 0 && d->Wo % 16 == 0);
    const int nb = band ? std::min(d->N - 1, (p.TH - 1 + d->Ho - 1) / d->Ho) : 0;      // image boundaries a band can straddle
    p.IW = (p.TW - 1) * S + d->KW;
    p.IH_max = (p.TH - 1) * S + d->KH + nb * (p.PR - d->Ho * S);
    p.NIT = (p.IH_max * p.IW * 4 + DP_THREADS - 1) / DP_THREADS;
    if (p.NIT > 10) return SISR_E_UNSUPPORTED;
    {   // two halo buffers + a ring of three weight stages must fit the CU's 160 KB: the large halos of stride 2 take 64-cout tiles
        const int in_bytes = (p.IH_max * p.IW * DP_PSB + 15) & ~15, wrb = d->KW * 64 + 16;
        if (2 * in_bytes + 3 * p.BN * wrb > 160 * 1024 && p.BN == 128) { p.BN = 64; p.n_ntiles = d->Cout / 64; }
        if (2 * in_bytes + 3 * p.BN * wrb > 160 * 1024) return SISR_E_UNSUPPORTED;
    }
    if ((int64_t)d->N * p.PR >= 65536 || p.IH_max * p.IW >= 65536) return SISR_E_TOOBIG;
    // K split: reach ~target workgroups, at least two chunks per slice (a slice pays a prologue, an epilogue-sized partial
    // store and its share of the finishing pass)
    const int base = p.tiles_x * p.tiles_q * p.n_ntiles * classes;
    int target = target_wg > 0 ? target_wg : 256;
    if (const char* e = getenv("SISR_DEEP_TARGET")) target = std::max(1, atoi(e));
    int min_cps = 2;
    if (const char* e = getenv("SISR_DEEP_MINCPS")) min_cps = std::max(1, atoi(e));
    int want = std::max(1, target / std::max(1, base));
    int cps = std::max(std::min(min_cps, p.n_chunk), (p.n_chunk + want - 1) / want);
    p.cps = cps;
    p.split = (p.n_chunk + cps - 1) / cps;
    if (p.split == 1) p.cps = p.n_chunk;
    p.wimg_elems = p.n_chunk * d->KH * d->Cout * (d->KW * 32 + 8);
    p.ws_bytes = p.split > 1 ? (int64_t)base * p.split * DP_BM * p.BN * 4 : 0;
    if (p.tiles_q >= 65536 || p.tiles_x * p.tiles_q >= 65536) return SISR_E_TOOBIG;
    p.m_tiles_x = fdiv_magic(p.tiles_x); p.m_tw = fdiv_magic(p.TW); p.m_ho = fdiv_magic(d->Ho);
    p.m_pr = fdiv_magic(p.PR); p.m_iw = fdiv_magic(p.IW);
    p.enabled = 1;
    return 0;
}

static int deep_main_lds(const SisrConvDesc* d, int KW) {
    const SisrDeepPlan& p = d->deep;
    const int in_bytes = (p.IH_max * p.IW * DP_PSB + 15) & ~15;
    return 2 * in_bytes + 3 * p.BN * (KW * 64 + 16);
}

template <int NSUB, int KW, int NITM, bool TWO>
static int launch_deep_t(const SisrConvDesc* d, hipStream_t st) {
    const SisrDeepPlan& p = d->deep;
    const bool images = d->res != nullptr || d->bnb_part != nullptr;
    const int epi = deep_epi_lds(p.BN, images);
    const int lds_main = std::max(deep_main_lds(d, KW), p.split > 1 ? 0 : epi);
    if (lds_main > 160 * 1024) return SISR_E_TOOBIG;
    static SisrLdsCap cap;
    if (int e = sisr_raise_lds_cap(cap, reinterpret_cast<const void*>(&conv_deep_kernel<NSUB, KW, NITM, TWO>), lds_main, 0)) return e;
    const dim3 grid(p.n_ntiles, p.tiles_x * p.tiles_q, p.split * p.classes);
    hipLaunchKernelGGL((conv_deep_kernel<NSUB, KW, NITM, TWO>), grid, dim3(DP_THREADS), lds_main, st, *d);
    SISR_CHECK_LAUNCH();
    if (p.split > 1) {
        static SisrLdsCap capf;
        if (int e = sisr_raise_lds_cap(capf, reinterpret_cast<const void*>(&conv_deep_finish_kernel<NSUB>), epi, 0)) return e;
        hipLaunchKernelGGL((conv_deep_finish_kernel<NSUB>), dim3(p.n_ntiles, p.tiles_x * p.tiles_q, p.classes), dim3(DP_THREADS), epi, st, *d);
        SISR_CHECK_LAUNCH();
    }
    return 0;
}

template <int NSUB, int KW>
static int launch_deep_k(const SisrConvDesc* d, hipStream_t st) {
    const bool two = operand_needs_x2(d->pro_mode);
    if (d->deep.NIT <= 4) return two ? launch_deep_t<NSUB, KW, 4, true>(d, st) : launch_deep_t<NSUB, KW, 4, false>(d, st);
    if (two) return SISR_E_UNSUPPORTED;                         // (the stride-2 halo: forward role only)
    return launch_deep_t<NSUB, KW, 10, false>(d, st);
}

extern "C" int sisr_conv2d_deep_eligible(const SisrConvDesc* d) {
    if (!d || !d->deep.enabled || !d->wdeep) return 0;
    if (d->deep.classes == 4 && !(d->wdeep_c[0] && d->wdeep_c[1] && d->wdeep_c[2] && d->wdeep_c[3])) return 0;
    if (!d->x_bf16 || !d->y_bf16 || d->x_mode != SISR_X_NHWC || d->y_mode != SISR_Y_NHWC || d->epi_act != SISR_EPI_NONE) return 0;
    if (d->pro_mode == SISR_PRO_RES_AFFINE || d->pro_mode == SISR_PRO_TANH_BWD || d->fin_stat) return 0;
    if ((d->res && !d->res_bf16) || (d->bnb_part && !d->bnbx_bf16)) return 0;
    if (d->deep.NIT > 4 && operand_needs_x2(d->pro_mode)) return 0;
    if (d->deep.split > 1 && !d->deep_ws) return 0;
    return 1;
}

int sisr_conv2d_deep_launch(const SisrConvDesc* d, hipStream_t st) {
    if (operand_needs_x2(d->pro_mode) && !d->x2) return SISR_E_BADARG;
    if (d->stat_part && !d->cnt_part) return SISR_E_BADARG;
    if (d->bnb_part && (!d->bnb_x || !d->bnb_scale || !d->bnb_shift || !d->bnb_mean || !d->bnb_invstd))
        return SISR_E_BADARG;
    const int nsub = d->deep.BN / 64;
    if (nsub == 2) {
        if (d->KW == 3) return launch_deep_k<2, 3>(d, st);
        if (d->KW == 2) return launch_deep_k<2, 2>(d, st);
        return launch_deep_k<2, 1>(d, st);
    }
    if (d->KW == 3) return launch_deep_k<1, 3>(d, st);
    if (d->KW == 2) return launch_deep_k<1, 2>(d, st);
    return launch_deep_k<1, 1>(d, st);
}

// rows of stat_part / cnt_part (one per pixel tile) or of bnb_part (one per (pixel tile, cout tile)) a launch writes
int sisr_conv2d_deep_parts(const SisrConvDesc* d) {
    const int mt = d->deep.tiles_x * d->deep.tiles_q * d->deep.classes;
    return d->bnb_part ? mt * d->deep.n_ntiles : mt;
}
