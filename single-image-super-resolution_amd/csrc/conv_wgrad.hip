// conv_wgrad.hip -- weight gradient of the direct convolution on gfx950 (fp32, exact-fp32 MFMA).
//
//   dW[r][s][ci][co] = sum over output pixels p of  in(p*stride + (r,s) - pad)[ci] * dy(p)[co]
//
// GEMM view per filter row r: rows i = contiguous K-row index (s*PS + ci) of the LDS input tile
// (same [pixel][PS] image as conv_fwd.hip), columns j = co, contraction over PIXELS (2 per
// v_mfma_f32_32x32x2_f32).  A workgroup owns one (channel chunk, cout tile) and walks pixel tiles
// grid-stride with all KH*NT accumulators (<= 9 x 16 VGPRs) resident, so the input tile is read
// from HBM once for all taps.  Wave w = (jsub = w % NJ : its 32 couts, ppart = w / NJ : its share
// of the tile's pixel rows); parts are summed through LDS and every workgroup writes ONE partial
// slab, reduced by sisr_slab_reduce_f32 (deterministic, no atomics).  Both operands take the lazy
// prologues of sisr_hip.h, e.g. in = PReLU(BN(c_prev)) and dy = BatchNorm-backward(g, c).
// Replaces the weight/bias half of convolution_backward for the nn.Conv2d call sites listed in
// conv_fwd.hip.
#include "sisr_dev.h"

#include <algorithm>
#include <cstring>

#define WG_NACC 9

// one K-step = 2 pixels: NACC A operands (one per (filter row, row tile)) against one B operand
template <int NACC, int KU>
__device__ __forceinline__ void wgrad_kblock(const float* ip, const float* dyp, const int (&aoff)[NACC], int istep,
                                             int dstep, f32x16 (&acc)[NACC]) {
    float a[KU][NACC], b[KU];
#pragma unroll
    for (int u = 0; u < KU; ++u) {
        b[u] = dyp[u * dstep];
#pragma unroll
        for (int i = 0; i < NACC; ++i) a[u][i] = ip[u * istep + aoff[i]];
    }
#pragma unroll
    for (int u = 0; u < KU; ++u)
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = mfma32(a[u][i], b[u], acc[i]);
    // pin the schedule: all LDS reads of the block first, then the MFMAs (counted lgkmcnt waits)
    __builtin_amdgcn_sched_group_barrier(0x100, KU * (NACC + 1), 0);
    __builtin_amdgcn_sched_group_barrier(0x008, KU * NACC, 0);
}

template <int NACC>
__global__ void __launch_bounds__(SISR_BLOCK, 2) wgrad_mfma_f32_kernel(const SisrWgradDesc d) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, kk = lane >> 5;
    const int S = d.stride;
    const int TWp = (d.TW + 1) & ~1;
    const int IH = (d.TH - 1) * S + d.KH, IW = (TWp - 1) * S + d.KW;
    const int in_elems = d.TN * IH * IW * d.PS;
    const int DSTR = d.NJ * 32;
    const int NP = 4 / d.NJ;
    const int jsub = wave % d.NJ, ppart = wave / d.NJ;
    const int q = blockIdx.y % d.n_chunk, cot = blockIdx.y / d.n_chunk;
    const int co_base = cot * DSTR;

    float* lds_in = smem;
    float* lds_dy = smem + ((in_elems + 64 + 3) & ~3);

    OperandView ox, og;
    ox.x1 = d.x1; ox.x2 = d.x2; ox.pa = d.pa; ox.pb = d.pb; ox.pd = d.pd; ox.ps = d.ps; ox.pt = d.pt;
    ox.N = d.N; ox.H = d.H; ox.W = d.W; ox.C = d.Cin; ox.mode = d.x_mode; ox.pro = d.pro_mode;
    ox.slope = d.pro_slope_p ? d.pro_slope_p[0] : d.pro_slope;
    og.x1 = d.g1; og.x2 = d.g2; og.pa = d.qa; og.pb = d.qb; og.pd = d.qd; og.ps = d.qs; og.pt = d.qt;
    og.N = d.N; og.H = d.Ho; og.W = d.Wo; og.C = d.Cout; og.mode = d.g_mode; og.pro = d.gpro_mode;
    og.slope = d.gpro_slope_p ? d.gpro_slope_p[0] : d.gpro_slope;
    ox.bf16 = d.x_bf16; og.bf16 = d.g_bf16;
    const bool xvec = (d.x_mode != SISR_X_NCHW) && !(d.CK & 3) && !((d.CK >> 2) & ((d.CK >> 2) - 1)) && !(d.Cin & 3) &&
                      !(d.x_mode == SISR_X_NHWC_UNSHUFFLE2 && ((d.Cin >> 2) & 3));
    const bool gvec = (d.g_mode != SISR_X_NCHW) && !(d.Cout & 3) &&
                      !(d.g_mode == SISR_X_NHWC_UNSHUFFLE2 && ((d.Cout >> 2) & 3));

    int aoff[NACC];
#pragma unroll
    for (int a = 0; a < NACC; ++a) {
        const int r = a / d.NT, tt = a - r * d.NT;
        aoff[a] = r * IW * d.PS + tt * d.TSTEP;
    }
    f32x16 acc[NACC];
#pragma unroll
    for (int a = 0; a < NACC; ++a)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[a][i] = 0.f;
    float bias_acc = 0.f;

    for (int t = blockIdx.x; t < d.n_tiles; t += gridDim.x) {
        int tt_ = t;
        const int txi = tt_ % d.tiles_x;
        tt_ /= d.tiles_x;
        const int tyi = tt_ % d.tiles_y, ng = tt_ / d.tiles_y;
        const int n0 = ng * d.TN, oy0 = tyi * d.TH, ox0 = txi * d.TW;
        __syncthreads();   // previous tile fully consumed
        stage_operand_tile<4>(ox, lds_in, d.PS, d.CK, q * d.CK, d.TN, IH, IW, n0, oy0 * S - d.pad_y,
                           ox0 * S - d.pad_x, xvec, 1 << 30, 64);
        // dy tile: TN x TH x TWp pixels, channels [co_base, co_base + DSTR); columns >= TW are zero
        stage_operand_tile<4>(og, lds_dy, DSTR, DSTR, co_base, d.TN, d.TH, TWp, n0, oy0, ox0, gvec, d.TW, 0);
        __syncthreads();
        if (d.bias_slab != nullptr && q == 0) {
            // bias-gradient partial: every thread sums a strided share of the tile's pixels for one channel
            // (all 256 threads, independent loads) -- combined across the pixel shares after the tile loop
            const int co = tid % DSTR, share = tid / DSTR, nshare = SISR_BLOCK / DSTR;
            const int npx = d.TN * d.TH * TWp;
            float s = 0.f;
#pragma unroll 8
            for (int px = share; px < npx; px += nshare) s += lds_dy[px * DSTR + co];
            bias_acc += s;
        }
        const int nrows = d.TN * d.TH;
        for (int row = ppart; row < nrows; row += NP) {
            const int tn = row / d.TH, ty = row - tn * d.TH;
            const float* dyp = lds_dy + (row * TWp + kk) * DSTR + jsub * 32 + l31;
            const float* inp = lds_in + ((tn * IH + ty * S) * IW + kk * S) * d.PS + l31;
            const int istep = 2 * S * d.PS, dstep = 2 * DSTR;      // one K-step = 2 pixels
            int tx0 = 0;
            for (; tx0 + 4 <= TWp; tx0 += 4)
                wgrad_kblock<NACC, 2>(inp + tx0 * S * d.PS, dyp + tx0 * DSTR, aoff, istep, dstep, acc);
            if (tx0 < TWp) wgrad_kblock<NACC, 1>(inp + tx0 * S * d.PS, dyp + tx0 * DSTR, aoff, istep, dstep, acc);
        }
    }

    // ---- sum the pixel parts of each jsub through LDS (one round per extra part) -----------------
    __syncthreads();
    for (int k = 1; k < NP; ++k) {
        float* buf = smem + (size_t)jsub * (NACC * 16 * 64);
        if (ppart == k) {
#pragma unroll
            for (int a = 0; a < NACC; ++a)
#pragma unroll
                for (int i = 0; i < 16; ++i) buf[(a * 16 + i) * 64 + lane] = acc[a][i];
        }
        __syncthreads();
        if (ppart == 0) {
#pragma unroll
            for (int a = 0; a < NACC; ++a)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[a][i] += buf[(a * 16 + i) * 64 + lane];
        }
        __syncthreads();
    }

    if (ppart == 0) {
        float* sl = d.slab + (int64_t)blockIdx.x * d.slab_stride;
        const int kvalid = d.KW * d.PS;
#pragma unroll
        for (int a = 0; a < NACC; ++a) {
            const int r = a / d.NT, tt = a - r * d.NT;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = mfma_row(i, lane);
                const int krow = tt * d.TSTEP + row;
                if (row < d.TVALID && krow < kvalid)
                    sl[((int64_t)(q * d.KH + r) * d.KROWP + krow) * d.CoutPad + co_base + jsub * 32 + l31] =
                        acc[a][i];
            }
        }
    }
    if (d.bias_slab != nullptr && q == 0) {
        __syncthreads();                                   // LDS is free: all tiles and the part reduction are done
        float* bsh = smem;
        bsh[tid] = bias_acc;
        __syncthreads();
        if (tid < DSTR) {
            float s = 0.f;
            for (int k = tid; k < SISR_BLOCK; k += DSTR) s += bsh[k];
            d.bias_slab[(int64_t)blockIdx.x * d.slab_stride + co_base + tid] = s;
        }
    }
}

// out[i] = sum_k slab[k][i]: 16 float4 columns x 16 slab-splits per workgroup, combined through LDS in a fixed order
// (deterministic).  The reduction is latency-bound, not bandwidth-bound (a few hundred slabs of ~150 KB, L2 / MALL
// resident): many short independent load chains beat few long ones.
__global__ void __launch_bounds__(SISR_BLOCK) slab_reduce_kernel(const float* __restrict__ slab,
                                                                float* __restrict__ out, int n_slabs,
                                                                int64_t elems, int64_t lead) {
    __shared__ f32x4 sh[SR_SPLITS][SR_COLS];
    slab_reduce_block(slab, out, n_slabs, elems, blockIdx.x, sh, lead);
}

// several independent reductions in ONE launch (the weight gradients of one backward pass through the discriminator: seven slab
// sets that are all final before the un-packing launch needs any of them); the jobs travel in the kernel arguments
#define SR_MAX_JOBS 8
struct SlabJobs {
    const float* slab[SR_MAX_JOBS];
    float* out[SR_MAX_JOBS];
    int64_t elems[SR_MAX_JOBS], lead[SR_MAX_JOBS];
    int n_slabs[SR_MAX_JOBS];
    int first_block[SR_MAX_JOBS + 1];
    int n;
};
__global__ void __launch_bounds__(SISR_BLOCK) slab_reduce_multi_kernel(const SlabJobs j) {
    __shared__ f32x4 sh[SR_SPLITS][SR_COLS];
    int k = 0;
#pragma unroll
    for (int i = 1; i < SR_MAX_JOBS; ++i) k += (i < j.n && (int)blockIdx.x >= j.first_block[i]) ? 1 : 0;
    // (uniform per workgroup: a job is served by one form or the other, see sisr_slab_reduce_multi's block counts)
    if (j.n_slabs[k] <= SR_SPLITS) slab_reduce_block_few(j.slab[k], j.out[k], j.n_slabs[k], j.elems[k], (int)blockIdx.x - j.first_block[k], j.lead[k]);
    else slab_reduce_block(j.slab[k], j.out[k], j.n_slabs[k], j.elems[k], (int)blockIdx.x - j.first_block[k], sh, j.lead[k]);
}

static inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

extern "C" int sisr_wgrad_plan(SisrWgradDesc* d, int32_t max_pixel_blocks) {
    if (!d || d->N <= 0 || d->Cin <= 0 || d->Cout <= 0 || d->KH <= 0 || d->KW <= 0) return SISR_E_BADARG;
    if (d->stride != 1 && d->stride != 2) return SISR_E_BADARG;
    if ((int64_t)d->N * d->H * d->W * d->Cin >= (1ll << 31) || (int64_t)d->N * d->Ho * d->Wo * d->Cout >= (1ll << 31))
        return SISR_E_TOOBIG;
    d->CK = d->Cin <= 32 ? d->Cin : 32;
    d->PS = d->CK | 1;
    d->KROWP = round_up(d->KW * d->PS, 4);
    d->n_chunk = (d->Cin + d->CK - 1) / d->CK;
    if (d->CK == 32) { d->NT = d->KW; d->TSTEP = d->PS; d->TVALID = 32; }
    else { d->NT = (d->KW * d->PS + 31) / 32; d->TSTEP = 32; d->TVALID = 32; }
    if (d->KH * d->NT > WG_NACC) return SISR_E_UNSUPPORTED;
    const int c32 = round_up(d->Cout, 32) / 32;
    d->NJ = c32 >= 4 ? 4 : (c32 >= 2 ? 2 : 1);
    d->NP = 4 / d->NJ;
    d->CoutPad = round_up(d->Cout, d->NJ * 32);
    const int S = d->stride, DSTR = d->NJ * 32;
    const int red_bytes = d->NP > 1 ? d->NJ * (d->KH * d->NT) * 16 * 64 * 4 : 0;
    double best = -1.0;
    for (int BMW = 128; BMW >= 32 && best < 0; BMW >>= 1) {
        for (int TW = 1; TW <= std::min(d->Wo, BMW); ++TW) {
            const int TWp = (TW + 1) & ~1;
            const int TH = std::min(d->Ho, BMW / TWp);
            if (TH < 1) continue;
            int TN = 1;
            if (TH == d->Ho && TW == d->Wo) TN = std::max(1, std::min(d->N, BMW / (TH * TWp)));
            const int IH = (TH - 1) * S + d->KH, IW = (TWp - 1) * S + d->KW;
            const int in_elems = TN * IH * IW * d->PS;
            const int lds = std::max((((in_elems + 64 + 3) & ~3) + TN * TH * TWp * DSTR + 8) * 4, red_bytes);
            if (lds > 80 * 1024) continue;
            const int ty = (d->Ho + TH - 1) / TH, tx = (d->Wo + TW - 1) / TW, ngr = (d->N + TN - 1) / TN;
            const double eff = (double)d->N * d->Ho * d->Wo / ((double)ty * tx * ngr * TN * TH * TWp);
            const double halo = (double)(TH * TW) * S * S / ((double)IH * IW);
            const double fill = (double)(TN * TH * TWp) / BMW;
            const double score = eff * (0.7 + 0.3 * halo) * (0.8 + 0.2 * fill);
            if (score > best + 1e-9) {
                best = score;
                d->TH = TH; d->TW = TW; d->TN = TN; d->tiles_y = ty; d->tiles_x = tx; d->n_groups = ngr;
                d->lds_bytes = lds;
            }
        }
    }
    if (best < 0) return SISR_E_TOOBIG;
    d->n_tiles = d->tiles_y * d->tiles_x * d->n_groups;
    const int per_pixel_block = d->n_chunk * (d->CoutPad / DSTR);
    int gx = std::max(1, max_pixel_blocks / per_pixel_block);
    d->grid_x = std::min(gx, d->n_tiles);
    d->n_slabs = d->grid_x;
    d->slab_elems = d->n_chunk * d->KH * d->KROWP * d->CoutPad;
    d->slab_stride = d->slab_elems;
    return 0;
}

template <int NACC>
static int launch_wgrad(const SisrWgradDesc* d, hipStream_t st) {
    static SisrLdsCap cap;
    if (int e = sisr_raise_lds_cap(cap, reinterpret_cast<const void*>(&wgrad_mfma_f32_kernel<NACC>), d->lds_bytes, 64 * 1024)) return e;
    const dim3 grid(d->grid_x, d->n_chunk * (d->CoutPad / (d->NJ * 32)));
    hipLaunchKernelGGL(wgrad_mfma_f32_kernel<NACC>, grid, dim3(SISR_BLOCK), d->lds_bytes, st, *d);
    SISR_CHECK_LAUNCH();
    return 0;
}

extern "C" int sisr_wgrad_trunk_f32_eligible(const SisrWgradDesc* d);
int sisr_wgrad_trunk_f32_launch(const SisrWgradDesc* d, hipStream_t st);      // wgrad_trunk_f32.hip
extern "C" int sisr_wgrad_thin_eligible(const SisrWgradDesc* d);
int sisr_wgrad_thin_launch(const SisrWgradDesc* d, hipStream_t st);           // wgrad_thin.hip
extern "C" int sisr_wgrad_toimage_f32_eligible(const SisrWgradDesc* d);
int sisr_wgrad_toimage_f32_launch(const SisrWgradDesc* d, hipStream_t st);    // wgrad_toimage.hip

extern "C" int sisr_conv2d_wgrad_f32(const SisrWgradDesc* d, void* stream) {
    if (!d || !d->x1 || !d->g1 || !d->slab) return SISR_E_BADARG;
    if (operand_needs_x2(d->pro_mode) && !d->x2) return SISR_E_BADARG;
    if (operand_needs_x2(d->gpro_mode) && !d->g2) return SISR_E_BADARG;
    if (d->slab_stride < d->slab_elems) return SISR_E_BADARG;
    if (d->grid_x <= 0 || d->lds_bytes <= 0 || d->lds_bytes > 160 * 1024 || d->KH * d->NT > WG_NACC)
        return SISR_E_BADARG;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (sisr_wgrad_trunk_f32_eligible(d)) return sisr_wgrad_trunk_f32_launch(d, st);
    if (sisr_wgrad_thin_eligible(d)) return sisr_wgrad_thin_launch(d, st);      // bf16 build: the first conv (3-channel image)
    if (sisr_wgrad_toimage_f32_eligible(d)) return sisr_wgrad_toimage_f32_launch(d, st);    // the last conv (64 -> 3), fp32 tensors
    switch (d->KH * d->NT) {
        case 1: return launch_wgrad<1>(d, st);
        case 2: return launch_wgrad<2>(d, st);
        case 3: return launch_wgrad<3>(d, st);
        case 4: return launch_wgrad<4>(d, st);
        case 5: return launch_wgrad<5>(d, st);
        case 6: return launch_wgrad<6>(d, st);
        case 7: return launch_wgrad<7>(d, st);
        case 8: return launch_wgrad<8>(d, st);
        case 9: return launch_wgrad<9>(d, st);
    }
    return SISR_E_UNSUPPORTED;
}

extern "C" int sisr_slab_reduce_f32(const float* slab, float* out, int32_t n_slabs, int64_t elems, int64_t lead_bf16,
                                    void* stream) {
    if (!slab || !out || n_slabs <= 0 || elems <= 0) return SISR_E_BADARG;
    if ((elems & 3) || (lead_bf16 & 3) || lead_bf16 < 0 || lead_bf16 > elems) return SISR_E_BADARG;   // 16-byte units
    const int blocks = (int)((elems / 4 + SR_COLS - 1) / SR_COLS);
    hipLaunchKernelGGL(slab_reduce_kernel, dim3(blocks), dim3(SISR_BLOCK), 0, reinterpret_cast<hipStream_t>(stream),
                       slab, out, n_slabs, elems, lead_bf16);
    SISR_CHECK_LAUNCH();
    return 0;
}

extern "C" int sisr_slab_reduce_multi(const void* const* slabs, void* const* outs, const int32_t* n_slabs, const int64_t* elems,
                                      const int64_t* leads, int32_t n_jobs, void* stream) {
    if (!slabs || !outs || !n_slabs || !elems || !leads || n_jobs <= 0) return SISR_E_BADARG;
    for (int i0 = 0; i0 < n_jobs; i0 += SR_MAX_JOBS) {
        SlabJobs j;
        std::memset(&j, 0, sizeof(j));
        j.n = std::min(SR_MAX_JOBS, n_jobs - i0);
        int blocks = 0;
        for (int k = 0; k < j.n; ++k) {
            const int i = i0 + k;
            if (!slabs[i] || !outs[i] || n_slabs[i] <= 0 || elems[i] <= 0) return SISR_E_BADARG;
            if ((elems[i] & 3) || (leads[i] & 3) || leads[i] < 0 || leads[i] > elems[i]) return SISR_E_BADARG;
            j.slab[k] = static_cast<const float*>(slabs[i]); j.out[k] = static_cast<float*>(outs[i]);
            j.n_slabs[k] = n_slabs[i]; j.elems[k] = elems[i]; j.lead[k] = leads[i];
            j.first_block[k] = blocks;
            const int cols = n_slabs[i] <= SR_SPLITS ? SR_FEW_COLS : SR_COLS;
            blocks += (int)((elems[i] / 4 + cols - 1) / cols);
        }
        j.first_block[j.n] = blocks;
        hipLaunchKernelGGL(slab_reduce_multi_kernel, dim3(blocks), dim3(SISR_BLOCK), 0, reinterpret_cast<hipStream_t>(stream), j);
        SISR_CHECK_LAUNCH();
    }
    return 0;
}
