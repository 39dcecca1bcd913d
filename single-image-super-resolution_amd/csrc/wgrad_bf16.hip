// wgrad_bf16.hip -- weight gradient on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16, fp32 accumulate).
//
//   dW[tap][ci][co] = sum over output pixels p of  in(p*stride + tap - pad)[ci] * dy(p)[co]
//
// The contraction runs over PIXELS, but both operands sit in LDS pixel-major ([pixel][channel], as
// staged from NHWC memory), so each MFMA fragment (8 consecutive pixels of ONE channel per lane)
// is fetched with the gfx950 transposing LDS read ds_read_b64_tr_b16: per 16-lane group it reads a
// 4-pixel x 16-channel block and hands lane i channel i of the 4 pixels (cdna_hip_programming.md T10);
// two such reads make one 8-deep fragment.  One K step = 16 consecutive pixels of a tile row.
// Workgroup = (32-channel chunk q, cout tile of NJ*32); wave = (jsub, pixel part); all KH*KW tap
// accumulators stay resident; partial slabs are reduced by sisr_slab_reduce_f32 (deterministic).
// Requirements: Cin % 32 == 0, KH*KW <= 9, Cout % 4 == 0.
#include "sisr_dev.h"

#include <algorithm>
#include <cstring>

#include "sisr_bf16_stage.h"

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ s16x4 lds_tr16(const __bf16* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
}

__device__ __forceinline__ bf16x8 frag8(const __bf16* p, int second_off) {
    const s16x4 lo = lds_tr16(p), hi = lds_tr16(p + second_off);
    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}

template <int NTAP>
__global__ void __launch_bounds__(SISR_BLOCK, 2) wgrad_mfma_bf16_kernel(const SisrWgradDesc d) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int S = d.stride;
    const int TWp = (d.TW + 15) & ~15;
    const int IH = (d.TH - 1) * S + d.KH, IW = (TWp - 1) * S + d.KW;
    const int npix_in = d.TN * IH * IW;
    const int DCH = d.NJ * 32, DS = DCH + 8;
    const int NP = 4 / d.NJ;
    const int jsub = wave % d.NJ, ppart = wave / d.NJ;
    const int q = blockIdx.y % d.n_chunk, cot = blockIdx.y / d.n_chunk;
    const int co_base = cot * DCH;

    __bf16* lds_in = reinterpret_cast<__bf16*>(smem);
    __bf16* lds_dy = lds_in + ((npix_in * BF_PS + 16 + 7) & ~7);

    OperandView ox, og;
    ox.x1 = d.x1; ox.x2 = d.x2; ox.pa = d.pa; ox.pb = d.pb; ox.pd = d.pd; ox.ps = d.ps; ox.pt = d.pt;
    ox.N = d.N; ox.H = d.H; ox.W = d.W; ox.C = d.Cin; ox.mode = d.x_mode; ox.pro = d.pro_mode;
    ox.slope = d.pro_slope_p ? d.pro_slope_p[0] : d.pro_slope;
    og.x1 = d.g1; og.x2 = d.g2; og.pa = d.qa; og.pb = d.qb; og.pd = d.qd; og.ps = d.qs; og.pt = d.qt;
    og.N = d.N; og.H = d.Ho; og.W = d.Wo; og.C = d.Cout; og.mode = d.g_mode; og.pro = d.gpro_mode;
    og.slope = d.gpro_slope_p ? d.gpro_slope_p[0] : d.gpro_slope;

    // transposing-read lane roles: group g = lane>>4 -> (channel half g&1, pixel half g>>1); inside the
    // group lane 4q+p addresses (pixel row q, channels 4p..4p+3)
    const int grp = lane >> 4, tq = (lane & 15) >> 2, tp = lane & 3;
    const int pix_l = 8 * (grp >> 1) + tq;                 // pixel of this lane's address inside the K step
    const int ch_l = 16 * (grp & 1) + 4 * tp;              // first channel of this lane's address

    int aoff[NTAP];
#pragma unroll
    for (int a = 0; a < NTAP; ++a) {
        const int r = a / d.KW, s = a - r * d.KW;
        aoff[a] = (r * IW + s) * BF_PS;
    }
    f32x16 acc[NTAP];
#pragma unroll
    for (int a = 0; a < NTAP; ++a)
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
        stage_operand_tile_bf16<4>(ox, lds_in, BF_PS, BF_CK, q * BF_CK, d.TN, IH, IW, n0, oy0 * S - d.pad_y,
                                ox0 * S - d.pad_x, 1 << 30);
        stage_operand_tile_bf16<4>(og, lds_dy, DS, DCH, co_base, d.TN, d.TH, TWp, n0, oy0, ox0, d.TW);
        __syncthreads();
        if (d.bias_slab != nullptr && q == 0) {
            // bias-gradient partial: every thread sums a strided share of the tile's pixels for one channel
            // (all 256 threads, independent loads) -- combined across the pixel shares after the tile loop
            const int co = tid % DCH, share = tid / DCH, nshare = SISR_BLOCK / DCH;
            const int npx = d.TN * d.TH * TWp;
            float s = 0.f;
#pragma unroll 8
            for (int px = share; px < npx; px += nshare) s += (float)lds_dy[px * DS + co];
            bias_acc += s;
        }
        const int nrows = d.TN * d.TH;
        for (int row = ppart; row < nrows; row += NP) {
            const int tn = row / d.TH, ty = row - tn * d.TH;
            const __bf16* dyp = lds_dy + (row * TWp + pix_l) * DS + jsub * 32 + ch_l;
            const __bf16* inp = lds_in + ((tn * IH + ty * S) * IW + pix_l * S) * BF_PS + ch_l;
            for (int tx0 = 0; tx0 < TWp; tx0 += 16) {
                const bf16x8 bfrag = frag8(dyp + tx0 * DS, 4 * DS);
                const __bf16* ip = inp + tx0 * S * BF_PS;
                // taps in two groups to bound the live fragment registers (9 taps: 5 + 4)
                constexpr int G0 = (NTAP + 1) / 2, G1 = NTAP - G0;
                {
                    bf16x8 af[G0];
#pragma unroll
                    for (int a = 0; a < G0; ++a) af[a] = frag8(ip + aoff[a], 4 * S * BF_PS);
#pragma unroll
                    for (int a = 0; a < G0; ++a)
                        acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a], bfrag, acc[a], 0, 0, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 2 * (G0 + 1), 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, G0, 0);
                }
                if (G1 > 0) {
                    bf16x8 af[G1 > 0 ? G1 : 1];
#pragma unroll
                    for (int a = 0; a < G1; ++a) af[a] = frag8(ip + aoff[G0 + a], 4 * S * BF_PS);
#pragma unroll
                    for (int a = 0; a < G1; ++a)
                        acc[G0 + a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a], bfrag, acc[G0 + a], 0, 0, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 2 * G1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, G1, 0);
                }
            }
        }
    }

    // ---- sum the pixel parts of each jsub through LDS ------------------------------------------------
    __syncthreads();
    for (int k = 1; k < NP; ++k) {
        float* buf = smem + (size_t)jsub * (NTAP * 16 * 64);
        if (ppart == k) {
#pragma unroll
            for (int a = 0; a < NTAP; ++a)
#pragma unroll
                for (int i = 0; i < 16; ++i) buf[(a * 16 + i) * 64 + lane] = acc[a][i];
        }
        __syncthreads();
        if (ppart == 0) {
#pragma unroll
            for (int a = 0; a < NTAP; ++a)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[a][i] += buf[(a * 16 + i) * 64 + lane];
        }
        __syncthreads();
    }

    if (ppart == 0) {
        // slab layout [chunk][tap][ci (32)][CoutPad]
        float* sl = d.slab + (int64_t)blockIdx.x * d.slab_stride;
#pragma unroll
        for (int a = 0; a < NTAP; ++a)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int ci = mfma_row(i, lane);
                sl[((int64_t)(q * NTAP + a) * 32 + ci) * d.CoutPad + co_base + jsub * 32 + (lane & 31)] = acc[a][i];
            }
    }
    if (d.bias_slab != nullptr && q == 0) {
        __syncthreads();                                   // LDS is free: all tiles and the part reduction are done
        float* bsh = smem;
        bsh[tid] = bias_acc;
        __syncthreads();
        if (tid < DCH) {
            float s = 0.f;
            for (int k = tid; k < SISR_BLOCK; k += DCH) s += bsh[k];
            d.bias_slab[(int64_t)blockIdx.x * d.slab_stride + co_base + tid] = s;
        }
    }
}

static inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

extern "C" int sisr_wgrad_plan_bf16(SisrWgradDesc* d, int32_t max_pixel_blocks) {
    if (!d || d->N <= 0 || d->Cin <= 0 || d->Cout <= 0 || d->KH <= 0 || d->KW <= 0) return SISR_E_BADARG;
    if (d->stride != 1 && d->stride != 2) return SISR_E_BADARG;
    if ((d->Cin % BF_CK) || d->KH * d->KW > 9 || (d->Cout & 3)) return SISR_E_UNSUPPORTED;
    if (d->x_mode == SISR_X_NCHW || d->g_mode == SISR_X_NCHW) return SISR_E_UNSUPPORTED;
    if ((int64_t)d->N * d->H * d->W * d->Cin >= (1ll << 31) || (int64_t)d->N * d->Ho * d->Wo * d->Cout >= (1ll << 31))
        return SISR_E_TOOBIG;
    d->CK = BF_CK; d->PS = BF_PS;
    d->KROWP = d->KH * d->KW * 32;        // rows per chunk in the slab: [tap][ci]
    d->n_chunk = d->Cin / BF_CK;
    d->NT = 1; d->TSTEP = 32; d->TVALID = 32;
    const int c32 = round_up(d->Cout, 32) / 32;
    d->NJ = c32 >= 4 ? 4 : (c32 >= 2 ? 2 : 1);
    d->NP = 4 / d->NJ;
    d->CoutPad = round_up(d->Cout, d->NJ * 32);
    const int S = d->stride, DS = d->NJ * 32 + 8;
    const int red_bytes = d->NP > 1 ? d->NJ * (d->KH * d->KW) * 16 * 64 * 4 : 0;
    double best = -1.0;
    for (int BMW = 256; BMW >= 64 && best < 0; BMW >>= 1) {
        for (int TW = 1; TW <= std::min(d->Wo, BMW); ++TW) {
            const int TWp = (TW + 15) & ~15;
            if (TWp > BMW) continue;
            const int TH = std::min(d->Ho, BMW / TWp);
            if (TH < 1) continue;
            int TN = 1;
            if (TH == d->Ho && TW == d->Wo) TN = std::max(1, std::min(d->N, BMW / (TH * TWp)));
            const int IH = (TH - 1) * S + d->KH, IW = (TWp - 1) * S + d->KW;
            const int in_elems = (TN * IH * IW * BF_PS + 16 + 7) & ~7;
            const int lds = std::max((in_elems + TN * TH * TWp * DS + 16) * 2, red_bytes);
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
    const int per_pixel_block = d->n_chunk * (d->CoutPad / (d->NJ * 32));
    const int gx = std::max(1, max_pixel_blocks / per_pixel_block);
    d->grid_x = std::min(gx, d->n_tiles);
    d->n_slabs = d->grid_x;
    d->slab_elems = d->n_chunk * d->KROWP * d->CoutPad;
    d->slab_stride = d->slab_elems;
    return 0;
}

template <int NTAP>
static int launch_wgrad_bf16(const SisrWgradDesc* d, hipStream_t st) {
    static int lds_max = 64 * 1024;
    if (d->lds_bytes > lds_max) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_mfma_bf16_kernel<NTAP>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, d->lds_bytes);
        if (e != hipSuccess) return (int)e;
        lds_max = d->lds_bytes;
    }
    const dim3 grid(d->grid_x, d->n_chunk * (d->CoutPad / (d->NJ * 32)));
    hipLaunchKernelGGL(wgrad_mfma_bf16_kernel<NTAP>, grid, dim3(SISR_BLOCK), d->lds_bytes, st, *d);
    SISR_CHECK_LAUNCH();
    return 0;
}

extern "C" int sisr_conv2d_wgrad_bf16(const SisrWgradDesc* d, void* stream) {
    if (!d || !d->x1 || !d->g1 || !d->slab) return SISR_E_BADARG;
    if (operand_needs_x2(d->pro_mode) && !d->x2) return SISR_E_BADARG;
    if (operand_needs_x2(d->gpro_mode) && !d->g2) return SISR_E_BADARG;
    if (d->slab_stride < d->slab_elems || d->CK != BF_CK || d->PS != BF_PS) return SISR_E_BADARG;
    if (d->grid_x <= 0 || d->lds_bytes <= 0 || d->lds_bytes > 160 * 1024) return SISR_E_BADARG;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    switch (d->KH * d->KW) {
        case 1: return launch_wgrad_bf16<1>(d, st);
        case 2: return launch_wgrad_bf16<2>(d, st);
        case 3: return launch_wgrad_bf16<3>(d, st);
        case 4: return launch_wgrad_bf16<4>(d, st);
        case 6: return launch_wgrad_bf16<6>(d, st);
        case 9: return launch_wgrad_bf16<9>(d, st);
    }
    return SISR_E_UNSUPPORTED;
}
