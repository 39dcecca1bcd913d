// wgrad_bf16.hip -- weight gradient on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16, fp32 accumulate).
//
//   dW[tap][ci][co] = sum over output pixels p of  in(p*stride + tap - pad)[ci] * dy(p)[co]
//
// The contraction runs over PIXELS, but both operands sit in LDS pixel-major ([pixel][channel], as
// staged from NHWC memory), so each MFMA fragment (8 consecutive pixels of ONE channel per lane)
// is fetched with the gfx950 transposing LDS read ds_read_b64_tr_b16: per 16-lane group it reads a
// 4-pixel x 16-channel block and hands lane i channel i of the 4 pixels (cdna_hip_programming.md T10);
// two such reads make one 8-deep fragment.  One K step = 16 consecutive pixels of a tile row.
// Workgroup = (32-channel chunk q, cout tile of NJ*32); wave = (jsub, tap part): a wave keeps the
// accumulators of its taps resident over all tiles of the workgroup; partial slabs (one per pixel block)
// are reduced by sisr_slab_reduce_f32 (deterministic).
// Requirements: Cin % 32 == 0, KH*KW <= 9, Cout % 4 == 0.
#include "sisr_dev.h"

#include <algorithm>
#include <cstring>

#include "sisr_bf16_stage.h"


__device__ __forceinline__ bf16x8 frag8(const __bf16* p, int second_off) {
    const s16x4 lo = lds_tr16(p), hi = lds_tr16(p + second_off);
    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}

// phase timeline, developer build only (see conv_bf16.hip / tools/trace_conv.py)
#ifdef SISR_CONV_TRACE
#define SISR_WTRACE_WG 2048
#define SISR_WTRACE_SLOTS 32
__device__ unsigned long long sisr_wtrace_buf[SISR_WTRACE_WG * SISR_WTRACE_SLOTS];
#define WTR(k)                                                                                             \
    do {                                                                                                   \
        const int wg_ = blockIdx.y * gridDim.x + blockIdx.x;                                               \
        if (threadIdx.x == 0 && wg_ < SISR_WTRACE_WG && (k) < SISR_WTRACE_SLOTS)                           \
            sisr_wtrace_buf[wg_ * SISR_WTRACE_SLOTS + (k)] = wall_clock64();                               \
    } while (0)
extern "C" int sisr_wtrace_read(void* dst, int n_u64) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(sisr_wtrace_buf), (size_t)n_u64 * 8, 0, hipMemcpyDeviceToHost);
}
#else
#define WTR(k)
#endif

// LDS pixel strides (bf16 elements) chosen for the transposing reads: the 32 lanes of one LDS pass address
// 4 pixels x 64 bytes, so a pixel stride of 16 or 48 banks (mod 64) puts them on 64 distinct banks.
#define WG_PSX 32
__host__ __device__ static inline int wg_dy_stride(int DCH) { return DCH == 32 ? 32 : DCH + 32; }

// Wave roles: jsub = wave % NJ picks 32 output channels, tpart = wave / NJ picks TPW filter taps; every wave
// walks ALL pixels of the tile, so no cross-wave reduction is needed and a wave keeps only TPW accumulators
// (9 taps, NJ = 2: 5 + 4).  A wave whose tap range runs past the filter repeats the last tap and drops it.
// K steps (16 consecutive pixels of a tile row) are software-pipelined through two fragment sets: the
// transposing LDS reads of step k+1 are issued before the MFMAs of step k.
template <int TPW>
__global__ void __launch_bounds__(SISR_BLOCK, 2) wgrad_mfma_bf16_kernel(const SisrWgradDesc d) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int S = d.stride;
    const int TWp = (d.TW + 15) & ~15;
    const int IH = (d.TH - 1) * S + d.KH, IW = (TWp - 1) * S + d.KW;
    const int npix_in = d.TN * IH * IW;
    const int DCH = d.NJ * 32, DS = wg_dy_stride(DCH);
    const int ntap = d.KH * d.KW;
    const int jsub = wave & (d.NJ - 1), tpart = wave >> (d.NJ >> 1);          // NJ is 1 or 2
    const int tap0 = tpart * TPW;
    const int q = blockIdx.y % d.n_chunk, cot = blockIdx.y / d.n_chunk;
    const int co_base = cot * DCH;

    __bf16* lds_in = reinterpret_cast<__bf16*>(smem);
    __bf16* lds_dy = lds_in + ((npix_in * WG_PSX + 16 + 7) & ~7);

    OperandView ox, og;
    ox.x1 = d.x1; ox.x2 = d.x2; ox.pa = d.pa; ox.pb = d.pb; ox.pd = d.pd; ox.ps = d.ps; ox.pt = d.pt;
    ox.N = d.N; ox.H = d.H; ox.W = d.W; ox.C = d.Cin; ox.mode = d.x_mode; ox.pro = d.pro_mode;
    ox.slope = d.pro_slope_p ? d.pro_slope_p[0] : d.pro_slope;
    ox.bf16 = d.x_bf16; og.bf16 = d.g_bf16;
    og.x1 = d.g1; og.x2 = d.g2; og.pa = d.qa; og.pb = d.qb; og.pd = d.qd; og.ps = d.qs; og.pt = d.qt;
    og.N = d.N; og.H = d.Ho; og.W = d.Wo; og.C = d.Cout; og.mode = d.g_mode; og.pro = d.gpro_mode;
    og.slope = d.gpro_slope_p ? d.gpro_slope_p[0] : d.gpro_slope;

    // transposing-read lane roles: group g = lane>>4 -> (channel half g&1, pixel half g>>1); inside the
    // group lane 4q+p addresses (pixel row q, channels 4p..4p+3)
    const int grp = lane >> 4, tq = (lane & 15) >> 2, tp = lane & 3;
    const int pix_l = 8 * (grp >> 1) + tq;                 // pixel of this lane's address inside the K step
    const int ch_l = 16 * (grp & 1) + 4 * tp;              // first channel of this lane's address

    int aoff[TPW];
#pragma unroll
    for (int a = 0; a < TPW; ++a) {
        const int tap = min(tap0 + a, ntap - 1);
        const int r = fdiv(tap, d.m_kw), s = tap - r * d.KW;
        aoff[a] = (r * IW + s) * WG_PSX;
    }
    f32x16 acc[TPW];
#pragma unroll
    for (int a = 0; a < TPW; ++a)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[a][i] = 0.f;
    f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
    const bool want_bias = d.bias_slab != nullptr && q == 0;
    WTR(0);
    int titer = 0;

    const int nkx = TWp >> 4, nrows = d.TN * d.TH, nks = nrows * nkx;
    const int dy_step = 16 * DS, in_step = 16 * S * WG_PSX;
    const int in_row = (S * IW - (nkx - 1) * 16 * S) * WG_PSX;           // last K step of a row -> next row
    const int in_img = ((IH - (d.TH - 1) * S) * IW - (nkx - 1) * 16 * S) * WG_PSX;   // ... -> next image
    const __bf16* dy0 = lds_dy + pix_l * DS + jsub * 32 + ch_l;
    const __bf16* in0 = lds_in + pix_l * S * WG_PSX + ch_l;
    const int a2 = 4 * S * WG_PSX, b2 = 4 * DS;

    for (int t = blockIdx.x; t < d.n_tiles; t += gridDim.x, ++titer) {
        const int trow = fdiv(t, d.m_tiles_x), txi = t - trow * d.tiles_x;       // t < 2^16 (planner)
        const int ng = fdiv(trow, d.m_tiles_y), tyi = trow - ng * d.tiles_y;
        const int n0 = ng * d.TN, oy0 = tyi * d.TH, ox0 = txi * d.TW;
        __syncthreads();   // previous tile fully consumed
        WTR(1 + 5 * titer);
        stage_operand_tile_bf16<8>(ox, lds_in, WG_PSX, BF_CK, q * BF_CK, d.TN, IH, IW, n0, oy0 * S - d.pad_y,
                                ox0 * S - d.pad_x, 1 << 30, d.m_iw);
        WTR(2 + 5 * titer);
        // the bias gradient is summed from the dy values while they pass through registers (thread = 4 channels of
        // group tid % (DCH / 4), a share of the pixels): no second pass over the LDS image
        if (want_bias) stage_operand_tile_bf16<8, -1, true>(og, lds_dy, DS, DCH, co_base, d.TN, d.TH, TWp, n0, oy0, ox0, d.TW, d.m_twp, &bias4);
        else stage_operand_tile_bf16<8>(og, lds_dy, DS, DCH, co_base, d.TN, d.TH, TWp, n0, oy0, ox0, d.TW, d.m_twp);
        WTR(3 + 5 * titer);
        __syncthreads();
        WTR(4 + 5 * titer);
        // ---- pipelined K loop: position (kx, ty) of the NEXT step to fetch; fetching stops at the last step
        const __bf16* dyp = dy0;
        const __bf16* inp = in0;
        int kfetch = 0, kx = 0, ty = 0;
        bf16x8 bA, bB, aA[TPW], aB[TPW];
        auto fetch = [&](bf16x8& bf, bf16x8 (&af)[TPW]) {
            bf = frag8(dyp, b2);
#pragma unroll
            for (int a = 0; a < TPW; ++a) af[a] = frag8(inp + aoff[a], a2);
            if (kfetch + 1 < nks) {                       // uniform: advance to the next K step
                ++kfetch;
                dyp += dy_step;
                int din = in_step;
                if (++kx == nkx) { kx = 0; din = in_row; if (++ty == d.TH) { ty = 0; din = in_img; } }
                inp += din;
            }
        };
        auto mma = [&](const bf16x8& bf, const bf16x8 (&af)[TPW]) {
#pragma unroll
            for (int a = 0; a < TPW; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a], bf, acc[a], 0, 0, 0);
        };
        fetch(bA, aA);
        for (int ks = 0; ks < nks; ks += 2) {
            fetch(bB, aB);
            mma(bA, aA);
            __builtin_amdgcn_sched_group_barrier(0x100, 2 * (TPW + 1), 0);
            __builtin_amdgcn_sched_group_barrier(0x008, TPW, 0);
            fetch(bA, aA);
            if (ks + 1 < nks) mma(bB, aB);
            __builtin_amdgcn_sched_group_barrier(0x100, 2 * (TPW + 1), 0);
            __builtin_amdgcn_sched_group_barrier(0x008, TPW, 0);
        }
        WTR(5 + 5 * titer);
    }
    WTR(28);

    {   // slab layout [chunk][tap][ci (32)][CoutPad]: each wave owns its taps, no cross-wave reduction
        float* sl = d.slab + (int64_t)blockIdx.x * d.slab_stride;
#pragma unroll
        for (int a = 0; a < TPW; ++a) {
            if (tap0 + a < ntap) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int ci = mfma_row(i, lane);
                    sl[((int64_t)(q * ntap + tap0 + a) * 32 + ci) * d.CoutPad + co_base + jsub * 32 + (lane & 31)] = acc[a][i];
                }
            }
        }
    }
    WTR(29);
    if (want_bias) {
        __syncthreads();                                   // LDS is free: all tiles are done
        float* bsh = smem;
        const int G = DCH >> 2;
        *reinterpret_cast<f32x4*>(bsh + tid * 4) = bias4;
        __syncthreads();
        if (tid < DCH) {
            const int gq = tid >> 2, j = tid & 3;
            float s = 0.f;
            for (int k = gq; k < SISR_BLOCK; k += G) s += bsh[k * 4 + j];        // fixed order: deterministic
            d.bias_slab[(int64_t)blockIdx.x * d.slab_stride + co_base + tid] = s;
        }
    }
    WTR(30);
}

static inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

extern "C" int sisr_wgrad_plan_bf16(SisrWgradDesc* d, int32_t max_pixel_blocks) {
    if (!d || d->N <= 0 || d->Cin <= 0 || d->Cout <= 0 || d->KH <= 0 || d->KW <= 0) return SISR_E_BADARG;
    if (d->stride != 1 && d->stride != 2) return SISR_E_BADARG;
    if ((d->Cin % BF_CK) || d->KH * d->KW > 9 || (d->Cout & 3)) return SISR_E_UNSUPPORTED;
    if (d->x_mode == SISR_X_NCHW || d->g_mode == SISR_X_NCHW) return SISR_E_UNSUPPORTED;
    if ((int64_t)d->N * d->H * d->W * d->Cin >= (1ll << 30) || (int64_t)d->N * d->Ho * d->Wo * d->Cout >= (1ll << 30))
        return SISR_E_TOOBIG;
    d->CK = BF_CK; d->PS = BF_PS;
    d->KROWP = d->KH * d->KW * 32;        // rows per chunk in the slab: [tap][ci]
    d->n_chunk = d->Cin / BF_CK;
    d->NT = 1; d->TSTEP = 32; d->TVALID = 32;
    const int c32 = round_up(d->Cout, 32) / 32;
    d->NJ = c32 >= 2 ? 2 : 1;             // 64-wide cout tiles: 5 + 4 taps per wave pair (9 resident taps would spill)
    d->NP = 4 / d->NJ;
    d->CoutPad = round_up(d->Cout, d->NJ * 32);
    const int S = d->stride, DS = wg_dy_stride(d->NJ * 32);
    const int red_bytes = SISR_BLOCK * 4;               // bias partial combine
    // tile choice: the workgroups of one pixel block walk ceil(n_tiles / grid_x) tiles in sequence, so the cost
    // of a plan is that round count times the work of one tile (dy pixels incl. padding, the half-as-wide
    // x halo tile, a fixed per-tile synchronisation charge) -- this also balances the tail round
    const int per_pixel_block = d->n_chunk * (d->CoutPad / (d->NJ * 32));
    const int gx_max = std::max(1, max_pixel_blocks / per_pixel_block);
    double best = -1.0;
    for (int BMW = 256; BMW >= 64; BMW >>= 1) {
        for (int TW16 = 16; TW16 <= BMW; TW16 += 16) {          // tile widths: multiples of the K step, or the image
            const int TW = std::min(d->Wo, TW16), TWp = (TW + 15) & ~15;
            if (TWp != TW16) break;
            for (int TH = std::min(d->Ho, BMW / TWp); TH >= std::min(d->Ho, 4); --TH) {
                int TN = 1;
                if (TH == d->Ho && TW == d->Wo) TN = std::max(1, std::min(d->N, BMW / (TH * TWp)));
                const int IH = (TH - 1) * S + d->KH, IW = (TWp - 1) * S + d->KW;
                const int in_elems = (TN * IH * IW * WG_PSX + 16 + 7) & ~7;
                const int lds = std::max((in_elems + TN * TH * TWp * DS + 16) * 2, red_bytes);
                if (lds > 80 * 1024) continue;
                const int ty = (d->Ho + TH - 1) / TH, tx = (d->Wo + TW - 1) / TW, ngr = (d->N + TN - 1) / TN;
                const int64_t nt = (int64_t)ty * tx * ngr;
                const int64_t gx = std::min<int64_t>(gx_max, nt);
                const double rounds = (double)((nt + gx - 1) / gx);
                const double work = (double)TN * TH * TWp + 0.5 * TN * IH * IW + 48.0;
                const double score = 1.0 / (rounds * work);
                if (score > best * (1.0 + 1e-9)) {
                    best = score;
                    d->TH = TH; d->TW = TW; d->TN = TN; d->tiles_y = ty; d->tiles_x = tx; d->n_groups = ngr;
                    d->lds_bytes = lds;
                }
                if (TH * TWp <= BMW / 2) break;           // smaller heights belong to the next BMW
            }
        }
    }
    if (best < 0) return SISR_E_TOOBIG;
    d->n_tiles = d->tiles_y * d->tiles_x * d->n_groups;
    if (d->n_tiles >= 65536) return SISR_E_TOOBIG;          // range of the reciprocal index arithmetic
    {
        const int TWp = (d->TW + 15) & ~15;
        d->m_tiles_x = fdiv_magic(d->tiles_x); d->m_tiles_y = fdiv_magic(d->tiles_y);
        d->m_iw = fdiv_magic((TWp - 1) * S + d->KW); d->m_twp = fdiv_magic(TWp); d->m_kw = fdiv_magic(d->KW);
    }
    d->grid_x = std::min(gx_max, d->n_tiles);
    d->n_slabs = d->grid_x;
    d->slab_elems = d->n_chunk * d->KROWP * d->CoutPad;
    d->slab_stride = d->slab_elems;
    return 0;
}

template <int TPW>
static int launch_wgrad_bf16(const SisrWgradDesc* d, hipStream_t st) {
    static SisrLdsCap cap;
    if (int e = sisr_raise_lds_cap(cap, reinterpret_cast<const void*>(&wgrad_mfma_bf16_kernel<TPW>), d->lds_bytes, 64 * 1024)) return e;
    const dim3 grid(d->grid_x, d->n_chunk * (d->CoutPad / (d->NJ * 32)));
    hipLaunchKernelGGL(wgrad_mfma_bf16_kernel<TPW>, grid, dim3(SISR_BLOCK), d->lds_bytes, st, *d);
    SISR_CHECK_LAUNCH();
    return 0;
}

extern "C" int sisr_wgrad_trunk_eligible(const SisrWgradDesc* d);
int sisr_wgrad_trunk_launch(const SisrWgradDesc* d, hipStream_t st);      // wgrad_trunk.hip
extern "C" int sisr_wgrad_toimage_eligible(const SisrWgradDesc* d);
int sisr_wgrad_toimage_launch(const SisrWgradDesc* d, hipStream_t st);    // wgrad_toimage.hip
extern "C" int sisr_wgrad_deep_eligible(const SisrWgradDesc* d);
int sisr_wgrad_deep_launch(const SisrWgradDesc* d, hipStream_t st);       // wgrad_deep.hip

extern "C" int sisr_conv2d_wgrad_bf16(const SisrWgradDesc* d, void* stream) {
    if (!d || !d->x1 || !d->g1 || !d->slab) return SISR_E_BADARG;
    if (operand_needs_x2(d->pro_mode) && !d->x2) return SISR_E_BADARG;
    if (operand_needs_x2(d->gpro_mode) && !d->g2) return SISR_E_BADARG;
    if (d->slab_stride < d->slab_elems || d->CK != BF_CK || d->PS != BF_PS) return SISR_E_BADARG;
    if (d->grid_x <= 0 || d->lds_bytes <= 0 || d->lds_bytes > 160 * 1024) return SISR_E_BADARG;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (sisr_wgrad_trunk_eligible(d)) return sisr_wgrad_trunk_launch(d, st);
    if (sisr_wgrad_toimage_eligible(d)) return sisr_wgrad_toimage_launch(d, st);    // the generator's last conv (64 -> 3)
    if (sisr_wgrad_deep_eligible(d)) return sisr_wgrad_deep_launch(d, st);          // 3x3, channels in 64s
    const int np = 4 / d->NJ, ntap = d->KH * d->KW;
    switch ((ntap + np - 1) / np) {                        // taps per wave
        case 1: return launch_wgrad_bf16<1>(d, st);
        case 2: return launch_wgrad_bf16<2>(d, st);
        case 3: return launch_wgrad_bf16<3>(d, st);
        case 4: return launch_wgrad_bf16<4>(d, st);
        case 5: return launch_wgrad_bf16<5>(d, st);
        case 6: return launch_wgrad_bf16<6>(d, st);
        case 9: return launch_wgrad_bf16<9>(d, st);
    }
    return SISR_E_UNSUPPORTED;
}
