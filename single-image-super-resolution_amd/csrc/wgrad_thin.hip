// wgrad_thin.hip -- weight gradient of the generator's first convolution (model_generator.py:32: 9x9, 3 -> 64, stride 1,
// pad 4) in the bf16 build: dW[ky][kx*3 + ci][co] = sum_pixels x(pixel + tap)[ci] * dy(pixel)[co] with x the NCHW fp32
// image and dy the bf16 NHWC gradient (activation-backward prologue: PReLU'(pre-activation) of model_generator.py:33).
//
// The generic fp32 kernel (conv_wgrad.hip) spends 103 us on these 4.6 GFLOP (profiles/r02_trace_step_order.txt): the
// 3-channel operand forces exact-fp32 MFMAs on 27-wide K rows.  Here the contraction runs on the bf16 matrix cores as
//     D[co 64][n = ky*27 + kx*3 + ci, 243 of 256] += A[co][pixel] * B[pixel][n],          K = pixels, 16 per MFMA
//   * A (dy^T): the dy tile lies in LDS as [pixel][64 ch]; ds_read_b64_tr_b16 hands every lane 8 consecutive pixels of
//     its channel (same fragment read as wgrad_trunk.hip);
//   * B: lane n needs 8 consecutive pixels of image row (r + ky), channel ci, starting at column (x + kx) -- any 2-byte
//     alignment.  The 3-channel halo is tiny, so it is kept as EIGHT copies per channel, copy s shifted left by s
//     elements: the fragment is one aligned 16-byte read from copy (kx & 7);
//   * 256 threads, one workgroup per CU, persistent over 8 x 32 pixel tiles; wave w owns columns n = 64w .. 64w+63 for
//     all 64 couts (4 accumulators); per 16-pixel K step: 4 transposing reads + 2 fragment reads feed 4 MFMAs;
//   * the next tile's operands are requested before the MFMA phase and committed to LDS after it;
//   * one fp32 slab [ky][krow 28][co 64] (+ bias row) per workgroup, summed by sisr_slab_reduce_f32 like every other
//     weight gradient -- the accumulation order is fixed, results are bit-reproducible.
// Requirements (sisr_wgrad_thin_eligible): the geometry above, H % 8 == 0, W % 32 == 0, NCHW fp32 x without prologue,
// bf16 NHWC dy with prologue NONE or ACT_BWD.
#include "sisr_dev.h"

#include <algorithm>
#include <cstdlib>

#include "sisr_bf16_stage.h"

#define WN_TH 8
#define WN_TW 32
#define WN_PS 192                          // LDS bytes per dy pixel: 64 bf16 + 64 bytes (bank spread of the transposing reads)
#define WN_DBYTES (WN_TH * WN_TW * WN_PS)  // 49152
#define WN_XROW 80                         // bytes per halo row of one (channel, shift) copy: 5 blocks of 8 bf16
#define WN_XBYTES (3 * 8 * 16 * WN_XROW)   // 30720
#define WN_KROWP 28
#define WN_SLAB (9 * WN_KROWP * 64)

struct WThinArgs {
    const float* x;
    const void *g1, *g2;
    float *slab, *bias_slab;
    const float* slope_p;
    float slope;
    int N, H, W;
    int tiles_x, per_img, total;
    long long slab_stride;
};

// KS = 9: the generator's first conv (above).  KS = 3: the discriminator's first conv (model_discriminator.py:36: 3x3, 3 -> 64,
// stride 1, pad 1): n = ky * 9 + kx * 3 + ci is 27 of ONE 32-column tile, so the four waves split the tile's ROWS instead of the
// columns (two accumulators each, summed across the waves in wave order at the end) and only the shift copies 0 .. 2 exist; the halo
// starts one column left of a 16-byte boundary, so a thread takes elements 3 .. 12 of its four aligned quads.
template <bool ACTB, int KS>
__global__ void __launch_bounds__(256, 1) wgrad_thin_kernel(const WThinArgs a) {
    constexpr int HALO = KS / 2, XROWS = WN_TH + KS - 1, XTHREADS = 3 * XROWS * 5, NQ = 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char* xs = lds;
    unsigned char* ds = lds + WN_XBYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, kk = lane >> 5, grp = lane >> 4, tq = (lane & 15) >> 2, tp = lane & 3;
    const float slope = a.slope_p ? a.slope_p[0] : a.slope;

    // ---- per-lane constants of the two B columns n = 64 wave + 32 nt + l31 -------------------------------------------
    int bbase[2], srow[2];
    bool zrow[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        if constexpr (KS == 9) {
            const int n = 64 * wave + 32 * nt + l31;
            const bool valid = n < 243;
            const int nn = valid ? n : 0;
            const int ky = nn / 27, krow = nn - 27 * ky, kx = krow / 3, ci = krow - 3 * kx;
            bbase[nt] = ((ci * 8 + (kx & 7)) * 16 + ky) * WN_XROW + ((kx >> 3) + kk) * 16;
            // columns 243 .. 251 write the (zero) padding rows krow = 27 of the slab instead
            zrow[nt] = n >= 243 && n < 252;
            srow[nt] = valid ? (ky * WN_KROWP + krow) * 64 : zrow[nt] ? ((n - 243) * WN_KROWP + 27) * 64 : -1;
        } else {
            const int n = l31, nn = n < 27 ? n : 0;
            const int ky = nn / 9, krow = nn - 9 * ky, kx = krow / 3, ci = krow - 3 * kx;
            bbase[nt] = ((ci * 8 + kx) * 16 + ky) * WN_XROW + kk * 16;
            zrow[nt] = false;
            srow[nt] = n < 27 ? (ky * 12 + krow) * 64 : -1;
        }
    }
    // A fragment (transposing read): lane 4q+p of a 16-lane group addresses pixel q (+4), channels 4p.. of its 16
    const int a_base = (8 * (grp >> 1) + tq) * WN_PS + (16 * (grp & 1) + 4 * tp) * 2;

    // ---- staging ---------------------------------------------------------------------------------------------------------
    const unsigned plane = (unsigned)(a.H * a.W);
    const __amdgpu_buffer_rsrc_t rx = sisr_rsrc(a.x, (unsigned)a.N * 3u * plane * 4u);
    const unsigned gbytes = (unsigned)a.N * plane * 128u;
    const __amdgpu_buffer_rsrc_t rg = sisr_rsrc(a.g1, gbytes), rp = sisr_rsrc(ACTB ? a.g2 : a.g1, gbytes);
    // x: thread < 3 * XROWS * 5 -> (channel, halo row, block of 8 columns); it loads columns 8 blk .. 8 blk + 15
    const int x_ci = tid / (5 * XROWS), x_row = (tid - 5 * XROWS * x_ci) / 5, x_blk = tid % 5;
    const int x_lds = ((x_ci * 8) * 16 + x_row) * WN_XROW + x_blk * 16;
    // dy: thread -> channels 8 oct .., pixels (row k, column p0), k = 0 .. 7
    const int oct = tid & 7, p0 = tid >> 3;
    u32x4 sx[NQ], sg[WN_TH], sp[WN_TH];
    float bsum[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) bsum[j] = 0.f;

    auto issue = [&](int T) {
        const int n = T / a.per_img, r = T - n * a.per_img;
        const int ty = r / a.tiles_x, tx = r - ty * a.tiles_x;
        const int live = T < a.total;
        // (selects only around the loads: an out-of-range item gets offset 2^31 and is dropped by the buffer unit)
        // (KS = 3: the aligned quad that holds halo column 0 = image column X0 + 3)
        const int Y = ty * WN_TH - HALO + x_row, X0 = tx * WN_TW - 4 + 8 * x_blk;
        const int rowok = live & (int)(tid < XTHREADS) & (int)((unsigned)Y < (unsigned)a.H);
        const int xoff = (((n * 3 + x_ci) * a.H + Y) * a.W + X0) * 4;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int ok = rowok & (int)((unsigned)(X0 + 4 * q) < (unsigned)a.W);
            sx[q] = __builtin_amdgcn_raw_buffer_load_b128(rx, ok ? (unsigned)(xoff + 16 * q) : 0x80000000u, 0, 0);
        }
        const int goff = (((n * a.H + ty * WN_TH) * a.W + tx * WN_TW + p0) * 64 + oct * 8) * 2;
#pragma unroll
        for (int k = 0; k < WN_TH; ++k) {
            const unsigned voff = live ? (unsigned)(goff + k * a.W * 128) : 0x80000000u;
            sg[k] = __builtin_amdgcn_raw_buffer_load_b128(rg, voff, 0, 0);
            if (ACTB) sp[k] = __builtin_amdgcn_raw_buffer_load_b128(rp, voff, 0, 0);
        }
    };
    auto commit = [&]() {
        // x: 16 floats -> 8 bf16 pairs d[i] = (e 2i, e 2i+1) and the 7 odd pairs o[i] = (e 2i+1, e 2i+2); copy s of the
        // block = elements s .. s + 7
        if constexpr (KS == 9) {
            if (tid < XTHREADS) {
                unsigned d[8], o[7];
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    d[i] = pack_bf16x2(__uint_as_float(sx[i >> 1][2 * (i & 1)]), __uint_as_float(sx[i >> 1][2 * (i & 1) + 1]));
#pragma unroll
                for (int i = 0; i < 7; ++i) o[i] = __builtin_amdgcn_alignbit(d[i + 1], d[i], 16);
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    *reinterpret_cast<u32x4*>(xs + x_lds + (2 * m) * 16 * WN_XROW) = u32x4{d[m], d[m + 1], d[m + 2], d[m + 3]};
                    *reinterpret_cast<u32x4*>(xs + x_lds + (2 * m + 1) * 16 * WN_XROW) = u32x4{o[m], o[m + 1], o[m + 2], o[m + 3]};
                }
            }
        } else {
            if (tid < XTHREADS) {
                // halo element e = loaded float e + 3; pairs d[i] = (e 2i, e 2i+1), o[i] = (e 2i+1, e 2i+2); copies 0, 1, 2
                auto f = [&](int e) { return __uint_as_float(sx[(e + 3) >> 2][(e + 3) & 3]); };
                unsigned d[5], o[4];
#pragma unroll
                for (int i = 0; i < 5; ++i) d[i] = pack_bf16x2(f(2 * i), f(2 * i + 1));
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = __builtin_amdgcn_alignbit(d[i + 1], d[i], 16);
                *reinterpret_cast<u32x4*>(xs + x_lds) = u32x4{d[0], d[1], d[2], d[3]};
                *reinterpret_cast<u32x4*>(xs + x_lds + 16 * WN_XROW) = u32x4{o[0], o[1], o[2], o[3]};
                *reinterpret_cast<u32x4*>(xs + x_lds + 2 * 16 * WN_XROW) = u32x4{d[1], d[2], d[3], d[4]};
            }
        }
#pragma unroll
        for (int k = 0; k < WN_TH; ++k) {
            u32x4 v;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float g0 = __uint_as_float(sg[k][j] << 16), g1 = __uint_as_float(sg[k][j] & 0xFFFF0000u);
                if (ACTB) {
                    const float b0 = __uint_as_float(sp[k][j] << 16), b1 = __uint_as_float(sp[k][j] & 0xFFFF0000u);
                    g0 = b0 > 0.f ? g0 : slope * g0;
                    g1 = b1 > 0.f ? g1 : slope * g1;
                }
                bsum[2 * j] += g0;
                bsum[2 * j + 1] += g1;
                v[j] = ACTB ? pack_bf16x2(g0, g1) : sg[k][j];
            }
            *reinterpret_cast<u32x4*>(ds + (k * WN_TW + p0) * WN_PS + oct * 16) = v;
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int mh = 0; mh < 2; ++mh)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[nt][mh][i] = 0.f;

    int T = blockIdx.x;
    issue(T);
    commit();
    __syncthreads();
    for (; T < a.total; T += gridDim.x) {
        issue(T + gridDim.x);
        if constexpr (KS == 9) {
#pragma unroll
            for (int r = 0; r < WN_TH; ++r)
#pragma unroll
                for (int xh = 0; xh < 2; ++xh) {
                    bf16x8 af[2], bf[2];
#pragma unroll
                    for (int mh = 0; mh < 2; ++mh) {
                        const __bf16* p = reinterpret_cast<const __bf16*>(ds + a_base + (r * WN_TW + 16 * xh) * WN_PS + 64 * mh);
                        const s16x4 lo = lds_tr16(p), hi = lds_tr16(p + 2 * WN_PS);      // (+4 pixels; p counts bf16)
                        af[mh] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
                    }
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
                        bf[nt] = *reinterpret_cast<const bf16x8*>(xs + bbase[nt] + r * WN_XROW + xh * 32);
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                        for (int mh = 0; mh < 2; ++mh)
                            acc[nt][mh] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mh], bf[nt], acc[nt][mh], 0, 0, 0);
                }
        } else {
            // wave w: tile rows 2 w, 2 w + 1, all 27 columns
#pragma unroll
            for (int rr = 0; rr < 2; ++rr)
#pragma unroll
                for (int xh = 0; xh < 2; ++xh) {
                    const int r = 2 * wave + rr;
                    bf16x8 af[2];
#pragma unroll
                    for (int mh = 0; mh < 2; ++mh) {
                        const __bf16* p = reinterpret_cast<const __bf16*>(ds + a_base + (r * WN_TW + 16 * xh) * WN_PS + 64 * mh);
                        const s16x4 lo = lds_tr16(p), hi = lds_tr16(p + 2 * WN_PS);
                        af[mh] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
                    }
                    const bf16x8 bf = *reinterpret_cast<const bf16x8*>(xs + bbase[0] + r * WN_XROW + xh * 32);
#pragma unroll
                    for (int mh = 0; mh < 2; ++mh)
                        acc[0][mh] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mh], bf, acc[0][mh], 0, 0, 0);
                }
        }
        __syncthreads();          // every wave has finished reading this tile
        commit();
        __syncthreads();
    }

    // ---- one slab per workgroup: [ky][krow][co]; a lane's register group q = couts 32 mh + 8 q + 4 kk .. +3 ------------
    float* sl = a.slab + (long long)blockIdx.x * a.slab_stride;
    if constexpr (KS == 9) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            if (srow[nt] < 0) continue;
#pragma unroll
            for (int mh = 0; mh < 2; ++mh)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4 v = {acc[nt][mh][4 * q], acc[nt][mh][4 * q + 1], acc[nt][mh][4 * q + 2], acc[nt][mh][4 * q + 3]};
                    if (zrow[nt]) v = f32x4{0.f, 0.f, 0.f, 0.f};
                    *reinterpret_cast<f32x4*>(sl + srow[nt] + 32 * mh + 8 * q + 4 * kk) = v;
                }
        }
    } else {
        // the four waves' partial sums meet in LDS (every tile is done: the loop ended on a barrier) and are added in wave order;
        // slab [ky][krow 12][co 64], rows krow = 9 .. 11 are padding and written as zeros
        float* red = reinterpret_cast<float*>(ds);
#pragma unroll
        for (int mh = 0; mh < 2; ++mh)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<f32x4*>(red + (((wave * 2 + mh) * 4 + q) * 64 + lane) * 4) =
                    f32x4{acc[0][mh][4 * q], acc[0][mh][4 * q + 1], acc[0][mh][4 * q + 2], acc[0][mh][4 * q + 3]};
        __syncthreads();
        for (int e = tid; e < 2 * 4 * 64; e += 256) {
            const int ln = e & 63, q = (e >> 6) & 3, mh = e >> 8;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int w = 0; w < 4; ++w) v += *reinterpret_cast<const f32x4*>(red + (((w * 2 + mh) * 4 + q) * 64 + ln) * 4);
            const int n = ln & 31;
            if (n < 27) {
                const int ky = n / 9, krow = n - 9 * ky;
                *reinterpret_cast<f32x4*>(sl + (ky * 12 + krow) * 64 + 32 * mh + 8 * q + 4 * (ln >> 5)) = v;
            }
        }
        for (int e = tid; e < 3 * 3 * 64; e += 256) sl[((e / 192) * 12 + 9 + (e % 192) / 64) * 64 + (e & 63)] = 0.f;
        __syncthreads();                                   // (the bias partials below reuse the front of the LDS)
    }
    if (a.bias_slab != nullptr) {                       // ... and its bias row: sum of the (activated) gradient per cout
        float* red = reinterpret_cast<float*>(lds);
#pragma unroll
        for (int j = 0; j < 8; ++j) red[tid * 8 + j] = bsum[j];
        __syncthreads();
        if (tid < 64) {
            float s = 0.f;
            for (int p = 0; p < 32; ++p) s += red[(p * 8 + (tid >> 3)) * 8 + (tid & 7)];
            a.bias_slab[(long long)blockIdx.x * a.slab_stride + tid] = s;
        }
    }
}

static int wn_cus() { return sisr_cu_slots(); }

// 9: the generator's first conv, 3: the discriminator's, 0: not taken
static int wn_ks(const SisrWgradDesc* d) {
    const char* sw = getenv("SISR_THIN");                       // A/B switch: SISR_THIN=0 keeps the generic kernel
    if ((sw && sw[0] == '0') || !d) return 0;
    if (d->KH != d->KW || (d->KH != 9 && d->KH != 3) || d->stride != 1 || d->pad_y != d->KH / 2 || d->pad_x != d->KH / 2) return 0;
    const int ks = d->KH;
    if (ks == 3) { const char* s3 = getenv("SISR_THIN3"); if (s3 && s3[0] == '0') return 0; }
    const int krowp = ks == 9 ? WN_KROWP : 12;
    if (d->Cin != 3 || d->Cout != 64 || d->CoutPad != 64 || d->n_chunk != 1 || d->PS != 3 || d->KROWP != krowp) return 0;
    if (d->x_mode != SISR_X_NCHW || d->x_bf16 || d->pro_mode != SISR_PRO_NONE) return 0;
    if (d->g_mode != SISR_X_NHWC || !d->g_bf16 || (d->gpro_mode != SISR_PRO_NONE && d->gpro_mode != SISR_PRO_ACT_BWD)) return 0;
    if (d->Ho != d->H || d->Wo != d->W || (d->H % WN_TH) || (d->W % WN_TW)) return 0;
    if (d->slab_elems != ks * krowp * 64 || (int64_t)d->N * d->H * d->W * 128 >= (1ll << 31)) return 0;
    return ks;
}
extern "C" int sisr_wgrad_thin_eligible(const SisrWgradDesc* d) { return wn_ks(d) != 0; }

static int wn_grid(const SisrWgradDesc* d) {
    const int total = d->N * (d->H / WN_TH) * (d->W / WN_TW), cus = wn_cus();
    const int rounds = (total + cus - 1) / cus;             // equal shares: every workgroup walks `rounds` tiles
    return (total + rounds - 1) / rounds;
}

int sisr_wgrad_thin_slabs(const SisrWgradDesc* d) { return wn_grid(d); }

int sisr_wgrad_thin_launch(const SisrWgradDesc* d, hipStream_t st) {
    constexpr int lds_bytes = WN_XBYTES + WN_DBYTES;
    const bool actb = d->gpro_mode == SISR_PRO_ACT_BWD;
    if (actb && !d->g2) return SISR_E_BADARG;
    const int ks = wn_ks(d);
    if (!ks) return SISR_E_UNSUPPORTED;
    static SisrLdsCap cap_t, cap_f, cap_t3, cap_f3;
    if (int e = sisr_raise_lds_cap(cap_t, reinterpret_cast<const void*>(&wgrad_thin_kernel<true, 9>), lds_bytes)) return e;
    if (int e = sisr_raise_lds_cap(cap_f, reinterpret_cast<const void*>(&wgrad_thin_kernel<false, 9>), lds_bytes)) return e;
    if (int e = sisr_raise_lds_cap(cap_t3, reinterpret_cast<const void*>(&wgrad_thin_kernel<true, 3>), lds_bytes)) return e;
    if (int e = sisr_raise_lds_cap(cap_f3, reinterpret_cast<const void*>(&wgrad_thin_kernel<false, 3>), lds_bytes)) return e;
    WThinArgs a;
    a.x = d->x1; a.g1 = d->g1; a.g2 = d->g2; a.slab = d->slab; a.bias_slab = d->bias_slab;
    a.slope_p = d->gpro_slope_p; a.slope = d->gpro_slope;
    a.N = d->N; a.H = d->H; a.W = d->W;
    a.tiles_x = d->W / WN_TW;
    a.per_img = a.tiles_x * (d->H / WN_TH);
    a.total = a.per_img * d->N;
    a.slab_stride = d->slab_stride;
    const int grid = wn_grid(d);
    if (ks == 9) {
        if (actb) hipLaunchKernelGGL((wgrad_thin_kernel<true, 9>), dim3(grid), dim3(256), lds_bytes, st, a);
        else hipLaunchKernelGGL((wgrad_thin_kernel<false, 9>), dim3(grid), dim3(256), lds_bytes, st, a);
    } else {
        if (actb) hipLaunchKernelGGL((wgrad_thin_kernel<true, 3>), dim3(grid), dim3(256), lds_bytes, st, a);
        else hipLaunchKernelGGL((wgrad_thin_kernel<false, 3>), dim3(grid), dim3(256), lds_bytes, st, a);
    }
    SISR_CHECK_LAUNCH();
    return 0;
}
