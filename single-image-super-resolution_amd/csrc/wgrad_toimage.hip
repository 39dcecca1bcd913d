// wgrad_toimage.hip -- weight gradient of the generator's LAST convolution (model_generator.py:52-53: 3x3, 64 -> 3,
// stride 1, pad 1, + Tanh) in the bf16 build:  dW[co][ci][ky][kx] = sum_p dy(p)[co] * x(p + (ky-1, kx-1))[ci]  with x the
// bf16 NHWC activations (PReLU of the upscale stage as prologue) and dy the NCHW fp32 image gradient (tanh' prologue).
//
// On the generic bf16 kernel (wgrad_bf16.hip) this launch costs 95 us plus a 6 us copy of the gradient into a 4-channel
// NHWC image (profiles/r02_trace_step_order.txt): that kernel shifts the 64-channel operand (nine transposed fragment
// reads per pixel block) against a 3-of-32-column gradient.  Here the sum runs over q = p + tap instead:
//     D[ci 64][n = (ky', kx', co), 27 of 32] = sum_q x(q)[ci] * dy(q + (ky'-1, kx'-1))[co],      ky' = 2 - ky, kx' = 2 - kx
// -- the structure of wgrad_thin.hip with the roles swapped: the 64-channel operand is read UNSHIFTED (transposing
// reads, no halo), the 3-channel gradient carries the shift: its halo lies in LDS as eight copies per channel, copy s
// shifted left by s elements, so a lane's fragment (8 consecutive pixels from any start column) is one aligned 16-byte
// read.  2 MFMAs per 16 pixels instead of 18.
//   * 256 threads, one workgroup per CU, persistent over 8 x 32 pixel tiles (9 per CU at HR 192); wave w contracts tile
//     rows 2w, 2w+1; the four partial 64 x 32 results meet in LDS at the end;
//   * the next tile's operands are requested before the MFMA phase and committed after it;
//   * one slab per workgroup in the generic kernel's layout [chunk 2][tap 9][ci 32][CoutPad 32] + bias row, every entry
//     written (zeros in the padding columns), summed by sisr_slab_reduce_f32: fixed order, bit-reproducible.
// Requirements (sisr_wgrad_toimage_eligible): the geometry above, H % 8 == 0, W % 32 == 0, bf16 NHWC x with prologue
// NONE / ACT, NCHW fp32 gradient with prologue NONE / TANH_BWD, descriptor planned for the 4-channel padded gradient.
#include "sisr_dev.h"

#include <algorithm>
#include <cstdlib>

#include "sisr_bf16_stage.h"

#define WI_TH 8
#define WI_TW 32
#define WI_PS 192                          // LDS bytes per x pixel: 64 bf16 + 64 bytes (bank spread of the transposing reads)
#define WI_XBYTES (WI_TH * WI_TW * WI_PS)  // 49152
#define WI_GROWS (WI_TH + 2)               // gradient halo rows
#define WI_GROW 64                         // bytes per halo row of one (channel, shift) copy: 4 blocks of 8 bf16
#define WI_GBYTES (3 * 8 * WI_GROWS * WI_GROW)   // 15360
#define WI_CP 32                           // CoutPad of the slab layout
#define WI_SLAB (2 * 9 * 32 * WI_CP)

struct WToImageArgs {
    const void* x;
    const float *g1, *g2;
    float *slab, *bias_slab;
    const float* slope_p;
    float slope;
    int N, H, W, act, tanhb;
    int tiles_x, per_img, total;
    long long slab_stride;
};

template <bool ACT, bool TANHB>
__global__ void __launch_bounds__(256, 1) wgrad_toimage_kernel(const WToImageArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char* xs = lds;                   // x tile [pixel][64 ch]
    unsigned char* gs = lds + WI_XBYTES;       // gradient halo copies [co][shift][row][4 blocks]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, kk = lane >> 5, grp = lane >> 4, tq = (lane & 15) >> 2, tp = lane & 3;
    const float slope = a.slope_p ? a.slope_p[0] : a.slope;

    // ---- B column n = (ky', kx', co): halo row r + ky', halo column x + kx' + 3 (halo origin = tile origin - (1, 4)) ------
    int bbase;
    {
        const int nn = l31 < 27 ? l31 : 0;
        const int kyf = nn / 9, kxf = (nn - 9 * kyf) / 3, co = nn - 9 * kyf - 3 * kxf;
        bbase = ((co * 8 + kxf + 3) * WI_GROWS + kyf) * WI_GROW + kk * 16;
    }
    const int a_base = (8 * (grp >> 1) + tq) * WI_PS + (16 * (grp & 1) + 4 * tp) * 2;

    // ---- staging ---------------------------------------------------------------------------------------------------------
    const unsigned plane = (unsigned)(a.H * a.W);
    const __amdgpu_buffer_rsrc_t rg = sisr_rsrc(a.g1, (unsigned)a.N * 3u * plane * 4u),
                                 ry = sisr_rsrc(TANHB ? a.g2 : a.g1, (unsigned)a.N * 3u * plane * 4u);
    const __amdgpu_buffer_rsrc_t rx = sisr_rsrc(a.x, (unsigned)a.N * plane * 128u);
    // gradient: thread < 120 -> (channel, halo row, block of 8 columns); it loads columns 8 blk .. 8 blk + 15
    const int g_co = tid / 40, g_row = (tid - 40 * g_co) / 4, g_blk = tid & 3;
    const int g_lds = ((g_co * 8) * WI_GROWS + g_row) * WI_GROW + g_blk * 16;
    // x: thread -> channels 8 oct .., pixels (row k, column p0), k = 0 .. 7
    const int oct = tid & 7, p0 = tid >> 3;
    u32x4 sg[4], sy[4], sx[WI_TH];
    float bsum = 0.f;

    auto issue = [&](int T) {
        const int n = T / a.per_img, r = T - n * a.per_img;
        const int ty = r / a.tiles_x, tx = r - ty * a.tiles_x;
        const int live = T < a.total;
        // (selects only around the loads: an out-of-range item gets offset 2^31 and is dropped by the buffer unit)
        const int Y = ty * WI_TH - 1 + g_row, X0 = tx * WI_TW - 4 + 8 * g_blk;
        const int rowok = live & (int)(tid < 120) & (int)((unsigned)Y < (unsigned)a.H);
        const int goff = (((n * 3 + g_co) * a.H + Y) * a.W + X0) * 4;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int ok = rowok & (int)((unsigned)(X0 + 4 * q) < (unsigned)a.W);
            const unsigned voff = ok ? (unsigned)(goff + 16 * q) : 0x80000000u;
            sg[q] = __builtin_amdgcn_raw_buffer_load_b128(rg, voff, 0, 0);
            if (TANHB) sy[q] = __builtin_amdgcn_raw_buffer_load_b128(ry, voff, 0, 0);
        }
        const int xoff = (((n * a.H + ty * WI_TH) * a.W + tx * WI_TW + p0) * 64 + oct * 8) * 2;
#pragma unroll
        for (int k = 0; k < WI_TH; ++k)
            sx[k] = __builtin_amdgcn_raw_buffer_load_b128(rx, live ? (unsigned)(xoff + k * a.W * 128) : 0x80000000u, 0, 0);
    };
    auto commit = [&]() {
        if (tid < 120) {
            float e[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float gq = __uint_as_float(sg[i >> 2][i & 3]);
                const float yq = TANHB ? __uint_as_float(sy[i >> 2][i & 3]) : 0.f;
                e[i] = TANHB ? gq * (1.f - yq * yq) : gq;
            }
            // bias partial: the tile's own pixels = halo rows 1 .. 8, halo columns 4 .. 35 (block 3 also owns 32 .. 35)
            if (g_row >= 1 && g_row <= WI_TH) {
#pragma unroll
                for (int i = 0; i < 12; ++i) {
                    const bool mine = i < 8 ? (g_blk > 0 || i >= 4) : g_blk == 3;
                    bsum += mine ? e[i] : 0.f;
                }
            }
            unsigned d[8], o[7];
#pragma unroll
            for (int i = 0; i < 8; ++i) d[i] = pack_bf16x2(e[2 * i], e[2 * i + 1]);
#pragma unroll
            for (int i = 0; i < 7; ++i) o[i] = __builtin_amdgcn_alignbit(d[i + 1], d[i], 16);
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                *reinterpret_cast<u32x4*>(gs + g_lds + (2 * m) * WI_GROWS * WI_GROW) = u32x4{d[m], d[m + 1], d[m + 2], d[m + 3]};
                *reinterpret_cast<u32x4*>(gs + g_lds + (2 * m + 1) * WI_GROWS * WI_GROW) = u32x4{o[m], o[m + 1], o[m + 2], o[m + 3]};
            }
        }
#pragma unroll
        for (int k = 0; k < WI_TH; ++k) {
            u32x4 v = sx[k];
            if (ACT) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float v0 = __uint_as_float(v[j] << 16), v1 = __uint_as_float(v[j] & 0xFFFF0000u);
                    v[j] = pack_bf16x2(v0 > 0.f ? v0 : slope * v0, v1 > 0.f ? v1 : slope * v1);
                }
            }
            *reinterpret_cast<u32x4*>(xs + (k * WI_TW + p0) * WI_PS + oct * 16) = v;
        }
    };

    f32x16 acc[2];
#pragma unroll
    for (int mh = 0; mh < 2; ++mh)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[mh][i] = 0.f;

    int T = blockIdx.x;
    issue(T);
    commit();
    __syncthreads();
    for (; T < a.total; T += gridDim.x) {
        issue(T + gridDim.x);
#pragma unroll
        for (int rr = 0; rr < 2; ++rr)
#pragma unroll
            for (int xh = 0; xh < 2; ++xh) {
                const int r = 2 * wave + rr;
                bf16x8 af[2];
#pragma unroll
                for (int mh = 0; mh < 2; ++mh) {
                    const __bf16* p = reinterpret_cast<const __bf16*>(xs + a_base + (r * WI_TW + 16 * xh) * WI_PS + 64 * mh);
                    const s16x4 lo = lds_tr16(p), hi = lds_tr16(p + 2 * WI_PS);      // (+4 pixels; p counts bf16)
                    af[mh] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
                }
                const bf16x8 bf = *reinterpret_cast<const bf16x8*>(gs + bbase + r * WI_GROW + xh * 32);
#pragma unroll
                for (int mh = 0; mh < 2; ++mh) acc[mh] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mh], bf, acc[mh], 0, 0, 0);
            }
        __syncthreads();          // every wave has finished reading this tile
        commit();
        __syncthreads();
    }

    // ---- the four waves' partial D[ci][n] meet in LDS; then every entry of the slab is written ----------------------------
    float* part = reinterpret_cast<float*>(lds);                  // [wave][ci 64][n 32]
    float* bred = part + 4 * 64 * 32;                              // [120] bias partials
#pragma unroll
    for (int mh = 0; mh < 2; ++mh)
#pragma unroll
        for (int i = 0; i < 16; ++i) part[(wave * 64 + 32 * mh + mfma_row(i, lane)) * 32 + l31] = acc[mh][i];
    if (tid < 120) bred[tid] = bsum;
    __syncthreads();
    float* sl = a.slab + (long long)blockIdx.x * a.slab_stride;
    for (int idx = tid; idx < WI_SLAB; idx += 256) {
        const int co = idx & (WI_CP - 1), row = idx >> 5, cl = row & 31, ct = row >> 5, chunk = ct / 9, tap = ct - 9 * chunk;
        float v = 0.f;
        if (co < 3) {
            const int n = (2 - tap / 3) * 9 + (2 - tap % 3) * 3 + co, ci = chunk * 32 + cl;
#pragma unroll
            for (int w = 0; w < 4; ++w) v += part[(w * 64 + ci) * 32 + n];
        }
        sl[idx] = v;
    }
    if (a.bias_slab != nullptr && tid < WI_CP) {
        float s = 0.f;
        if (tid < 3)
            for (int i = 0; i < 40; ++i) s += bred[tid * 40 + i];
        a.bias_slab[(long long)blockIdx.x * a.slab_stride + tid] = s;
    }
}

static int wi_cus() { return sisr_cu_slots(); }

extern "C" int sisr_wgrad_toimage_eligible(const SisrWgradDesc* d) {
    const char* sw = getenv("SISR_THIN");                       // A/B switch: SISR_THIN=0 keeps the generic kernel
    if ((sw && sw[0] == '0') || !d) return 0;
    if (d->KH != 3 || d->KW != 3 || d->stride != 1 || d->pad_y != 1 || d->pad_x != 1) return 0;
    // (planned by sisr_wgrad_plan_bf16 for the gradient padded to 4 channels: that fixes the slab layout)
    if (d->Cin != 64 || d->Cout != 4 || d->CoutPad != WI_CP || d->n_chunk != 2 || d->CK != 32 || d->slab_elems != WI_SLAB) return 0;
    if (d->x_mode != SISR_X_NHWC || !d->x_bf16 || (d->pro_mode != SISR_PRO_NONE && d->pro_mode != SISR_PRO_ACT)) return 0;
    if (d->g_mode != SISR_X_NCHW || d->g_bf16 || (d->gpro_mode != SISR_PRO_NONE && d->gpro_mode != SISR_PRO_TANH_BWD)) return 0;
    if (d->Ho != d->H || d->Wo != d->W || (d->H % WI_TH) || (d->W % WI_TW)) return 0;
    if ((int64_t)d->N * d->H * d->W * 128 >= (1ll << 31)) return 0;
    return 1;
}

static int wi_grid(const SisrWgradDesc* d) {
    const int total = d->N * (d->H / WI_TH) * (d->W / WI_TW), cus = wi_cus();
    const int rounds = (total + cus - 1) / cus;             // equal shares: every workgroup walks `rounds` tiles
    return (total + rounds - 1) / rounds;
}

int sisr_wgrad_toimage_slabs(const SisrWgradDesc* d) { return wi_grid(d); }

template <bool ACT, bool TANHB>
static int wi_launch(const WToImageArgs& a, int grid, hipStream_t st) {
    constexpr int lds_bytes = WI_XBYTES + WI_GBYTES;
    static SisrLdsCap cap;
    if (int e = sisr_raise_lds_cap(cap, reinterpret_cast<const void*>(&wgrad_toimage_kernel<ACT, TANHB>), lds_bytes)) return e;
    hipLaunchKernelGGL((wgrad_toimage_kernel<ACT, TANHB>), dim3(grid), dim3(256), lds_bytes, st, a);
    SISR_CHECK_LAUNCH();
    return 0;
}

int sisr_wgrad_toimage_launch(const SisrWgradDesc* d, hipStream_t st) {
    const bool act = d->pro_mode == SISR_PRO_ACT, tanhb = d->gpro_mode == SISR_PRO_TANH_BWD;
    if (tanhb && !d->g2) return SISR_E_BADARG;
    WToImageArgs a;
    a.x = d->x1; a.g1 = d->g1; a.g2 = d->g2; a.slab = d->slab; a.bias_slab = d->bias_slab;
    a.slope_p = d->pro_slope_p; a.slope = d->pro_slope;
    a.N = d->N; a.H = d->H; a.W = d->W; a.act = act; a.tanhb = tanhb;
    a.tiles_x = d->W / WI_TW;
    a.per_img = a.tiles_x * (d->H / WI_TH);
    a.total = a.per_img * d->N;
    a.slab_stride = d->slab_stride;
    const int grid = wi_grid(d);
    if (act) return tanhb ? wi_launch<true, true>(a, grid, st) : wi_launch<true, false>(a, grid, st);
    return tanhb ? wi_launch<false, true>(a, grid, st) : wi_launch<false, false>(a, grid, st);
}

// ---- the same layer with fp32 tensors (fp32 parity build): exact v_mfma_f32_32x32x2_f32 -------------------------------------
// D[ci 64][n = (ky', kx', co), 27 of 32] over K = pixels, 2 per MFMA.  An fp32 MFMA operand is one float per lane, so both
// operands are plain 4-byte LDS reads: A[ci][pixel] from the pixel-major x tile (lanes = consecutive channels), B[pixel][n]
// from the planar gradient halo at (row + ky', column + kx' + 3) -- no transposing reads, no shifted copies.  The generic
// fp32 kernel (conv_wgrad.hip) pads the 3 couts to 32 MFMA columns for each of the 9 x 64 K rows: ~340 us; here 590 K
// MFMAs (16 us of matrix time) over 151 MB of activations.  Slab in the generic fp32 layout [chunk][ky][kx * PS + cl][CoutPad].
#define WJ_PS 64                           // floats per x pixel in LDS
#define WJ_XBYTES (WI_TH * WI_TW * WJ_PS * 4)     // 65536
#define WJ_GW 40                           // gradient halo columns (origin = tile origin - 4: 16-byte aligned rows)
#define WJ_GBYTES (3 * WI_GROWS * WJ_GW * 4)      // 4800
#define WJ_XITEMS 16                       // 16-byte x items per thread and tile
#define WJ_GITEMS 5                        // gradient halo elements per thread: 1200 = 4.7 x 256

struct WToImageF32Args {
    const float *x, *g1, *g2;
    float *slab, *bias_slab;
    const float* slope_p;
    float slope;
    int N, H, W;
    int tiles_x, per_img, total;
    int CK, PS, KROWP, CoutPad, slab_elems;
    long long slab_stride;
};

template <bool ACT, bool TANHB>
__global__ void __launch_bounds__(256, 1) wgrad_toimage_f32_kernel(const WToImageF32Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    float* xs = reinterpret_cast<float*>(lds);                     // [pixel 256][64]
    float* gs = reinterpret_cast<float*>(lds + WJ_XBYTES);         // [co 3][row 10][col 40]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, kk = lane >> 5;
    const float slope = a.slope_p ? a.slope_p[0] : a.slope;

    int bbase;
    {
        const int nn = l31 < 27 ? l31 : 0;
        const int kyf = nn / 9, kxf = (nn - 9 * kyf) / 3, co = nn - 9 * kyf - 3 * kxf;
        bbase = (co * WI_GROWS + kyf) * WJ_GW + kxf + 3 + kk;
    }
    const unsigned plane = (unsigned)(a.H * a.W);
    const __amdgpu_buffer_rsrc_t rg = sisr_rsrc(a.g1, (unsigned)a.N * 3u * plane * 4u),
                                 ry = sisr_rsrc(TANHB ? a.g2 : a.g1, (unsigned)a.N * 3u * plane * 4u);
    const __amdgpu_buffer_rsrc_t rx = sisr_rsrc(a.x, (unsigned)a.N * plane * 256u);
    // x: item i of a thread = 16-byte group (tid + 256 i): pixel = group / 16, channels 4 (group % 16) ..
    // gradient halo: element (tid + 256 k) of [co][row][col]
    int g_co[WJ_GITEMS], g_row[WJ_GITEMS], g_col[WJ_GITEMS];
#pragma unroll
    for (int k = 0; k < WJ_GITEMS; ++k) {
        const int idx = tid + 256 * k;
        g_co[k] = idx / (WI_GROWS * WJ_GW);
        const int rem = idx - g_co[k] * (WI_GROWS * WJ_GW);
        g_row[k] = rem / WJ_GW;
        g_col[k] = rem - g_row[k] * WJ_GW;
    }
    f32x4 sx[WJ_XITEMS];
    float sg[WJ_GITEMS], sy[WJ_GITEMS];
    float bsum[3] = {0.f, 0.f, 0.f};

    auto issue = [&](int T) {
        const int n = T / a.per_img, r = T - n * a.per_img;
        const int ty = r / a.tiles_x, tx = r - ty * a.tiles_x;
        const int live = T < a.total;
#pragma unroll
        for (int k = 0; k < WJ_GITEMS; ++k) {
            const int Y = ty * WI_TH - 1 + g_row[k], X = tx * WI_TW - 4 + g_col[k];
            const int ok = live & (int)(g_co[k] < 3) & (int)((unsigned)Y < (unsigned)a.H) & (int)((unsigned)X < (unsigned)a.W);
            const unsigned voff = ok ? (unsigned)((((n * 3 + g_co[k]) * a.H + Y) * a.W + X) * 4) : 0x80000000u;
            sg[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rg, voff, 0, 0));
            if (TANHB) sy[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ry, voff, 0, 0));
        }
        const int origin = ((n * a.H + ty * WI_TH) * a.W + tx * WI_TW) * 256;
#pragma unroll
        for (int i = 0; i < WJ_XITEMS; ++i) {
            const int grp = tid + 256 * i, p = grp >> 4, c4 = grp & 15;
            const unsigned voff = live ? (unsigned)(origin + ((p >> 5) * a.W + (p & 31)) * 256 + c4 * 16) : 0x80000000u;
            sx[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, voff, 0, 0));
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int k = 0; k < WJ_GITEMS; ++k) {
            const float e = TANHB ? sg[k] * (1.f - sy[k] * sy[k]) : sg[k];
            if (g_co[k] < 3) gs[tid + 256 * k] = e;
            // bias partial: the tile's own pixels = halo rows 1 .. 8, halo columns 4 .. 35
            const bool mine = g_row[k] >= 1 && g_row[k] <= WI_TH && g_col[k] >= 4 && g_col[k] < 4 + WI_TW;
#pragma unroll
            for (int c = 0; c < 3; ++c) bsum[c] += (mine && g_co[k] == c) ? e : 0.f;
        }
#pragma unroll
        for (int i = 0; i < WJ_XITEMS; ++i) {
            f32x4 v = sx[i];
            if (ACT) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = v[j] > 0.f ? v[j] : slope * v[j];
            }
            *reinterpret_cast<f32x4*>(xs + (tid + 256 * i) * 4) = v;      // [pixel][64]: group index = pixel * 16 + c4
        }
    };

    f32x16 acc[2];
#pragma unroll
    for (int mh = 0; mh < 2; ++mh)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[mh][i] = 0.f;

    int T = blockIdx.x;
    issue(T);
    commit();
    __syncthreads();
    for (; T < a.total; T += gridDim.x) {
        issue(T + gridDim.x);
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int r = 2 * wave + rr;
#pragma unroll
            for (int sxp = 0; sxp < 16; ++sxp) {                     // pixels (r, 2 sxp + kk)
                const float* xp = xs + (r * WI_TW + 2 * sxp + kk) * WJ_PS + l31;
                const float b = gs[bbase + r * WJ_GW + 2 * sxp];
                acc[0] = mfma32(xp[0], b, acc[0]);
                acc[1] = mfma32(xp[32], b, acc[1]);
            }
        }
        __syncthreads();
        commit();
        __syncthreads();
    }

    float* part = reinterpret_cast<float*>(lds);                  // [wave][ci 64][n 32]
    float* bred = part + 4 * 64 * 32;                              // [256][3] bias partials
#pragma unroll
    for (int mh = 0; mh < 2; ++mh)
#pragma unroll
        for (int i = 0; i < 16; ++i) part[(wave * 64 + 32 * mh + mfma_row(i, lane)) * 32 + l31] = acc[mh][i];
#pragma unroll
    for (int c = 0; c < 3; ++c) bred[tid * 3 + c] = bsum[c];
    __syncthreads();
    float* sl = a.slab + (long long)blockIdx.x * a.slab_stride;
    for (int idx = tid; idx < a.slab_elems; idx += 256) {
        const int co = idx % a.CoutPad, t = idx / a.CoutPad, krow = t % a.KROWP, t2 = t / a.KROWP, ky = t2 % 3, chunk = t2 / 3;
        const int kx = krow / a.PS, cl = krow - kx * a.PS;
        float v = 0.f;
        if (co < 3 && kx < 3 && cl < a.CK) {
            const int n = (2 - ky) * 9 + (2 - kx) * 3 + co, ci = chunk * a.CK + cl;
#pragma unroll
            for (int w = 0; w < 4; ++w) v += part[(w * 64 + ci) * 32 + n];
        }
        sl[idx] = v;
    }
    if (a.bias_slab != nullptr && tid < a.CoutPad) {
        float s = 0.f;
        if (tid < 3)
            for (int i = 0; i < 256; ++i) s += bred[i * 3 + tid];
        a.bias_slab[(long long)blockIdx.x * a.slab_stride + tid] = s;
    }
}

extern "C" int sisr_wgrad_toimage_f32_eligible(const SisrWgradDesc* d) {
    const char* sw = getenv("SISR_THIN");                       // A/B switch: SISR_THIN=0 keeps the generic kernel
    if ((sw && sw[0] == '0') || !d) return 0;
    if (d->KH != 3 || d->KW != 3 || d->stride != 1 || d->pad_y != 1 || d->pad_x != 1) return 0;
    if (d->Cin != 64 || d->Cout != 3 || d->CoutPad < 3 || d->CoutPad > 256 || d->CK < 1 || d->n_chunk * d->CK != 64) return 0;
    if (d->PS < d->CK || d->KROWP < 2 * d->PS + d->CK || d->slab_elems != d->n_chunk * 3 * d->KROWP * d->CoutPad) return 0;
    if (d->x_mode != SISR_X_NHWC || d->x_bf16 || (d->pro_mode != SISR_PRO_NONE && d->pro_mode != SISR_PRO_ACT)) return 0;
    if (d->g_mode != SISR_X_NCHW || d->g_bf16 || (d->gpro_mode != SISR_PRO_NONE && d->gpro_mode != SISR_PRO_TANH_BWD)) return 0;
    if (d->Ho != d->H || d->Wo != d->W || (d->H % WI_TH) || (d->W % WI_TW)) return 0;
    if ((int64_t)d->N * d->H * d->W * 256 >= (1ll << 31)) return 0;
    return 1;
}

template <bool ACT, bool TANHB>
static int wj_launch(const WToImageF32Args& a, int grid, hipStream_t st) {
    constexpr int lds_bytes = WJ_XBYTES + WJ_GBYTES + 256;
    static SisrLdsCap cap;
    if (int e = sisr_raise_lds_cap(cap, reinterpret_cast<const void*>(&wgrad_toimage_f32_kernel<ACT, TANHB>), lds_bytes)) return e;
    hipLaunchKernelGGL((wgrad_toimage_f32_kernel<ACT, TANHB>), dim3(grid), dim3(256), lds_bytes, st, a);
    SISR_CHECK_LAUNCH();
    return 0;
}

int sisr_wgrad_toimage_f32_launch(const SisrWgradDesc* d, hipStream_t st) {
    const bool act = d->pro_mode == SISR_PRO_ACT, tanhb = d->gpro_mode == SISR_PRO_TANH_BWD;
    if (tanhb && !d->g2) return SISR_E_BADARG;
    WToImageF32Args a;
    a.x = d->x1; a.g1 = d->g1; a.g2 = d->g2; a.slab = d->slab; a.bias_slab = d->bias_slab;
    a.slope_p = d->pro_slope_p; a.slope = d->pro_slope;
    a.N = d->N; a.H = d->H; a.W = d->W;
    a.tiles_x = d->W / WI_TW;
    a.per_img = a.tiles_x * (d->H / WI_TH);
    a.total = a.per_img * d->N;
    a.CK = d->CK; a.PS = d->PS; a.KROWP = d->KROWP; a.CoutPad = d->CoutPad; a.slab_elems = d->slab_elems;
    a.slab_stride = d->slab_stride;
    const int grid = wi_grid(d);
    if (act) return tanhb ? wj_launch<true, true>(a, grid, st) : wj_launch<true, false>(a, grid, st);
    return tanhb ? wj_launch<false, true>(a, grid, st) : wj_launch<false, false>(a, grid, st);
}
