// conv_thin.hip -- KS x KS convolution (KS = 9, 3) of a 3-channel NCHW fp32 image into 64 bf16 NHWC channels, bf16 build.
//
// Two launches of a generator step have this shape: the first convolution (model_generator.py:32, 9x9, 3 -> 64 at LR
// size) and the data gradient of the last one (model_generator.py:52, 3x3, 64 -> 3 at HR size: a 3 -> 64 convolution of
// the image gradient, tanh' as its prologue, with the flipped weights).  On the generic fp32-storage kernel
// (conv_fwd.hip) they cost 72 us and 75 us (profiles/r02_trace_step_order.txt) for 4.6 and 2 GFLOP: that kernel's K
// loop walks the taps of a 3-channel chunk one row of taps at a time in 128-pixel workgroups, and its epilogue stores
// 4 bytes per lane.  Here (written for KS = 9; KS = 3 is the same with one K slice per tap row):
//   * K is re-ordered per tap row ky as kappa = 4 kx + c over 12 (9 real) taps x 4 (3 real) channels = 3 MFMA K slices
//     of 16: the LDS halo image stores 4 bf16 per pixel, so the operand fragment of a lane -- 2 neighbouring pixels x
//     4 channels -- is 16 contiguous bytes;
//   * A = weights (M = 32 couts), B = pixels (N = 32 pixels: 2 rows x 16 columns): a lane's accumulator registers are
//     groups of 4 consecutive couts of ONE pixel, and one v_permlane32_swap per register pair turns them into 16-byte
//     NHWC stores -- no transposition through LDS;
//   * 256 threads, one workgroup per CU, persistent over 16 x 16 pixel tiles; wave (h, g) = couts 32h.., tile rows
//     8g..8g+7 as 4 accumulators; its 27 weight fragments (108 VGPRs) are loaded once per launch;
//   * the fragment of (halo row hh, K slice j) serves every (sub-tile mt, tap row ky) with 2 mt + ky = hh: 45 LDS reads
//     feed 108 MFMAs per wave and tile;
//   * the 3-channel halo (<= 9 dwords per thread) of tile T + 2 is requested after the MFMA phase of tile T and committed
//     to the other LDS buffer after the MFMA phase of tile T + 1: one barrier per tile, memory latency never waited for.
// Requirements (sisr_conv2d_thin_eligible): NCHW fp32 input with Cin = 3, Cout = CoutPad = 64, 9x9 / 3x3, stride 1,
// "same" padding, bf16 NHWC output, prologue NONE or TANH_BWD, no residual / statistics / activation epilogue,
// H % 16 == W % 16 == 0.
#include "sisr_dev.h"

#include <algorithm>
#include <cstdlib>

#include "sisr_bf16_stage.h"

#define TN_T 16                          // tile edge (output pixels)

template <int KS> struct ThinGeom {
    static constexpr int NJ = (KS * 4 + 15) / 16;          // MFMA K slices (16) per tap row: taps padded to 4 NJ, channels to 4
    static constexpr int IH = TN_T + KS - 1;               // halo rows = halo columns that hold data
    static constexpr int IW = TN_T + 4 * NJ;               // ... + the zero columns the padded taps read (zero weights, finite data)
    static constexpr int HALO = IH * IW * 8;               // bytes (4 bf16 per pixel)
    static constexpr int ITEMS = (IH * IH + 255) / 256;    // halo pixels per thread
    static constexpr int KROWP = (KS * 3 + 3) / 4 * 4;     // packed weight row: KS taps x 3 channels, rounded up to 4 (conv_fwd.hip plan)
    static constexpr int WBYTES = KS * 64 * KROWP * 4;
    static constexpr int LDS = WBYTES + 64 > 2 * HALO ? WBYTES + 64 : 2 * HALO;
};

struct ThinArgs {
    const float *x1, *x2, *wpk, *bias;
    void* y;
    int N, H, W;
    int tiles_x, per_img, total, rounds;
};

template <int KS, bool TANHB>
__global__ void __launch_bounds__(256, 1) conv_thin_kernel(const ThinArgs a) {
    typedef ThinGeom<KS> G;
    constexpr int NJ = G::NJ, IH = G::IH, IW = G::IW, HALO = G::HALO, ITEMS = G::ITEMS, KROWP = G::KROWP, PAD = (KS - 1) / 2;
    __shared__ __attribute__((aligned(16))) unsigned char lds[G::LDS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, kk = lane >> 5, h = wave & 1, g = wave >> 1;
    const int prow = l31 >> 4, px = l31 & 15;

    // ---- weights: packed fp32 image [ky][cout][kx * 3 + c] -> LDS (coalesced) -> KS * NJ bf16 A fragments per lane ------
    {
        const f32x4* src = reinterpret_cast<const f32x4*>(a.wpk);
        f32x4* dst = reinterpret_cast<f32x4*>(lds);
        for (int i = tid; i < G::WBYTES / 16; i += 256) dst[i] = src[i];
    }
    __syncthreads();
    bf16x8 wf[KS][NJ];
    {
        // fragment (ky, j) of a lane = taps kx = 4j + 2kk, +1 of its cout: 6 consecutive floats of the packed row (the
        // reads of taps >= KS run into the next row -- or the 64 spare bytes behind the image -- and are discarded)
        const unsigned char* wl = lds + ((32 * h + l31) * KROWP + 6 * kk) * 4;
#pragma unroll
        for (int ky = 0; ky < KS; ++ky)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const unsigned char* p = wl + (ky * 64 * KROWP + 12 * j) * 4;
                const f32x2 w0 = *reinterpret_cast<const f32x2*>(p), w1 = *reinterpret_cast<const f32x2*>(p + 8),
                            w2 = *reinterpret_cast<const f32x2*>(p + 16);
                const bool t0 = 4 * j + 2 * kk < KS, t1 = 4 * j + 2 * kk + 1 < KS;
                const u32x4 f = {t0 ? pack_bf16x2(w0[0], w0[1]) : 0u, t0 ? pack_bf16x2(w1[0], 0.f) : 0u,
                                 t1 ? pack_bf16x2(w1[1], w2[0]) : 0u, t1 ? pack_bf16x2(w2[1], 0.f) : 0u};
                wf[ky][j] = __builtin_bit_cast(bf16x8, f);
            }
    }
    float bv[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) bv[i] = 0.f;
    if (a.bias != nullptr) {
#pragma unroll
        for (int i = 0; i < 16; ++i) bv[i] = a.bias[32 * h + mfma_row(i, lane)];
    }
    __syncthreads();
    // the pad columns of both halo buffers are written once
    for (int i = tid; i < 2 * HALO / 8; i += 256) reinterpret_cast<u32x2*>(lds)[i] = u32x2{0u, 0u};
    __syncthreads();

    // ---- staging: thread -> halo pixels tid, tid + 256, .. (< IH * IH) -------------------------------------------------
    const unsigned plane = (unsigned)(a.H * a.W);
    const unsigned xbytes = (unsigned)a.N * 3u * plane * 4u;
    const __amdgpu_buffer_rsrc_t rx = sisr_rsrc(a.x1, xbytes), rx2 = sisr_rsrc(TANHB ? a.x2 : a.x1, xbytes);
    int hy[ITEMS], hx[ITEMS], rel[ITEMS];
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
        const int id = tid + 256 * k;
        hy[k] = id / IH;
        hx[k] = id - hy[k] * IH;
        rel[k] = ((hy[k] - PAD) * a.W + hx[k] - PAD) * 4;      // byte offset from the tile's first pixel, within a plane
    }
    // a workgroup walks a CONTIGUOUS run of tiles (neighbours along x): the 128-byte lines of the planar image that two
    // neighbouring halos share are then fetched by one XCD's L2 once, not by two XCDs at the same moment
    const int t_first = blockIdx.x * a.rounds, t_end = min(t_first + a.rounds, a.total);
    float sv[ITEMS][3], sw[ITEMS][3];
    auto issue = [&](int T) {
        const int n = T / a.per_img, r = T - n * a.per_img;
        const int ty = r / a.tiles_x, tx = r - ty * a.tiles_x;
        const int origin = ((n * 3 * a.H + ty * TN_T) * a.W + tx * TN_T) * 4;
        const int live = T < t_end;
#pragma unroll
        for (int k = 0; k < ITEMS; ++k) {
            const int y = ty * TN_T - PAD + hy[k], x = tx * TN_T - PAD + hx[k];
            // (no control flow around the loads: selects only, out-of-range = offset 2^31 = dropped by the buffer unit)
            const int ok = live & (int)(hy[k] < IH) & (int)((unsigned)y < (unsigned)a.H) & (int)((unsigned)x < (unsigned)a.W);
            const unsigned base = (unsigned)(origin + rel[k]);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const unsigned voff = ok ? base + (unsigned)c * plane * 4u : 0x80000000u;
                sv[k][c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, voff, 0, 0));
                if (TANHB) sw[k][c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx2, voff, 0, 0));
            }
        }
    };
    auto commit = [&](unsigned char* buf) {
#pragma unroll
        for (int k = 0; k < ITEMS; ++k) {
            float v[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) v[c] = TANHB ? sv[k][c] * (1.f - sw[k][c] * sw[k][c]) : sv[k][c];
            const u32x2 o = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], 0.f)};
            if (hy[k] < IH) *reinterpret_cast<u32x2*>(buf + (hy[k] * IW + hx[k]) * 8) = o;
        }
    };

    const unsigned ybytes = (unsigned)a.N * plane * 128u;
    const __amdgpu_buffer_rsrc_t ry = sisr_rsrc(a.y, ybytes);
    // B fragment of (halo row hh, K slice j): pixels (8g + hh + prow, px + 4j + 2kk .. +1)
    const int b_base = ((8 * g + prow) * IW + px + 2 * kk) * 8;

    int T = t_first;
    issue(T);
    commit(lds);
    __syncthreads();
    issue(T + 1);
    int cur = 0;
    // per tile: MFMA phase | commit the next tile's halo (requested a tile ago) | request the one after | stores | barrier.
    // (Requests sit BEFORE the stores: the staging registers double as store operands, and a load into a register a
    // pending store still reads has to wait for that store -- this way the stores have a whole MFMA phase to drain.)
    for (; T < t_end; ++T, cur ^= 1) {
        f32x16 acc[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mt][i] = bv[i];
        const unsigned char* ib = lds + cur * HALO + b_base;
#pragma unroll
        for (int hh = 0; hh < 6 + KS; ++hh)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const unsigned char* p = ib + (hh * IW + 4 * j) * 8;
                const u32x2 lo = *reinterpret_cast<const u32x2*>(p), hi = *reinterpret_cast<const u32x2*>(p + 8);
                const bf16x8 f = __builtin_bit_cast(bf16x8, u32x4{lo[0], lo[1], hi[0], hi[1]});
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    const int ky = hh - 2 * mt;
                    if (ky < 0 || ky >= KS) continue;
                    acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ky][j], f, acc[mt], 0, 0, 0);
                }
            }
        __builtin_amdgcn_sched_barrier(0);        // (keeps the commit's conversions -- and their wait for the loads -- below the MFMAs)
        commit(lds + (cur ^ 1) * HALO);
        issue(T + 2);
        // ---- epilogue: registers 4q..4q+3 of a lane = couts 8q + 4kk .. +3 of its pixel; swapping the halves of register
        // groups (q, q + 1) between lanes l and l + 32 leaves 8 consecutive couts (16 bytes) in every lane -----------------
        const int n = T / a.per_img, r = T - n * a.per_img;
        const int ty = r / a.tiles_x, tx = r - ty * a.tiles_x;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int oy = ty * TN_T + 8 * g + 2 * mt + prow, ox = tx * TN_T + px;
            const unsigned pix = ((unsigned)(n * a.H + oy) * (unsigned)a.W + (unsigned)ox) * 128u;
#pragma unroll
            for (int qp = 0; qp < 2; ++qp) {
                const int q = 2 * qp;
                const unsigned p00 = pack_bf16x2(acc[mt][4 * q], acc[mt][4 * q + 1]), p01 = pack_bf16x2(acc[mt][4 * q + 2], acc[mt][4 * q + 3]);
                const unsigned p10 = pack_bf16x2(acc[mt][4 * q + 4], acc[mt][4 * q + 5]), p11 = pack_bf16x2(acc[mt][4 * q + 6], acc[mt][4 * q + 7]);
                const auto s0 = __builtin_amdgcn_permlane32_swap(p00, p10, false, false);
                const auto s1 = __builtin_amdgcn_permlane32_swap(p01, p11, false, false);
                const u32x4 o = {s0[0], s1[0], s0[1], s1[1]};
                __builtin_amdgcn_raw_buffer_store_b128(o, ry, pix + (unsigned)(32 * h + 16 * qp + 8 * kk) * 2u, 0, 0);
            }
        }
        __syncthreads();
    }
}

extern "C" int sisr_conv2d_thin_eligible(const SisrConvDesc* d) {
    const char* sw = getenv("SISR_THIN");                       // A/B switch: SISR_THIN=0 keeps the generic kernel
    if ((sw && sw[0] == '0') || !d) return 0;
    if (d->x_mode != SISR_X_NCHW || d->x_bf16 || d->Cin != 3 || d->Cout != 64 || d->plan.CoutPad != 64) return 0;
    if ((d->KH != 9 && d->KH != 3) || d->KW != d->KH || d->stride != 1 || d->pad_y != (d->KH - 1) / 2 || d->pad_x != d->pad_y) return 0;
    if (d->pro_mode != SISR_PRO_NONE && d->pro_mode != SISR_PRO_TANH_BWD) return 0;
    if (d->y_mode != SISR_Y_NHWC || !d->y_bf16 || d->res || d->stat_part || d->bnb_part || d->epi_act != SISR_EPI_NONE) return 0;
    if (d->y_sy != 1 || d->y_sx != 1 || d->y_oy != 0 || d->y_ox != 0 || d->y_H != d->Ho || d->y_W != d->Wo) return 0;
    if (d->Ho != d->H || d->Wo != d->W || (d->H % TN_T) || (d->W % TN_T)) return 0;
    if (d->plan.n_chunk != 1 || d->plan.PS != 3 || d->plan.KROWP != (d->KH == 9 ? ThinGeom<9>::KROWP : ThinGeom<3>::KROWP)) return 0;
    if ((int64_t)d->N * d->H * d->W * 128 >= (1ll << 31)) return 0;
    return 1;
}

int sisr_conv2d_thin_launch(const SisrConvDesc* d, hipStream_t st) {
    const int cus = sisr_cu_slots();
    const bool tanhb = d->pro_mode == SISR_PRO_TANH_BWD;
    if (tanhb && !d->x2) return SISR_E_BADARG;
    ThinArgs a;
    a.x1 = d->x1; a.x2 = d->x2; a.wpk = d->wpk; a.bias = d->bias; a.y = d->y;
    a.N = d->N; a.H = d->H; a.W = d->W;
    a.tiles_x = d->W / TN_T;
    a.per_img = a.tiles_x * (d->H / TN_T);
    a.total = a.per_img * d->N;
    // equal shares: every workgroup walks ceil(total / cus) tiles
    const int rounds = (a.total + cus - 1) / cus;
    a.rounds = rounds;
    const dim3 grid((a.total + rounds - 1) / rounds), block(256);
    if (d->KH == 9) {
        if (tanhb) hipLaunchKernelGGL((conv_thin_kernel<9, true>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((conv_thin_kernel<9, false>), grid, block, 0, st, a);
    } else {
        if (tanhb) hipLaunchKernelGGL((conv_thin_kernel<3, true>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((conv_thin_kernel<3, false>), grid, block, 0, st, a);
    }
    SISR_CHECK_LAUNCH();
    return 0;
}
