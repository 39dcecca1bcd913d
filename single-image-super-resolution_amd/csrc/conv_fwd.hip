// conv_fwd.hip -- im2col-free direct convolution on gfx950, fp32 storage, exact-fp32 MFMA.
//
// One 256-thread workgroup (4 wavefronts) computes a tile of BM = 4*MSUB*32 output pixels x
// BN = NSUB*32 output channels.  The input halo tile of one channel chunk lives in LDS as
// [pixel][PS] with PS = CK|1 (odd), so the K index of one filter row (s, ci) -> s*PS + ci is
// CONTIGUOUS in LDS: the implicit-GEMM A operand of output pixel m at K index k is simply
// lds_in[base(m) + r*IW*PS + k] -- no im2col buffer, one LDS word per lane per MFMA, and lanes on
// consecutive pixels hit distinct banks (odd stride).  Weights are pre-packed per (chunk, filter
// row) as [cout][KROWP] by sisr_weights_prepare (spectral.hip) with zeros in the pad slots.
// BatchNorm-apply / PReLU / LeakyReLU of the PRODUCER layer are applied while staging the tile
// (prologue), bias / tanh / PixelShuffle / residual and the BatchNorm batch statistics of THIS
// layer's output in the epilogue, so an activation tensor crosses HBM once per consumer.
//
// The same kernel is the data-gradient (flipped, transposed packed weights; BatchNorm-backward
// prologue) -- replaces nn.Conv2d fwd/dgrad at model_generator.py:10,13,33,39,45,52,123,
// model_discriminator.py:10,39 and the VGG19 convs behind model_content_extractor.py:43.
#include "sisr_dev.h"

#include <algorithm>
#include <cstring>

template <int MSUB, int NSUB, int KU>
__device__ __forceinline__ void conv_kblock(const float* const (&ap)[MSUB], const float* const (&bp)[NSUB], int k0,
                                            f32x16 (&acc)[MSUB][NSUB]) {
    float a[KU][MSUB], b[KU][NSUB];
#pragma unroll
    for (int u = 0; u < KU; ++u) {
#pragma unroll
        for (int ms = 0; ms < MSUB; ++ms) a[u][ms] = ap[ms][k0 + 2 * u];
#pragma unroll
        for (int ns = 0; ns < NSUB; ++ns) b[u][ns] = bp[ns][k0 + 2 * u];
    }
#pragma unroll
    for (int u = 0; u < KU; ++u)
#pragma unroll
        for (int ms = 0; ms < MSUB; ++ms)
#pragma unroll
            for (int ns = 0; ns < NSUB; ++ns) acc[ms][ns] = mfma32(a[u][ms], b[u][ns], acc[ms][ns]);
    // pin the schedule: all LDS reads of the block first, then the MFMAs (counted lgkmcnt waits)
    __builtin_amdgcn_sched_group_barrier(0x100, KU * (MSUB + NSUB), 0);
    __builtin_amdgcn_sched_group_barrier(0x008, KU * MSUB * NSUB, 0);
}

// TAG only changes the kernel SYMBOL: TAG 1 = the 3x3, 64->64, stride-1 trunk convolution of G (fwd and
// dgrad, 66 identical launches per training step), so that profiler per-kernel averages refer to one shape.
template <int MSUB, int NSUB, int TAG>
__global__ void __launch_bounds__(SISR_BLOCK, MSUB == 1 ? 3 : 2) conv_mfma_f32_kernel(const SisrConvDesc d) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const SisrConvPlan& p = d.plan;
    constexpr int BM = 4 * MSUB * 32, BN = NSUB * 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, kk = lane >> 5;
    const int S = d.stride;
    const int IH = (p.TH - 1) * S + d.KH, IW = (p.TW - 1) * S + d.KW;
    const int in_elems = p.TN * IH * IW * p.PS;
    const int WSTR = p.KROWP + 1;

    int* row_off = reinterpret_cast<int*>(smem);
    float* lds_in = smem + BM;
    float* lds_w = lds_in + ((in_elems + 8 + 3) & ~3);

    int t = blockIdx.x;
    const int txi = t % p.tiles_x;
    t /= p.tiles_x;
    const int tyi = t % p.tiles_y, ng = t / p.tiles_y;
    const int n0 = ng * p.TN, oy0 = tyi * p.TH, ox0 = txi * p.TW;
    const int cout_base = blockIdx.y * BN;
    const int thw = p.TH * p.TW, tile_rows = p.TN * thw;

    // ---- row table: output element offset of tile row m (or -1) ---------------------------------
    for (int m = tid; m < BM; m += SISR_BLOCK) {
        int off = -1;
        if (m < tile_rows) {
            const int tn = m / thw, rem = m - tn * thw;
            const int ty = rem / p.TW, tx = rem - ty * p.TW;
            const int n = n0 + tn, oy = oy0 + ty, ox = ox0 + tx;
            if (n < d.N && oy < d.Ho && ox < d.Wo) {
                const int py = oy * d.y_sy + d.y_oy, px = ox * d.y_sx + d.y_ox;
                if (d.y_mode == SISR_Y_NHWC)
                    off = ((n * d.y_H + py) * d.y_W + px) * d.Cout;
                else if (d.y_mode == SISR_Y_NCHW)
                    off = n * d.Cout * d.y_H * d.y_W + py * d.y_W + px;
                else
                    off = ((n * 2 * d.Ho + 2 * oy) * (2 * d.Wo) + 2 * ox) * (d.Cout >> 2);
            }
        }
        row_off[m] = off;
    }

    int a_base[MSUB];
#pragma unroll
    for (int ms = 0; ms < MSUB; ++ms) {
        const int m = wave * (MSUB * 32) + ms * 32 + l31;
        a_base[ms] = 0;
        if (m < tile_rows) {
            const int tn = m / thw, rem = m - tn * thw;
            const int ty = rem / p.TW, tx = rem - ty * p.TW;
            a_base[ms] = ((tn * IH + ty * S) * IW + tx * S) * p.PS;
        }
    }

    f32x16 acc[MSUB][NSUB];
#pragma unroll
    for (int ms = 0; ms < MSUB; ++ms)
#pragma unroll
        for (int ns = 0; ns < NSUB; ++ns)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[ms][ns][i] = 0.f;

    OperandView ov;
    ov.x1 = d.x1; ov.x2 = d.x2; ov.pa = d.pa; ov.pb = d.pb; ov.pd = d.pd; ov.ps = d.ps; ov.pt = d.pt;
    ov.N = d.N; ov.H = d.H; ov.W = d.W; ov.C = d.Cin;
    ov.mode = d.x_mode; ov.pro = d.pro_mode;
    ov.slope = d.pro_slope_p ? d.pro_slope_p[0] : d.pro_slope;
    ov.bf16 = d.x_bf16;
    const bool vec_ok = (d.x_mode != SISR_X_NCHW) && !(p.CK & 3) && !((p.CK >> 2) & ((p.CK >> 2) - 1)) && !(d.Cin & 3) &&
                        !(d.x_mode == SISR_X_NHWC_UNSHUFFLE2 && ((d.Cin >> 2) & 3));
    const int iy_org = oy0 * S - d.pad_y, ix_org = ox0 * S - d.pad_x;
    const int kr4 = p.KROWP >> 2;
    int wlg = 0;
    while ((1 << wlg) < kr4) ++wlg;

    for (int chunk = 0; chunk < p.n_chunk; ++chunk) {
        __syncthreads();   // all MFMA reads of the previous chunk are done
        stage_operand_tile(ov, lds_in, p.PS, p.CK, chunk * p.CK, p.TN, IH, IW, n0, iy_org, ix_org, vec_ok,
                           1 << 30, 8);
        for (int r = 0; r < d.KH; ++r) {
            if (r > 0) __syncthreads();   // reads of the previous filter row's weights are done
            {   // packed weights of (chunk, r): [BN][KROWP] contiguous -> LDS [BN][WSTR]; 2^wlg lanes per row
                const f32x4* src = reinterpret_cast<const f32x4*>(
                    d.wpk + ((int64_t)(chunk * d.KH + r) * p.CoutPad + cout_base) * p.KROWP);
                const int k4 = tid & ((1 << wlg) - 1);
                if (k4 < kr4) {
                    constexpr int WB = 8;                  // loads in flight per thread
                    const int jstep = SISR_BLOCK >> wlg;
                    for (int j0 = tid >> wlg; j0 < BN; j0 += WB * jstep) {
                        f32x4 wv[WB];
#pragma unroll
                        for (int u = 0; u < WB; ++u)
                            if (j0 + u * jstep < BN) wv[u] = src[(j0 + u * jstep) * kr4 + k4];
#pragma unroll
                        for (int u = 0; u < WB; ++u) {
                            const int j = j0 + u * jstep;
                            if (j < BN) {
                                float* dst = lds_w + j * WSTR + k4 * 4;
                                dst[0] = wv[u][0]; dst[1] = wv[u][1]; dst[2] = wv[u][2]; dst[3] = wv[u][3];
                            }
                        }
                    }
                }
            }
            __syncthreads();
            const float* ap[MSUB];
            const float* bp[NSUB];
#pragma unroll
            for (int ms = 0; ms < MSUB; ++ms) ap[ms] = lds_in + a_base[ms] + r * IW * p.PS + kk;
#pragma unroll
            for (int ns = 0; ns < NSUB; ++ns) bp[ns] = lds_w + (ns * 32 + l31) * WSTR + kk;
            // K loop in register blocks: all LDS reads of KU K-steps are issued, then their MFMAs (the
            // compiler emits counted lgkmcnt waits), so LDS latency is exposed once per 4*KU MFMAs.
            int k0 = 0;
            for (; k0 + 8 <= p.KROWP; k0 += 8) conv_kblock<MSUB, NSUB, 4>(ap, bp, k0, acc);
            for (; k0 < p.KROWP; k0 += 4) conv_kblock<MSUB, NSUB, 2>(ap, bp, k0, acc);   // KROWP % 4 == 0
        }
    }
    __syncthreads();   // LDS (weights region) is reused as reduction scratch below

    // ---- epilogue ---------------------------------------------------------------------------------
    int col_off[NSUB];
    bool col_ok[NSUB];
#pragma unroll
    for (int ns = 0; ns < NSUB; ++ns) {
        const int cp = cout_base + ns * 32 + l31;   // packed channel order
        col_ok[ns] = cp < d.Cout;
        int co = cp;
        col_off[ns] = cp;
        if (d.y_mode == SISR_Y_NHWC_SHUFFLE2) {
            const int Cq = d.Cout >> 2;
            const int ij = cp / Cq, c = cp - ij * Cq;
            co = c * 4 + ij;
            col_off[ns] = ((ij >> 1) * (2 * d.Wo) + (ij & 1)) * Cq + c;
        } else if (d.y_mode == SISR_Y_NCHW) {
            col_off[ns] = cp * d.y_H * d.y_W;
        }
        const float bv = (d.bias != nullptr && col_ok[ns]) ? d.bias[co] : 0.f;
#pragma unroll
        for (int ms = 0; ms < MSUB; ++ms)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[ms][ns][i] += bv;
    }

    if (d.stat_part != nullptr) {
        // per-tile, per-channel (mean, M2) of the biased conv output over the tile's valid pixels
        float* red = lds_w;             // [4][BN]
        float* meanb = lds_w + 4 * BN;  // [BN]
        const int vn = min(p.TN, d.N - n0), vh = min(p.TH, d.Ho - oy0), vw = min(p.TW, d.Wo - ox0);
        const float cnt = (float)(vn * vh * vw);
        float s[NSUB];
#pragma unroll
        for (int ns = 0; ns < NSUB; ++ns) s[ns] = 0.f;
#pragma unroll
        for (int ms = 0; ms < MSUB; ++ms)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = wave * (MSUB * 32) + ms * 32 + mfma_row(i, lane);
                if (row_off[row] >= 0) {
#pragma unroll
                    for (int ns = 0; ns < NSUB; ++ns) s[ns] += acc[ms][ns][i];
                }
            }
#pragma unroll
        for (int ns = 0; ns < NSUB; ++ns) {
            s[ns] += __shfl_xor(s[ns], 32);
            if (kk == 0) red[wave * BN + ns * 32 + l31] = s[ns];
        }
        __syncthreads();
        if (tid < BN) meanb[tid] = (red[tid] + red[BN + tid] + red[2 * BN + tid] + red[3 * BN + tid]) / cnt;
        __syncthreads();
#pragma unroll
        for (int ns = 0; ns < NSUB; ++ns) {
            const float mu = meanb[ns * 32 + l31];
            float q = 0.f;
#pragma unroll
            for (int ms = 0; ms < MSUB; ++ms)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int row = wave * (MSUB * 32) + ms * 32 + mfma_row(i, lane);
                    if (row_off[row] >= 0) {
                        const float dv = acc[ms][ns][i] - mu;
                        q += dv * dv;
                    }
                }
            s[ns] = q + __shfl_xor(q, 32);
        }
        __syncthreads();   // everyone has read meanb/red of phase 1
#pragma unroll
        for (int ns = 0; ns < NSUB; ++ns)
            if (kk == 0) red[wave * BN + ns * 32 + l31] = s[ns];
        __syncthreads();
        if (tid < BN && cout_base + tid < d.Cout) {
            const float m2 = red[tid] + red[BN + tid] + red[2 * BN + tid] + red[3 * BN + tid];
            float* sp = d.stat_part + (int64_t)blockIdx.x * 2 * d.Cout + cout_base + tid;
            sp[0] = meanb[tid];
            sp[d.Cout] = m2;
        }
        if (tid == 0 && blockIdx.y == 0) d.cnt_part[blockIdx.x] = cnt;
    }

    // output (and residual) through raw buffer accesses with 32-bit byte offsets: rows outside the image carry the
    // out-of-range marker 2^31 (tensors are < 2 GB), so marker + channel offset is dropped by the hardware
    unsigned rb[MSUB][16];
#pragma unroll
    for (int ms = 0; ms < MSUB; ++ms)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int o = row_off[wave * (MSUB * 32) + ms * 32 + mfma_row(i, lane)];
            rb[ms][i] = o >= 0 ? (unsigned)o * 4u : 0x80000000u;
        }
    if (d.y_bf16) {
        // bf16 output tensor (edge layers of the bf16 build: the 3-channel convs keep the exact-fp32 contraction but
        // hand a bf16 NHWC tensor to the bf16 kernels): 2-byte stores, no residual / tanh (checked by the launcher)
        const unsigned yb2 = (unsigned)max(d.N * d.y_H * d.y_W, d.N * d.Ho * d.Wo) * (unsigned)d.Cout * 2u;
        const __amdgpu_buffer_rsrc_t ry2 = sisr_rsrc(d.y, yb2);
#pragma unroll
        for (int ns = 0; ns < NSUB; ++ns)
            if (col_ok[ns]) {
#pragma unroll
                for (int ms = 0; ms < MSUB; ++ms)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const float val = acc[ms][ns][i];
                        __builtin_amdgcn_raw_buffer_store_b16((short)f32_to_bf16_bits(val), ry2, (rb[ms][i] >> 1) + (rb[ms][i] >> 31 ? 0x80000000u : 0u) + (unsigned)col_off[ns] * 2u, 0, 0);
                    }
            }
        return;
    }
    const unsigned ybytes = (unsigned)max(d.N * d.y_H * d.y_W, d.N * d.Ho * d.Wo) * (unsigned)d.Cout * 4u;
    if (d.res != nullptr) {
        const __amdgpu_buffer_rsrc_t rr = sisr_rsrc(d.res, ybytes);
#pragma unroll
        for (int ns = 0; ns < NSUB; ++ns)
            if (col_ok[ns]) {
#pragma unroll
                for (int ms = 0; ms < MSUB; ++ms)
#pragma unroll
                    for (int i = 0; i < 16; ++i)
                        acc[ms][ns][i] += __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rr, rb[ms][i] + (unsigned)col_off[ns] * 4u, 0, 0));
            }
    }
    if (d.epi_act == SISR_EPI_TANH) {
#pragma unroll
        for (int ms = 0; ms < MSUB; ++ms)
#pragma unroll
            for (int ns = 0; ns < NSUB; ++ns)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[ms][ns][i] = tanhf(acc[ms][ns][i]);
    }
    const __amdgpu_buffer_rsrc_t ry = sisr_rsrc(d.y, ybytes);
#pragma unroll
    for (int ns = 0; ns < NSUB; ++ns)
        if (col_ok[ns]) {
#pragma unroll
            for (int ms = 0; ms < MSUB; ++ms)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float val = acc[ms][ns][i];      // (scalar temporary: see conv_bf16.hip)
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(val), ry, rb[ms][i] + (unsigned)col_off[ns] * 4u, 0, 0);
                }
        }
}

// ------------------------------------------------------------------------------------------------
// host: planner + launcher
// ------------------------------------------------------------------------------------------------
static inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

static int conv_lds_bytes(int BM, int TN, int TH, int TW, int S, int KH, int KW, int PS, int KROWP,
                          int BN) {
    const int IH = (TH - 1) * S + KH, IW = (TW - 1) * S + KW;
    const int in_elems = TN * IH * IW * PS;
    const int w_elems = std::max(BN * (KROWP + 1), 5 * BN);
    return (BM + ((in_elems + 8 + 3) & ~3) + w_elems + 4) * 4;
}

extern "C" int sisr_conv2d_plan(SisrConvDesc* d) {
    if (!d || d->N <= 0 || d->Cin <= 0 || d->Cout <= 0 || d->KH <= 0 || d->KW <= 0) return SISR_E_BADARG;
    if (d->stride != 1 && d->stride != 2) return SISR_E_BADARG;
    if (d->y_mode == SISR_Y_NHWC_SHUFFLE2 && (d->Cout & 3)) return SISR_E_BADARG;
    if (d->x_mode == SISR_X_NHWC_UNSHUFFLE2 && (d->Cin & 3)) return SISR_E_BADARG;
    SisrConvPlan& p = d->plan;
    std::memset(&p, 0, sizeof(p));
    const int64_t ypix = std::max((int64_t)d->N * d->y_H * d->y_W, (int64_t)d->N * d->Ho * d->Wo);
    if (ypix * d->Cout >= (1ll << 29) || (int64_t)d->N * d->H * d->W * d->Cin >= (1ll << 31))
        return SISR_E_TOOBIG;
    p.nsub = d->Cout <= 32 ? 1 : 2;
    const int BN = p.nsub * 32;
    p.CoutPad = round_up(d->Cout, BN);
    const int64_t out_pix = (int64_t)d->N * d->Ho * d->Wo;
    const int S = d->stride;
    const int ck_opts[4] = {32, 16, 8, 4};
    // For each workgroup height (BM = 256 or 128 pixels) find the best tile shape, then keep the
    // height whose last wave of workgroups wastes least: cost = ceil(blocks / 256 CUs) * BM.
    SisrConvPlan cand[2];
    int cand_lds[2] = {0, 0};
    double cand_cost[2] = {1e30, 1e30};
    for (int mi = 0; mi < 2; ++mi) {
        const int msub = 2 - mi;
        const int BM = 4 * msub * 32;
        SisrConvPlan q = p;
        q.msub = msub;
        int best_lds = 1 << 30;
        double best_score = -1.0;
        for (int pass = 0; pass < 2 && best_score < 0; ++pass) {
            const int lds_cap = pass == 0 ? 80 * 1024 : 160 * 1024;
            for (int oi = 0; oi < 4; ++oi) {
                int CK = d->Cin <= 32 ? d->Cin : ck_opts[oi];
                if (d->Cin <= 32 && oi > 0) break;
                const int PS = CK | 1;
                const int KROWP = round_up(d->KW * PS, 4);
                for (int TW = 1; TW <= std::min(d->Wo, BM); ++TW) {
                    const int TH = std::min(d->Ho, BM / TW);
                    int TN = 1;
                    if (TH == d->Ho && TW == d->Wo) TN = std::max(1, std::min(d->N, BM / (TH * TW)));
                    const int lds = conv_lds_bytes(BM, TN, TH, TW, S, d->KH, d->KW, PS, KROWP, BN);
                    if (lds > lds_cap) continue;
                    const int ty = (d->Ho + TH - 1) / TH, tx = (d->Wo + TW - 1) / TW, ngr = (d->N + TN - 1) / TN;
                    const double eff = (double)out_pix / ((double)ty * tx * ngr * BM);
                    const double halo = (double)(TH * TW) * S * S /
                                        ((double)((TH - 1) * S + d->KH) * ((TW - 1) * S + d->KW));
                    const double keff = (double)(d->KW * CK) / KROWP;
                    const double score = eff * (0.75 + 0.25 * halo) * (0.5 + 0.5 * keff);
                    if (score > best_score + 1e-9) {
                        best_score = score; best_lds = lds;
                        q.TH = TH; q.TW = TW; q.TN = TN; q.tiles_y = ty; q.tiles_x = tx; q.n_groups = ngr;
                        q.CK = CK; q.PS = PS; q.KROWP = KROWP;
                    }
                }
                if (best_score >= 0 && pass == 0 && d->Cin > 32) break;   // largest chunk that fits wins
            }
        }
        if (best_score < 0) continue;
        const int64_t blocks = (int64_t)q.tiles_y * q.tiles_x * q.n_groups * (p.CoutPad / BN);
        cand[mi] = q;
        cand_lds[mi] = best_lds;
        cand_cost[mi] = (double)((blocks + 255) / 256) * BM * (mi == 0 ? 1.0 : 1.02);
    }
    double best_score = -1.0;
    int best_lds = 0;
    {
        const int pick = cand_cost[1] < cand_cost[0] ? 1 : 0;
        if (cand_cost[pick] < 1e29) {
            p = cand[pick];
            best_lds = cand_lds[pick];
            best_score = 1.0;
        }
    }
    if (best_score < 0) return SISR_E_TOOBIG;
    p.n_chunk = (d->Cin + p.CK - 1) / p.CK;
    p.n_tiles = p.tiles_y * p.tiles_x * p.n_groups;
    p.lds_bytes = best_lds;
    p.wpk_elems = p.n_chunk * d->KH * p.CoutPad * p.KROWP;
    return 0;
}

template <int MSUB, int NSUB, int TAG>
static int launch_conv(const SisrConvDesc* d, hipStream_t st) {
    static SisrLdsCap cap;   // raise the dynamic-LDS cap only when a plan needs it
    if (int e = sisr_raise_lds_cap(cap, reinterpret_cast<const void*>(&conv_mfma_f32_kernel<MSUB, NSUB, TAG>), d->plan.lds_bytes, 64 * 1024)) return e;
    const dim3 grid(d->plan.n_tiles, d->plan.CoutPad / (NSUB * 32));
    hipLaunchKernelGGL((conv_mfma_f32_kernel<MSUB, NSUB, TAG>), grid, dim3(SISR_BLOCK), d->plan.lds_bytes, st, *d);
    SISR_CHECK_LAUNCH();
    return 0;
}

extern "C" int sisr_conv2d_trunk_f32_eligible(const SisrConvDesc* d);
int sisr_conv2d_trunk_f32_launch(const SisrConvDesc* d, hipStream_t st);      // conv_trunk_f32.hip
extern "C" int sisr_conv2d_thin_eligible(const SisrConvDesc* d);
int sisr_conv2d_thin_launch(const SisrConvDesc* d, hipStream_t st);           // conv_thin.hip
extern "C" int sisr_conv2d_toimage_f32_eligible(const SisrConvDesc* d);
int sisr_conv2d_toimage_launch(const SisrConvDesc* d, hipStream_t st);        // conv_toimage.hip

extern "C" int sisr_conv2d_f32(const SisrConvDesc* d, void* stream) {
    // fused BatchNorm-backward partials: bf16 kernels and the persistent fp32 trunk kernel only
    if (d && d->bnb_part && sisr_conv2d_trunk_f32_eligible(d) != 2) return SISR_E_UNSUPPORTED;
    if (!d || !d->x1 || !d->wpk || !d->y) return SISR_E_BADARG;
    if (operand_needs_x2(d->pro_mode) && !d->x2) return SISR_E_BADARG;
    if (d->stat_part && !d->cnt_part) return SISR_E_BADARG;
    if (d->stat_part && d->y_mode != SISR_Y_NHWC) return SISR_E_UNSUPPORTED;
    if (d->y_bf16 && (d->y_mode == SISR_Y_NCHW || d->res || d->epi_act != SISR_EPI_NONE)) return SISR_E_UNSUPPORTED;
    if (d->res && d->res_bf16) return SISR_E_UNSUPPORTED;
    const SisrConvPlan& p = d->plan;
    if (p.n_tiles <= 0 || p.lds_bytes <= 0 || p.lds_bytes > 160 * 1024) return SISR_E_BADARG;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (sisr_conv2d_trunk_f32_eligible(d)) return sisr_conv2d_trunk_f32_launch(d, st);
    if (sisr_conv2d_thin_eligible(d)) return sisr_conv2d_thin_launch(d, st);    // bf16 build: 9x9 over a 3-channel image
    if (sisr_conv2d_toimage_f32_eligible(d)) return sisr_conv2d_toimage_launch(d, st);     // the generator's last conv (64 -> 3)
    if (d->pro_mode == SISR_PRO_RES_AFFINE) return SISR_E_UNSUPPORTED;      // persistent trunk kernels only
    const bool trunk = d->Cin == 64 && d->Cout == 64 && d->KH == 3 && d->KW == 3 && d->stride == 1;
    if (p.msub == 2 && p.nsub == 2) return trunk ? launch_conv<2, 2, 1>(d, st) : launch_conv<2, 2, 0>(d, st);
    if (p.msub == 2 && p.nsub == 1) return launch_conv<2, 1, 0>(d, st);
    if (p.msub == 1 && p.nsub == 2) return trunk ? launch_conv<1, 2, 1>(d, st) : launch_conv<1, 2, 0>(d, st);
    if (p.msub == 1 && p.nsub == 1) return launch_conv<1, 1, 0>(d, st);
    return SISR_E_BADARG;
}
