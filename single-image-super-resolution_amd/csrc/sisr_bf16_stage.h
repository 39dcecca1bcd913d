// sisr_bf16_stage.h -- bf16 LDS tile staging shared by conv_bf16.hip and wgrad_bf16.hip
#pragma once
#include "sisr_dev.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

#define BF_CK 32
#define BF_PS 40

// float4 staging with the prologue fixed at compile time, bf16 LDS image (see stage_tile_vec)
template <int PRO, int SBQ>
__device__ __forceinline__ void stage_tile_bf16(const OperandView& o, __bf16* lds, int PS, int CK, int c0, int TN,
                                                int IH, int IW, int n0, int iy_org, int ix_org, int valid_w) {
    constexpr bool need2 = PRO == SISR_PRO_BNBWD || PRO == SISR_PRO_BNACT_BWD || PRO == SISR_PRO_ACT_BWD ||
                           PRO == SISR_PRO_TANH_BWD;
    const int tid = threadIdx.x;
    const int G = CK >> 2;
    int lg = 0;
    while ((1 << lg) < G) ++lg;
    const int g = tid & (G - 1);
    const int c = c0 + g * 4;
    const bool c_ok = c < o.C;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    f32x4 ka = zero, kb = zero, kd = zero, ks = zero, kt = zero;
    if (c_ok) {
        if (PRO == SISR_PRO_AFFINE_ACT || PRO == SISR_PRO_BNBWD || PRO == SISR_PRO_BNACT_BWD) {
            ka = *reinterpret_cast<const f32x4*>(o.pa + c);
            kd = *reinterpret_cast<const f32x4*>(o.pd + c);
        }
        if (PRO == SISR_PRO_BNBWD || PRO == SISR_PRO_BNACT_BWD) kb = *reinterpret_cast<const f32x4*>(o.pb + c);
        if (PRO == SISR_PRO_BNACT_BWD) {
            ks = *reinterpret_cast<const f32x4*>(o.ps + c);
            kt = *reinterpret_cast<const f32x4*>(o.pt + c);
        }
    }
    int coff = c, ysh = 0, xsh = 0, Cp = o.C, Wp = o.W, Hp = o.H, mul = 1;
    if (o.mode == SISR_X_NHWC_UNSHUFFLE2) {
        const int Cq = o.C >> 2;
        const int ij = c / Cq;
        coff = c - ij * Cq; ysh = ij >> 1; xsh = ij & 1; Cp = Cq; Wp = 2 * o.W; Hp = 2 * o.H; mul = 2;
    }
    const int xstep = mul * Cp;
    // flat item loop, SB items per thread per batch: all global loads of a batch are issued before the first
    // of them is consumed, so a tile costs ~one memory latency instead of one per item.  (row, column) of a
    // thread's next pixel advance incrementally -- no per-item division.
    constexpr int SB = need2 ? SBQ / 2 : SBQ;      // loads in flight per thread and operand
    const int ppi = SISR_BLOCK >> lg;                    // pixels advanced per item step
    const int step_rows = ppi / IW, step_cols = ppi - step_rows * IW;
    const int rows = TN * IH, npix = rows * IW;
    int pix = tid >> lg;
    int row = pix / IW, ixl = pix - row * IW;
    for (; pix < npix; ) {
        f32x4 a[SB], b[need2 ? SB : 1];
        int lds_off[SB];
        bool live[SB], ok[SB];
#pragma unroll
        for (int u = 0; u < SB; ++u) {
            live[u] = pix < npix;
            int tn = 0, iyl = row;
            if (TN > 1) { tn = row / IH; iyl = row - tn * IH; }
            const int n = n0 + tn, iy = iy_org + iyl, ix = ix_org + ixl;
            ok[u] = live[u] && c_ok && n < o.N && iy >= 0 && iy < o.H && ix >= 0 && ix < o.W && ixl < valid_w;
            lds_off[u] = pix * PS + g * 4;
            a[u] = zero;
            if (need2) b[u] = zero;
            if (ok[u]) {
                const int off = ((n * Hp + iy * mul + ysh) * Wp + xsh) * Cp + coff + ix * xstep;
                a[u] = *reinterpret_cast<const f32x4*>(o.x1 + off);
                if (need2) b[u] = *reinterpret_cast<const f32x4*>(o.x2 + off);
            }
            pix += ppi; row += step_rows; ixl += step_cols;
            if (ixl >= IW) { ixl -= IW; ++row; }
        }
#pragma unroll
        for (int u = 0; u < SB; ++u) {
            if (live[u]) {
                f32x4 v = zero;
                if (ok[u]) v = apply4<PRO>(a[u], need2 ? b[u] : zero, ka, kb, kd, ks, kt, o.slope);
                bf16x4 h;
                h[0] = (__bf16)v[0]; h[1] = (__bf16)v[1]; h[2] = (__bf16)v[2]; h[3] = (__bf16)v[3];
                *reinterpret_cast<bf16x4*>(lds + lds_off[u]) = h;
            }
        }
    }
}

template <int SBQ = 8>
__device__ __forceinline__ void stage_operand_tile_bf16(const OperandView& o, __bf16* lds, int PS, int CK, int c0,
                                                        int TN, int IH, int IW, int n0, int iy_org, int ix_org,
                                                        int valid_w) {
    switch (o.pro) {
#define SISR_STAGE_CASE(P) \
    case P: stage_tile_bf16<P, SBQ>(o, lds, PS, CK, c0, TN, IH, IW, n0, iy_org, ix_org, valid_w); break;
        SISR_STAGE_CASE(SISR_PRO_NONE)
        SISR_STAGE_CASE(SISR_PRO_ACT)
        SISR_STAGE_CASE(SISR_PRO_AFFINE_ACT)
        SISR_STAGE_CASE(SISR_PRO_BNBWD)
        SISR_STAGE_CASE(SISR_PRO_BNACT_BWD)
        SISR_STAGE_CASE(SISR_PRO_ACT_BWD)
        SISR_STAGE_CASE(SISR_PRO_TANH_BWD)
#undef SISR_STAGE_CASE
    }
}


