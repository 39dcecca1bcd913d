// sisr_bf16_stage.h -- bf16 LDS tile staging shared by conv_bf16.hip and wgrad_bf16.hip
#pragma once
#include "sisr_dev.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

#define BF_CK 32
#define BF_PS 40

// float4 staging with the prologue fixed at compile time, bf16 LDS image (see stage_tile_vec)
template <int PRO>
__device__ __forceinline__ void stage_tile_bf16(const OperandView& o, __bf16* lds, int PS, int CK, int c0, int TN,
                                                int IH, int IW, int n0, int iy_org, int ix_org, int valid_w) {
    constexpr bool need2 = PRO == SISR_PRO_BNBWD || PRO == SISR_PRO_BNACT_BWD || PRO == SISR_PRO_ACT_BWD ||
                           PRO == SISR_PRO_TANH_BWD;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int G = CK >> 2;
    int lg = 0;
    while ((1 << lg) < G) ++lg;
    const int g = lane & (G - 1);
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
    const int row_items = IW << lg;
    for (int row = wave; row < TN * IH; row += SISR_BLOCK / 64) {
        const int tn = row / IH, iyl = row - tn * IH;
        const int n = n0 + tn, iy = iy_org + iyl;
        const bool row_ok = c_ok && n < o.N && iy >= 0 && iy < o.H;
        const int rbase = ((n * Hp + iy * mul + ysh) * Wp + xsh) * Cp + coff;
        __bf16* lrow = lds + row * IW * PS;
        for (int item = lane; item < row_items; item += 64) {
            const int ixl = item >> lg;
            const int ix = ix_org + ixl;
            f32x4 v = zero;
            if (row_ok && ix >= 0 && ix < o.W && ixl < valid_w) {
                const int off = rbase + ix * xstep;
                const f32x4 a = *reinterpret_cast<const f32x4*>(o.x1 + off);
                f32x4 b = zero;
                if (need2) b = *reinterpret_cast<const f32x4*>(o.x2 + off);
                v = apply4<PRO>(a, b, ka, kb, kd, ks, kt, o.slope);
            }
            bf16x4 h;
            h[0] = (__bf16)v[0]; h[1] = (__bf16)v[1]; h[2] = (__bf16)v[2]; h[3] = (__bf16)v[3];
            *reinterpret_cast<bf16x4*>(lrow + ixl * PS + g * 4) = h;
        }
    }
}

__device__ __forceinline__ void stage_operand_tile_bf16(const OperandView& o, __bf16* lds, int PS, int CK, int c0,
                                                        int TN, int IH, int IW, int n0, int iy_org, int ix_org,
                                                        int valid_w) {
    switch (o.pro) {
#define SISR_STAGE_CASE(P) \
    case P: stage_tile_bf16<P>(o, lds, PS, CK, c0, TN, IH, IW, n0, iy_org, ix_org, valid_w); break;
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

