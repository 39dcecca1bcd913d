// sisr_bf16_stage.h -- bf16 LDS tile staging shared by conv_bf16.hip and wgrad_bf16.hip
#pragma once
#include "sisr_dev.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

#define BF_CK 32
#define BF_PS 40

__device__ __forceinline__ __amdgpu_buffer_rsrc_t bf_rsrc(const void* p, unsigned bytes) { return sisr_rsrc(p, bytes); }

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

// gfx950 transposing LDS read (cdna_hip_programming.md T10): per 16-lane group a block of 4 rows x 16 columns of
// 16-bit elements; lane 4q+p of the group supplies the address of row q, columns 4p..4p+3; lane i receives column i
// of the 4 rows.  Needs all 64 lanes active and 8-byte aligned addresses.
__device__ __forceinline__ s16x4 lds_tr16(const __bf16* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
}
// four bf16 values (two dwords as loaded) -> four floats: a bf16 is the upper half of the float with the same value
__device__ __forceinline__ f32x4 bf16x4_bits_to_f32(u32x2 w) {
    f32x4 v;
    v[0] = __uint_as_float(w[0] << 16); v[1] = __uint_as_float(w[0] & 0xFFFF0000u);
    v[2] = __uint_as_float(w[1] << 16); v[3] = __uint_as_float(w[1] & 0xFFFF0000u);
    return v;
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
// two floats -> one dword of two bf16 (round to nearest even): a single v_cvt_pk_bf16_f32
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
// leaky ReLU.  EASY (0 <= slope <= 1, every slope this model family uses): max(v, slope v), one multiply and one
// v_max (as an instruction: fmaxf() adds a canonicalising v_max per operand); otherwise compare and select.
// EASY == 2: slope is exactly 1 (no activation on this operand): the identity.
template <int EASY>
__device__ __forceinline__ float lrelu_t(float v, float slope) {
    if (EASY == 2) return v;
    if (EASY) {
        float r;
        const float sv = slope * v;
        asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(v), "v"(sv));
        return r;
    }
    return v > 0.f ? v : slope * v;
}

// float4 staging with the prologue fixed at compile time, bf16 LDS image.
// Addressing is the cheap part by construction: a thread's items walk the tile in steps of `ppi` pixels, so
// its (row, column) and the two offset terms advance by adds; loads are raw buffer loads with a 32-bit byte
// offset (tensor < 4 GB, checked by the planners) and pixels outside the image get an out-of-range offset,
// which the hardware answers with zeros -- no divergent branch, no 64-bit address math, no multiplies.
// XBF: the tensor behind x1 / x2 is stored as bf16 (8-byte loads of the thread's 4 channels) instead of fp32.
//
// Staging is split in two halves so that a kernel can keep the loads of one tile (or of two operands) in flight
// while it does something else: stage_issue() computes the addresses of one ROUND of NIT items per thread and
// issues their loads into a StageRegs; stage_commit() applies the prologue, rounds to bf16 and writes the LDS image.
template <int PRO>
struct StageTraits {
    static constexpr bool need2 = PRO == SISR_PRO_BNBWD || PRO == SISR_PRO_BNACT_BWD || PRO == SISR_PRO_ACT_BWD ||
                                  PRO == SISR_PRO_TANH_BWD;
    // prologues with f(0) != 0 need the halo forced to zero after the transform
    static constexpr bool mask_after = PRO == SISR_PRO_AFFINE_ACT || PRO == SISR_PRO_BNBWD || PRO == SISR_PRO_BNACT_BWD;
};

template <bool XBF> struct StageRaw { typedef u32x4 type; };
template <> struct StageRaw<true> { typedef u32x2 type; };

template <bool NEED2, bool XBF, int NIT>
struct StageRegs {
    typename StageRaw<XBF>::type a[NIT];
    typename StageRaw<XBF>::type b[NEED2 ? NIT : 1];
    unsigned ok;                                   // bit u: item u lies inside the image
};

__device__ __forceinline__ f32x4 stage_unpack(u32x2 w) { return bf16x4_bits_to_f32(w); }
__device__ __forceinline__ f32x4 stage_unpack(u32x4 w) { return __builtin_bit_cast(f32x4, w); }

template <bool NEED2, bool XBF, int NIT>
__device__ __forceinline__ void stage_issue(const OperandView& o, StageRegs<NEED2, XBF, NIT>& r, int round, int CK, int c0,
                                            int TN, int IH, int IW, int n0, int iy_org, int ix_org, int valid_w,
                                            uint32_t m_iw) {
    constexpr bool need2 = NEED2;
    const int tid = threadIdx.x;
    const int G = CK >> 2;
    int lg = 0;
    while ((1 << lg) < G) ++lg;
    const int g = tid & (G - 1);
    const int c = c0 + g * 4;
    const bool c_ok = c < o.C;
    int coff = c, ysh = 0, xsh = 0, Cp = o.C, Wp = o.W, Hp = o.H, mul = 1;
    if (o.mode == SISR_X_NHWC_UNSHUFFLE2) {
        const int Cq = o.C >> 2;
        const int ij = c / Cq;
        coff = c - ij * Cq; ysh = ij >> 1; xsh = ij & 1; Cp = Cq; Wp = 2 * o.W; Hp = 2 * o.H; mul = 2;
    }
    constexpr int EB = XBF ? 2 : 4;                                                                    // bytes per element
    const int col_step = mul * Cp * EB, row_step = mul * Wp * Cp * EB, img_step = Hp * Wp * Cp * EB;   // bytes
    const unsigned nbytes = (unsigned)o.N * (unsigned)img_step;
    const __amdgpu_buffer_rsrc_t r1 = bf_rsrc(o.x1, nbytes);
    const __amdgpu_buffer_rsrc_t r2 = bf_rsrc(need2 ? o.x2 : o.x1, nbytes);
    // byte offset of the tile origin for this thread's channel group (may be "negative": only used in range)
    const int base = (((n0 * Hp + iy_org * mul + ysh) * Wp + ix_org * mul + xsh) * Cp + coff) * EB;
    const int ppi = SISR_BLOCK >> lg;              // pixels advanced per item step
    const int step_rows = fdiv(ppi, m_iw), step_cols = ppi - step_rows * IW;      // m_iw = fdiv_magic(IW)
    const int npix = TN * IH * IW;
    const int d_roff = step_rows * row_step, d_xoff = step_cols * col_step, wrap_xoff = IW * col_step;
    int pix = (tid >> lg) + round * (NIT * ppi);
    int row = fdiv(pix, m_iw), ixl = pix - row * IW;
    int roff = row * row_step, xoff = ixl * col_step;
    unsigned okm = 0;
#pragma unroll
    for (int u = 0; u < NIT; ++u) {
        int iyl = row, ro = roff;
        bool n_ok = true;
        if (TN > 1) {                           // whole small images per tile: rows run over (image, row)
            const int tn = row / IH;
            iyl = row - tn * IH;
            ro = tn * img_step + iyl * row_step;
            n_ok = n0 + tn < o.N;
        }
        const int iy = iy_org + iyl, ix = ix_org + ixl;
        const bool ok = pix < npix && c_ok && n_ok && (unsigned)iy < (unsigned)o.H && (unsigned)ix < (unsigned)o.W &&
                        ixl < valid_w;
        const unsigned voff = ok ? (unsigned)(base + ro + xoff) : 0xFFFFFFF0u;
        if constexpr (XBF) {
            r.a[u] = __builtin_amdgcn_raw_buffer_load_b64(r1, voff, 0, 0);
            if constexpr (need2) r.b[u] = __builtin_amdgcn_raw_buffer_load_b64(r2, voff, 0, 0);
        } else {
            r.a[u] = __builtin_amdgcn_raw_buffer_load_b128(r1, voff, 0, 0);
            if constexpr (need2) r.b[u] = __builtin_amdgcn_raw_buffer_load_b128(r2, voff, 0, 0);
        }
        okm |= ok ? (1u << u) : 0u;
        pix += ppi; row += step_rows; ixl += step_cols; roff += d_roff; xoff += d_xoff;
        if (ixl >= IW) { ixl -= IW; ++row; roff += row_step; xoff -= wrap_xoff; }
    }
    r.ok = okm;
}

// SUM: also add the committed values (after the prologue, before the bf16 rounding) of the thread's 4 channels into
// `sum` -- the bias gradient of the weight-gradient kernel, for free while its dy operand passes through registers
template <int PRO, bool XBF, int NIT, bool SUM = false>
__device__ __forceinline__ void stage_commit(const OperandView& o, const StageRegs<StageTraits<PRO>::need2, XBF, NIT>& r,
                                             __bf16* lds, int round, int PS, int CK, int c0, int npix,
                                             f32x4* sum = nullptr) {
    constexpr bool need2 = StageTraits<PRO>::need2;
    constexpr bool mask_after = StageTraits<PRO>::mask_after;
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
    const int ppi = SISR_BLOCK >> lg;
    const int pix0 = (tid >> lg) + round * (NIT * ppi);
    __bf16* dst = lds + pix0 * PS + g * 4;
    const int dst_step = ppi * PS;
#pragma unroll
    for (int u = 0; u < NIT; ++u) {
        const bool ok = (r.ok >> u) & 1u;
        f32x4 v = apply4<PRO>(stage_unpack(r.a[u]), need2 ? stage_unpack(r.b[u]) : zero, ka, kb, kd, ks, kt, o.slope);
        if (mask_after || SUM) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = ok ? v[j] : 0.f;
        }
        if (SUM) *sum += v;
        bf16x4 h;
        h[0] = (__bf16)v[0]; h[1] = (__bf16)v[1]; h[2] = (__bf16)v[2]; h[3] = (__bf16)v[3];
        if (pix0 + u * ppi < npix) *reinterpret_cast<bf16x4*>(dst + u * dst_step) = h;
        // one item at a time: the loads are already in registers, interleaving the items only multiplies temporaries
#ifndef SISR_AB_NO_COMMIT_FENCE
        if (NIT > 4) __builtin_amdgcn_sched_barrier(0);
#endif
    }
}

template <int PRO, int SBQ, bool XBF, bool SUM = false>
__device__ __forceinline__ void stage_tile_bf16(const OperandView& o, __bf16* lds, int PS, int CK, int c0, int TN,
                                                int IH, int IW, int n0, int iy_org, int ix_org, int valid_w,
                                                uint32_t m_iw, f32x4* sum = nullptr) {
    constexpr int SB = StageTraits<PRO>::need2 ? SBQ / 2 : SBQ;      // loads in flight per thread and operand
    const int G = CK >> 2;
    int lg = 0;
    while ((1 << lg) < G) ++lg;
    const int ppi = SISR_BLOCK >> lg, npix = TN * IH * IW;
    for (int round = 0; round * (SB * ppi) < npix; ++round) {
        StageRegs<StageTraits<PRO>::need2, XBF, SB> r;
        stage_issue<StageTraits<PRO>::need2, XBF, SB>(o, r, round, CK, c0, TN, IH, IW, n0, iy_org, ix_org, valid_w, m_iw);
        stage_commit<PRO, XBF, SB, SUM>(o, r, lds, round, PS, CK, c0, npix, sum);
    }
}

// XSEL: 0 = fp32 tensor, 1 = bf16 tensor, -1 = decided at run time by o.bf16
// SUM: also accumulate the staged values (after the prologue, before the bf16 rounding; 0 outside the image) of the
// thread's 4 channels into *sum
template <int SBQ = 8, int XSEL = -1, bool SUM = false>
__device__ __forceinline__ void stage_operand_tile_bf16(const OperandView& o, __bf16* lds, int PS, int CK, int c0,
                                                        int TN, int IH, int IW, int n0, int iy_org, int ix_org,
                                                        int valid_w, uint32_t m_iw, f32x4* sum = nullptr) {
    switch (o.pro) {
#define SISR_STAGE_CASE(P)                                                                                              \
    case P:                                                                                                             \
        if (XSEL == 1 || (XSEL < 0 && o.bf16))                                                                          \
            stage_tile_bf16<P, SBQ, true, SUM>(o, lds, PS, CK, c0, TN, IH, IW, n0, iy_org, ix_org, valid_w, m_iw, sum); \
        else                                                                                                            \
            stage_tile_bf16<P, SBQ, false, SUM>(o, lds, PS, CK, c0, TN, IH, IW, n0, iy_org, ix_org, valid_w, m_iw, sum);\
        break;
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
