// conv_trunk.hip -- the generator's trunk convolution (3x3, 64 -> 64, stride 1, pad 1, bf16 NHWC tensors) as a
// PERSISTENT, weights-in-registers kernel for gfx950.  It is the same arithmetic as conv_bf16.hip's generic kernel
// (bf16 MFMA operands, fp32 accumulate, fp32 BatchNorm statistics / reductions) specialised for the one geometry that
// carries 66 of a training step's ~100 convolution launches (model_generator.py:10,13,39 forward and data gradient).
//
// Why a second kernel: the generic kernel is bound by instruction issue and per-workgroup fixed costs, not by HBM or
// the matrix cores (profiles/r02_pmc_trunk_kernels_generic_per_launch.json: 1,925 VALU + 675 SALU per wave and
// 128-pixel tile for 72 MFMAs; 1,152 workgroups = 1.5 rounds of 3 per CU, each re-staging 73 KB of weights).  Here:
//   * ONE workgroup per CU (256 threads, launch bounds 1 wave per SIMD = the whole 512-register file) walks
//     ~4.5 tiles of 8 x 16 output pixels; wave (h, g) owns output channels 32h..32h+31 of tile rows 4g..4g+3;
//   * its B operands -- W[32 couts][64 cin][9 taps] = 36 MFMA fragments = 144 VGPRs -- are loaded ONCE per launch and
//     stay in registers: no weight image in LDS, no weight traffic per tile, half the LDS fragment reads per MFMA;
//   * the geometry is compile-time (tile 8 x 16, halo 10 x 18, 144-byte LDS pixel stride), so every LDS fragment
//     address is base + immediate: the MFMA phase is 72 ds_read_b128 + 72 MFMAs and nothing else;
//   * staging works on 16-byte items (8 channels); the next tile's loads are issued before the MFMA phase of the
//     current one and committed to the other LDS buffer after it: one barrier per tile, memory latency off the
//     critical path;
//   * BatchNorm statistics are carried across the workgroup's tiles in registers: ONE (count, mean, M2) partial per
//     workgroup (256 per launch instead of 1,152), so the finalize kernels read a quarter of the rows.
// Requirements (checked by sisr_conv2d_trunk_eligible): Cin = Cout = 64, 3x3, stride 1, pad 1, bf16 NHWC input and
// output, H % 8 == 0, W % 16 == 0, plain output mode.  Everything else runs on the generic kernels.
#include "sisr_dev.h"

#include <algorithm>
#include <type_traits>
#include <cstdlib>

#include "sisr_bf16_stage.h"

#define TK_TH 8
#define TK_TW 16
#define TK_IH (TK_TH + 2)
#define TK_IW (TK_TW + 2)
#define TK_NPIX (TK_IH * TK_IW)          // 180 halo pixels
// LDS halo image: pixel (py, pxx) at py * TK_RP + pxx * TK_PSB.  160 bytes per pixel and 16 bytes of padding per row
// make the consumers' ds_read_b128 of an A fragment (lanes 0-15: 16 pixels of one halo row, lanes 16-31: the row below,
// served in the lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}) conflict-free; a plain 144-byte pixel stride
// costs two extra LDS cycles per read (44 % of all LDS cycles, SQ_LDS_BANK_CONFLICT).
#define TK_PSB 160
#define TK_RP (TK_IW * TK_PSB + 16)
#define TK_HALO_BYTES (TK_IH * TK_RP)     // 28960
#define TK_ITEMS ((TK_NPIX * 8 + 255) / 256)   // 16-byte staging items per producer thread (6)
#define TK_YS 68                          // output image: [32 couts][64 pixels + 4] bf16 per wave

// phase timeline, developer build only (make trace; tools/trace_trunk.py): thread 0 stamps the 100 MHz wall clock
#ifdef SISR_CONV_TRACE
#define TT_WG 512
#define TT_SLOTS 128
__device__ unsigned long long sisr_ttrace_buf[TT_WG * TT_SLOTS];
#define TT(k)                                                                                                   \
    do {                                                                                                        \
        if (threadIdx.x == 0 && blockIdx.x < TT_WG && (k) < TT_SLOTS) sisr_ttrace_buf[blockIdx.x * TT_SLOTS + (k)] = wall_clock64(); \
    } while (0)
// producer timeline: thread 256 (first producer wave), slots 64 ..
#define TTP(k)                                                                                                  \
    do {                                                                                                        \
        if (threadIdx.x == 256 && blockIdx.x < TT_WG && 64 + (k) < TT_SLOTS) sisr_ttrace_buf[blockIdx.x * TT_SLOTS + 64 + (k)] = wall_clock64(); \
    } while (0)
#define TTC(k) do { if (threadIdx.x == 0 && blockIdx.x < TT_WG) sisr_ttrace_buf[blockIdx.x * TT_SLOTS + (k)] = clock64(); } while (0)
extern "C" int sisr_ttrace_read(void* dst, int n_u64) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(sisr_ttrace_buf), (size_t)n_u64 * 8, 0, hipMemcpyDeviceToHost);
}
#else
#define TT(k)
#define TTP(k)
#define TTC(k)
#endif

// barrier-wait accounting, developer build only (-DSISR_BARRIER_ACCT; tools/barrier_acct.py): every tile-loop barrier of the
// forward kernel is bracketed by two s_memtime reads whose difference is summed in SGPRs -- no stores, no LDS drain inside
// the loop -- and lane 0 of wave 0 (consumer) / wave 4 (producer) stores {loop cycles, cycles spent waiting at barriers}
#ifdef SISR_BARRIER_ACCT
__device__ unsigned long long sisr_bacct_buf[512 * 4];
extern "C" int sisr_bacct_read(void* dst, int n_u64) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(sisr_bacct_buf), (size_t)n_u64 * 8, 0, hipMemcpyDeviceToHost);
}
#define BA_DECL unsigned long long ba_wait = 0, ba_t0 = clock64()
#define BA_SYNC() do { const unsigned long long b0_ = clock64(); __syncthreads(); ba_wait += clock64() - b0_; } while (0)
#define BA_STORE(slot) do { if ((threadIdx.x & 63) == 0 && blockIdx.x < 512) { sisr_bacct_buf[blockIdx.x * 4 + (slot)] = clock64() - ba_t0; sisr_bacct_buf[blockIdx.x * 4 + (slot) + 1] = ba_wait; } } while (0)
#else
#define BA_DECL
#define BA_SYNC() __syncthreads()
#define BA_STORE(slot)
#endif

struct TrunkArgs {
    const void *x1, *x2;                  // input operand(s), bf16 NHWC [N][H][W][64]
    void* x_out;                          // skip-sum prologue: the materialised operand
    BnFinArgs fin;                        // deferred BatchNorm finalisation (fin.stat != nullptr): pa / pd come from here
    const float *pa, *pb, *pd, *ps, *pt;  // per-channel prologue constants
    const float* slope_p; float slope;
    const void* wpk;                      // bf16 image [2 chunks][64 couts][9 taps][32 cin]
    const void* wln;                      // the same weights in the consumers' load order (SisrWeightDesc.bf_f_lanes), or nullptr
    const float* bias;
    void* y;                              // bf16 NHWC [N][H][W][64]
    float *stat_part, *cnt_part;          // forward role: [grid][2][64], [grid]
    // data-gradient role: residual (skip gradient) added to the output; the output is the gradient arriving at a
    // BatchNorm whose input is bnb_x: its backward reductions (SisrConvDesc.bnb_*), one row [2*64+1] per workgroup
    const void *res, *bnb_x;
    const float *bnb_scale, *bnb_shift, *bnb_mean, *bnb_invstd, *bnb_slope_p;
    float bnb_slope; int bnb_act;
    float* bnb_part;
    int N, H, W;
    int tiles_x, per_img, total;          // tiles per row, per image, in all
    uint32_t m_tiles_x, m_per_img;        // reciprocals (fdiv_magic)
    int pro;
    // forward role with Cout = 256 + PixelShuffle(2) store (the generator's upscale conv, model_generator.py:43-48): 4 groups
    // of 64 packed couts = the 4 shuffle phases; workgroup b serves group b % 4 on tile stream b / 4
    int glog, cout_pad, shuffle;
};

// The producers are bound by VALU issue (they share a SIMD's issue port with the consumer's MFMAs), so everything about
// a staging item that does not depend on the tile is computed once per thread: item k of producer thread ptid is halo
// pixel (ptid + 256 k) / 8, channel octet ptid % 8.
//   rel[k]   byte offset of the item from the tile's first pixel in the NHWC tensor (negative in the top / left halo)
//   ldso[k]  byte offset inside the LDS halo image
//   flags    5 bits per item: halo row 0, halo row 9, halo column 0, halo column 17, beyond the 180 halo pixels
// A tile turns its own edge pattern (5 bits: image above / below / left / right missing, 1) into a mask replicated over
// the items; flags & mask is non-zero exactly for the items outside the image.
struct HaloMap {
    int rel[TK_ITEMS], ldso[TK_ITEMS];
    unsigned flags;
};
__device__ __forceinline__ void halo_map_init(HaloMap& m, int ptid, int W) {
    const int oct = ptid & 7;
    m.flags = 0;
#pragma unroll
    for (int k = 0; k < TK_ITEMS; ++k) {
        const int px = (ptid + k * 256) >> 3;
        const int py = px / TK_IW, pxx = px - py * TK_IW;
        m.rel[k] = ((py - 1) * W + (pxx - 1)) * 128 + oct * 16;
        m.ldso[k] = py * TK_RP + pxx * TK_PSB + oct * 16;
        const unsigned f = (py == 0 ? 1u : 0u) | (py == TK_IH - 1 ? 2u : 0u) | (pxx == 0 ? 4u : 0u) | (pxx == TK_IW - 1 ? 8u : 0u) |
                           (px >= TK_NPIX ? 16u : 0u);
        m.flags |= f << (5 * k);
    }
}
__device__ __forceinline__ unsigned tile_edge_mask(int ty, int tx, int tiles_y, int tiles_x) {
    const unsigned e = (ty == 0 ? 1u : 0u) | (ty == tiles_y - 1 ? 2u : 0u) | (tx == 0 ? 4u : 0u) | (tx == tiles_x - 1 ? 8u : 0u) | 16u;
    return e * 0x02108421u;                                  // replicated at bits 0, 5, .., 25
}

// prologue of 8 consecutive channels of one pixel: a (and b) hold 8 bf16; returns 8 bf16 packed
template <int PRO, int EASY>
__device__ __forceinline__ u32x4 trunk_apply8(u32x4 a, u32x4 b, const f32x8& ka, const f32x8& kb, const f32x8& kd,
                                              const f32x8& ks, const f32x8& kt, float slope, bool ok) {
    if (PRO == SISR_PRO_NONE) return a;                       // zeros outside the image already (hardware OOB)
    u32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float a0 = __uint_as_float(a[j] << 16), a1 = __uint_as_float(a[j] & 0xFFFF0000u);
        float b0 = 0.f, b1 = 0.f;
        if (PRO == SISR_PRO_BNBWD || PRO == SISR_PRO_BNACT_BWD || PRO == SISR_PRO_RES_AFFINE) {
            b0 = __uint_as_float(b[j] << 16); b1 = __uint_as_float(b[j] & 0xFFFF0000u);
        }
        float r0, r1;
        if (PRO == SISR_PRO_ACT) { r0 = lrelu_t<EASY>(a0, slope); r1 = lrelu_t<EASY>(a1, slope); }
        else if (PRO == SISR_PRO_RES_AFFINE) {                 // as sisr_eltwise_res_affine: lrelu(x1) + (a x2 + d)
            r0 = lrelu_t<EASY>(a0, slope) + (ka[2 * j] * b0 + kd[2 * j]); r1 = lrelu_t<EASY>(a1, slope) + (ka[2 * j + 1] * b1 + kd[2 * j + 1]);
        }
        else if (PRO == SISR_PRO_AFFINE_ACT) {
            r0 = lrelu_t<EASY>(ka[2 * j] * a0 + kd[2 * j], slope); r1 = lrelu_t<EASY>(ka[2 * j + 1] * a1 + kd[2 * j + 1], slope);
        } else if (PRO == SISR_PRO_BNBWD) {
            r0 = ka[2 * j] * a0 + kb[2 * j] * b0 + kd[2 * j];
            r1 = ka[2 * j + 1] * a1 + kb[2 * j + 1] * b1 + kd[2 * j + 1];
        } else {                                               // BNACT_BWD
            const float z0 = ks[2 * j] * b0 + kt[2 * j], z1 = ks[2 * j + 1] * b1 + kt[2 * j + 1];
            const float g0 = z0 > 0.f ? a0 : slope * a0, g1 = z1 > 0.f ? a1 : slope * a1;
            r0 = ka[2 * j] * g0 + kb[2 * j] * b0 + kd[2 * j];
            r1 = ka[2 * j + 1] * g1 + kb[2 * j + 1] * b1 + kd[2 * j + 1];
        }
        // prologues with f(0) != 0: the halo must be zero AFTER the transform
        const unsigned pk = pack_bf16x2(r0, r1);
        o[j] = (PRO == SISR_PRO_ACT || ok) ? pk : 0u;
    }
    return o;
}

// Forward role: one-tensor prologue (NONE / ACT / AFFINE_ACT), bias, BatchNorm statistics.
// 512 threads: waves 0-3 are CONSUMERS (MFMA + epilogue, weights in registers), waves 4-7 are PRODUCERS (they stage
// the next tile into the other LDS buffer while the consumers work on the current one).  Wave k and wave k + 4 share a
// SIMD (the dispatcher deals a workgroup's waves over the SIMDs cyclically), so each SIMD overlaps one matrix-heavy
// and one memory / VALU-heavy wave.  One workgroup barrier per tile.
#define TK_THREADS 512
#ifndef TK_PF
#define TK_PF 3                           // A-fragment prefetch distance of the data-gradient consumers' MFMA loop, in steps of 2 MFMAs
#endif
#ifndef TK_DEFER
#define TK_DEFER 0                        // A/B variants: 1 = also defer sub-tile 1's epilogue under the next tile's first phase; 2 = one
                                          // accumulator set, each sub-tile's epilogue right behind its own phase
#endif
#ifndef TK_EARLY
#define TK_EARLY 0
#endif
#ifndef TK_PFA
#define TK_PFA 4                          // ... of the forward consumers' phases, in MFMAs (5 and more spill: 256 VGPRs are in use)
#endif
template <int PRO>
__global__ void __launch_bounds__(TK_THREADS, 2) conv_trunk_fwd_kernel(const TrunkArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    __bf16* out_img = reinterpret_cast<__bf16*>(lds + 2 * TK_HALO_BYTES);            // [4 waves][32][TK_YS]
    float* red = reinterpret_cast<float*>(lds + 2 * TK_HALO_BYTES + 4 * 32 * TK_YS * 2);   // [4 waves][32][3]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);         // provably wave-uniform
    const bool consumer = wave < 4;
    const int l31 = lane & 31, kk = lane >> 5;
    const int h = wave & 1, g = (wave >> 1) & 1;
    const unsigned xbytes = (unsigned)a.N * (unsigned)a.H * (unsigned)a.W * 128u;
    const int cg = blockIdx.x & ((1 << a.glog) - 1);                    // cout group (shuffle phase) of this workgroup
    const int t_first = blockIdx.x >> a.glog, t_step = gridDim.x >> a.glog;
    auto tile_coords = [&](int T, int& n, int& ty, int& tx) {
        n = fdiv(T, a.m_per_img);
        const int rem = T - n * a.per_img;
        ty = fdiv(rem, a.m_tiles_x);
        tx = rem - ty * a.tiles_x;
    };

    // ---- consumer state: B operands (this wave's 32 output channels, 9 taps x 64 input channels) in registers -------
    bf16x8 bw[9][4];
    int a_base[2] = {0, 0};
    float bv = 0.f;
    float st_shift = 0.f, st_s1 = 0.f, st_s2 = 0.f;     // running statistics of this lane's values, shifted sums
    f32x2 st_s1v = {0.f, 0.f}, st_s2v = {0.f, 0.f};     // (two partial sums each, packed arithmetic; joined into st_s1 / st_s2 after the loop)
    int st_n = 0;
    // ---- producer state: items idx = ptid + 256 k -> halo pixel idx / 8, channel octet ptid % 8 ----------------------
    const int ptid = tid & 255, oct = tid & 7;
    const f32x8 zero8 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    f32x8 ka = zero8, kd = zero8;
    float slope = 1.f;
    // a staging set: the raw items of one tile (x2 only for the two-tensor skip-sum prologue), its "outside" mask and
    // the byte offset of its first pixel (the skip-sum prologue stores the materialised sum back through it)
    constexpr bool TWO = PRO == SISR_PRO_RES_AFFINE;
    struct Stage { u32x4 a[TK_ITEMS], b[TWO ? TK_ITEMS : 1]; unsigned bad, origin; };
    Stage stA, stB;
    HaloMap hm;
    bool easy_slope = true;

    TT(0);
    TTC(60);
    // deferred BatchNorm finalisation: scale / shift of the prologue's BatchNorm from its statistics rows, in LDS (the halo
    // buffers are free yet: 48 KB of scratch, the constants behind the reduction scratch)
    float* kfin = red + 4 * 32 * 3;
    const bool fin = (PRO == SISR_PRO_AFFINE_ACT || PRO == SISR_PRO_RES_AFFINE) && a.fin.stat != nullptr;
    // (role state is set up INSIDE the role branches below: set up ahead of the split, every register of both roles
    // meets in one merge block and the allocator spills weights at load time)
    auto init_consumer = [&]() {
        const unsigned wbytes = 2u * (unsigned)a.cout_pad * 9u * 32u * 2u;
        if (a.wln != nullptr) {
            // lane-order image: [32-cout block 2 cg + h][tap][k slice][lane] x 16 bytes -- every load instruction is 1 KB contiguous
            const __amdgpu_buffer_rsrc_t wrs = bf_rsrc(a.wln, wbytes);
            const unsigned base = (unsigned)(((2 * cg + h) * 36 * 64 + lane) * 16);
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    bw[t][j] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wrs, base + (unsigned)((t * 4 + j) * 1024), 0, 0));
        } else {
            const __amdgpu_buffer_rsrc_t wrs = bf_rsrc(a.wpk, wbytes);
            const int co = 64 * cg + 32 * h + l31;                              // packed cout
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const unsigned off = (unsigned)((((j >> 1) * a.cout_pad + co) * 9 + t) * 32 + (j & 1) * 16 + 8 * kk) * 2u;
                    bw[t][j] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wrs, off, 0, 0));
                }
        }
        // A operand of sub-tile ms: lane (l31, kk) = pixel (tile row 4g + 2ms + (l31 >> 4), column l31 & 15), channels 8kk..
#pragma unroll
        for (int ms = 0; ms < 2; ++ms) a_base[ms] = (4 * g + 2 * ms + (l31 >> 4)) * TK_RP + (l31 & 15) * TK_PSB + kk * 16;
        // (bias is in ORIGINAL channel order: packed cout (phase cg, channel c) of a shuffled layer = original c * 4 + cg)
        bv = a.bias != nullptr ? a.bias[a.shuffle ? (32 * h + l31) * 4 + cg : 32 * h + l31] : 0.f;
    };
    auto init_producer = [&]() {
        slope = a.slope_p ? a.slope_p[0] : a.slope;
        easy_slope = slope >= 0.f && slope <= 1.f;
        halo_map_init(hm, ptid, a.W);
    };
    auto init_constants = [&]() {
        if (PRO == SISR_PRO_AFFINE_ACT || TWO) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                ka[j] = fin ? kfin[oct * 8 + j] : a.pa[oct * 8 + j];
                kd[j] = fin ? kfin[64 + oct * 8 + j] : a.pd[oct * 8 + j];
            }
        }
    };
    // two staging register sets: the loads of tile T + 2 are in flight while tile T + 1 is transformed and written to
    // LDS, so the producers never sit out a full memory latency (they are the critical path otherwise)
    const int tiles_y = a.per_img / a.tiles_x;
    auto issue = [&](int T, Stage& st) {
        const __amdgpu_buffer_rsrc_t rx = bf_rsrc(a.x1, xbytes), rx2 = bf_rsrc(TWO ? a.x2 : a.x1, xbytes);
        int n, ty, tx;
        tile_coords(T, n, ty, tx);
        st.origin = (unsigned)(((n * a.H + ty * TK_TH) * a.W + tx * TK_TW) * 128);
        // (always executed, so that the tile loop stays free of control flow around loads: past the last tile every
        // item is "outside", which costs an instruction and no memory traffic.  With a branch around the loads the
        // compiler's wait-count bookkeeping gives up at the merge and drains every load before the next commit.)
        st.bad = T < a.total ? hm.flags & tile_edge_mask(ty, tx, tiles_y, a.tiles_x) : 0xFFFFFFFFu;
#pragma unroll
        for (int k = 0; k < TK_ITEMS; ++k) {
            const bool ok = ((st.bad >> (5 * k)) & 31u) == 0u;
            const unsigned voff = ok ? st.origin + (unsigned)hm.rel[k] : 0x80000000u;
            st.a[k] = __builtin_amdgcn_raw_buffer_load_b128(rx, voff, 0, 0);
            if (TWO) st.b[k] = __builtin_amdgcn_raw_buffer_load_b128(rx2, voff, 0, 0);
        }
    };
    auto commit_t = [&](auto easy, unsigned char* buf, const Stage& st) {
        constexpr int EASY = decltype(easy)::value;
        const __amdgpu_buffer_rsrc_t ro = bf_rsrc(TWO ? a.x_out : a.x1, xbytes);
#pragma unroll
        for (int k = 0; k < TK_ITEMS; ++k) {
            const bool ok = ((st.bad >> (5 * k)) & 31u) == 0u;
            const u32x4 v = trunk_apply8<PRO, EASY>(st.a[k], st.b[TWO ? k : 0], ka, ka, kd, ka, ka, slope, ok);
            // (only the last item of a thread can lie beyond the 180 halo pixels)
            if (k < TK_ITEMS - 1 || !((hm.flags >> (5 * k + 4)) & 1u)) *reinterpret_cast<u32x4*>(buf + hm.ldso[k]) = v;
            // skip-sum prologue: the tiles partition the image, so the items of a tile's own 8 x 16 pixels (no halo flag
            // set) store every pixel of the materialised sum exactly once
            if (TWO && ((hm.flags >> (5 * k)) & 31u) == 0u) __builtin_amdgcn_raw_buffer_store_b128(v, ro, st.origin + (unsigned)hm.rel[k], 0, 0);
        }
    };
    auto commit = [&](unsigned char* buf, const Stage& st) {
        // (the skip sum of every block but the first has no activation on its residual: slope 1, nothing to compute)
        if (TWO && slope == 1.f) commit_t(std::integral_constant<int, 2>{}, buf, st);
        else if (easy_slope) commit_t(std::integral_constant<int, 1>{}, buf, st);
        else commit_t(std::integral_constant<int, 0>{}, buf, st);
    };
    __bf16* my_out = out_img + (wave & 3) * (32 * TK_YS);
    const int grp = lane >> 4, tq = (lane & 15) >> 2, tp = lane & 3;

#if !TK_EARLY
    if (fin) bn_finalize_in_kernel(a.fin, reinterpret_cast<double*>(lds), kfin, blockIdx.x == 0);
#endif
    // Two role-specific tile loops with matching barrier counts (a barrier only counts arriving waves).  Written as ONE
    // loop with a role branch inside, the register allocator carries the consumers' 144 weight registers through the
    // producers' code (and the producers' staging registers through the consumers').
    if (!consumer) {
        // ---- producers: tile T + 1 into the other buffer while the consumers work on tile T ------------------------------
        int T = t_first;
#if TK_EARLY
        // every wave's first memory requests go out BEFORE the BatchNorm finalisation and the per-channel constants: the
        // producers' first two tiles (and the consumers' weights) are raw loads that depend on nothing computed here.  (The
        // finalisation is instantiated once per role: as common code between two role branches it would keep both roles'
        // registers live at once.)
        halo_map_init(hm, ptid, a.W);
        issue(T, stA);
        issue(T + t_step, stB);
        if (fin) bn_finalize_in_kernel(a.fin, reinterpret_cast<double*>(lds), kfin, blockIdx.x == 0);
        slope = a.slope_p ? a.slope_p[0] : a.slope;
        easy_slope = slope >= 0.f && slope <= 1.f;
        init_constants();
        TTP(0);
#else
        init_producer();
        init_constants();
        TTP(0);
        issue(T, stA);
        issue(T + t_step, stB);
#endif
        TTP(1);
        if (T < a.total) commit(lds, stA);
        TTP(2);
        __syncthreads();
        BA_DECL;
        // unrolled by two: each staging set has a fixed name in each half (stB holds tile T + grid in the first)
        int cur = 0;
        [[maybe_unused]] int it = 0;
        while (T < a.total) {
            TTP(4 + 6 * it);
            issue(T + 2 * t_step, stA);
            TTP(5 + 6 * it);
            if (T + t_step < a.total) commit(lds + (cur ^ 1) * TK_HALO_BYTES, stB);
            TTP(8 + 6 * it);
            BA_SYNC();
            TTP(9 + 6 * it);
            T += t_step; cur ^= 1; ++it;
            if (T >= a.total) break;
            TTP(4 + 6 * it);
            issue(T + 2 * t_step, stB);
            TTP(5 + 6 * it);
            if (T + t_step < a.total) commit(lds + (cur ^ 1) * TK_HALO_BYTES, stA);
            TTP(8 + 6 * it);
            BA_SYNC();
            TTP(9 + 6 * it);
            T += t_step; cur ^= 1; ++it;
        }
        // (the always-executed prefetch loads past the last tile are still in flight: wait for them HERE, so that the
        // role join below carries no pending load -- with one, the compiler's wait-count bookkeeping makes the consumers
        // drain their output stores, ~2 us, before the statistics tail)
        __builtin_amdgcn_s_waitcnt(0x0F70);                  // vmcnt(0)
        if (wave == 4) BA_STORE(2);
    } else {
        init_consumer();
#if TK_EARLY
        if (fin) bn_finalize_in_kernel(a.fin, reinterpret_cast<double*>(lds), kfin, blockIdx.x == 0);
#endif
        TT(2);
#ifdef SISR_CONV_TRACE
        __builtin_amdgcn_s_waitcnt(0x0F70);                  // (trace build: when the weights have landed)
        TT(58);
#endif
        __syncthreads();
        __builtin_amdgcn_s_waitcnt(0x0F70);                  // vmcnt(0): the weights (long here: the barrier took microseconds)
        // ---- consumers.  A tile is two sub-tiles of 32 pixels (tile rows 4g + 2ms, 4g + 2ms + 1) x this wave's 32 couts,
        // 36 MFMAs each (9 taps x 4 K slices; every A address is base + immediate, B is in registers).  The epilogue of a
        // sub-tile -- statistics, bf16 conversion, transpose through LDS, 16-byte stores -- is software-pipelined UNDER the
        // MFMAs of the next sub-tile: sub-tile 0's under sub-tile 1's of the same tile, sub-tile 1's under sub-tile 0's of
        // the NEXT tile (its accumulators live across the barrier).  Written out in six slices that are placed between
        // fixed MFMAs (sched_barrier pins the source order), so the matrix pipe never waits for an epilogue: before, every
        // tile ended with 0.62 us of epilogue while the pipe was idle (profiles/r02_trace_trunk_fwd.txt).
        const __amdgpu_buffer_rsrc_t ry = bf_rsrc(a.y, a.shuffle ? 4u * xbytes : xbytes);
        auto consumer_loop = [&](auto STATS_) {
            constexpr bool STATS = decltype(STATS_)::value;
            f32x16 acc0, acc1;
            // slice sl (0 .. 5) of the epilogue of sub-tile ms whose accumulators are `acc`, tile (n, ty, tx):
            //   0 .. 3  register group q = sl: statistics of its 4 values, bf16, 8-byte LDS store into the wave's
            //           [32 couts][64 pixels] image (pixels 32 ms + 8 q + 4 kk .. + 3 of cout l31)
            //   4, 5    16-pixel block pb = 2 ms + (sl - 4) = tile row 4 g + pb: transposing reads, one 16-byte store
            auto epi = [&](int sl, const f32x16& acc, int ms, bool first, int n, int ty, int tx) {
#ifdef SISR_ABLATE_EPI          // timing-only ablation (no output): the consumers' epilogue is skipped
                if (a.N > 0) { asm volatile("" :: "v"(acc[0])); return; }
#endif
                if (sl < 4) {
                    const int q = sl;
                    // (the accumulators start from zero -- a literal operand of the first MFMA -- and the bias joins here: an
                    // accumulator set initialised to the bias keeps a 16-register copy of it alive for the whole launch)
                    if (STATS) {
                        if (first && q == 0) {                      // shift = mean of the first sub-tile's values of this lane
                            float sm = 0.f;
#pragma unroll
                            for (int i = 0; i < 16; ++i) sm += acc[i];
                            st_shift = sm * (1.f / 16.f) + bv;
                        }
                        // (written on pairs so that it compiles to packed fp32 instructions: 6 per 4 values instead of ~14 -- this
                        // wave's VALU instructions and the MFMA stream add up on the SIMD; two partial sums per lane, joined at the end)
                        const float st_c = bv - st_shift;
                        const f32x2 c2 = {st_c, st_c};
                        const f32x2 d01 = f32x2{acc[4 * q], acc[4 * q + 1]} + c2, d23 = f32x2{acc[4 * q + 2], acc[4 * q + 3]} + c2;
                        st_s1v += d01; st_s1v += d23;
                        st_s2v = __builtin_elementwise_fma(d01, d01, st_s2v);
                        st_s2v = __builtin_elementwise_fma(d23, d23, st_s2v);
                        if (q == 3) st_n += 16;
                    }
                    bf16x4 hv;
#pragma unroll
                    for (int j = 0; j < 4; ++j) hv[j] = (__bf16)(acc[4 * q + j] + bv);
                    *reinterpret_cast<bf16x4*>(my_out + l31 * TK_YS + 32 * ms + 8 * q + 4 * kk) = hv;
                } else {
                    const int pb = 2 * ms + (sl - 4);               // group = channel octet
                    const __bf16* src = my_out + (8 * grp + tq) * TK_YS + 16 * pb + 4 * tp;
                    const s16x4 lo = lds_tr16(src), hi = lds_tr16(src + 4 * TK_YS);
                    const s16x8 v8 = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                    const int oy = ty * TK_TH + 4 * g + pb, ox = tx * TK_TW + (lane & 15);
                    // PixelShuffle(2) on store: phase cg = (i, j) of this workgroup's couts lands on pixel (2 oy + i, 2 ox + j)
                    const unsigned vo = a.shuffle ? (unsigned)(((n * 2 * a.H + 2 * oy + (cg >> 1)) * 2 * a.W + 2 * ox + (cg & 1)) * 128 + (32 * h + 8 * grp) * 2)
                                                  : (unsigned)(((n * a.H + oy) * a.W + ox) * 128 + (32 * h + 8 * grp) * 2);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v8), ry, vo, 0, 0);
                }
            };
            // after which MFMA of a 36-MFMA phase each slice goes (the head of a phase stays MFMA-only: the matrix pipe
            // starts at once behind the barrier; the stores go last so the data gradient's / next layer's loads see them early
            // enough without crowding the head)
            auto slice_at = [](int c) { return c == 5 ? 0 : c == 10 ? 1 : c == 15 ? 2 : c == 20 ? 3 : c == 27 ? 4 : c == 33 ? 5 : -1; };
            // one phase: 36 MFMAs into `acc` over sub-tile `ms` of halo image `ib`; `under(c)` runs after MFMA c
            auto phase = [&](f32x16& acc, const unsigned char* ib, int ms, auto&& under) {
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] = 0.f;
                bf16x8 af[36];
                const unsigned char* pa = ib + a_base[ms];
                auto fetch = [&](int st) {
                    const int t = st >> 2, j = st & 3;
                    af[st] = *reinterpret_cast<const bf16x8*>(pa + (t / 3) * TK_RP + (t % 3) * TK_PSB + j * 32);
                };
#pragma unroll
#ifdef SISR_ABLATE_AREADS
                for (int st = 0; st < TK_PFA; st += 2) fetch(st);
#else
                for (int st = 0; st < TK_PFA; ++st) fetch(st);
#endif
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int st = 0; st < 36; ++st) {
#ifdef SISR_ABLATE_AREADS      // timing-only ablation (wrong results): every second A fragment is not fetched but reused
                    if (st + TK_PFA < 36 && ((st + TK_PFA) & 1) == 0) fetch(st + TK_PFA);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[st & ~1], bw[st >> 2][st & 3], acc, 0, 0, 0);
#elif defined(SISR_ABLATE_MFMA)  // timing-only ablation (wrong results): fragment reads without the matrix instructions
                    if (st + TK_PFA < 36) fetch(st + TK_PFA);
                    asm volatile("" :: "v"(af[st]), "v"(bw[st >> 2][st & 3]));
#else
                    if (st + TK_PFA < 36) fetch(st + TK_PFA);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[st], bw[st >> 2][st & 3], acc, 0, 0, 0);
#endif
                    under(st);
                    __builtin_amdgcn_sched_barrier(0);              // the source order IS the software pipeline
                }
            };
            int cur = 0, it = 0;
            int n, ty, tx;
            BA_DECL;
#if TK_DEFER == 1
            // sub-tile 1's epilogue deferred under the NEXT tile's first phase (its accumulators live across the barrier)
            int T = t_first, pn = 0, pty = 0, ptx = 0;
            if (T < a.total) {
                // first tile (peeled): nothing to finish under its first phase
                TT(4);
                tile_coords(T, n, ty, tx);
                phase(acc0, lds, 0, [&](int) {});
                TT(6);
                phase(acc1, lds, 1, [&](int c) { const int sl = slice_at(c); if (sl >= 0) epi(sl, acc0, 0, true, n, ty, tx); });
                TT(8);
                BA_SYNC();            // the next tile's image is complete; the consumers have finished reading this one
                TT(9);
                pn = n; pty = ty; ptx = tx;
                T += t_step; cur ^= 1; ++it;
            }
            for (; T < a.total; T += t_step, cur ^= 1, ++it) {
                TT(4 + 6 * it);
                tile_coords(T, n, ty, tx);
                const unsigned char* ib = lds + cur * (TK_HALO_BYTES);
                phase(acc0, ib, 0, [&](int c) { const int sl = slice_at(c); if (sl >= 0) epi(sl, acc1, 1, false, pn, pty, ptx); });
                TT(6 + 6 * it);
                phase(acc1, ib, 1, [&](int c) { const int sl = slice_at(c); if (sl >= 0) epi(sl, acc0, 0, false, n, ty, tx); });
                TT(8 + 6 * it);
                BA_SYNC();
                TT(9 + 6 * it);
                pn = n; pty = ty; ptx = tx;
            }
            if (it > 0) {                                           // the last tile's second sub-tile
#pragma unroll
                for (int sl = 0; sl < 6; ++sl) epi(sl, acc1, 1, false, pn, pty, ptx);
            }
#elif TK_DEFER == 2
            // ONE accumulator set: each sub-tile's epilogue follows its own 36 MFMAs.  The 16 registers this frees go to the
            // A-fragment prefetch (8 instead of 4 in flight): the consumers are bound by LDS LATENCY at a prefetch of 4 --
            // ablations (tools/barrier_acct.py builds): without any MFMA the 72 fragment reads of a tile still take ~2,000
            // cycles (4 in flight x ~110 cycles each), halving the reads at the same depth changes nothing
            for (int T = t_first; T < a.total; T += t_step, cur ^= 1, ++it) {
                TT(4 + 6 * it);
                tile_coords(T, n, ty, tx);
                const unsigned char* ib = lds + cur * (TK_HALO_BYTES);
                const bool first = it == 0;
                phase(acc0, ib, 0, [&](int) {});
#pragma unroll
                for (int sl = 0; sl < 6; ++sl) epi(sl, acc0, 0, first, n, ty, tx);
                TT(6 + 6 * it);
                phase(acc0, ib, 1, [&](int) {});
#pragma unroll
                for (int sl = 0; sl < 6; ++sl) epi(sl, acc0, 1, false, n, ty, tx);
                TT(8 + 6 * it);
                BA_SYNC();
                TT(9 + 6 * it);
            }
#else
            // sub-tile 0's epilogue under sub-tile 1's MFMAs; sub-tile 1's own epilogue follows its phase (no accumulator
            // is carried around the loop: that costs the allocator 16 registers, which buy a deeper A prefetch here)
            for (int T = t_first; T < a.total; T += t_step, cur ^= 1, ++it) {
                TT(4 + 6 * it);
                tile_coords(T, n, ty, tx);
                const unsigned char* ib = lds + cur * (TK_HALO_BYTES);
                phase(acc0, ib, 0, [&](int) {});
                TT(6 + 6 * it);
                const bool first = it == 0;
                phase(acc1, ib, 1, [&](int c) { const int sl = slice_at(c); if (sl >= 0) epi(sl, acc0, 0, first, n, ty, tx); });
#pragma unroll
                for (int sl = 0; sl < 6; ++sl) epi(sl, acc1, 1, false, n, ty, tx);
                TT(8 + 6 * it);
                BA_SYNC();
                TT(9 + 6 * it);
            }
#endif
            TT(1);
            if (wave == 0) BA_STORE(0);
        };
        if (a.stat_part != nullptr) consumer_loop(std::true_type{});
        else consumer_loop(std::false_type{});
    }
    TT(3);

    // ---- one (count, mean, M2) partial per workgroup and channel ------------------------------------------------------
    if (a.stat_part != nullptr) {
        if (consumer) {
            st_s1 = st_s1v[0] + st_s1v[1];
            st_s2 = st_s2v[0] + st_s2v[1];
            float n = (float)st_n, mu = 0.f, m2 = 0.f;
            if (st_n > 0) {
                const float m1 = st_s1 / n;
                mu = st_shift + m1;
                m2 = st_s2 - st_s1 * m1;
            }
            // lane halves (kk), then the two row-group waves of this channel half, merged with Chan's formula
            const float nb = __shfl_xor(n, 32), mub = __shfl_xor(mu, 32), m2b = __shfl_xor(m2, 32);
            const float nt = n + nb;
            if (nt > 0.f) { const float dl = mub - mu, f = nb / nt; mu += dl * f; m2 += m2b + dl * dl * n * f; }
            n = nt;
            if (kk == 0) { float* r = red + (wave * 32 + l31) * 3; r[0] = n; r[1] = mu; r[2] = m2; }
        }
        __syncthreads();
        if (tid < 64) {
            const int hh = tid >> 5, c = tid & 31;               // waves hh (g = 0) and hh + 2 (g = 1)
            const float* r0 = red + (hh * 32 + c) * 3;
            const float* r1 = red + ((hh + 2) * 32 + c) * 3;
            float nn = r0[0], mm = r0[1], qq = r0[2];
            const float nb = r1[0], nt = nn + nb;
            if (nt > 0.f) { const float dl = r1[1] - mm, f = nb / nt; mm += dl * f; qq += r1[2] + dl * dl * nn * f; }
            float* sp = a.stat_part + (int64_t)blockIdx.x * 128 + 32 * hh + c;
            sp[0] = mm;
            sp[64] = qq;
            if (tid == 0) a.cnt_part[blockIdx.x] = nt;
        }
    }
    TT(63);
    TTC(61);
}


// Data-gradient role: two-tensor prologue (BNBWD / BNACT_BWD: the gradient and the forward activation of the BatchNorm
// being differentiated), optional residual (the skip gradient) and optional fused backward reductions of the NEXT
// BatchNorm of the chain.  Same producer / consumer structure; the producers also copy the residual and BatchNorm-input
// tiles of the NEXT tile into pixel-major LDS images ([128 pixels][64] bf16, double-buffered) with LDS-direct loads
// (buffer_load_dwordx4 ... lds: no staging registers), from which the consumers fetch them in accumulator layout with
// ds_read_b64_tr_b16.  A direct load lays a wave's 64 x 16 bytes down linearly, so rows cannot be padded; instead the
// 16-byte slot s of pixel p holds channel octet s ^ 2 (p & 3), which spreads the 4 pixel rows of a transposing read
// over the banks.  The reductions are carried across the workgroup's tiles in registers: one partial row per workgroup.
#define TK_RS 64
#define TK_IMG (128 * TK_RS * 2)                 // bytes of one residual / BatchNorm-input image
#define TK_KOFF (4 * 32 * 2 + 4)                 // prologue constants inside the reduction scratch: [5][64] floats
#define TK_RED_BYTES 4096
template <int PRO>
__global__ void __launch_bounds__(TK_THREADS, 2) conv_trunk_bwd_kernel(const TrunkArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    __bf16* out_img = reinterpret_cast<__bf16*>(lds + 2 * TK_HALO_BYTES);            // [4 waves][32][TK_YS]
    float* red = reinterpret_cast<float*>(lds + 2 * TK_HALO_BYTES + 4 * 32 * TK_YS * 2);   // [4 waves][32][2] + [4]
    unsigned char* img0 = lds + 2 * TK_HALO_BYTES + 4 * 32 * TK_YS * 2 + TK_RED_BYTES;   // [2 buffers][res, bnb_x][TK_IMG]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool consumer = wave < 4;
    const int l31 = lane & 31, kk = lane >> 5;
    const int h = wave & 1, g = (wave >> 1) & 1;
    const bool has_r = a.res != nullptr, has_x = a.bnb_part != nullptr;                 // uniform
    const unsigned xbytes = (unsigned)a.N * (unsigned)a.H * (unsigned)a.W * 128u;
    auto tile_coords = [&](int T, int& n, int& ty, int& tx) {
        n = fdiv(T, a.m_per_img);
        const int rem = T - n * a.per_img;
        ty = fdiv(rem, a.m_tiles_x);
        tx = rem - ty * a.tiles_x;
    };

    // ---- consumer state ---------------------------------------------------------------------------------------------
    bf16x8 bw[9][4];
    int a_base[2] = {0, 0};
    float b_sc = 0.f, b_sf = 0.f, b_mu = 0.f, b_is = 0.f, b_slope = 1.f;
    // running sums of this lane's channel (two partial sums each, added at the end): g, g * xhat, slope term
    f32x2 rs1v = {0.f, 0.f}, rs2v = {0.f, 0.f}, rslv = {0.f, 0.f};
    f32x2 sc2 = {0.f, 0.f}, sf2 = {0.f, 0.f}, is2 = {0.f, 0.f}, nm2 = {0.f, 0.f}, sl2 = {1.f, 1.f};
    // ---- producer state ---------------------------------------------------------------------------------------------
    const int ptid = tid & 255, oct = tid & 7;
    float slope = 1.f;
    u32x4 sa[TK_ITEMS], sb[TK_ITEMS];
    unsigned sok = 0;

    auto init_consumer = [&]() {
        const int co = 32 * h + l31;
        if (a.wln != nullptr) {
            const __amdgpu_buffer_rsrc_t wrs = bf_rsrc(a.wln, 2u * 64u * 9u * 32u * 2u);
            const unsigned base = (unsigned)((h * 36 * 64 + lane) * 16);
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    bw[t][j] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wrs, base + (unsigned)((t * 4 + j) * 1024), 0, 0));
        } else {
            const __amdgpu_buffer_rsrc_t wrs = bf_rsrc(a.wpk, 2u * 64u * 9u * 32u * 2u);
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const unsigned off = (unsigned)((((j >> 1) * 64 + co) * 9 + t) * 32 + (j & 1) * 16 + 8 * kk) * 2u;
                    bw[t][j] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wrs, off, 0, 0));
                }
        }
#pragma unroll
        for (int ms = 0; ms < 2; ++ms) a_base[ms] = (4 * g + 2 * ms + (l31 >> 4)) * TK_RP + (l31 & 15) * TK_PSB + kk * 16;
        if (has_x) {
            b_sc = a.bnb_scale[co]; b_sf = a.bnb_shift[co]; b_mu = a.bnb_mean[co]; b_is = a.bnb_invstd[co];
            b_slope = a.bnb_slope_p ? a.bnb_slope_p[0] : a.bnb_slope;
            sc2 = f32x2{b_sc, b_sc}; sf2 = f32x2{b_sf, b_sf}; is2 = f32x2{b_is, b_is}; sl2 = f32x2{b_slope, b_slope};
            nm2 = f32x2{-b_mu * b_is, -b_mu * b_is};                   // xhat = x * invstd - mean * invstd
        }
    };
    // two staging register sets: the loads of tile T + 2 are in flight while tile T + 1 is transformed and written to LDS
    u32x4 sa2[TK_ITEMS], sb2[TK_ITEMS];
    unsigned sok2 = 0;
    HaloMap hm;
    const int tiles_y = a.per_img / a.tiles_x;
    auto issue = [&](int T, u32x4 (&ra)[TK_ITEMS], u32x4 (&rb)[TK_ITEMS], unsigned& bad) {
        const __amdgpu_buffer_rsrc_t r1 = bf_rsrc(a.x1, xbytes), r2 = bf_rsrc(a.x2, xbytes);
        int n, ty, tx;
        tile_coords(T, n, ty, tx);
        const unsigned origin = (unsigned)(((n * a.H + ty * TK_TH) * a.W + tx * TK_TW) * 128);
        bad = T < a.total ? hm.flags & tile_edge_mask(ty, tx, tiles_y, a.tiles_x) : 0xFFFFFFFFu;   // (always executed: see the forward kernel)
#pragma unroll
        for (int k = 0; k < TK_ITEMS; ++k) {
            const bool ok = ((bad >> (5 * k)) & 31u) == 0u;
            const unsigned voff = ok ? origin + (unsigned)hm.rel[k] : 0x80000000u;
            ra[k] = __builtin_amdgcn_raw_buffer_load_b128(r1, voff, 0, 0);
            rb[k] = __builtin_amdgcn_raw_buffer_load_b128(r2, voff, 0, 0);
        }
    };
    // the tile's own 128 pixels x 8 octets of the residual and of the BatchNorm input, LDS-direct into image buffer b:
    // producer wave pw lays down 1 KB chunks 4 pw + k (8 pixels each); lane = (pixel lane >> 3, slot lane & 7) fetches
    // octet slot ^ 2 (pixel & 3).  Issued for tile T + 1 only (the images are double-buffered, not triple-buffered).
    auto issue_images = [&](int T, int b) {
        if (!(has_r || has_x)) return;
        int n, ty, tx;
        tile_coords(T, n, ty, tx);
        int ln_ = lane;
        asm volatile("" : "+v"(ln_));
        const __amdgpu_buffer_rsrc_t q1 = bf_rsrc(has_r ? a.res : a.bnb_x, xbytes), q2 = bf_rsrc(has_x ? a.bnb_x : a.res, xbytes);
        unsigned char* ir = img0 + b * (2 * TK_IMG);
        const int pw = wave & 3;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = 4 * pw + k, m = 8 * c + (ln_ >> 3);                   // tile pixel m = row (m >> 4), col (m & 15)
            const int o = (ln_ & 7) ^ (2 * (m & 3));
            const unsigned voff = (unsigned)(((n * a.H + ty * TK_TH + (m >> 4)) * a.W + tx * TK_TW + (m & 15)) * 128 + o * 16);
            if (has_r)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(q1, (__attribute__((address_space(3))) void*)(ir + c * 1024),
                                                         16, voff, 0, 0, 0);
            if (has_x)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(q2, (__attribute__((address_space(3))) void*)(ir + TK_IMG + c * 1024),
                                                         16, voff, 0, 0, 0);
        }
    };
    // per-channel prologue constants of this thread's octet, in registers for the whole tile loop (the producers' loop has
    // its own register budget, and a producer that reads LDS would wait for its LDS-direct loads first)
    const f32x8 zero8 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    f32x8 ka = zero8, kb = zero8, kd = zero8, ks = zero8, kt = zero8;
    auto commit = [&](int b, const u32x4 (&ra)[TK_ITEMS], const u32x4 (&rb)[TK_ITEMS], unsigned bad) {
        unsigned char* buf = lds + b * (TK_HALO_BYTES);
#pragma unroll
        for (int k = 0; k < TK_ITEMS; ++k) {
            const bool ok = ((bad >> (5 * k)) & 31u) == 0u;
            const u32x4 v = trunk_apply8<PRO, 0>(ra[k], rb[k], ka, kb, kd, ks, kt, slope, ok);
            // (only the last item of a thread can lie beyond the 180 halo pixels)
            if (k < TK_ITEMS - 1 || !((hm.flags >> (5 * k + 4)) & 1u)) *reinterpret_cast<u32x4*>(buf + hm.ldso[k]) = v;
        }
    };
    __bf16* my_out = out_img + (wave & 3) * (32 * TK_YS);
    const int grp = lane >> 4, tq = (lane & 15) >> 2, tp = lane & 3;

    // role-specific tile loops with matching barrier counts (see the forward kernel)
    if (!consumer) {
        slope = a.slope_p ? a.slope_p[0] : a.slope;
        halo_map_init(hm, ptid, a.W);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            ka[j] = a.pa[oct * 8 + j]; kb[j] = a.pb[oct * 8 + j]; kd[j] = a.pd[oct * 8 + j];
            if (PRO == SISR_PRO_BNACT_BWD) { ks[j] = a.ps[oct * 8 + j]; kt[j] = a.pt[oct * 8 + j]; }
        }
        int T = blockIdx.x;
        if (T < a.total) issue_images(T, 0);
        issue(T, sa, sb, sok);
        issue(T + gridDim.x, sa2, sb2, sok2);
        if (T < a.total) commit(0, sa, sb, sok);
        __syncthreads();
        // Per half: register loads of tile T + 2 grid and LDS-direct loads of tile T + grid go out back to back, then tile
        // T + grid is committed from the set filled one half ago while they fly.  The barrier's fence drains them all (it
        // must, for the LDS-direct ones), which costs little: they were issued together.  Unrolled by two: each staging
        // set has a fixed name in each half.
        int cur = 0;
        while (T < a.total) {
            issue(T + 2 * gridDim.x, sa, sb, sok);
            if (T + (int)gridDim.x < a.total) { issue_images(T + gridDim.x, cur ^ 1); commit(cur ^ 1, sa2, sb2, sok2); }
            __syncthreads();
            T += gridDim.x; cur ^= 1;
            if (T >= a.total) break;
            issue(T + 2 * gridDim.x, sa2, sb2, sok2);
            if (T + (int)gridDim.x < a.total) { issue_images(T + gridDim.x, cur ^ 1); commit(cur ^ 1, sa, sb, sok); }
            __syncthreads();
            T += gridDim.x; cur ^= 1;
        }
    } else {
        TT(0);
        init_consumer();
        TT(2);
        __syncthreads();
        int cur = 0, it = 0;
        for (int T = blockIdx.x; T < a.total; T += gridDim.x, cur ^= 1, ++it) {
            TT(4 + 6 * it);
            {
            f32x16 acc[2];
#pragma unroll
            for (int ms = 0; ms < 2; ++ms)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[ms][i] = 0.f;
            const unsigned char* ib = lds + cur * (TK_HALO_BYTES);
            // software pipeline: the A fragments of step s + TK_PF are requested before the two MFMAs of step s
            // (step = tap t, K slice j; left to itself the compiler keeps one step of reads in flight, ~64 cycles of cover)
            bf16x8 af[36][2];
            auto fetch = [&](int st) {
                const int t = st >> 2, j = st & 3;
                const int toff = (t / 3) * TK_RP + (t % 3) * TK_PSB;
#pragma unroll
                for (int ms = 0; ms < 2; ++ms) af[st][ms] = *reinterpret_cast<const bf16x8*>(ib + a_base[ms] + toff + j * 32);
            };
#pragma unroll
            for (int st = 0; st < TK_PF; ++st) fetch(st);
            __builtin_amdgcn_sched_group_barrier(0x100, 2 * TK_PF, 0);
#pragma unroll
            for (int st = 0; st < 36; ++st) {
                if (st + TK_PF < 36) fetch(st + TK_PF);
#pragma unroll
                for (int ms = 0; ms < 2; ++ms)
                    acc[ms] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[st][ms], bw[st >> 2][st & 3], acc[ms], 0, 0, 0);
                if (st + TK_PF < 36) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            }
            TT(6 + 6 * it);
            // ---- epilogue: residual, BatchNorm-backward reductions, bf16, transposed store -------------------------------
            const __amdgpu_buffer_rsrc_t ry = bf_rsrc(a.y, xbytes);
            int n, ty, tx;
            tile_coords(T, n, ty, tx);
            // residual add and BatchNorm-backward reductions, specialised at compile time on what the launch asked for (with
            // the three flags tested per register group the epilogue was a chain of 40 short branches: 2.1 us per tile)
            auto res_bnb = [&](auto R_, auto X_, auto A_) {
                constexpr bool R = decltype(R_)::value, X = decltype(X_)::value, A = decltype(A_)::value;
                const __bf16* ir = reinterpret_cast<const __bf16*>(img0 + cur * (2 * TK_IMG));
                const __bf16* ix_ = ir + TK_IMG / 2;
                // register group u = 4 ms + q of this lane: pixels 64g + 32ms + 8q + 4kk .. +3 (rows of the transposing read),
                // channels 32h + 16 (l31 >> 4) ..; row tq of a read = pixel with (pixel & 3) == tq, whose octets sit at slot
                // octet ^ 2 tq.  The reads of group u + 1 are issued before the arithmetic of group u.
                const int oct_r = (4 * h + 2 * (grp & 1) + (tp >> 1)) ^ (2 * tq);
                const int off0 = (64 * g + 4 * (grp >> 1) + tq) * TK_RS + 8 * oct_r + 4 * (tp & 1);
                s16x4 pr_n = {0, 0, 0, 0}, px_n = {0, 0, 0, 0};
                auto fetch = [&](int u) {
                    const int off = off0 + (32 * (u >> 2) + 8 * (u & 3)) * TK_RS;
                    if (R) pr_n = lds_tr16(ir + off);
                    if (X) px_n = lds_tr16(ix_ + off);
                };
                fetch(0);
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const s16x4 pr_c = pr_n, px_c = px_n;
                    if (u + 1 < 8) fetch(u + 1);
                    __builtin_amdgcn_sched_barrier(0);          // (keeps the compiler from hoisting all 16 reads: 32 registers it does not have)
                    const int ms = u >> 2, q = u & 3;
                    // packed fp32 arithmetic on the 4 values of the register group
                    f32x4 rv = {0.f, 0.f, 0.f, 0.f}, xv = {0.f, 0.f, 0.f, 0.f};
                    if (R) rv = bf16x4_bits_to_f32(__builtin_bit_cast(u32x2, pr_c));
                    if (X) xv = bf16x4_bits_to_f32(__builtin_bit_cast(u32x2, px_c));
#pragma unroll
                    for (int jp = 0; jp < 2; ++jp) {
                        f32x2 gv = {acc[ms][4 * q + 2 * jp], acc[ms][4 * q + 2 * jp + 1]};
                        if (R) {
                            gv += f32x2{rv[2 * jp], rv[2 * jp + 1]};
                            acc[ms][4 * q + 2 * jp] = gv[0];
                            acc[ms][4 * q + 2 * jp + 1] = gv[1];
                        }
                        if (X) {
                            const f32x2 x2 = {xv[2 * jp], xv[2 * jp + 1]};
                            if (A) {
                                const f32x2 z = sc2 * x2 + sf2, gz = gv * z, gs = gv * sl2;
                                const bool n0 = !(z[0] > 0.f), n1 = !(z[1] > 0.f);
                                rslv += f32x2{n0 ? gz[0] : 0.f, n1 ? gz[1] : 0.f};
                                gv = f32x2{n0 ? gs[0] : gv[0], n1 ? gs[1] : gv[1]};
                            }
                            rs1v += gv;
                            rs2v += gv * (x2 * is2 + nm2);
                        }
                    }
                }
            };
            {
                using T_ = std::true_type; using F_ = std::false_type;
                if (has_r && has_x) { if (a.bnb_act) res_bnb(T_{}, T_{}, T_{}); else res_bnb(T_{}, T_{}, F_{}); }
                else if (has_x) { if (a.bnb_act) res_bnb(F_{}, T_{}, T_{}); else res_bnb(F_{}, T_{}, F_{}); }
                else if (has_r) res_bnb(T_{}, F_{}, F_{});
            }
#pragma unroll
            for (int ms = 0; ms < 2; ++ms)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    bf16x4 hv;
#pragma unroll
                    for (int j = 0; j < 4; ++j) { const float v = acc[ms][4 * q + j]; hv[j] = (__bf16)v; }
                    *reinterpret_cast<bf16x4*>(my_out + l31 * TK_YS + 32 * ms + 8 * q + 4 * kk) = hv;
                }
#pragma unroll
            for (int pb = 0; pb < 4; ++pb) {
                const __bf16* src = my_out + (8 * grp + tq) * TK_YS + 16 * pb + 4 * tp;
                const s16x4 lo = lds_tr16(src), hi = lds_tr16(src + 4 * TK_YS);
                const s16x8 v8 = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                const unsigned vo = (unsigned)(((n * a.H + ty * TK_TH + 4 * g + pb) * a.W + tx * TK_TW + (lane & 15)) * 128 +
                                               (32 * h + 8 * grp) * 2);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v8), ry, vo, 0, 0);
            }
            }
            TT(8 + 6 * it);
            __syncthreads();
            TT(9 + 6 * it);
        }
        TT(3);
    }

    // ---- one row of BatchNorm-backward partial sums per workgroup ----------------------------------------------------
    if (has_x) {
        if (consumer) {
            float rs1 = rs1v[0] + rs1v[1], rs2 = rs2v[0] + rs2v[1];
            const float rsl = rslv[0] + rslv[1];
            rs1 += __shfl_xor(rs1, 32);
            rs2 += __shfl_xor(rs2, 32);
            if (kk == 0) { red[(wave * 32 + l31) * 2] = rs1; red[(wave * 32 + l31) * 2 + 1] = rs2; }
            const float ws = wave_sum(rsl);
            if (lane == 0) red[256 + wave] = ws;
        }
        __syncthreads();
        float* wk = a.bnb_part + (int64_t)blockIdx.x * 129;
        if (tid < 64) {
            const int hh = tid >> 5, c = tid & 31;               // waves hh (g = 0) and hh + 2 (g = 1), fixed order
            wk[tid] = red[(hh * 32 + c) * 2] + red[((hh + 2) * 32 + c) * 2];
            wk[64 + tid] = red[(hh * 32 + c) * 2 + 1] + red[((hh + 2) * 32 + c) * 2 + 1];
        }
        if (tid == 0) wk[128] = (red[256] + red[257]) + (red[258] + red[259]);
    }
    TT(63);
}

// ---- host ----------------------------------------------------------------------------------------------------------
static int trunk_groups(const SisrConvDesc* d) { return d->Cout == 256 ? 4 : 1; }
static int trunk_grid(const SisrConvDesc* d) {
    const int total = d->N * (d->H / TK_TH) * (d->W / TK_TW);
    const int cus = sisr_cu_slots();
    int n_cu = cus;
    // equal shares: ceil(total / rounds) workgroups, rounds = ceil(total / (CUs x workgroups per CU))
    int per_cu = 1;
    if (const char* e = getenv("SISR_TRUNK_WG_PER_CU")) per_cu = std::max(1, std::min(2, atoi(e)));
    n_cu *= per_cu;
    const int G = trunk_groups(d);                 // cout groups: each tile stream is served by G workgroups
    n_cu = std::max(1, n_cu / G);
    const int rounds = (total + n_cu - 1) / n_cu;
    return G * ((total + rounds - 1) / rounds);
}

// 1 when this descriptor (geometry + storage flags + fusions requested) can run on the trunk kernel
extern "C" int sisr_conv2d_trunk_eligible(const SisrConvDesc* d) {
    const char* sw = getenv("SISR_TRUNK");                      // A/B switch: SISR_TRUNK=0 keeps the generic kernel
    if (!d || (sw && sw[0] == '0')) return 0;
    if (d->Cin != 64 || d->KH != 3 || d->KW != 3 || d->stride != 1 || d->pad_y != 1 || d->pad_x != 1) return 0;
    // Cout = 64 (trunk), or 256 stored through PixelShuffle(2) -- the upscale conv, forward role without statistics
    const char* swu = getenv("SISR_TRUNK_UP");                 // A/B switch for the upscale conv alone
    const bool up = !(swu && swu[0] == '0') && d->Cout == 256 && d->y_mode == SISR_Y_NHWC_SHUFFLE2 && d->plan.CoutPad == 256 && !d->stat_part && !d->res &&
                    !d->bnb_part && d->pro_mode != SISR_PRO_RES_AFFINE && !d->fin_stat &&
                    (d->pro_mode == SISR_PRO_NONE || d->pro_mode == SISR_PRO_ACT || d->pro_mode == SISR_PRO_AFFINE_ACT);
    if (!up && (d->Cout != 64 || d->y_mode != SISR_Y_NHWC)) return 0;
    if (d->x_mode != SISR_X_NHWC || !d->x_bf16 || !d->y_bf16) return 0;
    if (d->Ho != d->H || d->Wo != d->W || (d->H % TK_TH) || (d->W % TK_TW)) return 0;
    if (!up && (d->y_sy != 1 || d->y_sx != 1 || d->y_oy || d->y_ox || d->y_H != d->Ho || d->y_W != d->Wo)) return 0;
    if (up && (int64_t)d->N * d->H * d->W * 512 >= (1ll << 31)) return 0;
    if (d->epi_act != SISR_EPI_NONE) return 0;
    if ((int64_t)d->N * d->H * d->W * 128 >= (1ll << 31)) return 0;
    if (d->N * (d->H / TK_TH) * (d->W / TK_TW) >= 65536) return 0;
    const bool fwd_pro = d->pro_mode == SISR_PRO_NONE || d->pro_mode == SISR_PRO_ACT || d->pro_mode == SISR_PRO_AFFINE_ACT ||
                         (d->pro_mode == SISR_PRO_RES_AFFINE && d->x2 && d->x_out && ((d->pa && d->pd) || d->fin_stat));
    if (d->fin_stat && !((d->pro_mode == SISR_PRO_AFFINE_ACT || d->pro_mode == SISR_PRO_RES_AFFINE) && d->fin_cnt && d->fin_gamma &&
                         d->fin_beta && d->fin_rm && d->fin_rv && d->fin_k && d->fin_rows > 0))
        return 0;
    if (fwd_pro && !d->res && !d->bnb_part) return 1;           // forward role
    const bool bwd_pro = d->pro_mode == SISR_PRO_BNBWD || d->pro_mode == SISR_PRO_BNACT_BWD;
    if (bwd_pro && !d->stat_part && !d->bias && (!d->res || d->res_bf16) && (!d->bnb_part || d->bnbx_bf16)) return 2;   // data-gradient role
    return 0;
}

// rows of stat_part / cnt_part (or bnb_part) a launch of this descriptor writes: the trunk kernel writes one per
// workgroup, the generic kernels one per tile (plan.n_tiles)
int sisr_conv2d_deep_parts(const SisrConvDesc* d);            // conv_deep.hip
extern "C" int sisr_conv2d_bf16_parts(const SisrConvDesc* d) {
    if (!d) return SISR_E_BADARG;
    if (d->deep.enabled && d->wdeep) return sisr_conv2d_deep_parts(d);
    if (sisr_conv2d_trunk_eligible(d)) return trunk_grid(d);
    return d->plan.n_tiles;
}

template <int PRO>
static int launch_trunk_fwd(const TrunkArgs& a, int grid, hipStream_t st) {
    constexpr int lds_bytes = 2 * TK_HALO_BYTES + 4 * 32 * TK_YS * 2 + 4 * 32 * 3 * 4 + 128 * 4;
    static SisrLdsCap cap;
    if (int e = sisr_raise_lds_cap(cap, reinterpret_cast<const void*>(&conv_trunk_fwd_kernel<PRO>), lds_bytes)) return e;
    hipLaunchKernelGGL((conv_trunk_fwd_kernel<PRO>), dim3(grid), dim3(TK_THREADS), lds_bytes, st, a);
    SISR_CHECK_LAUNCH();
    return 0;
}

template <int PRO>
static int launch_trunk_bwd(const TrunkArgs& a, int grid, bool images, hipStream_t st) {
    const int lds_bytes = 2 * TK_HALO_BYTES + 4 * 32 * TK_YS * 2 + TK_RED_BYTES + (images ? 4 * TK_IMG : 0);
    static SisrLdsCap cap;
    if (int e = sisr_raise_lds_cap(cap, reinterpret_cast<const void*>(&conv_trunk_bwd_kernel<PRO>), lds_bytes, 0)) return e;
    hipLaunchKernelGGL((conv_trunk_bwd_kernel<PRO>), dim3(grid), dim3(TK_THREADS), lds_bytes, st, a);
    SISR_CHECK_LAUNCH();
    return 0;
}

// called by sisr_conv2d_bf16 for eligible descriptors
int sisr_conv2d_trunk_launch(const SisrConvDesc* d, hipStream_t st) {
    TrunkArgs a;
    a.fin.stat = d->fin_stat; a.fin.cnt = d->fin_cnt; a.fin.gamma = d->fin_gamma; a.fin.beta = d->fin_beta;
    a.fin.rm = d->fin_rm; a.fin.rv = d->fin_rv; a.fin.k = d->fin_k; a.fin.rows = d->fin_rows; a.fin.momentum = d->fin_momentum; a.fin.eps = d->fin_eps;
    a.x1 = d->x1; a.x2 = d->x2; a.x_out = d->x_out; a.pa = d->pa; a.pb = d->pb; a.pd = d->pd; a.ps = d->ps; a.pt = d->pt;
    a.slope_p = d->pro_slope_p; a.slope = d->pro_slope;
    a.wpk = d->wpk; a.bias = d->bias;
    a.wln = (d->plan.variant & 1) ? static_cast<const char*>(static_cast<const void*>(d->wpk)) + (size_t)(d->Cout == 256 ? 256 : 64) * 1152 : nullptr; a.y = d->y; a.stat_part = d->stat_part; a.cnt_part = d->cnt_part;
    a.N = d->N; a.H = d->H; a.W = d->W;
    a.tiles_x = d->W / TK_TW; a.per_img = (d->H / TK_TH) * a.tiles_x; a.total = d->N * a.per_img;
    a.m_tiles_x = fdiv_magic(a.tiles_x); a.m_per_img = fdiv_magic(a.per_img);
    a.pro = d->pro_mode;
    a.glog = d->Cout == 256 ? 2 : 0; a.cout_pad = d->Cout == 256 ? 256 : 64; a.shuffle = d->y_mode == SISR_Y_NHWC_SHUFFLE2 ? 1 : 0;
    a.res = d->res; a.bnb_x = d->bnb_x; a.bnb_scale = d->bnb_scale; a.bnb_shift = d->bnb_shift; a.bnb_mean = d->bnb_mean;
    a.bnb_invstd = d->bnb_invstd; a.bnb_slope_p = d->bnb_slope_p; a.bnb_slope = d->bnb_slope; a.bnb_act = d->bnb_act;
    a.bnb_part = d->bnb_part;
    const int grid = trunk_grid(d);
    if (d->pro_mode == SISR_PRO_BNBWD) return launch_trunk_bwd<SISR_PRO_BNBWD>(a, grid, d->res || d->bnb_part, st);
    if (d->pro_mode == SISR_PRO_BNACT_BWD) return launch_trunk_bwd<SISR_PRO_BNACT_BWD>(a, grid, d->res || d->bnb_part, st);
    switch (d->pro_mode) {
        case SISR_PRO_NONE: return launch_trunk_fwd<SISR_PRO_NONE>(a, grid, st);
        case SISR_PRO_ACT: return launch_trunk_fwd<SISR_PRO_ACT>(a, grid, st);
        case SISR_PRO_AFFINE_ACT: return launch_trunk_fwd<SISR_PRO_AFFINE_ACT>(a, grid, st);
        case SISR_PRO_RES_AFFINE: return launch_trunk_fwd<SISR_PRO_RES_AFFINE>(a, grid, st);
    }
    return SISR_E_UNSUPPORTED;
}
