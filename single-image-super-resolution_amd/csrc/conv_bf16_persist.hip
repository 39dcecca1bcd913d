// conv_bf16_persist.hip -- persistent, weights-resident bf16 convolution for Cin = 64 (G's trunk and
// upscale layers: 3x3, stride 1), built for HBM streaming on gfx950.
//
//   * one 768-thread workgroup per CU, alive for the whole launch (grid = 256 / cout-tiles);
//   * the complete packed weight image of its 64-cout tile ([64][taps*64 + 8] bf16, 75 KB for 3x3)
//     is loaded into LDS ONCE;
//   * SPECIALISED WAVES: waves 4-11 are loaders in two groups that alternate tiles -- a group puts the
//     global loads of a whole 4x32-pixel input tile (6x34 halo, all 64 channels) in flight TWO tile
//     iterations before it is needed (two tiles, >= 104 KB, outstanding per CU: the kernel is
//     latency-bound otherwise), then applies the producer's BatchNorm/activation (or the
//     BatchNorm-backward transform), converts to bf16 and writes the free LDS buffer; waves 0-3 are
//     consumers -- 32 pixels x 64 couts each on v_mfma_f32_32x32x16_bf16, then bias / BN statistics /
//     residual / PixelShuffle and the store.  One workgroup barrier per tile; loader VALU+VMEM and
//     consumer MFMA+LDS overlap on every SIMD (one wave of each kind per SIMD);
//   * tiles are walked in vertical strips, consecutive tiles of a workgroup are vertical
//     neighbours, so 2 of the 6 halo rows were just read by the same CU (L2 hits).
// With fp32 tensors in HBM the layer moves 75.5 MB per launch at B=16, 96x96 and needs only ~4.5 us
// of MFMA time: it is HBM-bound by construction.
#include "sisr_dev.h"

#include <algorithm>
#include <cstring>

#include "sisr_bf16_stage.h"

#define PC_CIN 64
#define PC_PS 72            // 144-byte pixel stride: 16-byte fragments of 16 consecutive pixels conflict-free
#define PC_TH 4
#define PC_TW 32
#define PC_THREADS 768      // 4 consumer waves + 2 loader groups of 4 waves

// Workgroup barrier that waits only for this wave's LDS traffic (lgkmcnt), NOT for its global loads /
// stores: __syncthreads() would emit s_waitcnt vmcnt(0) and thereby (a) drain the loaders' prefetch
// that must stay in flight across the barrier and (b) stall the consumers on the write-acks of the
// tile they just stored.  LDS hand-off only needs the ds_writes/ds_reads of this wave to be complete.
__device__ __forceinline__ void pc_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

__device__ __forceinline__ f32x16 pc_mfma(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t pc_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 pc_ld128(__amdgpu_buffer_rsrc_t r, unsigned voff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 0));
}

// loader state: one whole halo tile (<= 6 x 34 pixels x 64 channels) as float4 per thread, PC_NL loads per
// operand in flight.  Addressing is 32-bit buffer offsets: voff[u] (relative to the tile origin, fixed per
// thread for the whole launch) + the tile base -- one v_add per load; no per-tile divisions.
#define PC_NL 13
template <int PRO>
struct PcLoader {
    static constexpr bool need2 = PRO == SISR_PRO_BNBWD || PRO == SISR_PRO_BNACT_BWD || PRO == SISR_PRO_ACT_BWD ||
                                  PRO == SISR_PRO_TANH_BWD;
    f32x4 a[PC_NL], b[need2 ? PC_NL : 1];
    int rel[PC_NL];            // byte offset of pixel u relative to the tile's (iy_org, ix_org), incl. channel
    int yx[PC_NL];             // (iyl << 8) | ixl, or -1 if the slot is beyond the tile
    unsigned okmask;
    f32x4 ka, kb, kd, ks, kt;
    int c, Cp, Wp, mul;

    __device__ __forceinline__ void init(const OperandView& o, int lt, int IH, int IW) {
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        c = (lt & 15) * 4;                 // 16 channel groups of 4: fixed per thread
        ka = kb = kd = ks = kt = zero;
        if (PRO == SISR_PRO_AFFINE_ACT || PRO == SISR_PRO_BNBWD || PRO == SISR_PRO_BNACT_BWD) {
            ka = *reinterpret_cast<const f32x4*>(o.pa + c);
            kd = *reinterpret_cast<const f32x4*>(o.pd + c);
        }
        if (PRO == SISR_PRO_BNBWD || PRO == SISR_PRO_BNACT_BWD) kb = *reinterpret_cast<const f32x4*>(o.pb + c);
        if (PRO == SISR_PRO_BNACT_BWD) {
            ks = *reinterpret_cast<const f32x4*>(o.ps + c);
            kt = *reinterpret_cast<const f32x4*>(o.pt + c);
        }
        int coff = c, ysh = 0, xsh = 0;
        Cp = o.C; Wp = o.W; mul = 1;
        if (o.mode == SISR_X_NHWC_UNSHUFFLE2) {
            const int Cq = o.C >> 2;
            const int ij = c / Cq;
            coff = c - ij * Cq; ysh = ij >> 1; xsh = ij & 1; Cp = Cq; Wp = 2 * o.W; mul = 2;
        }
        const int npix = IH * IW;
#pragma unroll
        for (int u = 0; u < PC_NL; ++u) {
            const int pix = (lt >> 4) + 16 * u;
            const int iyl = pix / IW, ixl = pix - iyl * IW;
            yx[u] = pix < npix ? ((iyl << 8) | ixl) : -1;
            rel[u] = (((iyl * mul + ysh) * Wp + ixl * mul + xsh) * Cp + coff) * 4;
        }
    }
    // issue the global loads of the tile whose halo starts at image n, row iy_org, column ix_org
    __device__ __forceinline__ void issue(const OperandView& o, __amdgpu_buffer_rsrc_t r1, __amdgpu_buffer_rsrc_t r2,
                                          int n, int iy_org, int ix_org) {
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        const int base = ((n * o.H * mul + iy_org * mul) * Wp + ix_org * mul) * Cp * 4;   // may be "before" the image
        okmask = 0u;
#pragma unroll
        for (int u = 0; u < PC_NL; ++u) {
            const int iy = iy_org + (yx[u] >> 8), ix = ix_org + (yx[u] & 255);
            const bool ok = yx[u] >= 0 && iy >= 0 && iy < o.H && ix >= 0 && ix < o.W;
            a[u] = zero;
            if (need2) b[u] = zero;
            if (ok) {
                okmask |= 1u << u;
                const unsigned voff = (unsigned)(base + rel[u]);
                a[u] = pc_ld128(r1, voff);
                if (need2) b[u] = pc_ld128(r2, voff);
            }
        }
    }
    // prologue -> bf16 -> LDS image [pixel][PC_PS]
    __device__ __forceinline__ void commit(const OperandView& o, __bf16* lds, int lt) {
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < PC_NL; ++u) {
            if (yx[u] >= 0) {
                f32x4 v = zero;
                if (okmask & (1u << u)) v = apply4<PRO>(a[u], need2 ? b[u] : zero, ka, kb, kd, ks, kt, o.slope);
                bf16x4 h;
                h[0] = (__bf16)v[0]; h[1] = (__bf16)v[1]; h[2] = (__bf16)v[2]; h[3] = (__bf16)v[3];
                *reinterpret_cast<bf16x4*>(lds + ((lt >> 4) + 16 * u) * PC_PS + c) = h;
            }
        }
    }
};

template <int PRO>
__global__ void __launch_bounds__(PC_THREADS, 1) conv_c64_bf16_persist_kernel(const SisrConvDesc d) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const SisrConvPlan& p = d.plan;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, kk = lane >> 5;
    const bool loader = wave >= 4;
    const int taps = d.KH * d.KW;
    const int WSG = taps * PC_CIN, WS = WSG + 8;
    const int IH = PC_TH + d.KH - 1, IW = PC_TW + d.KW - 1;
    const int tile_elems = (IH * IW * PC_PS + 7) & ~7;

    __bf16* lds_w = reinterpret_cast<__bf16*>(smem);
    __bf16* lds_in0 = lds_w + 64 * WS;
    __bf16* lds_in1 = lds_in0 + tile_elems;

    const int cout_base = blockIdx.y * 64;
    const int n_tiles = p.tiles_x * p.tiles_y * d.N;
    // contiguous share of the tile list (strip order: image, strip tx, then ty)
    const int t_begin = (int)((int64_t)n_tiles * blockIdx.x / gridDim.x);
    const int t_end = (int)((int64_t)n_tiles * (blockIdx.x + 1) / gridDim.x);
    const int K = t_end - t_begin;                       // tiles of this workgroup; local index k = t - t_begin
    const int epi_act = d.epi_act & 255;

    OperandView ov;
    ov.x1 = d.x1; ov.x2 = d.x2; ov.pa = d.pa; ov.pb = d.pb; ov.pd = d.pd; ov.ps = d.ps; ov.pt = d.pt;
    ov.N = d.N; ov.H = d.H; ov.W = d.W; ov.C = d.Cin;
    ov.mode = d.x_mode; ov.pro = d.pro_mode;
    ov.slope = d.pro_slope_p ? d.pro_slope_p[0] : d.pro_slope;

    {   // resident weights: packed [CoutPad][WSG] bf16 rows of this cout tile -> LDS [64][WS];
        // all 16-byte loads of a thread are issued before the first LDS write (<= 6 per thread for 3x3)
        const bf16x8* src = reinterpret_cast<const bf16x8*>(reinterpret_cast<const __bf16*>(d.wpk) +
                                                            (int64_t)cout_base * WSG);
        const int vrow = WSG >> 3, nv = 64 * vrow;
        constexpr int WB = 6;
        for (int v0 = tid; v0 < nv; v0 += WB * PC_THREADS) {
            bf16x8 wv[WB];
#pragma unroll
            for (int u = 0; u < WB; ++u) {
                const int v = v0 + u * PC_THREADS;
                if (v < nv) wv[u] = src[v];
            }
#pragma unroll
            for (int u = 0; u < WB; ++u) {
                const int v = v0 + u * PC_THREADS;
                if (v < nv) {
                    const int j = v / vrow, k8 = v - j * vrow;
                    *reinterpret_cast<bf16x8*>(lds_w + j * WS + k8 * 8) = wv[u];
                }
            }
        }
    }
    auto tile_origin = [&](int t, int& n, int& oy0, int& ox0) {
        const int per_img = p.tiles_x * p.tiles_y;
        n = t / per_img;
        const int r = t - n * per_img;
        const int sx = r / p.tiles_y;
        oy0 = (r - sx * p.tiles_y) * PC_TH;
        ox0 = sx * PC_TW;
    };
    // running BatchNorm statistics of this wave over all its tiles (Chan merge in registers)
    float st_n = 0.f, st_m0 = 0.f, st_q0 = 0.f, st_m1 = 0.f, st_q1 = 0.f;
    const int cp0 = cout_base + l31, cp1 = cp0 + 32;
    const bool col_ok0 = cp0 < d.Cout, col_ok1 = cp1 < d.Cout;

    // Role-separated loops: every wave executes exactly K + 1 workgroup barriers, but loaders and
    // consumers run different code between them, so their register sets do not add up.
    if (loader) {
        // group g (0/1) owns tiles k = g, g+2, ...; tile k lives in LDS buffer k & 1
        const int grp = (wave - 4) >> 2;
        const int lt = tid - 256 - 256 * grp;
        const unsigned xbytes = (unsigned)((int64_t)d.N * d.H * d.W * d.Cin * 4);
        const __amdgpu_buffer_rsrc_t r1 = pc_rsrc(d.x1, xbytes);
        const __amdgpu_buffer_rsrc_t r2 = pc_rsrc(d.x2 ? d.x2 : d.x1, xbytes);
        PcLoader<PRO> ld;
        ld.init(ov, lt, IH, IW);
        int n, oy0, ox0;
        if (grp == 0) {
            if (K > 0) {
                tile_origin(t_begin, n, oy0, ox0);
                ld.issue(ov, r1, r2, n, oy0 - d.pad_y, ox0 - d.pad_x);
                ld.commit(ov, lds_in0, lt);
            }
            if (K > 2) {
                tile_origin(t_begin + 2, n, oy0, ox0);
                ld.issue(ov, r1, r2, n, oy0 - d.pad_y, ox0 - d.pad_x);
            }
        } else if (K > 1) {
            tile_origin(t_begin + 1, n, oy0, ox0);
            ld.issue(ov, r1, r2, n, oy0 - d.pad_y, ox0 - d.pad_x);
        }
        pc_barrier();
        for (int k = 0; k < K; ++k) {
            // during iteration k (consumers on tile k) tile k+1 is committed by its owner group, whose
            // loads were issued two iterations ago; then that group puts tile k+3 in flight
            if (((k + 1) & 1) == grp && k + 1 < K) {
                ld.commit(ov, ((k + 1) & 1) ? lds_in1 : lds_in0, lt);
                if (k + 3 < K) {
                    tile_origin(t_begin + k + 3, n, oy0, ox0);
                    ld.issue(ov, r1, r2, n, oy0 - d.pad_y, ox0 - d.pad_x);
                }
            }
            pc_barrier();
        }
    } else {
        // consumer: wave w owns tile row ty = w (32 pixels) x 64 couts
        const __bf16* bp0 = lds_w + l31 * WS + 8 * kk;
        const __bf16* bp1 = bp0 + 32 * WS;
        const int a_lane = (wave * IW + l31) * PC_PS + 8 * kk;
        float bias0 = 0.f, bias1 = 0.f;
        int col_off0 = cp0, col_off1 = cp1, co0 = cp0, co1 = cp1;
        if (d.y_mode == SISR_Y_NHWC_SHUFFLE2) {
            const int Cq = d.Cout >> 2;
            const int ij0 = cp0 / Cq, c0 = cp0 - ij0 * Cq, ij1 = cp1 / Cq, c1 = cp1 - ij1 * Cq;
            co0 = c0 * 4 + ij0; co1 = c1 * 4 + ij1;
            col_off0 = ((ij0 >> 1) * (2 * d.Wo) + (ij0 & 1)) * Cq + c0;
            col_off1 = ((ij1 >> 1) * (2 * d.Wo) + (ij1 & 1)) * Cq + c1;
        } else if (d.y_mode == SISR_Y_NCHW) {
            col_off0 = cp0 * d.Ho * d.Wo;
            col_off1 = cp1 * d.Ho * d.Wo;
        }
        if (d.bias != nullptr) {
            if (col_ok0) bias0 = d.bias[co0];
            if (col_ok1) bias1 = d.bias[co1];
        }
        const int pstep = d.y_mode == SISR_Y_NHWC_SHUFFLE2 ? 2 * (d.Cout >> 2) : (d.y_mode == SISR_Y_NCHW ? 1 : d.Cout);
        // buffer addressing of the output (and residual): per-lane byte offset fixed for the launch,
        // per-element offsets are wave-uniform scalars
        const unsigned ybytes = (unsigned)((int64_t)d.N * d.Ho * d.Wo * d.Cout * 4);
        const __amdgpu_buffer_rsrc_t ry = pc_rsrc(d.y, ybytes);
        const __amdgpu_buffer_rsrc_t rr = pc_rsrc(d.res ? d.res : d.y, ybytes);
        const unsigned voff0 = (unsigned)((4 * kk * pstep + col_off0) * 4);
        const unsigned voff1 = (unsigned)((4 * kk * pstep + col_off1) * 4);
        pc_barrier();
        for (int k = 0; k < K; ++k) {
            const int t = t_begin + k;
            const __bf16* cur = (k & 1) ? lds_in1 : lds_in0;
            int n, oy0, ox0;
            tile_origin(t, n, oy0, ox0);
            f32x16 acc0, acc1;
#pragma unroll
            for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
            const __bf16* ap = cur + a_lane;
            int tap = 0;
            for (int r = 0; r < d.KH; ++r)
                for (int sx = 0; sx < d.KW; ++sx, ++tap) {
                    const __bf16* a_t = ap + (r * IW + sx) * PC_PS;
                    const int boff = tap * PC_CIN;
#pragma unroll
                    for (int kh = 0; kh < 2; ++kh) {       // 64 channels = 2 groups of two K=16 steps
                        bf16x8 a[2], b0[2], b1[2];
#pragma unroll
                        for (int kc = 0; kc < 2; ++kc) {
                            a[kc] = *reinterpret_cast<const bf16x8*>(a_t + (kh * 2 + kc) * 16);
                            b0[kc] = *reinterpret_cast<const bf16x8*>(bp0 + boff + (kh * 2 + kc) * 16);
                            b1[kc] = *reinterpret_cast<const bf16x8*>(bp1 + boff + (kh * 2 + kc) * 16);
                        }
#pragma unroll
                        for (int kc = 0; kc < 2; ++kc) {
                            acc0 = pc_mfma(a[kc], b0[kc], acc0);
                            acc1 = pc_mfma(a[kc], b1[kc], acc1);
                        }
                        __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                    }
                }
            // ---- epilogue of this wave's 32 pixels (tile row `wave`) ------------------------------
            const int oy = oy0 + wave;
            const bool row_ok = oy < d.Ho;
            const int vw = min(PC_TW, d.Wo - ox0);
            const bool full = vw == PC_TW;
#pragma unroll
            for (int i = 0; i < 16; ++i) { acc0[i] += bias0; acc1[i] += bias1; }
            if (d.stat_part != nullptr && row_ok) {
                const float cnt = (float)vw;
                float s0 = 0.f, s1 = 0.f;
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    if (full || mfma_row(i, lane) < vw) { s0 += acc0[i]; s1 += acc1[i]; }
                s0 += __shfl_xor(s0, 32); s1 += __shfl_xor(s1, 32);
                const float m0 = s0 / cnt, m1 = s1 / cnt;
                float q0 = 0.f, q1 = 0.f;
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    if (full || mfma_row(i, lane) < vw) {
                        const float dv0 = acc0[i] - m0, dv1 = acc1[i] - m1;
                        q0 += dv0 * dv0; q1 += dv1 * dv1;
                    }
                q0 += __shfl_xor(q0, 32); q1 += __shfl_xor(q1, 32);
                // merge (cnt, m, q) of this tile row into the wave's running statistics
                const float nn = st_n + cnt, f = cnt / nn, g = st_n * f;
                const float d0 = m0 - st_m0, d1 = m1 - st_m1;
                st_m0 += d0 * f; st_q0 += q0 + d0 * d0 * g;
                st_m1 += d1 * f; st_q1 += q1 + d1 * d1 * g;
                st_n = nn;
            }
            if (row_ok) {
                unsigned rbase;                            // wave-uniform byte offset of (n, oy, ox0)
                if (d.y_mode == SISR_Y_NHWC_SHUFFLE2)
                    rbase = (unsigned)(((n * 2 * d.Ho + 2 * oy) * (2 * d.Wo) + 2 * ox0) * (d.Cout >> 2)) * 4u;
                else if (d.y_mode == SISR_Y_NCHW)
                    rbase = (unsigned)(n * d.Cout * d.Ho * d.Wo + oy * d.Wo + ox0) * 4u;
                else
                    rbase = (unsigned)(((n * d.Ho + oy) * d.Wo + ox0) * d.Cout) * 4u;
                const unsigned ps4 = (unsigned)pstep * 4u;
                if (d.res != nullptr) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const unsigned so = rbase + (unsigned)((i & 3) + 8 * (i >> 2)) * ps4;
                        if (full || mfma_row(i, lane) < vw) {
                            if (col_ok0) acc0[i] += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rr, voff0, so, 0));
                            if (col_ok1) acc1[i] += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rr, voff1, so, 0));
                        }
                    }
                }
                if (epi_act == SISR_EPI_TANH) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) { acc0[i] = tanhf(acc0[i]); acc1[i] = tanhf(acc1[i]); }
                }
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const unsigned so = rbase + (unsigned)((i & 3) + 8 * (i >> 2)) * ps4;
                    if (full || mfma_row(i, lane) < vw) {
                        if (col_ok0) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint((float)acc0[i]), ry, voff0, so, 0);
                        if (col_ok1) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint((float)acc1[i]), ry, voff1, so, 0);
                    }
                }
            }
            pc_barrier();   // tile k consumed, tile k+1 staged
        }
    }
    if (d.stat_part != nullptr) {
        // one (count, mean, M2) partial per workgroup: waves 1-3 publish through LDS, wave 0 merges
        float* sh = reinterpret_cast<float*>(lds_in0);       // [3][5][64] floats, tile buffers are free now
        if (!loader && wave > 0 && kk == 0) {
            float* q = sh + (wave - 1) * 5 * 64;
            q[l31] = st_m0; q[64 + l31] = st_q0; q[128 + l31] = st_m1; q[192 + l31] = st_q1; q[256 + l31] = st_n;
        }
        __syncthreads();
        if (wave == 0 && kk == 0) {
            for (int w = 0; w < 3; ++w) {
                const float* q = sh + w * 5 * 64;
                const float nb = q[256 + l31];
                if (nb > 0.f) {
                    const float nn = st_n + nb, f = nb / nn, g = st_n * f;
                    const float d0 = q[l31] - st_m0, d1 = q[128 + l31] - st_m1;
                    st_m0 += d0 * f; st_q0 += q[64 + l31] + d0 * d0 * g;
                    st_m1 += d1 * f; st_q1 += q[192 + l31] + d1 * d1 * g;
                    st_n = nn;
                }
            }
            float* sp = d.stat_part + (int64_t)blockIdx.x * 2 * d.Cout;
            if (col_ok0) { sp[cout_base + l31] = st_m0; sp[d.Cout + cout_base + l31] = st_q0; }
            if (col_ok1) { sp[cout_base + 32 + l31] = st_m1; sp[d.Cout + cout_base + 32 + l31] = st_q1; }
            if (lane == 0 && blockIdx.y == 0) d.cnt_part[blockIdx.x] = st_n;
        }
    }
}

// ---- host ------------------------------------------------------------------------------------------
static int pc_lds_bytes(int KH, int KW) {
    const int taps = KH * KW, WS = taps * PC_CIN + 8;
    const int IH = PC_TH + KH - 1, IW = PC_TW + KW - 1;
    const int tile_elems = (IH * IW * PC_PS + 7) & ~7;
    return (64 * WS + 2 * tile_elems) * 2 + 64;
}

// variant 1 of sisr_conv2d_plan_bf16: called from conv_bf16.hip's planner when the layer qualifies
extern "C" int sisr_conv2d_plan_bf16_persist(SisrConvDesc* d) {
    if (!d || d->Cin != PC_CIN || d->stride != 1 || d->KH * d->KW > 9) return SISR_E_UNSUPPORTED;
    if (d->x_mode == SISR_X_NCHW) return SISR_E_UNSUPPORTED;
    if (d->y_sy != 1 || d->y_sx != 1 || d->y_oy != 0 || d->y_ox != 0 || d->y_H != d->Ho || d->y_W != d->Wo)
        return SISR_E_UNSUPPORTED;
    const int lds = pc_lds_bytes(d->KH, d->KW);
    if (lds > 160 * 1024) return SISR_E_UNSUPPORTED;
    const int64_t ypix = (int64_t)d->N * d->Ho * d->Wo;
    // 32-bit buffer offsets: tensors below 4 GB (2^30 fp32 elements)
    if (ypix * d->Cout >= (1ll << 30) || (int64_t)d->N * d->H * d->W * d->Cin >= (1ll << 30)) return SISR_E_UNSUPPORTED;
    SisrConvPlan& p = d->plan;
    std::memset(&p, 0, sizeof(p));
    p.variant = 1;
    p.TH = PC_TH; p.TW = PC_TW; p.TN = 1;
    p.tiles_y = (d->Ho + PC_TH - 1) / PC_TH;
    p.tiles_x = (d->Wo + PC_TW - 1) / PC_TW;
    p.n_groups = d->N;
    p.CoutPad = (d->Cout + 63) / 64 * 64;
    // BN-statistics partials: one per workgroup (grid.x)
    p.n_tiles = std::max(1, std::min(p.tiles_y * p.tiles_x * d->N, 256 / (p.CoutPad / 64)));
    p.CK = PC_CIN; p.PS = PC_PS; p.KROWP = d->KH * d->KW * PC_CIN; p.n_chunk = 1;
    p.nsub = 2; p.msub = 1;
    p.CoutPad = (d->Cout + 63) / 64 * 64;
    p.lds_bytes = lds;
    p.wpk_elems = p.CoutPad * p.KROWP;                  // bf16 elements, layout [cout][tap*64 + ci]
    return 0;
}

template <int PRO>
static int launch_pc(const SisrConvDesc* d, hipStream_t st) {
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_c64_bf16_persist_kernel<PRO>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        attr = true;
    }
    const int cout_tiles = d->plan.CoutPad / 64;
    const int gx = d->plan.n_tiles;
    hipLaunchKernelGGL(conv_c64_bf16_persist_kernel<PRO>, dim3(gx, cout_tiles), dim3(PC_THREADS), d->plan.lds_bytes,
                       st, *d);
    SISR_CHECK_LAUNCH();
    return 0;
}

extern "C" int sisr_conv2d_bf16_persist(const SisrConvDesc* d, void* stream) {
    if (d && d->bnb_part) return SISR_E_UNSUPPORTED;     // fused BatchNorm-backward partials: generic bf16 kernel only
    if (!d || !d->x1 || !d->wpk || !d->y || d->plan.variant != 1) return SISR_E_BADARG;
    if (operand_needs_x2(d->pro_mode) && !d->x2) return SISR_E_BADARG;
    if (d->stat_part && (!d->cnt_part || d->y_mode != SISR_Y_NHWC)) return SISR_E_BADARG;
    if (d->x_mode == SISR_X_NCHW || d->Cin != PC_CIN) return SISR_E_UNSUPPORTED;
    if (d->x_mode == SISR_X_NHWC_UNSHUFFLE2 && ((d->Cin >> 2) & 3)) return SISR_E_UNSUPPORTED;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    switch (d->pro_mode) {
        case SISR_PRO_NONE: return launch_pc<SISR_PRO_NONE>(d, st);
        case SISR_PRO_ACT: return launch_pc<SISR_PRO_ACT>(d, st);
        case SISR_PRO_AFFINE_ACT: return launch_pc<SISR_PRO_AFFINE_ACT>(d, st);
        case SISR_PRO_BNBWD: return launch_pc<SISR_PRO_BNBWD>(d, st);
        case SISR_PRO_BNACT_BWD: return launch_pc<SISR_PRO_BNACT_BWD>(d, st);
        case SISR_PRO_ACT_BWD: return launch_pc<SISR_PRO_ACT_BWD>(d, st);
        case SISR_PRO_TANH_BWD: return launch_pc<SISR_PRO_TANH_BWD>(d, st);
    }
    return SISR_E_BADARG;
}
