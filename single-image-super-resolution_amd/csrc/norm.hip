// norm.hip -- BatchNorm2d (training mode) pieces that are not fused into the convolutions, the
// residual/BatchNorm-apply elementwise pass, and the small reductions of the backward pass.
// Replaces nn.BatchNorm2d fwd/bwd (model_generator.py:11,14,40; model_discriminator.py:11), the
// shared-slope nn.PReLU gradient (model_generator.py:12,34,48) and the residual adds
// (model_generator.py:19,93).  All reductions are deterministic (no atomics): wave shuffles ->
// LDS -> per-workgroup partials -> a single finishing workgroup.
#include "sisr_dev.h"

#include <algorithm>

// ---- forward statistics: Chan-merge of the per-tile (count, mean, M2) partials ------------------
// Merge of the per-tile (count, mean_b, M2_b) partials in two fully parallel passes (no serial chain):
//   N = sum n_b ; mean = sum n_b*mean_b / N ; M2 = sum [ M2_b + n_b*(mean_b - mean)^2 ]
// (algebraically Chan et al.'s pairwise update summed over all tiles; all sums in double).
// Workgroup = 4 channels x 256 tile-splits (1024 threads): C/4 workgroups, a few tiles per thread.
#ifndef BNF_SPLITS
#define BNF_SPLITS 256
#define BNF_CH 4
#endif
// sum over the 256 tile-splits of each of the 4 channels: lanes of one channel combine by shuffles (thread =
// channel + 4 * split, so a wave holds 16 splits of every channel), the 16 waves through LDS; fixed order
__device__ __forceinline__ void bnf_reduce2(double& a, double& b, double (*sh)[2][BNF_CH], int cl) {
#pragma unroll
    for (int o = BNF_CH; o < 64; o <<= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
    const int wave = threadIdx.x >> 6;
    __syncthreads();                                   // previous use of sh is over
    if ((threadIdx.x & 63) < BNF_CH) { sh[wave][0][cl] = a; sh[wave][1][cl] = b; }
    __syncthreads();
    a = 0.0; b = 0.0;
#pragma unroll
    for (int w = 0; w < (BNF_SPLITS * BNF_CH) / 64; ++w) { a += sh[w][0][cl]; b += sh[w][1][cl]; }
}

__global__ void __launch_bounds__(BNF_SPLITS * BNF_CH) bn_finalize_kernel(
    const float* __restrict__ stat_part, const float* __restrict__ cnt_part, int n_tiles, int C,
    const float* __restrict__ gamma, const float* __restrict__ beta, float* running_mean,
    float* running_var, float momentum, float eps, float* scale, float* shift, float* save_mean,
    float* save_invstd) {
    __shared__ double sh[(BNF_SPLITS * BNF_CH) / 64][2][BNF_CH];
    const int cl = threadIdx.x & (BNF_CH - 1), split = threadIdx.x / BNF_CH;
    const int c = min(blockIdx.x * BNF_CH + cl, C - 1);
    double n = 0.0, s = 0.0;
    for (int t = split; t < n_tiles; t += BNF_SPLITS) {
        const double nb = cnt_part[t];
        n += nb;
        s += nb * (double)stat_part[(int64_t)t * 2 * C + c];
    }
    bnf_reduce2(n, s, sh, cl);
    const double mean = s / n;
    double m2 = 0.0, unused = 0.0;
    for (int t = split; t < n_tiles; t += BNF_SPLITS) {
        const double nb = cnt_part[t];
        const double dm = (double)stat_part[(int64_t)t * 2 * C + c] - mean;
        m2 += (double)stat_part[(int64_t)t * 2 * C + C + c] + nb * dm * dm;
    }
    bnf_reduce2(m2, unused, sh, cl);
    if (split == 0 && blockIdx.x * BNF_CH + cl < C) {
        const double var = m2 / n;                       // biased (normalisation)
        const float invstd = (float)(1.0 / sqrt(var + (double)eps));
        const float sc = gamma[c] * invstd;
        scale[c] = sc;
        shift[c] = beta[c] - (float)mean * sc;
        save_mean[c] = (float)mean;
        save_invstd[c] = invstd;
        const double unb = n > 1.0 ? m2 / (n - 1.0) : var;   // unbiased (running estimate)
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
    }
}

__global__ void bn_eval_consts_kernel(const float* gamma, const float* beta, const float* rm,
                                      const float* rv, float eps, int C, float* scale, float* shift) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) {
        const float sc = gamma[c] / sqrtf(rv[c] + eps);
        scale[c] = sc;
        shift[c] = beta[c] - rm[c] * sc;
    }
}

// ---- backward reductions ------------------------------------------------------------------------
// thread = (channel group of 4, pixel lane); workgroup partials work[blk][2*C+1]
__global__ void __launch_bounds__(SISR_BLOCK) bn_bwd_reduce_kernel(const SisrBnBwdDesc d) {
    extern __shared__ __attribute__((aligned(16))) float sh[];   // [PL][2*C + PLpad]
    __shared__ float scratch[8];
    const int G = d.C >> 2, PL = SISR_BLOCK / G;
    const int g = threadIdx.x % G, pl = threadIdx.x / G;
    const int c = g * 4;
    f32x4 sc, sf, mu, is;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        sc[j] = d.scale[c + j]; sf[j] = d.shift[c + j]; mu[j] = d.mean[c + j]; is[j] = d.invstd[c + j];
    }
    f32x4 sg = {0.f, 0.f, 0.f, 0.f}, sgx = {0.f, 0.f, 0.f, 0.f};
    float ssl = 0.f;
    const float slope = d.slope_p ? d.slope_p[0] : d.slope;
    for (int64_t p = (int64_t)blockIdx.x * PL + pl; p < d.P; p += (int64_t)gridDim.x * PL) {
        const int64_t i4 = (p * d.C + c) >> 2;
        const f32x4 dy = d.dy_bf16 ? ld4<true>(d.dy, i4) : ld4<false>(d.dy, i4);
        const f32x4 x = d.x_bf16 ? ld4<true>(d.x, i4) : ld4<false>(d.x, i4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float gg = dy[j];
            if (d.act_mode) {
                const float z = sc[j] * x[j] + sf[j];
                if (!(z > 0.f)) { ssl += dy[j] * z; gg = slope * dy[j]; }
            }
            sg[j] += gg;
            sgx[j] += gg * ((x[j] - mu[j]) * is[j]);
        }
    }
    // reduce over the PL pixel lanes through LDS
    float* a = sh;   // [PL][2*C]
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        a[pl * 2 * d.C + c + j] = sg[j];
        a[pl * 2 * d.C + d.C + c + j] = sgx[j];
    }
    const float tot_sl = block_sum(ssl, scratch);   // contains the barrier that publishes `a`
    float* wk = d.work + (int64_t)blockIdx.x * (2 * d.C + 1);
    for (int i = threadIdx.x; i < 2 * d.C; i += SISR_BLOCK) {
        float s = 0.f;
        for (int k = 0; k < PL; ++k) s += a[k * 2 * d.C + i];
        wk[i] = s;
    }
    if (threadIdx.x == 0) wk[2 * d.C] = tot_sl;
}

#ifndef BWF_CH
#define BWF_CH 4
#endif
__device__ __forceinline__ void bn_bwd_finalize_block(const SisrBnBwdDesc& d, int block) {
    // workgroup = 4 channels x 64 row-splits over the per-workgroup partial rows of the reduce kernel: a thread
    // sums ~grid/64 rows with independent loads (double accumulation), lanes of one channel combine by
    // shuffles, the 4 waves through LDS -- fixed order, deterministic
    __shared__ double sh[2][4][BWF_CH];
    __shared__ float scratch[8];
    const int stride = 2 * d.C + 1;
    const double inv_n = 1.0 / (double)d.P;
    const int cl = threadIdx.x & (BWF_CH - 1), split = threadIdx.x / BWF_CH, wave = threadIdx.x >> 6;
    const int c = block * BWF_CH + cl;
    double s1 = 0.0, s2 = 0.0;
    if (c < d.C) {
#pragma unroll 4
        for (int b = split; b < d.grid; b += SISR_BLOCK / BWF_CH) {
            s1 += (double)d.work[(int64_t)b * stride + c];
            s2 += (double)d.work[(int64_t)b * stride + d.C + c];
        }
    }
#pragma unroll
    for (int o = BWF_CH; o < 64; o <<= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
    if ((threadIdx.x & 63) < BWF_CH) { sh[0][wave][cl] = s1; sh[1][wave][cl] = s2; }
    __syncthreads();
    if (threadIdx.x < BWF_CH && c < d.C) {
        s1 = sh[0][0][cl] + sh[0][1][cl] + sh[0][2][cl] + sh[0][3][cl];
        s2 = sh[1][0][cl] + sh[1][1][cl] + sh[1][2][cl] + sh[1][3][cl];
        const float ga = d.gamma[c], is = d.invstd[c], mu = d.mean[c];
        const float qa = ga * is;
        const float qb = (float)(-(double)ga * is * is * (s2 * inv_n));
        d.qa[c] = qa;
        d.qb[c] = qb;
        d.qd[c] = (float)(-(double)qa * (s1 * inv_n)) - qb * mu;
        d.dgamma[c] = (float)s2;
        d.dbeta[c] = (float)s1;
    }
    if (d.dslope != nullptr && block == 0) {
        float part = 0.f;
        for (int b = threadIdx.x; b < d.grid; b += SISR_BLOCK) part += d.work[(int64_t)b * stride + 2 * d.C];
        const float tot = block_sum(part, scratch);
        if (threadIdx.x == 0) d.dslope[0] = tot;
    }
}

__global__ void __launch_bounds__(SISR_BLOCK) bn_bwd_finalize_kernel(const SisrBnBwdDesc d) { bn_bwd_finalize_block(d, blockIdx.x); }

// one launch, two independent jobs: workgroups [0, fin_blocks) finish the BatchNorm backward, the rest sum the slabs of the
// weight gradient computed just before (sisr_bn_bwd_finalize_slab)
__global__ void __launch_bounds__(SISR_BLOCK) bn_bwd_finalize_slab_kernel(const SisrBnBwdDesc d, int fin_blocks,
                                                                         const float* __restrict__ slab, float* __restrict__ out,
                                                                         int n_slabs, int64_t elems, int64_t lead) {
    if ((int)blockIdx.x < fin_blocks) {
        bn_bwd_finalize_block(d, blockIdx.x);
    } else {
        __shared__ f32x4 sh[SR_SPLITS][SR_COLS];
        slab_reduce_block(slab, out, n_slabs, elems, (int)blockIdx.x - fin_blocks, sh, lead);
    }
}

// ---- elementwise --------------------------------------------------------------------------------
template <bool X1B, bool X2B, bool YB>
__global__ void eltwise_res_affine_kernel(const float* __restrict__ x1, const float* slope1_p, float slope1,
                                          const float* __restrict__ x2, const float* __restrict__ pa,
                                          const float* __restrict__ pd, float* __restrict__ y,
                                          int64_t n4, int C) {
    if (slope1_p != nullptr) slope1 = slope1_p[0];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
         i += (int64_t)gridDim.x * blockDim.x) {
        const f32x4 a = ld4<X1B>(x1, i);
        f32x4 r;
#pragma unroll
        for (int j = 0; j < 4; ++j) r[j] = lrelu(a[j], slope1);
        if (x2 != nullptr) {
            const f32x4 b = ld4<X2B>(x2, i);
            if (pa != nullptr) {
                const int c = (int)((i * 4) % C);
#pragma unroll
                for (int j = 0; j < 4; ++j) r[j] += pa[c + j] * b[j] + pd[c + j];
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) r[j] += b[j];
            }
        }
        st4<YB>(y, i, r);
    }
}

// all three tensors bf16, 8 | C, C | 2048: 16-byte accesses (8 elements per thread and step).  The grid stride is a
// multiple of C / 8, so a thread meets the same 8 channels at every step and keeps their scale / shift in registers.
__global__ void __launch_bounds__(256) eltwise_res_affine_bf16x8_kernel(const float* __restrict__ x1, const float* slope1_p,
                                                                        float slope1, const float* __restrict__ x2,
                                                                        const float* __restrict__ pa,
                                                                        const float* __restrict__ pd, float* __restrict__ y,
                                                                        int64_t n8, int C) {
    if (slope1_p != nullptr) slope1 = slope1_p[0];
    const int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int c = (int)((i0 * 8) % C);
    f32x8 ka, kd;
#pragma unroll
    for (int j = 0; j < 8; ++j) { ka[j] = pa != nullptr ? pa[c + j] : 1.f; kd[j] = pa != nullptr ? pd[c + j] : 0.f; }
    const int64_t step = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = i0; i < n8; i += step) {
        const f32x8 a = ld8_bf16(x1, i);
        f32x8 r;
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = lrelu(a[j], slope1);
        if (x2 != nullptr) {
            const f32x8 b = ld8_bf16(x2, i);
#pragma unroll
            for (int j = 0; j < 8; ++j) r[j] += ka[j] * b[j] + kd[j];
        }
        st8_bf16(y, i, r);
    }
}

template <bool GB, bool PB>
__global__ void __launch_bounds__(SISR_BLOCK) prelu_slope_partial_kernel(const float* __restrict__ dy,
                                                                         const float* __restrict__ pre,
                                                                         int64_t n, float* work) {
    __shared__ float scratch[8];
    float s = 0.f;
    const int64_t n4 = n >> 2;                               // 16-byte loads; buffers come from the allocator (aligned)
    for (int64_t i = (int64_t)blockIdx.x * SISR_BLOCK + threadIdx.x; i < n4; i += (int64_t)gridDim.x * SISR_BLOCK) {
        const f32x4 p = ld4<PB>(pre, i), g = ld4<GB>(dy, i);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (!(p[j] > 0.f)) s += g[j] * p[j];
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const int64_t i = (n4 << 2) + threadIdx.x;
        const float p = ld_elem(pre, i, PB);
        if (!(p > 0.f)) s += ld_elem(dy, i, GB) * p;
    }
    const float t = block_sum(s, scratch);
    if (threadIdx.x == 0) work[blockIdx.x] = t;
}

__global__ void __launch_bounds__(SISR_BLOCK) sum_partials_kernel(const float* work, int n, float* out) {
    __shared__ float scratch[8];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += SISR_BLOCK) s += work[i];
    const float t = block_sum(s, scratch);
    if (threadIdx.x == 0) out[0] = t;
}

template <bool AB, bool BB, bool YB>
__global__ void add_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ y,
                           int64_t n) {
    const int64_t n4 = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x)
        st4<YB>(y, i, ld4<AB>(a, i) + ld4<BB>(b, i));
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const int64_t i = (n4 << 2) + threadIdx.x;
        st_elem(y, i, YB, ld_elem(a, i, AB) + ld_elem(b, i, BB));
    }
}

// ---- C ABI ----------------------------------------------------------------------------------------
static inline hipStream_t S_(void* s) { return reinterpret_cast<hipStream_t>(s); }

extern "C" int sisr_bn_finalize(const float* stat_part, const float* cnt_part, int32_t n_tiles, int32_t C,
                                const float* gamma, const float* beta, float* running_mean,
                                float* running_var, float momentum, float eps, float* scale, float* shift,
                                float* save_mean, float* save_invstd, void* stream) {
    if (!stat_part || !cnt_part || n_tiles <= 0 || C <= 0 || !gamma || !beta || !running_mean || !running_var ||
        !scale || !shift || !save_mean || !save_invstd)
        return SISR_E_BADARG;
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + BNF_CH - 1) / BNF_CH), dim3(BNF_SPLITS * BNF_CH), 0, S_(stream), stat_part,
                       cnt_part, n_tiles, C, gamma, beta, running_mean, running_var, momentum, eps, scale, shift,
                       save_mean, save_invstd);
    SISR_CHECK_LAUNCH();
    return 0;
}

extern "C" int sisr_bn_eval_consts(const float* gamma, const float* beta, const float* running_mean,
                                   const float* running_var, float eps, int32_t C, float* scale, float* shift,
                                   void* stream) {
    if (!gamma || !beta || !running_mean || !running_var || !scale || !shift || C <= 0) return SISR_E_BADARG;
    hipLaunchKernelGGL(bn_eval_consts_kernel, dim3((C + 255) / 256), dim3(256), 0, S_(stream), gamma, beta,
                       running_mean, running_var, eps, C, scale, shift);
    SISR_CHECK_LAUNCH();
    return 0;
}

extern "C" int sisr_bn_bwd_plan(SisrBnBwdDesc* d) {
    if (!d || d->C < 4 || (d->C & 3) || d->P <= 0) return SISR_E_BADARG;
    const int G = d->C >> 2;
    if (G > SISR_BLOCK || (SISR_BLOCK % G) != 0) return SISR_E_UNSUPPORTED;
    const int PL = SISR_BLOCK / G;
    const int64_t want = (d->P + (int64_t)PL * 16 - 1) / ((int64_t)PL * 16);   // >= 16 pixels per thread
    d->grid = (int)std::max<int64_t>(1, std::min<int64_t>(want, 512));
    return 0;
}

extern "C" int sisr_bn_bwd(const SisrBnBwdDesc* d, void* stream) {
    if (!d || !d->dy || !d->x || !d->scale || !d->shift || !d->mean || !d->invstd || !d->gamma || !d->work ||
        !d->qa || !d->qb || !d->qd || !d->dgamma || !d->dbeta || d->grid <= 0)
        return SISR_E_BADARG;
    const int G = d->C >> 2, PL = SISR_BLOCK / G;
    const int lds = PL * 2 * d->C * 4;
    if (lds > 64 * 1024) return SISR_E_UNSUPPORTED;
    hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(d->grid), dim3(SISR_BLOCK), lds, S_(stream), *d);
    SISR_CHECK_LAUNCH();
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((d->C + BWF_CH - 1) / BWF_CH), dim3(SISR_BLOCK), 0, S_(stream), *d);
    SISR_CHECK_LAUNCH();
    return 0;
}

extern "C" int sisr_bn_bwd_finalize(const SisrBnBwdDesc* d, void* stream) {
    if (!d || !d->invstd || !d->mean || !d->gamma || !d->work || !d->qa || !d->qb || !d->qd || !d->dgamma || !d->dbeta ||
        d->grid <= 0 || d->C <= 0 || d->P <= 0)
        return SISR_E_BADARG;
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((d->C + BWF_CH - 1) / BWF_CH), dim3(SISR_BLOCK), 0, S_(stream), *d);
    SISR_CHECK_LAUNCH();
    return 0;
}

extern "C" int sisr_bn_bwd_finalize_slab(const SisrBnBwdDesc* d, const float* slab, float* out, int32_t n_slabs, int64_t elems,
                                         int64_t lead_bf16, void* stream) {
    if (!d || !d->invstd || !d->mean || !d->gamma || !d->work || !d->qa || !d->qb || !d->qd || !d->dgamma || !d->dbeta ||
        d->grid <= 0 || d->C <= 0 || d->P <= 0)
        return SISR_E_BADARG;
    if (!slab || !out || n_slabs <= 0 || elems <= 0 || (elems & 3) || (lead_bf16 & 3) || lead_bf16 < 0 || lead_bf16 > elems) return SISR_E_BADARG;
    const int fin_blocks = (d->C + BWF_CH - 1) / BWF_CH;
    const int slab_blocks = (int)((elems / 4 + SR_COLS - 1) / SR_COLS);
    hipLaunchKernelGGL(bn_bwd_finalize_slab_kernel, dim3(fin_blocks + slab_blocks), dim3(SISR_BLOCK), 0, S_(stream), *d, fin_blocks,
                       slab, out, n_slabs, elems, lead_bf16);
    SISR_CHECK_LAUNCH();
    return 0;
}

extern "C" int sisr_eltwise_res_affine(const float* x1, const float* slope1_p, float slope1, const float* x2,
                                       const float* pa, const float* pd, float* y, int64_t P, int32_t C, int32_t dt,
                                       void* stream) {
    if (!x1 || !y || P <= 0 || C <= 0 || (C & 3) || (pa && !pd) || (pa && !x2)) return SISR_E_BADARG;
    const int64_t n4 = P * C / 4;
    const bool x2b = x2 ? (dt & 2) != 0 : (dt & 1) != 0;       // no second operand: its flag does not matter
    const int key = (dt & 1) | (x2b ? 2 : 0) | (dt & 4);
    if (key == 7 && !(C & 7) && (2048 % C) == 0) {
        const int64_t n8 = n4 / 2;
        // 2048 blocks x 256 threads: the grid stride (a multiple of 2048 octets) keeps a thread on the same channels
        const int blocks = (int)std::min<int64_t>((n8 + 255) / 256, 2048);
        hipLaunchKernelGGL(eltwise_res_affine_bf16x8_kernel, dim3(blocks), dim3(256), 0, S_(stream), x1, slope1_p, slope1,
                           x2, pa, pd, y, n8, C);
        SISR_CHECK_LAUNCH();
        return 0;
    }
    const int blocks = (int)std::min<int64_t>((n4 + 255) / 256, 4096);
#define SISR_ELT_CASE(K, A, B, Y)                                                                                      \
    case K:                                                                                                           \
        hipLaunchKernelGGL((eltwise_res_affine_kernel<A, B, Y>), dim3(blocks), dim3(256), 0, S_(stream), x1, slope1_p, \
                           slope1, x2, pa, pd, y, n4, C);                                                             \
        break;
    switch (key) {
        SISR_ELT_CASE(0, false, false, false) SISR_ELT_CASE(1, true, false, false) SISR_ELT_CASE(2, false, true, false)
        SISR_ELT_CASE(3, true, true, false) SISR_ELT_CASE(4, false, false, true) SISR_ELT_CASE(5, true, false, true)
        SISR_ELT_CASE(6, false, true, true) SISR_ELT_CASE(7, true, true, true)
    }
#undef SISR_ELT_CASE
    SISR_CHECK_LAUNCH();
    return 0;
}

extern "C" int sisr_prelu_slope_grad(const float* dy, const float* pre, int64_t n, float* work, float* out,
                                     int32_t dt, void* stream) {
    if (!dy || !pre || !work || !out || n <= 0) return SISR_E_BADARG;
    if ((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(pre)) & 15) return SISR_E_BADARG;   // 16-byte loads
    const int blocks = (int)std::min<int64_t>((n + 8191) / 8192, 1024);
    switch (dt & 3) {
        case 0: hipLaunchKernelGGL((prelu_slope_partial_kernel<false, false>), dim3(blocks), dim3(SISR_BLOCK), 0, S_(stream), dy, pre, n, work); break;
        case 1: hipLaunchKernelGGL((prelu_slope_partial_kernel<true, false>), dim3(blocks), dim3(SISR_BLOCK), 0, S_(stream), dy, pre, n, work); break;
        case 2: hipLaunchKernelGGL((prelu_slope_partial_kernel<false, true>), dim3(blocks), dim3(SISR_BLOCK), 0, S_(stream), dy, pre, n, work); break;
        default: hipLaunchKernelGGL((prelu_slope_partial_kernel<true, true>), dim3(blocks), dim3(SISR_BLOCK), 0, S_(stream), dy, pre, n, work); break;
    }
    SISR_CHECK_LAUNCH();
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(SISR_BLOCK), 0, S_(stream), work, blocks, out);
    SISR_CHECK_LAUNCH();
    return 0;
}

extern "C" int sisr_add(const float* a, const float* b, float* y, int64_t n, int32_t dt, void* stream) {
    if (!a || !b || !y || n <= 0) return SISR_E_BADARG;
    if ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(y)) & 15) return SISR_E_BADARG;
    const int blocks = (int)std::min<int64_t>((n / 4 + 255) / 256 + 1, 4096);
    if ((dt & 7) == 0) hipLaunchKernelGGL((add_kernel<false, false, false>), dim3(blocks), dim3(256), 0, S_(stream), a, b, y, n);
    else if ((dt & 7) == 7) hipLaunchKernelGGL((add_kernel<true, true, true>), dim3(blocks), dim3(256), 0, S_(stream), a, b, y, n);
    else if ((dt & 7) == 4) hipLaunchKernelGGL((add_kernel<false, false, true>), dim3(blocks), dim3(256), 0, S_(stream), a, b, y, n);
    else return SISR_E_UNSUPPORTED;
    SISR_CHECK_LAUNCH();
    return 0;
}
