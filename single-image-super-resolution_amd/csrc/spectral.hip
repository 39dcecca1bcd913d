// spectral.hip -- weight-side kernels: spectral-norm power iteration, weight packing for the
// direct-conv kernels, and the weight-gradient epilogue (un-packing + spectral-norm quotient rule).
//
// Restates torch.nn.utils.spectral_norm's hook (SpectralNorm.compute_weight), which the reference
// applies to every trunk convolution of G and every convolution of D (model_generator.py:3,10,13,
// 33,39,45,52,123; model_discriminator.py:2,10,39):
//   training: v <- normalize(W_mat^T u), u <- normalize(W_mat v)  (one iteration, eps 1e-12, in place)
//   sigma = u . (W_mat v);  W = W_orig / sigma;  autograd treats u, v as constants.
// "Multi-tensor": one launch serves every convolution of a network (descriptor table in HBM).
#include "sisr_dev.h"

#include <algorithm>

#include "sisr_bf16_stage.h"

#define SN_EPS 1e-12f

// Power iteration in four short multi-tensor launches (grid = (weights, jobs); surplus workgroups exit), so
// the 512x4608 matrices of the discriminator are spread over the chip instead of one workgroup per matrix:
//   A  t_part[rb][j] = sum_{i in row block rb} W[i][j] u[i]        (workgroup = 16 rows x 1024 columns)
//   B  t = sum_rb t_part                                           (thread per column)
//   C  s'[i] = W[i,:] . t                                          (wave per row, 4 rows per workgroup)
//   D  v = t / max(|t|, eps) ; s = s' / max(|t|, eps) ; u = s / max(|s|, eps) ; sigma = u . s   (one workgroup per weight)
// Evaluation mode skips A/B (C uses the stored v) and D keeps the stored u.  sn_work per weight: [RB][cols] + [rows] floats.
#define SN_RB 16
#define SN_CB 1024
__device__ __forceinline__ int sn_rblocks(int rows) { return (rows + SN_RB - 1) / SN_RB; }

__global__ void __launch_bounds__(SISR_BLOCK) sn_wt_u_kernel(const SisrWeightDesc* table) {
    __shared__ float us[SN_RB];
    const SisrWeightDesc w = table[blockIdx.x];
    if (w.u == nullptr || !w.training) return;
    const int rows = w.Cout, cols = w.Cin * w.KH * w.KW;
    const int ncb = (cols + SN_CB - 1) / SN_CB, job = blockIdx.y;
    if (job >= sn_rblocks(rows) * ncb) return;
    const int rb = job / ncb, cb = job - rb * ncb;
    const int r0 = rb * SN_RB, nr = min(SN_RB, rows - r0);
    if (threadIdx.x < nr) us[threadIdx.x] = w.u[r0 + threadIdx.x];
    __syncthreads();
    if ((cols & 3) == 0 && nr == SN_RB) {
        // whole row block, rows of whole float4s: a thread owns four consecutive columns and has all 16 rows' loads in flight at
        // once (the scalar loop below waits for each row before it asks for the next: the phase was latency-bound at 31 us)
        const int j = cb * SN_CB + threadIdx.x * 4;
        if (j >= cols) return;
        const float* wp = w.w_orig + (int64_t)r0 * cols + j;
        f32x4 v[SN_RB];
#pragma unroll
        for (int i = 0; i < SN_RB; ++i) v[i] = *reinterpret_cast<const f32x4*>(wp + (int64_t)i * cols);
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < SN_RB; ++i) a += v[i] * us[i];             // row order, as the scalar path: same bits
        *reinterpret_cast<f32x4*>(w.sn_work + (int64_t)rb * cols + j) = a;
        return;
    }
    float acc[SN_CB / SISR_BLOCK];
#pragma unroll
    for (int k = 0; k < SN_CB / SISR_BLOCK; ++k) acc[k] = 0.f;
    const int j0 = cb * SN_CB + threadIdx.x;
    const float* wp = w.w_orig + (int64_t)r0 * cols;
    for (int i = 0; i < nr; ++i, wp += cols) {
        const float ui = us[i];
#pragma unroll
        for (int k = 0; k < SN_CB / SISR_BLOCK; ++k) {
            const int j = j0 + k * SISR_BLOCK;
            if (j < cols) acc[k] += wp[j] * ui;
        }
    }
#pragma unroll
    for (int k = 0; k < SN_CB / SISR_BLOCK; ++k) {
        const int j = j0 + k * SISR_BLOCK;
        if (j < cols) w.sn_work[(int64_t)rb * cols + j] = acc[k];
    }
}

__global__ void __launch_bounds__(SISR_BLOCK) sn_t_sum_kernel(const SisrWeightDesc* table) {
    const SisrWeightDesc w = table[blockIdx.x];
    if (w.u == nullptr || !w.training) return;
    const int rows = w.Cout, cols = w.Cin * w.KH * w.KW;
    const int j = blockIdx.y * SISR_BLOCK + threadIdx.x;
    if (j >= cols) return;
    const int nrb = sn_rblocks(rows);
    float t = 0.f;
    for (int rb = 0; rb < nrb; ++rb) t += w.sn_work[(int64_t)rb * cols + j];
    w.sn_work[j] = t;                                   // row block 0 now holds t = W^T u (not yet normalised)
}

// s' = W t (training; the 1/|t| factor is applied by the finishing kernel) or s = W v (evaluation)
__global__ void __launch_bounds__(SISR_BLOCK) sn_w_v_kernel(const SisrWeightDesc* table) {
    const SisrWeightDesc w = table[blockIdx.x];
    if (w.u == nullptr) return;
    const int rows = w.Cout, cols = w.Cin * w.KH * w.KW;
    const int lane = threadIdx.x & 63, i = blockIdx.y * (SISR_BLOCK / 64) + (threadIdx.x >> 6);
    if (i >= rows) return;
    const float* wr = w.w_orig + (int64_t)i * cols;
    const float* vec = w.training ? w.sn_work : w.v;
    float a = 0.f;
    // twelve loads of each operand in flight per round; the adds keep the order of the plain loop (j = lane, lane + 64, ...), so the
    // result is bit-identical to it (the plain loop waited for every load before asking for the next: 72 dependent rounds for the
    // discriminator's 4,608 columns)
    for (int j0 = lane; j0 < cols; j0 += 64 * 12) {
        float wv[12], vv[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) {
            const int jj = min(j0 + 64 * k, cols - 1);
            wv[k] = wr[jj]; vv[k] = vec[jj];
        }
#pragma unroll
        for (int k = 0; k < 12; ++k)
            if (j0 + 64 * k < cols) a += wv[k] * vv[k];
    }
    a = wave_sum(a);
    if (lane == 0) w.sn_work[(int64_t)sn_rblocks(rows) * cols + i] = a;
}

__global__ void __launch_bounds__(SISR_BLOCK) sn_u_finish_kernel(const SisrWeightDesc* table) {
    __shared__ float scratch[8];
    const SisrWeightDesc w = table[blockIdx.x];
    const int tid = threadIdx.x;
    if (w.u == nullptr) {
        if (tid == 0 && w.sigma) { w.sigma[0] = 1.f; w.sigma[1] = 1.f; }
        return;
    }
    const int rows = w.Cout, cols = w.Cin * w.KH * w.KW;
    const float* s = w.sn_work + (int64_t)sn_rblocks(rows) * cols;
    float iv = 1.f, iu = 0.f;
    if (w.training) {
        // v = t / max(|t|, eps) ; s = W v = s' / max(|t|, eps) ; u = s / max(|s|, eps)
        float part = 0.f;
        for (int j = tid; j < cols; j += SISR_BLOCK) part += w.sn_work[j] * w.sn_work[j];
        iv = 1.f / fmaxf(sqrtf(block_sum(part, scratch)), SN_EPS);
        for (int j = tid; j < cols; j += SISR_BLOCK) {
            const float v = w.sn_work[j] * iv;
            w.v[j] = v;
            if (w.v_used) w.v_used[j] = v;
        }
        part = 0.f;
        for (int i = tid; i < rows; i += SISR_BLOCK) { const float si = s[i] * iv; part += si * si; }
        iu = 1.f / fmaxf(sqrtf(block_sum(part, scratch)), SN_EPS);
    } else if (w.v_used) {
        for (int j = tid; j < cols; j += SISR_BLOCK) w.v_used[j] = w.v[j];
    }
    float part = 0.f;
    for (int i = tid; i < rows; i += SISR_BLOCK) {
        const float si = s[i] * iv;
        const float u = w.training ? si * iu : w.u[i];
        part += u * si;
        if (w.training) w.u[i] = u;
        if (w.u_used) w.u_used[i] = u;
    }
    const float sigma = block_sum(part, scratch);
    if (tid == 0 && w.sigma) { w.sigma[0] = sigma; w.sigma[1] = 1.f / sigma; }      // [1]: the conv_deep.hip epilogue scale
}

// packed channel index -> original output channel (PixelShuffle(2) consumers use (i,j)-major order)
__device__ __forceinline__ int unpermute_cout(int cp, int Cout, int shuffle2) {
    if (!shuffle2) return cp;
    const int Cq = Cout >> 2;
    const int ij = cp / Cq, c = cp - ij * Cq;
    return c * 4 + ij;
}

// phase 2: pack W/sigma.  grid (n_weights, parts); grid-stride over the packed elements.
__global__ void __launch_bounds__(SISR_BLOCK) weights_pack_kernel(const SisrWeightDesc* table) {
    const SisrWeightDesc w = table[blockIdx.x];
    const float inv = 1.f / (w.sigma ? w.sigma[0] : 1.f);
    const int64_t stride = (int64_t)gridDim.y * SISR_BLOCK;
    const int64_t start = (int64_t)blockIdx.y * SISR_BLOCK + threadIdx.x;
    if (w.wpk_fwd) {
        // [chunk][r][cp][krow], krow = s*PS + (ci - chunk*CK)
        const int64_t total = (int64_t)w.f_n_chunk * w.KH * w.f_CoutPad * w.f_KROWP;
        for (int64_t e = start; e < total; e += stride) {
            unsigned tq = (unsigned)e;          // (packed images are far below 2^32 elements: 32-bit index arithmetic)
            const int krow = (int)(tq % w.f_KROWP); tq /= w.f_KROWP;
            const int cp = (int)(tq % w.f_CoutPad); tq /= w.f_CoutPad;
            const int r = (int)(tq % w.KH);
            const int chunk = (int)(tq / w.KH);
            const int s = krow / w.f_PS, cl = krow - s * w.f_PS;
            const int ci = chunk * w.f_CK + cl;
            float val = 0.f;
            if (s < w.KW && cl < w.f_CK && ci < w.Cin && cp < w.Cout) {
                const int co = unpermute_cout(cp, w.Cout, w.shuffle2);
                val = w.w_orig[(((int64_t)co * w.Cin + ci) * w.KH + r) * w.KW + s] * inv;
            }
            w.wpk_fwd[e] = val;
        }
    }
    if (w.wpk_dgrad) {
        // data gradient = conv over dy (channels = couts in packed order) with flipped taps:
        // Wd[op = ci][ip = cp][r'][s'] = W[co(cp)][ci][KH-1-r'][KW-1-s']
        const int64_t total = (int64_t)w.d_n_chunk * w.KH * w.d_CoutPad * w.d_KROWP;
        for (int64_t e = start; e < total; e += stride) {
            unsigned tq = (unsigned)e;
            const int krow = (int)(tq % w.d_KROWP); tq /= w.d_KROWP;
            const int op = (int)(tq % w.d_CoutPad); tq /= w.d_CoutPad;
            const int r = (int)(tq % w.KH);
            const int chunk = (int)(tq / w.KH);
            const int s = krow / w.d_PS, il = krow - s * w.d_PS;
            const int ip = chunk * w.d_CK + il;
            float val = 0.f;
            if (s < w.KW && il < w.d_CK && ip < w.Cout && op < w.Cin) {
                const int co = unpermute_cout(ip, w.Cout, w.shuffle2);
                val = w.w_orig[(((int64_t)co * w.Cin + op) * w.KH + (w.KH - 1 - r)) * w.KW + (w.KW - 1 - s)] * inv;
            }
            w.wpk_dgrad[e] = val;
        }
    }
    // the fp32-tensor trunk conv's LDS-order images (SisrWeightDesc.f_ldsimg / d_ldsimg), behind the standard fp32 images
    for (int role = 0; role < 2; ++role) {
        const int mode = role == 0 ? w.f_ldsimg : w.d_ldsimg;
        float* base = role == 0 ? w.wpk_fwd : w.wpk_dgrad;
        if (mode == 0 || base == nullptr) continue;
        const int64_t std_elems = role == 0 ? (int64_t)w.f_n_chunk * w.KH * w.f_CoutPad * w.f_KROWP
                                            : (int64_t)w.d_n_chunk * w.KH * w.d_CoutPad * w.d_KROWP;
        unsigned* dst = reinterpret_cast<unsigned*>(base + std_elems);
        // value of (packed output channel oc, packed input channel ic, tap): the same numbers as the standard images hold
        auto val = [&](int oc, int ic, int tap) {
            const int r = tap / 3, sx = tap - 3 * r;
            if (role == 0) return w.w_orig[(((int64_t)oc * w.Cin + ic) * w.KH + r) * w.KW + sx] * inv;
            return w.w_orig[(((int64_t)ic * w.Cin + oc) * w.KH + (w.KH - 1 - r)) * w.KW + (w.KW - 1 - sx)] * inv;
        };
        for (int64_t e = start; e < SISR_WLDS_WORDS; e += stride) {
            unsigned tq = (unsigned)e;
            const int wd = (int)(tq % 36u); tq /= 36u;
            const int co = (int)(tq & 31u); tq >>= 5;
            const int tap = (int)(tq % 9u); tq /= 9u;
            const int q = (int)(tq & 1u), hc = (int)(tq >> 1);
            unsigned word = 0u;
            if (wd < 32) {
                const int oc = 32 * hc + co;
                if (mode == 1) {
                    word = __float_as_uint(val(oc, 32 * q + wd, tap));
                } else {
                    const int m = wd & 15;
                    const float v0 = val(oc, 32 * q + 2 * m, tap), v1 = val(oc, 32 * q + 2 * m + 1, tap);
                    const __bf16 h0 = (__bf16)v0, h1 = (__bf16)v1;
                    if (wd < 16) {
                        word = (unsigned)__builtin_bit_cast(unsigned short, h0) | ((unsigned)__builtin_bit_cast(unsigned short, h1) << 16);
                    } else {
                        const __bf16 l0 = (__bf16)(v0 - (float)h0), l1 = (__bf16)(v1 - (float)h1);
                        word = (unsigned)__builtin_bit_cast(unsigned short, l0) | ((unsigned)__builtin_bit_cast(unsigned short, l1) << 16);
                    }
                }
            }
            dst[e] = word;
        }
    }
    // bf16 images for conv_bf16.hip: [chunk32][cp][tap*32 + cl]
    if (w.wbf_fwd) {
        __bf16* dst = reinterpret_cast<__bf16*>(w.wbf_fwd);
        const int CKb = w.bf_f_CK;
        const int taps = w.KH * w.KW, WSG = taps * CKb;
        const int64_t total = (int64_t)(w.Cin / CKb) * w.bf_f_CoutPad * WSG;
        for (int64_t e = start; e < total; e += stride) {
            unsigned tq = (unsigned)e;
            const int kidx = (int)(tq % WSG); tq /= WSG;
            const int cp = (int)(tq % w.bf_f_CoutPad);
            const int chunk = (int)(tq / w.bf_f_CoutPad);
            const int tap = kidx / CKb, cl = kidx - tap * CKb;
            const int r = tap / w.KW, sx = tap - r * w.KW, ci = chunk * CKb + cl;
            float val = 0.f;
            if (cp < w.Cout) {
                const int co = unpermute_cout(cp, w.Cout, w.shuffle2);
                val = w.w_orig[(((int64_t)co * w.Cin + ci) * w.KH + r) * w.KW + sx] * inv;
            }
            dst[e] = (__bf16)val;
        }
        if (w.bf_f_lanes) {
            // the same values in the persistent trunk kernels' load order (see SisrWeightDesc.bf_f_lanes), behind the image above
            for (int64_t e = start; e < total; e += stride) {
                unsigned tq = (unsigned)e;
                const int el = (int)(tq & 7); tq >>= 3;
                const int lane = (int)(tq & 63); tq >>= 6;
                const int j = (int)(tq & 3); tq >>= 2;
                const int tap = (int)(tq % 9);
                const int cp = (int)(tq / 9) * 32 + (lane & 31);
                const int ci = (j >> 1) * 32 + (j & 1) * 16 + 8 * (lane >> 5) + el;
                const int r = tap / w.KW, sx = tap - r * w.KW;
                float val = 0.f;
                if (cp < w.Cout) {
                    const int co = unpermute_cout(cp, w.Cout, w.shuffle2);
                    val = w.w_orig[(((int64_t)co * w.Cin + ci) * w.KH + r) * w.KW + sx] * inv;
                }
                dst[total + e] = (__bf16)val;
            }
        }
    }
    if (w.wbf_dgrad) {
        __bf16* dst = reinterpret_cast<__bf16*>(w.wbf_dgrad);
        const int CKb = w.bf_d_CK;
        const int taps = w.KH * w.KW, WSG = taps * CKb;
        const int64_t total = (int64_t)(w.Cout / CKb) * w.bf_d_CoutPad * WSG;
        for (int64_t e = start; e < total; e += stride) {
            unsigned tq = (unsigned)e;
            const int kidx = (int)(tq % WSG); tq /= WSG;
            const int op = (int)(tq % w.bf_d_CoutPad);
            const int chunk = (int)(tq / w.bf_d_CoutPad);
            const int tap = kidx / CKb, il = kidx - tap * CKb;
            const int r = tap / w.KW, sx = tap - r * w.KW, ip = chunk * CKb + il;
            float val = 0.f;
            if (op < w.Cin) {
                const int co = unpermute_cout(ip, w.Cout, w.shuffle2);
                val = w.w_orig[(((int64_t)co * w.Cin + op) * w.KH + (w.KH - 1 - r)) * w.KW + (w.KW - 1 - sx)] * inv;
            }
            dst[e] = (__bf16)val;
        }
        if (w.bf_d_lanes) {
            for (int64_t e = start; e < total; e += stride) {
                unsigned tq = (unsigned)e;
                const int el = (int)(tq & 7); tq >>= 3;
                const int lane = (int)(tq & 63); tq >>= 6;
                const int j = (int)(tq & 3); tq >>= 2;
                const int tap = (int)(tq % 9);
                const int op = (int)(tq / 9) * 32 + (lane & 31);
                const int ip = (j >> 1) * 32 + (j & 1) * 16 + 8 * (lane >> 5) + el;
                const int r = tap / w.KW, sx = tap - r * w.KW;
                float val = 0.f;
                if (op < w.Cin) {
                    const int co = unpermute_cout(ip, w.Cout, w.shuffle2);
                    val = w.w_orig[(((int64_t)co * w.Cin + op) * w.KH + (w.KH - 1 - r)) * w.KW + (w.KW - 1 - sx)] * inv;
                }
                dst[total + e] = (__bf16)val;
            }
        }
    }
    // stride-2 data gradient: one packed image per output parity class (bf16 image for the bf16 kernels)
    for (int cls = 0; cls < 4; ++cls) {
        __bf16* dst = reinterpret_cast<__bf16*>(w.wbf_dcls[cls]);
        if (dst == nullptr) continue;
        const int KHc = w.c_KH[cls], KWc = w.c_KW[cls], WSG = KHc * KWc * 32, CoutPad = w.bf_c_CoutPad[cls];
        const int64_t total = (int64_t)(w.Cout / 32) * CoutPad * WSG;
        for (int64_t e = start; e < total; e += stride) {
            unsigned tq = (unsigned)e;
            const int kidx = (int)(tq % WSG); tq /= WSG;
            const int op = (int)(tq % CoutPad);
            const int chunk = (int)(tq / CoutPad);
            const int tap = kidx >> 5, il = kidx & 31;
            const int rp = tap / KWc, sp = tap - rp * KWc, ip = chunk * 32 + il;
            const int r = w.c_R0y[cls] - 2 * rp, sx = w.c_R0x[cls] - 2 * sp;
            float val = 0.f;
            if (op < w.Cin && r >= 0 && r < w.KH && sx >= 0 && sx < w.KW)
                val = w.w_orig[(((int64_t)ip * w.Cin + op) * w.KH + r) * w.KW + sx] * inv;
            dst[e] = (__bf16)val;
        }
    }
    for (int cls = 0; cls < 4; ++cls) {
        float* dst = w.wpk_dcls[cls];
        if (dst == nullptr) continue;
        const int KHc = w.c_KH[cls], KWc = w.c_KW[cls];
        const int CK = w.c_CK[cls], PS = w.c_PS[cls], KROWP = w.c_KROWP[cls], CoutPad = w.c_CoutPad[cls];
        const int64_t total = (int64_t)w.c_n_chunk[cls] * KHc * CoutPad * KROWP;
        for (int64_t e = start; e < total; e += stride) {
            unsigned tq = (unsigned)e;
            const int krow = (int)(tq % KROWP); tq /= KROWP;
            const int op = (int)(tq % CoutPad); tq /= CoutPad;
            const int rp = (int)(tq % KHc);
            const int chunk = (int)(tq / KHc);
            const int sp = krow / PS, il = krow - sp * PS;
            const int ip = chunk * CK + il;
            const int r = w.c_R0y[cls] - 2 * rp, sx = w.c_R0x[cls] - 2 * sp;
            float val = 0.f;
            if (sp < KWc && il < CK && ip < w.Cout && op < w.Cin && r >= 0 && r < w.KH && sx >= 0 && sx < w.KW)
                val = w.w_orig[(((int64_t)ip * w.Cin + op) * w.KH + r) * w.KW + sx] * inv;
            dst[e] = val;
        }
    }
}

// conv_deep.hip images (SisrWeightDesc.wdp_*): [32-channel chunk][tap row][cout][KW * 32 + 8] bf16, element kx * 32 + ci, the 8
// padding elements zero.  Workgroup = one 32 cout x 32 cin tile of a 3x3 weight: its 32 x 288 floats are read ONCE, as 32
// contiguous 1,152-byte pieces (the generic pack kernel gathers every element with a 36-byte stride, once per image), pass
// through LDS and leave as whole rows of every image that is asked for -- forward, data gradient (channels swapped, taps
// flipped), the four output-parity classes of a stride-2 layer's data gradient.  grid (weights, cout chunks, cin chunks).
__global__ void __launch_bounds__(SISR_BLOCK) weights_pack_deep_kernel(const SisrWeightDesc* table) {
    __shared__ float tile[32][9 * 32 + 1];                         // [cout][ci * 9 + tap]
    const SisrWeightDesc w = table[blockIdx.x];
    if (w.wdp_fwd == nullptr && w.wdp_dgrad == nullptr && w.wdp_dcls[0] == nullptr && w.wdp_dcls[1] == nullptr &&
        w.wdp_dcls[2] == nullptr && w.wdp_dcls[3] == nullptr)
        return;
    const int cb = blockIdx.y, kb = blockIdx.z;                    // cout chunk, cin chunk
    if (cb * 32 >= w.Cout || kb * 32 >= w.Cin || w.KH != 3 || w.KW != 3) return;
    const float sc = w.wdp_scaled ? 1.f / (w.sigma ? w.sigma[0] : 1.f) : 1.f;
    const int tid = threadIdx.x;
    // 32 couts x 288 contiguous floats (32 ci x 9 taps; 1,152-byte rows: 16-byte aligned), one float4 per item
    for (int i = tid; i < 32 * 72; i += SISR_BLOCK) {
        const int co = i / 72, e4 = i - co * 72;
        const f32x4 v = *reinterpret_cast<const f32x4*>(w.w_orig + ((int64_t)(cb * 32 + co) * w.Cin + kb * 32) * 9 + e4 * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) tile[co][e4 * 4 + j] = v[j] * sc;
    }
    __syncthreads();
    // every image row is written in 16-byte items of 8 bf16: 8 consecutive channels of one tap column (or the 8 padding slots)
    auto pack8 = [](const float (&v)[8]) {
        return u32x4{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
    };
    constexpr int RW = 3 * 32 + 8, G8 = RW / 8;
    if (w.wdp_fwd) {
        // rows (ky, co) of chunk kb: element kx * 32 + ci = tile[co][ci][ky][kx]
        __bf16* dst = reinterpret_cast<__bf16*>(w.wdp_fwd);
        for (int i = tid; i < 3 * 32 * G8; i += SISR_BLOCK) {
            const int row = i / G8, k8 = i - row * G8, ky = row >> 5, co = row & 31;
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = k8 < 12 ? tile[co][((k8 & 3) * 8 + j) * 9 + ky * 3 + (k8 >> 2)] : 0.f;
            *reinterpret_cast<u32x4*>(dst + ((int64_t)(kb * 3 + ky) * w.Cout + cb * 32 + co) * RW + k8 * 8) = pack8(v);
        }
    }
    if (w.wdp_dgrad) {
        // conv over dy: chunk = cb (over the forward couts), rows (ky', op = ci), element kx' * 32 + co = tile[co][ci][2 - ky'][2 - kx']
        __bf16* dst = reinterpret_cast<__bf16*>(w.wdp_dgrad);
        for (int i = tid; i < 3 * 32 * G8; i += SISR_BLOCK) {
            const int row = i / G8, k8 = i - row * G8, ky = row >> 5, ci = row & 31;
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = k8 < 12 ? tile[(k8 & 3) * 8 + j][ci * 9 + (2 - ky) * 3 + (2 - (k8 >> 2))] : 0.f;
            *reinterpret_cast<u32x4*>(dst + ((int64_t)(cb * 3 + ky) * w.Cin + kb * 32 + ci) * RW + k8 * 8) = pack8(v);
        }
    }
    for (int cls = 0; cls < 4; ++cls) {
        __bf16* dst = reinterpret_cast<__bf16*>(w.wdp_dcls[cls]);
        if (dst == nullptr) continue;
        const int KHc = w.c_KH[cls], KWc = w.c_KW[cls], RWc = (w.wdp_cls_kw ? w.wdp_cls_kw : KWc) * 32 + 8, G8c = RWc >> 3;
        for (int i = tid; i < KHc * 32 * G8c; i += SISR_BLOCK) {
            const int row = i / G8c, k8 = i - row * G8c, rp = row >> 5, ci = row & 31;
            const int r = w.c_R0y[cls] - 2 * rp, sx = w.c_R0x[cls] - 2 * (k8 >> 2);
            const bool ok = k8 < KWc * 4 && r >= 0 && r < 3 && sx >= 0 && sx < 3;
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = ok ? tile[(k8 & 3) * 8 + j][ci * 9 + r * 3 + sx] : 0.f;
            *reinterpret_cast<u32x4*>(dst + ((int64_t)(cb * KHc + rp) * w.Cin + kb * 32 + ci) * RWc + k8 * 8) = pack8(v);
        }
    }
}

// weight-gradient epilogue: the reduced packed gradient is a [K rows][CoutPad] matrix (K = chunk, tap, channel
// in the conv kernels' order), the parameter gradient is OIHW -- a transpose with a row permutation.  A workgroup
// takes one tile (32 packed couts x one channel chunk x all taps) through LDS: coalesced 128-byte reads along the
// couts, contiguous writes along (ci, r, s).  Two launches over grid (weights, max tiles per weight):
//   1. partial <G, W_orig> per tile  -> dot_part[w][tile]   (deterministic order)
//   2. every tile sums the partials of its weight and writes its share of the OIHW gradient
#define WGT_ROWS 64                                      // LDS rows per tile (taps x channels of the chunk, wgt_tg)
// A weight's un-packing is cut into tiles of 32 packed couts x one channel chunk x a GROUP OF TAPS of ~32 LDS rows, one
// workgroup each: a trunk layer (64 x 64 x 9) is 36 tiles with one memory round trip per phase (it was 4 tiles walking 9
// dependent rounds each: 26-30 us for the 37 layers of the generator, latency-bound with 150 busy workgroups).
__host__ __device__ __forceinline__ int wgt_tg(const SisrWeightGradDesc& w) {      // taps per tile
    const int ck = w.layout == 1 ? 32 : w.CK, nci = ck < w.Cin ? ck : w.Cin, taps = w.KH * w.KW;
    const int tg = nci > 0 ? 32 / nci : 1;
    return tg < 1 ? 1 : (tg > taps ? taps : tg);
}
__host__ __device__ __forceinline__ int wgt_base_tiles(const SisrWeightGradDesc& w) {
    const int ck = w.layout == 1 ? 32 : w.CK;
    return ((w.Cout + 31) / 32) * ((w.Cin + ck - 1) / ck);
}
__host__ __device__ __forceinline__ int wgt_tiles(const SisrWeightGradDesc& w) {
    const int tg = wgt_tg(w);
    return wgt_base_tiles(w) * ((w.KH * w.KW + tg - 1) / tg);
}

// mode 0: returns this thread's share of <G, W_orig> over the tile; mode 1: writes the gradient
template <int MODE>
__device__ __forceinline__ float wgt_tile(const SisrWeightGradDesc& w, int tile, float* lds, float gw, float inv) {
    const int tid = threadIdx.x;
    const int n_cot = (w.Cout + 31) / 32, n_base = wgt_base_tiles(w);
    const int base = tile % n_base, tgi = tile / n_base;
    const int cot = base % n_cot, chunk = base / n_cot;
    const int ck = w.layout == 1 ? 32 : w.CK;
    const int ci0 = chunk * ck, nci = min(ck, w.Cin - ci0);
    const int taps = w.KH * w.KW, cols = w.Cin * taps, Cq = w.Cout >> 2;
    const int tgroup = wgt_tg(w);                                   // taps of this tile (rows = taps x channels <= WGT_ROWS)
    float part = 0.f;
    for (int t0 = tgi * tgroup; t0 < min(taps, (tgi + 1) * tgroup); t0 += tgroup) {
        const int nt = min(tgroup, taps - t0), rows = nt * nci;
        __syncthreads();
        // 4 independent 128-byte-coalesced loads in flight per thread before the LDS stores (the tile is otherwise
        // one memory latency per 8 rows)
        for (int idx0 = tid; idx0 < rows * 32; idx0 += 4 * SISR_BLOCK) {
            float val[4];
            int off[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int idx = idx0 + k * SISR_BLOCK;
                val[k] = 0.f; off[k] = -1;
                if (idx < rows * 32) {
                    const int row = idx >> 5, c = idx & 31;
                    const int tl = row / nci, cl = row - tl * nci, tap = t0 + tl;
                    const int cp = cot * 32 + c;
                    int64_t prow;
                    if (w.layout == 1) prow = (int64_t)(chunk * taps + tap) * 32 + cl;
                    else { const int r = tap / w.KW, sx = tap - r * w.KW; prow = (int64_t)(chunk * w.KH + r) * w.KROWP + sx * w.PS + cl; }
                    off[k] = row * 33 + c;
                    if (cp < w.CoutPad) val[k] = w.dwpk[prow * w.CoutPad + cp];
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (off[k] >= 0) lds[off[k]] = val[k];
        }
        __syncthreads();
        for (int idx0 = tid; idx0 < rows * 32; idx0 += 4 * SISR_BLOCK) {
            float g[4], wv[4];
            int64_t e[4];
            int co_[4], col_[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int idx = idx0 + k * SISR_BLOCK;
                e[k] = -1; g[k] = wv[k] = 0.f; co_[k] = col_[k] = 0;
                if (idx < rows * 32) {
                    const int c = idx / rows, q = idx - c * rows;        // q runs over (cl, tap): contiguous in OIHW
                    const int cl = q / nt, tl = q - cl * nt;
                    const int cp = cot * 32 + c;
                    if (cp < w.Cout) {
                        co_[k] = w.shuffle2 ? ((cp % Cq) * 4 + cp / Cq) : cp;
                        col_[k] = (ci0 + cl) * taps + t0 + tl;
                        e[k] = (int64_t)co_[k] * cols + col_[k];
                        g[k] = lds[(tl * nci + cl) * 33 + c];
                        if (MODE == 0) wv[k] = w.w_orig[e[k]];
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (e[k] < 0) continue;
                if (MODE == 0) part += g[k] * wv[k];
                else w.grad[e[k]] = w.u_used ? (g[k] - gw * w.u_used[co_[k]] * w.v_used[col_[k]]) * inv : g[k];
            }
        }
    }
    return part;
}

__global__ void __launch_bounds__(SISR_BLOCK) weights_grad_dot_kernel(const SisrWeightGradDesc* table, float* dot_part) {
    __shared__ float lds[WGT_ROWS * 33];
    __shared__ float scratch[8];
    const SisrWeightGradDesc w = table[blockIdx.x];
    if ((int)blockIdx.y >= wgt_tiles(w)) return;
    float part = 0.f;
    if (w.u_used != nullptr && w.grad != nullptr) part = wgt_tile<0>(w, blockIdx.y, lds, 0.f, 0.f);
    const float tot = block_sum(part, scratch);
    if (threadIdx.x == 0) dot_part[(int64_t)blockIdx.x * gridDim.y + blockIdx.y] = tot;
}

__global__ void __launch_bounds__(SISR_BLOCK) weights_grad_kernel(const SisrWeightGradDesc* table, const float* dot_part) {
    __shared__ float lds[WGT_ROWS * 33];
    __shared__ float scratch[8];
    const SisrWeightGradDesc w = table[blockIdx.x];
    const int tid = threadIdx.x, n_tiles = wgt_tiles(w);
    if ((int)blockIdx.y >= n_tiles) return;
    const int Cq = w.Cout >> 2;
    if (blockIdx.y == 0 && w.grad_bias != nullptr && w.dbias_pk != nullptr) {
        for (int co = tid; co < w.Cout; co += SISR_BLOCK)
            w.grad_bias[co] = w.dbias_pk[w.shuffle2 ? ((co & 3) * Cq + (co >> 2)) : co];
    }
    if (w.grad == nullptr) return;
    float gw = 0.f, inv = 1.f;
    if (w.u_used != nullptr) {
        const float sigma = w.sigma[0];
        // (a large layer has thousands of tiles: the partial dots are summed by the workgroup, fixed partition and order,
        // not by every thread on its own)
        float dot = 0.f;
        for (int k = tid; k < n_tiles; k += SISR_BLOCK) dot += dot_part[(int64_t)blockIdx.x * gridDim.y + k];
        dot = block_sum(dot, scratch);
        gw = dot / sigma;          // <G, W> with W = W_orig / sigma
        inv = 1.f / sigma;
    }
    wgt_tile<1>(w, blockIdx.y, lds, gw, inv);
}

// ---- fast path of the weight-gradient epilogue for the 3x3 layers of the bf16 build (slab layout 1, channels in 32s) -------------
// The generic pair above cuts a 512 x 512 x 3 x 3 weight into 2,304 tiles of 32 rows and reads it twice (dot, then gradient): 166 us
// per discriminator backward.  Here a workgroup takes a whole 32-cout x 32-cin x 9-tap tile (288 packed rows of 128 bytes in,
// 32 OIHW rows of 1,152 contiguous bytes out) through LDS ONCE: it writes grad = G / sigma and its share of <G, W_orig>; the
// rank-one spectral-norm term -<G, W> u v^T / sigma is an elementwise pass over the finished OIHW gradient afterwards.
__global__ void __launch_bounds__(SISR_BLOCK) weights_grad_fast_kernel(const SisrWeightGradDesc* table, float* dot_part) {
    __shared__ float tile[288][33];
    __shared__ float scratch[8];
    const SisrWeightGradDesc w = table[blockIdx.x];
    const int cb = blockIdx.y, kb = blockIdx.z, tid = threadIdx.x;
    const int ncb = (w.Cout + 31) >> 5, nkb = w.Cin >> 5;
    if (cb >= ncb || kb >= nkb) return;
    if (cb == 0 && kb == 0 && w.grad_bias != nullptr && w.dbias_pk != nullptr)
        for (int co = tid; co < w.Cout; co += SISR_BLOCK) w.grad_bias[co] = w.dbias_pk[co];
    float part = 0.f;
    if (w.grad != nullptr) {
        const float inv = w.u_used != nullptr ? 1.f / w.sigma[0] : 1.f;
        // packed rows (tap, ci) of chunk kb: dwpk[((kb * 9 + tap) * 32 + ci) * CoutPad + cb * 32 + c]; CoutPad % 32 == 0 here
        // (the caller's dispatch rule), so a row's 32 couts are eight aligned float4s
        const float* src = w.dwpk + (int64_t)kb * 288 * w.CoutPad + cb * 32;
        const bool full = cb * 32 + 32 <= w.CoutPad;
        for (int i = tid; i < 288 * 8; i += SISR_BLOCK) {
            const int row = i >> 3, c4 = i & 7;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (full) v = *reinterpret_cast<const f32x4*>(src + (int64_t)row * w.CoutPad + c4 * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) tile[row][c4 * 4 + j] = v[j];
        }
        __syncthreads();
        // OIHW rows: grad[co][kb * 32 + ci][tap], 288 contiguous floats per cout (1,152-byte rows); element e = ci * 9 + tap
        // <- tile[tap * 32 + ci][co]; one float4 of the row per item
        for (int i = tid; i < 32 * 72; i += SISR_BLOCK) {
            const int co = i / 72, e4 = i - co * 72;
            if (cb * 32 + co >= w.Cout) continue;
            const int64_t o = ((int64_t)(cb * 32 + co) * w.Cin + kb * 32) * 9 + e4 * 4;
            f32x4 g;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int e = e4 * 4 + j, ci = e / 9, tap = e - ci * 9;
                g[j] = tile[tap * 32 + ci][co];
            }
            if (w.u_used != nullptr) {
                const f32x4 wo = *reinterpret_cast<const f32x4*>(w.w_orig + o);
                part += (g[0] * wo[0] + g[1] * wo[1]) + (g[2] * wo[2] + g[3] * wo[3]);
            }
            *reinterpret_cast<f32x4*>(w.grad + o) = g * inv;
        }
    }
    const float tot = block_sum(part, scratch);
    if (tid == 0) dot_part[((int64_t)blockIdx.x * gridDim.y + cb) * gridDim.z + kb] = tot;
}

// grad[co][j] -= (<G, W_orig> / sigma^2) u[co] v[j]      (W = W_orig / sigma; dW_orig = (G - <G, W> u v^T) / sigma)
__global__ void __launch_bounds__(SISR_BLOCK) weights_grad_rank1_kernel(const SisrWeightGradDesc* table, const float* dot_part,
                                                                        int tiles_y, int tiles_z) {
    __shared__ float scratch[8];
    const SisrWeightGradDesc w = table[blockIdx.x];
    if (w.u_used == nullptr || w.grad == nullptr) return;
    const int ncb = (w.Cout + 31) >> 5, nkb = w.Cin >> 5, tid = threadIdx.x;
    float dot = 0.f;
    for (int k = tid; k < ncb * nkb; k += SISR_BLOCK) dot += dot_part[((int64_t)blockIdx.x * tiles_y + k / nkb) * tiles_z + k % nkb];
    dot = block_sum(dot, scratch);                                      // fixed partition and order: every workgroup gets the same bits
    const float sigma = w.sigma[0];
    const float coef = dot / (sigma * sigma);
    const int cols = w.Cin * 9;
    const int64_t total = (int64_t)w.Cout * cols;
    for (int64_t e = (int64_t)blockIdx.y * SISR_BLOCK + tid; e < total; e += (int64_t)gridDim.y * SISR_BLOCK) {
        const int co = (int)(e / cols), j = (int)(e - (int64_t)co * cols);
        w.grad[e] -= coef * w.u_used[co] * w.v_used[j];
    }
}

static int parts_for(int64_t elems) {           // workgroups per weight for the element-wise multi-tensor kernels
    const int64_t want = (elems + 4095) / 4096;   // ~16 elements per thread
    return (int)std::max<int64_t>(16, std::min<int64_t>(want, 1024));
}

// the three parts of sisr_weights_prepare as entry points of their own: a caller that keeps packed images across forwards
// (W_orig does not change between two optimizer steps; only u, v, sigma do) runs the power iteration alone
extern "C" int sisr_weights_sn(const SisrWeightDesc* table_dev, int32_t n, int32_t max_rows, int32_t max_cols, void* stream) {
    if (!table_dev || n <= 0 || max_rows <= 0 || max_cols <= 0) return SISR_E_BADARG;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int jobs_a = ((max_rows + SN_RB - 1) / SN_RB) * ((max_cols + SN_CB - 1) / SN_CB);
    hipLaunchKernelGGL(sn_wt_u_kernel, dim3(n, jobs_a), dim3(SISR_BLOCK), 0, st, table_dev);
    SISR_CHECK_LAUNCH();
    hipLaunchKernelGGL(sn_t_sum_kernel, dim3(n, (max_cols + SISR_BLOCK - 1) / SISR_BLOCK), dim3(SISR_BLOCK), 0, st, table_dev);
    SISR_CHECK_LAUNCH();
    hipLaunchKernelGGL(sn_w_v_kernel, dim3(n, (max_rows + 3) / 4), dim3(SISR_BLOCK), 0, st, table_dev);
    SISR_CHECK_LAUNCH();
    hipLaunchKernelGGL(sn_u_finish_kernel, dim3(n), dim3(SISR_BLOCK), 0, st, table_dev);
    SISR_CHECK_LAUNCH();
    return 0;
}

extern "C" int sisr_weights_pack(const SisrWeightDesc* table_dev, int32_t n, int32_t max_rows, int32_t max_cols, void* stream) {
    if (!table_dev || n <= 0 || max_rows <= 0 || max_cols <= 0) return SISR_E_BADARG;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    // packed images are a little larger than the matrix (padding slots): parts from 2x its size
    hipLaunchKernelGGL(weights_pack_kernel, dim3(n, parts_for(2ll * max_rows * max_cols)), dim3(SISR_BLOCK), 0, st, table_dev);
    SISR_CHECK_LAUNCH();
    return 0;
}

// the conv_deep.hip images (wdp_*) of every 3x3 weight of the table that asks for one; max_cout / max_cin over those weights
extern "C" int sisr_weights_pack_deep(const SisrWeightDesc* table_dev, int32_t n, int32_t max_cout, int32_t max_cin, void* stream) {
    if (!table_dev || n <= 0 || max_cout <= 0 || max_cin <= 0) return SISR_E_BADARG;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(weights_pack_deep_kernel, dim3(n, (max_cout + 31) / 32, (max_cin + 31) / 32), dim3(SISR_BLOCK), 0, st, table_dev);
    SISR_CHECK_LAUNCH();
    return 0;
}

extern "C" int sisr_weights_prepare(const SisrWeightDesc* table_dev, int32_t n, int32_t max_rows, int32_t max_cols,
                                    void* stream) {
    if (int e = sisr_weights_sn(table_dev, n, max_rows, max_cols, stream)) return e;
    return sisr_weights_pack(table_dev, n, max_rows, max_cols, stream);
}

// tiles (= workgroups along grid.y, = dot_work entries) one weight of the table needs; `parts` of sisr_weights_grad must
// be >= the maximum over the table
extern "C" int sisr_weights_grad_tiles(const SisrWeightGradDesc* w) {
    if (!w || w->Cout <= 0 || w->Cin <= 0 || w->KH <= 0 || w->KW <= 0 || (w->layout != 1 && w->CK <= 0)) return SISR_E_BADARG;
    return wgt_tiles(*w);
}

// every weight of the table: 3x3, Cin % 32 == 0, layout 1 (bf16-kernel slabs), no PixelShuffle permutation.
// dot_work: n * ceil(max_cout / 32) * (max_cin / 32) floats.
extern "C" int sisr_weights_grad_fast(const SisrWeightGradDesc* table_dev, int32_t n, float* dot_work, int32_t max_cout,
                                      int32_t max_cin, void* stream) {
    if (!table_dev || n <= 0 || !dot_work || max_cout <= 0 || max_cin < 32) return SISR_E_BADARG;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int ty = (max_cout + 31) / 32, tz = max_cin / 32;
    hipLaunchKernelGGL(weights_grad_fast_kernel, dim3(n, ty, tz), dim3(SISR_BLOCK), 0, st, table_dev, dot_work);
    SISR_CHECK_LAUNCH();
    const int parts = (int)std::max<int64_t>(8, std::min<int64_t>(((int64_t)max_cout * max_cin * 9 + 4095) / 4096, 512));
    hipLaunchKernelGGL(weights_grad_rank1_kernel, dim3(n, parts), dim3(SISR_BLOCK), 0, st, table_dev, dot_work, ty, tz);
    SISR_CHECK_LAUNCH();
    return 0;
}

extern "C" int sisr_weights_grad(const SisrWeightGradDesc* table_dev, int32_t n, float* dot_work, int32_t parts,
                                 void* stream) {
    if (!table_dev || n <= 0 || !dot_work || parts <= 0 || parts > 65535) return SISR_E_BADARG;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(weights_grad_dot_kernel, dim3(n, parts), dim3(SISR_BLOCK), 0, st, table_dev, dot_work);
    SISR_CHECK_LAUNCH();
    hipLaunchKernelGGL(weights_grad_kernel, dim3(n, parts), dim3(SISR_BLOCK), 0, st, table_dev, dot_work);
    SISR_CHECK_LAUNCH();
    return 0;
}
