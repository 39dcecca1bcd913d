// fc_head.hip -- the discriminator's classifier head (model_discriminator.py:47-53):
//     Linear(fc_in, 2 f) -> LeakyReLU -> Linear(2 f, 1) -> Sigmoid          fc_in = 18,432 (HR 96) / 73,728 (HR 192), 2 f = 1,024
// forward and backward, as four kernels around the ONE tensor that matters -- W1 [1024][fc_in] fp32, 75 / 302 MB, streamed
// exactly once per pass at 16 bytes per lane:
//   fc1_forward_kernel   h1 partials = x W1^T on the exact-fp32 matrix instruction (v_mfma_f32_16x16x4_f32: rows = 16 weight rows,
//                        columns = the 16 batch rows, one K = 4 step per dword of a lane's 16-byte loads).  The round-3 kernel did
//                        these 604 MFLOP on the vector ALU (4,608 dependent-free but issue-bound FMAs per thread, one wave per
//                        SIMD): 107 us for 75 MB.  Here a wave's arithmetic is 4 MFMAs per KB of weights (4 us of matrix pipe in
//                        all) and the kernel is what it should be: a weight stream with 8 loads of 1 KB in flight per wave.
//                        Workgroup = 16 weight rows x a quarter of K (4 waves x a sixteenth each), so 64 x 4 = 256 workgroups.
//   fc_head_finish_kernel  sums the K quarters in order, adds b1, stores h1 (the backward needs the pre-activation), and runs
//                        the whole second layer: out = sigmoid(lrelu(h1) . W2 + b2) -- one workgroup per batch row.
//   fc_head_bwd_kernel   everything of the backward that is not W1-sized: d2 = g out (1 - out), dW2, db2, d1 = d2 W2 lrelu'(h1), db1.
//   fc1_dgrad_kernel     dx = d1 W1 on the same instruction: workgroup = 64 columns of K, its 4 waves split the 1,024 rows and
//                        meet in LDS (no global partials, fixed order); 4 weight rows x 256 contiguous bytes per load.
// dW1 = d1^T x is an outer product with a 16-deep contraction: write-bound, it stays on the vector kernel of layout_fc.hip.
// All sums are fp32 in a fixed order (deterministic); exact-fp32 products, so the same kernels serve the fp32 parity build
// and the bf16 build (whose classifier head stays fp32: its tensors are 16 x 1,024).
#include "sisr_dev.h"

#include <algorithm>

typedef float fcx4 __attribute__((ext_vector_type(4)));

#define FH_B 16                      // batch rows of one call (the MFMA's 16 columns); rows >= B are zeros
#define FH_KQ 4                      // K quarters (workgroups per 16-row group)
#define FH_UNROLL 8                  // 16-byte loads in flight per lane and operand

__device__ __forceinline__ fcx4 fh_mfma(float a, float b, fcx4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// x: [B][K] (already activated: in_slope applied here on load when != 1), W: [N][K]; part: [FH_KQ][FH_B][N]
__global__ void __launch_bounds__(256) fc1_forward_kernel(const float* __restrict__ x, float in_slope, const float* __restrict__ W,
                                                          float* __restrict__ part, int B, int K, int N) {
    __shared__ float red[4][16 * 16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n0 = blockIdx.x * 16, kq = blockIdx.y;
    const int li = lane & 15, kk = lane >> 4;
    // K steps of 16 floats: step s covers k = 16 s + 4 kk + j for MFMA j (the same permutation of K on both operands)
    const int steps = K >> 4, nsl = FH_KQ * 4, sl = kq * 4 + wave;
    const int s0 = (int)((int64_t)steps * sl / nsl), s1 = (int)((int64_t)steps * (sl + 1) / nsl);
    const float* wp = W + (int64_t)(n0 + li) * K + 4 * kk;
    const float* xp = x + (int64_t)li * K + 4 * kk;
    const bool xok = li < B;
    fcx4 acc = {0.f, 0.f, 0.f, 0.f};
    const fcx4 zero = {0.f, 0.f, 0.f, 0.f};
    int s = s0;
    for (; s + FH_UNROLL <= s1; s += FH_UNROLL) {
        fcx4 wv[FH_UNROLL], xv[FH_UNROLL];
#pragma unroll
        for (int u = 0; u < FH_UNROLL; ++u) {
            wv[u] = __builtin_nontemporal_load(reinterpret_cast<const fcx4*>(wp + (int64_t)(s + u) * 16));
            xv[u] = xok ? *reinterpret_cast<const fcx4*>(xp + (int64_t)(s + u) * 16) : zero;
        }
#pragma unroll
        for (int u = 0; u < FH_UNROLL; ++u)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc = fh_mfma(wv[u][j], lrelu(xv[u][j], in_slope), acc);
    }
    for (; s < s1; ++s) {
        const fcx4 wv = *reinterpret_cast<const fcx4*>(wp + (int64_t)s * 16);
        const fcx4 xv = xok ? *reinterpret_cast<const fcx4*>(xp + (int64_t)s * 16) : zero;
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = fh_mfma(wv[j], lrelu(xv[j], in_slope), acc);
    }
    // acc[r]: weight row n0 + 4 kk + r, batch column li.  The 4 waves (K sixteenths of this quarter) meet in LDS, fixed order.
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][(4 * kk + r) * 16 + li] = acc[r];
    __syncthreads();
    {
        const int row = tid >> 4, b = tid & 15;
        const float v = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
        part[((int64_t)kq * FH_B + b) * N + n0 + row] = v;
    }
}

// one workgroup per batch row: h1[b][:] = sum of the K quarters + b1; out[b] = sigmoid(sum_n lrelu(h1[b][n], slope) W2[n] + b2)
// (W2 == nullptr: first layer only)
__global__ void __launch_bounds__(256) fc_head_finish_kernel(const float* __restrict__ part, const float* __restrict__ b1,
                                                             float* __restrict__ h1, const float* __restrict__ W2,
                                                             const float* __restrict__ b2, float slope, float* __restrict__ out,
                                                             int N) {
    __shared__ float scratch[8];
    const int b = blockIdx.x;
    float dot = 0.f;
    for (int n = threadIdx.x; n < N; n += 256) {
        float v = 0.f;
#pragma unroll
        for (int q = 0; q < FH_KQ; ++q) v += part[((int64_t)q * FH_B + b) * N + n];
        v += b1 ? b1[n] : 0.f;
        h1[(int64_t)b * N + n] = v;
        if (W2) dot += lrelu(v, slope) * W2[n];
    }
    if (W2) {
        const float t = block_sum(dot, scratch) + (b2 ? b2[0] : 0.f);
        if (threadIdx.x == 0) out[b] = 1.f / (1.f + expf(-t));
    }
}

// backward of everything behind h1 (thread = one of the N hidden units; B <= 16 batch rows looped):
//   d2[b] = g[b] out[b] (1 - out[b]);  dW2[n] = sum_b d2[b] lrelu(h1[b][n]);  db2 = sum_b d2[b]
//   d1[b][n] = d2[b] W2[n] (h1[b][n] > 0 ? 1 : slope);  db1[n] = sum_b d1[b][n]
__global__ void __launch_bounds__(256) fc_head_bwd_kernel(const float* __restrict__ g, const float* __restrict__ out,
                                                          const float* __restrict__ h1, const float* __restrict__ W2, float slope,
                                                          float* __restrict__ d1, float* __restrict__ dW2, float* __restrict__ db2,
                                                          float* __restrict__ db1, int B, int N) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    float d2[FH_B];
    float sb2 = 0.f;
#pragma unroll
    for (int b = 0; b < FH_B; ++b) {
        d2[b] = 0.f;
        if (b < B) { const float o = out[b]; d2[b] = g[b] * o * (1.f - o); sb2 += d2[b]; }
    }
    if (n == 0 && db2) db2[0] = sb2;
    if (n >= N) return;
    const float w = W2[n];
    float sw = 0.f, sb1 = 0.f;
#pragma unroll
    for (int b = 0; b < FH_B; ++b) {
        if (b < B) {
            const float h = h1[(int64_t)b * N + n];
            sw += d2[b] * lrelu(h, slope);
            const float dv = d2[b] * w * (h > 0.f ? 1.f : slope);
            d1[(int64_t)b * N + n] = dv;
            sb1 += dv;
        }
    }
    dW2[n] = sw;
    if (db1) db1[n] = sb1;
}

// dx[b][k] = sum_n d1[b][n] W[n][k].  MFMA j of a step: A[i = b][kk] = d1[b][n + kk], B[kk][jj] = W[n + kk][k0 + 4 jj + j]
// (a lane's 16-byte load = 4 consecutive k of row n + kk): accumulator j holds columns k0 + 4 jj + j, jj = 0..15.
__global__ void __launch_bounds__(256) fc1_dgrad_kernel(const float* __restrict__ d1, const float* __restrict__ W,
                                                        float* __restrict__ dx, int B, int K, int N) {
    __shared__ float red[4][FH_B * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int k0 = blockIdx.x * 64;
    const int li = lane & 15, kk = lane >> 4;
    const int rows = N >> 2;                              // rows of this wave: [wave * rows, (wave + 1) * rows)
    const int nb = wave * rows;
    fcx4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = fcx4{0.f, 0.f, 0.f, 0.f};
    const float* wp = W + (int64_t)(nb + kk) * K + k0 + 4 * li;
    const bool bok = li < B;
    for (int n = 0; n < rows; n += 4 * FH_UNROLL) {
        fcx4 wv[FH_UNROLL];
        float av[FH_UNROLL];
#pragma unroll
        for (int u = 0; u < FH_UNROLL; ++u) {
            const int r = n + 4 * u;
            const bool ok = r + kk < rows;
            wv[u] = ok ? __builtin_nontemporal_load(reinterpret_cast<const fcx4*>(wp + (int64_t)r * K)) : fcx4{0.f, 0.f, 0.f, 0.f};
            av[u] = (ok && bok) ? d1[(int64_t)li * N + nb + r + kk] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < FH_UNROLL; ++u)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = fh_mfma(av[u], wv[u][j], acc[j]);
    }
    // acc[j][r]: batch row 4 kk + r, column k0 + 4 li + j
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave][(4 * kk + r) * 64 + 4 * li + j] = acc[j][r];
    __syncthreads();
    for (int i = tid; i < FH_B * 64; i += 256) {
        const int b = i >> 6, c = i & 63;
        if (b < B && k0 + c < K) dx[(int64_t)b * K + k0 + c] = red[0][i] + red[1][i] + red[2][i] + red[3][i];
    }
}

// ---- host ---------------------------------------------------------------------------------------------------------------------
#define S_(s) reinterpret_cast<hipStream_t>(s)

extern "C" int sisr_fc_head_ws_floats(int32_t N) { return FH_KQ * FH_B * N; }

// forward of the whole head.  x [B][K] fp32 (activated), W1 [N][K], b1 [N], W2 [N] (Linear(N, 1).weight), b2 [1];
// out: h1 [B][N] (pre-activation of the hidden layer), y [B] (sigmoid output); ws: sisr_fc_head_ws_floats(N) floats.
extern "C" int sisr_fc_head_forward(const float* x, const float* W1, const float* b1, const float* W2, const float* b2, float slope,
                                    float* h1, float* y, float* ws, int32_t B, int32_t K, int32_t N, void* stream) {
    if (!x || !W1 || !h1 || !ws || B <= 0 || B > FH_B || K <= 0 || (K & 15) || N <= 0 || (N & 15)) return SISR_E_BADARG;
    if (W2 && !y) return SISR_E_BADARG;
    hipLaunchKernelGGL(fc1_forward_kernel, dim3(N / 16, FH_KQ), dim3(256), 0, S_(stream), x, 1.f, W1, ws, B, K, N);
    SISR_CHECK_LAUNCH();
    hipLaunchKernelGGL(fc_head_finish_kernel, dim3(B), dim3(256), 0, S_(stream), ws, b1, h1, W2, b2, slope, y, N);
    SISR_CHECK_LAUNCH();
    return 0;
}

// backward of the head behind h1: g [B] = dL/dy; -> d1 [B][N] = dL/dh1, dW2 [N], db2 [1], db1 [N]
extern "C" int sisr_fc_head_backward(const float* g, const float* y, const float* h1, const float* W2, float slope, float* d1,
                                     float* dW2, float* db2, float* db1, int32_t B, int32_t N, void* stream) {
    if (!g || !y || !h1 || !W2 || !d1 || !dW2 || B <= 0 || B > FH_B || N <= 0) return SISR_E_BADARG;
    hipLaunchKernelGGL(fc_head_bwd_kernel, dim3((N + 255) / 256), dim3(256), 0, S_(stream), g, y, h1, W2, slope, d1, dW2, db2, db1, B, N);
    SISR_CHECK_LAUNCH();
    return 0;
}

// dx [B][K] = d1 [B][N] W1 [N][K]
extern "C" int sisr_fc1_dgrad(const float* d1, const float* W1, float* dx, int32_t B, int32_t K, int32_t N, void* stream) {
    if (!d1 || !W1 || !dx || B <= 0 || B > FH_B || K <= 0 || (K & 63) || N <= 0 || (N % (16 * FH_UNROLL))) return SISR_E_BADARG;
    hipLaunchKernelGGL(fc1_dgrad_kernel, dim3(K / 64), dim3(256), 0, S_(stream), d1, W1, dx, B, K, N);
    SISR_CHECK_LAUNCH();
    return 0;
}

// ---- dW1 from MANY rows: the data-parallel exchange of the classifier head's weight gradient --------------------------------------
// dW[n][k] = scale * sum_b dy[b][n] * x[b][k] is a rank-B product of two small factors (B = 16 rows per rank).  Exchanging the 75-302 MB
// product with an all-reduce costs 2 (N - 1) / N of it per GPU and pass; exchanging the FACTORS costs N x 1.2-4.8 MB, after which every
// rank forms the mean gradient itself from all N x 16 rows (distributed.GradReducer.gather, discriminator_engine.run_backward).
// Exact-fp32 matrix instruction (v_mfma_f32_32x32x2_f32: A = dy^T tile 32 n x 2 b, B = x tile 2 b x 32 k).
// B <= 256; K % 128 == 0; N % 64 == 0.
#define FR_MAXROWS 256
#define FR_CHUNK 64                   // rows staged per round
#define FR_DP 96                      // LDS row pitch (floats) of the dy^T tile: 64 columns + 32 (the two K rows of a step 32 banks apart)
#define FR_XP 160                     // ... of the x tile: 128 columns + 32
// Workgroup = 64 rows n x 128 columns k of dW; wave (wn, wk) = 32 n x 64 k (two accumulators).  The factors' rows are staged 64 at a time
// into LDS (dy^T tile [64][64], x tile [64][128]: 64 KB, two workgroups per CU -- one loads while the other multiplies): a value of
// either factor is fetched from memory once per workgroup instead of once per wave and tile.
__global__ void __launch_bounds__(256, 2) fc_wgrad_rows_kernel(const float* __restrict__ dy, const float* __restrict__ x, float scale,
                                                               float* __restrict__ dW, int B, int K, int N) {
    __shared__ __attribute__((aligned(16))) float dys[FR_CHUNK * FR_DP];
    __shared__ __attribute__((aligned(16))) float xs[FR_CHUNK * FR_XP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, kk = lane >> 5, wn = wave >> 1, wk = wave & 1;
    const int n0 = (int)blockIdx.x * 64, k0 = (int)blockIdx.y * 128;      // (the workgroups that share an x tile are neighbours)
    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
    for (int b0 = 0; b0 < B; b0 += FR_CHUNK) {
        const int nb = min(FR_CHUNK, B - b0);
        if (b0 > 0) __syncthreads();                          // the previous chunk has been consumed
        // dy rows: 16 float4 per row; x rows: 32 float4 per row
        for (int i = tid; i < FR_CHUNK * 16; i += 256) {
            const int r = i >> 4, c4 = i & 15;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (r < nb) v = *reinterpret_cast<const f32x4*>(dy + (int64_t)(b0 + r) * N + n0 + c4 * 4);
            *reinterpret_cast<f32x4*>(dys + r * FR_DP + c4 * 4) = v;
        }
        for (int i = tid; i < FR_CHUNK * 32; i += 256) {
            const int r = i >> 5, c4 = i & 31;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (r < nb) v = *reinterpret_cast<const f32x4*>(x + (int64_t)(b0 + r) * K + k0 + c4 * 4);
            *reinterpret_cast<f32x4*>(xs + r * FR_XP + c4 * 4) = v;
        }
        __syncthreads();
        const float* ap = dys + kk * FR_DP + wn * 32 + l31;
        const float* bp = xs + kk * FR_XP + wk * 64 + l31;
        const int ns = (nb + 1) >> 1;                          // (an odd last row pairs with a zero row in LDS)
#pragma unroll 8
        for (int s = 0; s < ns; ++s) {
            const float a = ap[2 * s * FR_DP];
            acc[0] = mfma32(a, bp[2 * s * FR_XP], acc[0]);
            acc[1] = mfma32(a, bp[2 * s * FR_XP + 32], acc[1]);
        }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 16; ++i)
            dW[(int64_t)(n0 + wn * 32 + mfma_row(i, lane)) * K + k0 + wk * 64 + j * 32 + l31] = acc[j][i] * scale;
}

extern "C" int sisr_fc_wgrad_rows(const float* dy, const float* x, float scale, float* dW, int32_t B, int32_t K, int32_t N, void* stream) {
    if (!dy || !x || !dW || B <= 0 || B > FR_MAXROWS || K <= 0 || (K & 127) || N <= 0 || (N & 63)) return SISR_E_BADARG;
    hipLaunchKernelGGL(fc_wgrad_rows_kernel, dim3(N / 64, K / 128), dim3(256), 0, S_(stream), dy, x, scale, dW, B, K, N);
    SISR_CHECK_LAUNCH();
    return 0;
}
