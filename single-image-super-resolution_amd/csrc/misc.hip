// misc.hip -- device query and an MFMA lane-layout self test (run by the GPU test-suite before any
// parity test so that a wrong operand/accumulator map is reported as such).
#include "sisr_dev.h"

#include <cstdlib>
#include <cstring>

// C[32][32] = A[32][K=8] * B[8][32] with asymmetric integer data; out row-major [row][col]
__global__ void mfma_selftest_kernel(float* out) {
    const int lane = threadIdx.x & 63;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    for (int k0 = 0; k0 < 8; k0 += 2) {
        const int k = k0 + (lane >> 5);
        const float a = (float)((lane & 31) * 3 + k * 7 + 1);     // A[i][k] = 3i + 7k + 1
        const float b = (float)(k * 5 - (lane & 31) * 2 + 11);    // B[k][j] = 5k - 2j + 11
        acc = mfma32(a, b, acc);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) out[mfma_row(i, lane) * 32 + (lane & 31)] = acc[i];
}

extern "C" int sisr_mfma_selftest(float* out_dev, void* stream) {
    if (!out_dev) return SISR_E_BADARG;
    hipLaunchKernelGGL(mfma_selftest_kernel, dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), out_dev);
    SISR_CHECK_LAUNCH();
    return 0;
}

int sisr_device_index() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) dev = 0;
    return dev < SISR_MAX_DEVICES ? dev : SISR_MAX_DEVICES - 1;
}

int sisr_cu_slots() {
    static int cus[SISR_MAX_DEVICES];              // per device id; 0 = not queried yet
    int& c = cus[sisr_device_index()];
    if (c == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0)
            c = v;
        else
            c = 256;
    }
    int n = c;
    if (const char* e = getenv("SISR_PERSIST_MAX_WG")) {
        const int cap = atoi(e);
        if (cap > 0 && cap < n) n = cap;
    }
    return n;
}

extern "C" int sisr_device_info(int32_t* n_cu, int32_t* lds_per_cu, char* arch, int32_t arch_len) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, dev);
    if (e != hipSuccess) return (int)e;
    if (n_cu) *n_cu = prop.multiProcessorCount;
    if (lds_per_cu) *lds_per_cu = (int32_t)prop.maxSharedMemoryPerMultiProcessor;
    if (arch && arch_len > 0) {
        std::strncpy(arch, prop.gcnArchName, arch_len - 1);
        arch[arch_len - 1] = 0;
    }
    return 0;
}

// ds_read_b64_tr_b16 lane-role check: LDS holds a [16 pixels][32 channels] bf16-sized tile of shorts
// (value = pixel*64 + channel); every lane addresses (pixel 8*(g>>1)+q, channel 16*(g&1)+4p) exactly as
// wgrad_bf16.hip does and writes back the 8 shorts it received (two reads, pixels +0..3 and +4..7).
typedef short ts4 __attribute__((ext_vector_type(4)));
__global__ void tr16_selftest_kernel(short* out) {
    __shared__ __attribute__((aligned(16))) short tile[16 * 40];
    for (int i = threadIdx.x; i < 16 * 40; i += 64) tile[i] = (short)((i / 40) * 64 + (i % 40));
    __syncthreads();
    const int lane = threadIdx.x, grp = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    const short* a = tile + (8 * (grp >> 1) + q) * 40 + 16 * (grp & 1) + 4 * p;
    const ts4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) ts4*)(a));
    const ts4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) ts4*)(a + 4 * 40));
    for (int j = 0; j < 4; ++j) { out[lane * 8 + j] = lo[j]; out[lane * 8 + 4 + j] = hi[j]; }
}

extern "C" int sisr_tr16_selftest(short* out_dev, void* stream) {
    if (!out_dev) return SISR_E_BADARG;
    hipLaunchKernelGGL(tr16_selftest_kernel, dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), out_dev);
    SISR_CHECK_LAUNCH();
    return 0;
}

extern "C" int sisr_struct_sizes(int32_t* out, int32_t cap) {
    const int32_t v[8] = {(int32_t)sizeof(SisrConvDesc), (int32_t)sizeof(SisrWgradDesc), (int32_t)sizeof(SisrWeightDesc),
                          (int32_t)sizeof(SisrWeightGradDesc), (int32_t)sizeof(SisrBnBwdDesc),
                          (int32_t)sizeof(SisrConvPlan), (int32_t)sizeof(SisrDeepPlan), (int32_t)sizeof(SisrWgradDeepPlan)};
    for (int i = 0; i < 8 && i < cap; ++i) out[i] = v[i];
    return 8;
}

// hipGetLastError() is per host thread and sticky until fetched: a failed stream capture (or any other failed
// runtime call made by the host framework) leaves its code behind and the next kernel launch of this library
// would report it as its own.  Returns the error that was pending (0: none) and clears it.
extern "C" int sisr_clear_last_error(void) { return (int)hipGetLastError(); }

extern "C" const char* sisr_version(void) { return "sisr_hip 0.4 (gfx950)"; }

