// abi.hip -- error reporting / versioning shared by every entry point of libstraps_hip.so
#include <stdarg.h>

#include "common.h"

static thread_local char g_err[512] = "";

void straps_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---- calibration kernel: register-resident fp32 MFMA stream (what the matrix pipe sustains on THIS board with
// non-trivial operand data; the roofline "peak" stays the 157.3 TFLOP/s spec, this is the practical ceiling) ----
__global__ __launch_bounds__(256) void mfma_peak_kernel(const float* __restrict__ seed, float* __restrict__ out, int iters) {
    f32x16 a0, a1, a2, a3;
    const float s0 = seed[threadIdx.x], s1 = seed[256 + threadIdx.x];
#pragma unroll
    for (int r = 0; r < 16; ++r) { a0[r] = s0; a1[r] = s1; a2[r] = -s0; a3[r] = -s1; }
    float x = s0, y = s1;
    for (int it = 0; it < iters; ++it) {
        a0 = mfma32(x, y, a0);
        a1 = mfma32(y, x, a1);
        a2 = mfma32(x, x, a2);
        a3 = mfma32(y, y, a3);
        x = -x; y = -y;             // keeps the accumulators bounded, operands toggling
    }
    float t = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) t += a0[r] + a1[r] + a2[r] + a3[r];
    out[blockIdx.x * 256 + threadIdx.x] = t;
}

extern "C" int straps_selftest_mfma_peak(const float* seed512, float* out, int blocks, int iters, void* stream) {
    STRAPS_REQUIRE(seed512 && out && blocks > 0 && iters > 0, "straps_selftest_mfma_peak: bad arguments");
    hipLaunchKernelGGL(mfma_peak_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, seed512, out, iters);
    STRAPS_CHECK_LAUNCH("mfma_peak_kernel");
    return STRAPS_OK;
}

// ---- sustained shader clock of the convolution kernels: with an accumulator set, workgroup 0 of every implicit-GEMM launch adds the
// shader-clock ticks (s_memtime: one per shader cycle, it follows DVFS) and the constant-rate wall ticks (s_memrealtime) of its lifetime
// to the pair; MHz = ticks ratio x wall-clock rate.  The chip clocks to its power budget, so the same binary reads several per cent
// apart on two boards -- bench.py reports this number so that a difference can be attributed.  (A lane spinning on a side stream, the
// first form of this probe, stalled whatever stream shared its hardware queue.)
unsigned long long* g_straps_clk_acc = nullptr;

extern "C" int straps_wall_clock_khz(void) {
    int dev = 0, khz = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess) return 0;
    return khz;
}

extern "C" int straps_set_clock_accumulator(unsigned long long* acc2) {
    g_straps_clk_acc = acc2;          // (kernel arguments are fixed at launch / graph-capture time: set it before capturing)
    return STRAPS_OK;
}

extern "C" int straps_abi_version(void) { return STRAPS_ABI_VERSION; }
extern "C" const char* straps_last_error(void) { return g_err; }
extern "C" int straps_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
