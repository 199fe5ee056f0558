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

// ---- sustained shader clock: ONE lane spins for a given wall time next to whatever else runs on the chip and reports how many
// shader-clock ticks (s_memtime: one per shader cycle, it follows DVFS) passed per constant-rate wall tick (s_memrealtime).  bench.py
// launches it on a side stream across its timed region: the chip clocks to its power budget, so the same binary reads 8 % apart on
// two boards -- this number says which part of a difference is the board's clock.
__global__ __launch_bounds__(64) void clock_probe_kernel(unsigned long long* __restrict__ out, unsigned long long spin_wall_ticks) {
    if (threadIdx.x != 0) return;
    const unsigned long long w0 = wall_clock64(), c0 = clock64();
    unsigned long long w1 = w0;
    while (w1 - w0 < spin_wall_ticks) {
        __builtin_amdgcn_s_sleep(64);
        w1 = wall_clock64();
    }
    const unsigned long long c1 = clock64();
    out[0] = c1 - c0;
    out[1] = w1 - w0;
}

extern "C" int straps_wall_clock_khz(void) {
    int dev = 0, khz = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess) return 0;
    return khz;
}

extern "C" int straps_clock_probe(unsigned long long* out2, double spin_seconds, void* stream) {
    STRAPS_REQUIRE(out2 && spin_seconds > 0.0 && spin_seconds <= 5.0, "straps_clock_probe: need an output pair and 0 < spin_seconds <= 5");
    const int khz = straps_wall_clock_khz();
    STRAPS_REQUIRE(khz > 0, "straps_clock_probe: the device reports no wall-clock rate");
    hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, out2, (unsigned long long)(spin_seconds * 1e3 * khz));
    STRAPS_CHECK_LAUNCH("clock_probe_kernel");
    return STRAPS_OK;
}

extern "C" int straps_abi_version(void) { return STRAPS_ABI_VERSION; }
extern "C" const char* straps_last_error(void) { return g_err; }
extern "C" int straps_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
