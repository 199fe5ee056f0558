// common.h -- shared helpers for the gfx950 kernels behind include/straps_hip.h
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/straps_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// error text shared by all translation units (defined in abi.hip)
void straps_set_error(const char* fmt, ...);

#define STRAPS_REQUIRE(cond, ...)             \
    do {                                      \
        if (!(cond)) {                        \
            straps_set_error(__VA_ARGS__);    \
            return STRAPS_EINVAL;             \
        }                                     \
    } while (0)

#define STRAPS_CHECK_LAUNCH(name)                                                        \
    do {                                                                                 \
        hipError_t e__ = hipGetLastError();                                              \
        if (e__ != hipSuccess) {                                                         \
            straps_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));     \
            return STRAPS_EHIP;                                                          \
        }                                                                                \
    } while (0)

// fp32-input MFMA, 32x32 output tile, K=2 per instruction (exact fmaf chain, 64 cycles / SIMD).
// A: lane l holds A[i = l&31][k = l>>5]; B: lane l holds B[k = l>>5][n = l&31];
// C/D: lane l, reg r -> C[row = (r&3) + 8*(r>>2) + 4*(l>>5)][col = l&31].
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ int mfma_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// XCD-aware bijective remap of a 1-D block id: consecutive logical ids land on the same XCD
// (hardware places block b on XCD b % 8; speed only, never correctness).
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }
