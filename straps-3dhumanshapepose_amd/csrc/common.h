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
// device pair the convolution kernels add their (shader ticks, wall ticks) to: the CURRENT device's, NULL unless straps_set_clock_accumulator set it (abi.hip)
unsigned long long* straps_clk_acc_current();

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

// Kernels in which the compiler would form a packed fp32 instruction whose LOW result reads the HIGH register of a source (VOP3P op_sel, e.g.
// v_pk_fma_f32 d, a, b, c op_sel:[0,1,0]) are compiled WITHOUT packed fp32 instructions.  Round 5, DESIGN section 1: on MI355X such an instruction
// returns a wrong low result in lanes 48..63 (the selected operand reads as zero: fma -> c, mul -> 0, add -> a) while a bf16x3 convolution workgroup
// runs on the same compute unit -- measured with a victim of nothing but such instructions, each checked against the plain instruction on the same
// registers (profiles/r05_packed_fp32_victim.txt: 0.12 % of the executions fail beside the convolution, none of 23 million alone; only the src1 select,
// only the low half, only the last sixteen lanes).  tests/test_packed_fp32_audit.py disassembles the built library and fails if ANY kernel holds one.
#if defined(__HIP_DEVICE_COMPILE__) && !(defined(STRAPS_TOOLS) && defined(STRAPS_ALLOW_PACKED_FP32))      // (tools build of a reproducer's victim: the kernels as they were)
#define STRAPS_NO_PACKED_FP32 __attribute__((target("no-packed-fp32-ops")))
#else
#define STRAPS_NO_PACKED_FP32      // (the host pass does not know the feature)
#endif

// Dynamic-LDS limit of a kernel above the 64 KiB default.  Function attributes are per device: `done` is the caller's static bit mask
// of the devices this kernel has been raised on (a process that drives several GPUs sets each once).
inline hipError_t straps_raise_dynamic_lds(const void* fn, size_t bytes, unsigned long long& done) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const unsigned long long bit = 1ull << (dev & 63);
    if (done & bit) return hipSuccess;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess) done |= bit;
    return e;
}
#define STRAPS_RAISE_LDS(fn, bytes, name)                                                                  \
    do {                                                                                                   \
        static unsigned long long done__ = 0;                                                              \
        hipError_t e__ = straps_raise_dynamic_lds((const void*)(fn), (size_t)(bytes), done__);             \
        if (e__ != hipSuccess) {                                                                           \
            straps_set_error("%s: cannot raise dynamic LDS to %zu: %s", name, (size_t)(bytes), hipGetErrorString(e__)); \
            return STRAPS_EHIP;                                                                            \
        }                                                                                                  \
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

// BatchNorm finalize helper: fixed-order fp64 sum of the per-block (s1, s2) partials part[k][C][2], k < nblocks, for 4
// consecutive channels per 256-thread block (grid = ceil(C/4)).  Lane group pl = tid>>2 walks k = pl, pl+64, ... with
// 32-byte coalesced segments; the 64 group sums are then added in order by the pl == 0 threads, which get `true`.
template <typename T>
__device__ __forceinline__ bool bn_partials_sum4(const T* __restrict__ part, int nblocks, int C, double& s1, double& s2, int& c) {
    __shared__ double red[256][2];
    const int cl = threadIdx.x & 3, pl = threadIdx.x >> 2;
    c = blockIdx.x * 4 + cl;
    double a1 = 0.0, a2 = 0.0;
    if (c < C) {
        // (sixteen independent loads in flight per trip -- round 5: with four, the 32 partial blocks a thread walks for a layer1 BatchNorm
        //  were eight dependent round trips, and the finalize launches, 40 per resnet18 step and 106 per resnet50 step, ran 5.4 us each; the
        //  additions stay in ascending k: same sums, bit for bit)
        int k = pl;
        for (; k + 15 * 64 < nblocks; k += 16 * 64) {
            T v0[16], v1[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const T* v = part + ((long long)(k + q * 64) * C + c) * 2;      // (sum, second sum) pair of a block: one 8- or 16-byte access
                v0[q] = v[0];
                v1[q] = v[1];
            }
#pragma unroll
            for (int q = 0; q < 16; ++q) { a1 += (double)v0[q]; a2 += (double)v1[q]; }
        }
#pragma unroll 8
        for (; k < nblocks; k += 64) {
            const T* v = part + ((long long)k * C + c) * 2;
            a1 += (double)v[0];
            a2 += (double)v[1];
        }
    }
    red[threadIdx.x][0] = a1;
    red[threadIdx.x][1] = a2;
    __syncthreads();
    if (pl != 0 || c >= C) return false;
    s1 = 0.0; s2 = 0.0;
    for (int q = 0; q < 64; ++q) { s1 += red[q * 4 + cl][0]; s2 += red[q * 4 + cl][1]; }
    return true;
}

// ---- three-plane bf16 representation of an fp32 value (csrc/conv_x3.hip): x = b1 + b2 + b3 exactly, round-to-nearest-even splits
typedef unsigned short u16;
typedef u16 u16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u16 bf16_rn(float x) {
    unsigned u = __float_as_uint(x);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (u16)(u >> 16);
}
__device__ __forceinline__ void split3(float x, u16& b1, u16& b2, u16& b3) {
    b1 = bf16_rn(x);
    // (an infinite x keeps its leading plane and zero residues -- inf - inf would put a NaN into the low planes, and a NaN times a zero
    //  weight poisons outputs the fp32 chain would leave finite; a NaN stays a NaN)
    const float r1 = (__float_as_uint(x) & 0x7fffffffu) == 0x7f800000u ? 0.f : x - __uint_as_float((unsigned)b1 << 16);       // exact
    b2 = bf16_rn(r1);
    const float r2 = r1 - __uint_as_float((unsigned)b2 << 16);      // exact; at most 8 significant bits are left
    b3 = bf16_rn(r2);
}
// four consecutive values -> the three planes at element offset i (i % 4 == 0, planes 8-byte aligned)
__device__ __forceinline__ void store_planes4(u16* __restrict__ planes, long long ps, long long i, const f32x4& v) {
    u16x4 q1, q2, q3;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        u16 b1, b2, b3;
        split3(v[e], b1, b2, b3);
        q1[e] = b1; q2[e] = b2; q3[e] = b3;
    }
    *reinterpret_cast<u16x4*>(planes + i) = q1;
    *reinterpret_cast<u16x4*>(planes + ps + i) = q2;
    *reinterpret_cast<u16x4*>(planes + 2 * ps + i) = q3;
}

// ---- chunk-major plane layout (what every bf16x3 kernel reads): a [rows][C] tensor's plane stores its 32-channel chunks outermost,
//   element (r, c) at ((c >> 5) * rows + r) * 32 + (c & 31),
// so that the 64 bytes one K chunk of one pixel contributes sit next to the neighbouring pixels' -- an LDS-DMA instruction that fetches a
// chunk for 16 consecutive rows reads 1 KiB of whole 128-byte lines.  (In the plain NHWC order of round 2 the same fetch touched HALF of
// each of 16 lines: tools/l2_line_probe.hip measures 16-18 TB/s of useful L2 -> LDS bytes for that pattern against 31-36 TB/s for whole
// lines -- the "13 TB/s copy ceiling" of the implicit GEMM was this.)  C % 32 == 0.
__device__ __forceinline__ long long cm_index(long long r, int c, long long rows) { return ((long long)(c >> 5) * rows + r) * 32 + (c & 31); }
// four consecutive channels c .. c+3 (c % 4 == 0) of row r -> the three planes
__device__ __forceinline__ void store_planes4_cm(u16* __restrict__ planes, long long ps, long long r, int c, long long rows, const f32x4& v) {
    store_planes4(planes, ps, cm_index(r, c, rows), v);
}

// Measurement switches (ablation instantiations that compute WRONG results, environment-selected A/B variants) exist only in the tools
// build: `python tools/build_tools_lib.py` compiles the same sources with -DSTRAPS_TOOLS into tools/bin/libstraps_hip_tools.so.  The
// product library never reads the environment: STRAPS_TOOL_ENV_INT is its compile-time default there.
#ifdef STRAPS_TOOLS
#include <stdlib.h>
#define STRAPS_TOOL_ENV_INT(name, dflt) (getenv(name) ? atoi(getenv(name)) : (dflt))
#else
#define STRAPS_TOOL_ENV_INT(name, dflt) (dflt)
#endif
// (the same macro serves switches a sweep changes from call to call: use it in a non-static expression)

static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// ---- launch geometry of the streaming BatchNorm kernels (bn_apply_kernel, csrc/elementwise.hip; bn_bwd_apply_kernel, csrc/backward.hip) ----
// 256-thread workgroups over n float4 elements, at most 4096 of them (grid-stride loops)
static inline unsigned straps_grid256(long long n) {
    const long long g = (n + 255) / 256;
    return (unsigned)(g > 256 * 16 ? 256 * 16 : (g < 1 ? 1 : g));
}
// the same, with grid x 256 a multiple of the row length C4 (float4 units) whenever a grid under the cap allows it: a grid-stride thread then
// keeps its four channels, and the kernels hoist the per-channel constants out of their loops
static inline unsigned straps_grid256_rows(long long n, int C4) {
    unsigned g = straps_grid256(n);
    if (C4 > 0 && (256 % C4) != 0) {
        long long a = C4, b = 256;
        while (b) { const long long t = a % b; a = b; b = t; }
        const long long m = C4 / a;            // smallest m with (m * 256) % C4 == 0
        if (m <= 256 * 16) { const long long up = ((g + m - 1) / m) * m; g = (unsigned)(up > 256 * 16 ? (256 * 16 / m) * m : up); }
    }
    return g;
}
// tiled form (a wave = 4 rows x 2 chunks of 32 channels: 256-byte runs of the chunk-major planes per wave store): 0 = use the linear form,
// else the column groups of 64 channels a workgroup's four waves sit on side by side (1, 2 or 4).  Product: C >= 256 with the rows in fours.
static inline int straps_bn_tiled(long long rows, int C4) {
    static const int mode = STRAPS_TOOL_ENV_INT("STRAPS_BN_TILED", 1);      // (A/B switch of the tools build: 0 off, 1 C >= 256 only, 2 also C = 64 / 128)
    if (!mode || (C4 & 15)) return 0;
    const int ncg = C4 >> 4;
    if (!(ncg == 1 || ncg == 2 || (ncg & 3) == 0) || (mode == 1 && ncg < 4)) return 0;
    const int wcg = ncg < 4 ? ncg : 4;
    return (rows % (16 / wcg)) == 0 ? wcg : 0;
}
// grid of the tiled form: a multiple of the column blocks, so that a thread's channels stay fixed over its trips
static inline unsigned straps_bn_tiled_grid(long long rows, int C4, int wcg) {
    const long long ncb = (C4 >> 4) / wcg, tiles = rows / (16 / wcg) * ncb;
    long long g = tiles < 256 * 16 ? tiles : 256 * 16;
    g = g / ncb * ncb;
    return (unsigned)(g < ncb ? ncb : g);
}
