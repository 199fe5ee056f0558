// ief.hip -- the IEF regressor's fully connected layers on the fp32 MFMA
// (replaces nn.Linear x3 per iteration, models/ief_module.py:16-18,55-58).
//
// The batch is tiny (M = 64 rows per GPU) so each layer is a skinny GEMM: one 32x32 output tile
// per workgroup, the K dimension split across the 4 waves and reduced through LDS, operands read
// straight from L2 in MFMA fragment order (x and w are both k-contiguous).  The concatenation
// [features, estimate] of the reference (models/ief_module.py:54,58) is never materialised: fc1's
// weight is split into its two column blocks and the feature half is applied once, outside the
// iteration loop (host side, ief_module.py).
#include "common.h"

namespace {

constexpr int LW = 8;      // waves per workgroup: the K groups of one 32x32 output tile are dealt round-robin to them (these GEMMs are
                           // tiny and latency-bound: 8 short dependent load chains instead of 4 longer ones)
__global__ __launch_bounds__(64 * LW) void linear_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ w, int ldw,
                                                     const float* __restrict__ bias, const float* addend, float* out, int ldo,
                                                     int M, int N, int K, int relu) {
    __shared__ float red[LW][32][33];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
    const int i = lane & 31, h = lane >> 5;
    const int mr = min(m0 + i, M - 1);
    const float* xp = x + (long long)mr * ldx + 4 * h;
    const float* wp = w + (long long)(n0 + i) * ldw + 4 * h;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int G = K >> 3;
#pragma unroll 4
    for (int g = wave; g < G; g += LW) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(xp + 8 * g);
        const f32x4 b = *reinterpret_cast<const f32x4*>(wp + 8 * g);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = mfma32(a[e], b[e], acc);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][mfma_row(r, lane)][i] = acc[r];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 1024 / (64 * LW); ++q) {
        const int idx = tid + 64 * LW * q;
        const int row = idx >> 5, col = idx & 31;
        const int m = m0 + row, n = n0 + col;
        if (m < M && n < N) {
            float v = ((red[0][row][col] + red[1][row][col]) + (red[2][row][col] + red[3][row][col])) +
                      ((red[4][row][col] + red[5][row][col]) + (red[6][row][col] + red[7][row][col]));
            if (bias) v += bias[n];
            if (addend) v += addend[(long long)m * ldo + n];
            if (relu) v = fmaxf(v, 0.f);
            out[(long long)m * ldo + n] = v;
        }
    }
}

__global__ __launch_bounds__(256) void pad_copy_kernel(const float* __restrict__ src, int ld_src, int col0, int rows, int cols,
                                                       float* __restrict__ dst, int ld_dst, int rows_pad) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)rows_pad * ld_dst) return;
    const int r = (int)(i / ld_dst), c = (int)(i - (long long)r * ld_dst);
    dst[i] = (r < rows && c < cols) ? src[(long long)r * ld_src + col0 + c] : 0.f;
}

__global__ __launch_bounds__(256) void broadcast_rows_kernel(const float* __restrict__ row, int cols, float* __restrict__ dst,
                                                             int ld_dst, int m) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)m * ld_dst) return;
    const int c = (int)(i % ld_dst);
    dst[i] = c < cols ? row[c] : 0.f;
}

}  // namespace

extern "C" int straps_linear_fwd(const float* x, int ldx, const float* w, int ldw, const float* bias, const float* addend,
                                 float* out, int ldo, int m, int n, int kdim, int relu, void* stream) {
    STRAPS_REQUIRE(x && w && out, "straps_linear_fwd: null pointer");
    STRAPS_REQUIRE(m > 0 && n > 0 && kdim > 0, "straps_linear_fwd: empty problem m=%d n=%d k=%d", m, n, kdim);
    STRAPS_REQUIRE((kdim & 7) == 0 && (ldx & 3) == 0 && (ldw & 3) == 0 && ldx >= kdim && ldw >= kdim,
                   "straps_linear_fwd: need kdim%%8==0, ldx%%4==0, ldw%%4==0 (kdim=%d ldx=%d ldw=%d)", kdim, ldx, ldw);
    STRAPS_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)w & 15) == 0, "straps_linear_fwd: x and w must be 16-byte aligned");
    dim3 grid((n + 31) / 32, (m + 31) / 32);
    hipLaunchKernelGGL(linear_kernel, grid, dim3(64 * LW), 0, (hipStream_t)stream, x, ldx, w, ldw, bias, addend, out, ldo, m, n, kdim, relu);
    STRAPS_CHECK_LAUNCH("linear_kernel");
    return STRAPS_OK;
}

extern "C" int straps_pad_copy(const float* src, int ld_src, int col0, int rows, int cols, float* dst, int ld_dst, int rows_pad,
                               void* stream) {
    STRAPS_REQUIRE(src && dst && rows > 0 && cols > 0 && ld_dst >= cols && rows_pad >= rows, "straps_pad_copy: bad arguments");
    const long long n = (long long)rows_pad * ld_dst;
    hipLaunchKernelGGL(pad_copy_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, ld_src, col0, rows, cols,
                       dst, ld_dst, rows_pad);
    STRAPS_CHECK_LAUNCH("pad_copy_kernel");
    return STRAPS_OK;
}

extern "C" int straps_broadcast_rows(const float* row, int cols, float* dst, int ld_dst, int m, void* stream) {
    STRAPS_REQUIRE(row && dst && cols > 0 && ld_dst >= cols && m > 0, "straps_broadcast_rows: bad arguments");
    const long long n = (long long)m * ld_dst;
    hipLaunchKernelGGL(broadcast_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, row, cols, dst, ld_dst, m);
    STRAPS_CHECK_LAUNCH("broadcast_rows_kernel");
    return STRAPS_OK;
}
