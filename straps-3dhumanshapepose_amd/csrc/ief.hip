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

__global__ __launch_bounds__(256) void broadcast_rows_kernel(const float* __restrict__ row, int cols, float* __restrict__ dst,
                                                             int ld_dst, int m) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)m * ld_dst) return;
    const int c = (int)(i % ld_dst);
    dst[i] = c < cols ? row[c] : 0.f;
}

// ---- several small strided GEMMs in ONE launch (blockIdx.z = problem): the IEF backward's weight / bias gradients of all three
// iterations (K = the three iterations' rows stacked: one reduction instead of three accumulating launches), the feature gradient, and --
// with one problem -- each link of its dependent chain with the elementwise work around it fused into the epilogue (ReLU mask, the
// addend of est_out = est_in + ..., the running sum dc1 += dh1).  One 32x32 output tile per
// workgroup, the K groups of 8 dealt round-robin to 8 waves, partial tiles combined through LDS in a fixed order.
struct GemmBatch {
    straps_gemm_desc_t d[STRAPS_GEMM_MULTI_MAX];
};

__global__ __launch_bounds__(64 * LW) void gemm_multi_kernel(GemmBatch batch) {
    const straps_gemm_desc_t& p = batch.d[blockIdx.z];
    const int M = p.m, N = p.n, K = p.k;
    const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
    if (m0 >= M || n0 >= N) return;                        // a smaller problem of the same launch
    __shared__ float red[LW][32][33];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int mr = min(m0 + i, M - 1), nr = min(n0 + i, N - 1);
    const float* ap = p.a + (long long)mr * p.sam;
    const float* bp = p.b + (long long)nr * p.sbn;
    const long long sak = p.sak, sbk = p.sbk;
    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
    const int G = (K + 7) >> 3;
    for (int g = wave; g < G; g += 2 * LW) {
        float av[8], bv[8];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k = (g + LW * u) * 8 + 4 * h + e;
                const bool ok = k < K;
                av[u * 4 + e] = ok ? ap[(long long)k * sak] : 0.f;
                bv[u * 4 + e] = ok ? bp[(long long)k * sbk] : 0.f;
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) acc = mfma32(av[e], bv[e], acc);
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) red[wave][mfma_row(q, lane)][i] = acc[q];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 1024 / (64 * LW); ++q) {
        const int idx = threadIdx.x + 64 * LW * q;
        const int row = idx >> 5, col = idx & 31;
        const int m = m0 + row, n = n0 + col;
        if (m < M && n < N) {
            float v = ((red[0][row][col] + red[1][row][col]) + (red[2][row][col] + red[3][row][col])) +
                      ((red[4][row][col] + red[5][row][col]) + (red[6][row][col] + red[7][row][col]));
            if (p.addend) v += p.addend[(long long)m * p.ldadd + n];
            if (p.mask) v = p.mask[(long long)m * p.ldmask + n] > 0.f ? v : 0.f;
            float* o = p.c + (long long)m * p.ldc + n;
            *o = p.accumulate ? *o + v : v;
            if (p.c2) {
                float* o2 = p.c2 + (long long)m * p.ldc2 + n;
                *o2 = p.accumulate2 ? *o2 + v : v;
            }
        }
    }
}

// the three repacked views of the IEF weights in one launch: fc1.weight [H1][F + P] -> w1f [H1][F] and w1e [H1][ld_e] (zero padded),
// fc3.weight [P][H2] -> w3 [P rounded up to 32][H2] (zero rows)
__global__ __launch_bounds__(256) void ief_pack_kernel(const float* __restrict__ fc1, const float* __restrict__ fc3, float* __restrict__ w1f,
                                                       float* __restrict__ w1e, float* __restrict__ w3, int F, int P, int H1, int H2, int ld_e, int p_pad) {
    const long long n1 = (long long)H1 * F, n2 = (long long)H1 * ld_e, n3 = (long long)p_pad * H2;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n1 + n2 + n3; i += (long long)gridDim.x * 256) {
        if (i < n1) {
            const int r = (int)(i / F), c = (int)(i - (long long)r * F);
            w1f[i] = fc1[(long long)r * (F + P) + c];
        } else if (i < n1 + n2) {
            const long long j = i - n1;
            const int r = (int)(j / ld_e), c = (int)(j - (long long)r * ld_e);
            w1e[j] = c < P ? fc1[(long long)r * (F + P) + F + c] : 0.f;
        } else {
            const long long j = i - n1 - n2;
            const int r = (int)(j / H2);
            w3[j] = r < P ? fc3[j] : 0.f;
        }
    }
}

}  // namespace

extern "C" int straps_gemm_multi(const straps_gemm_desc_t* descs_host, int n, void* stream) {
    STRAPS_REQUIRE(descs_host && n >= 1 && n <= STRAPS_GEMM_MULTI_MAX, "straps_gemm_multi: need 1..%d problem descriptors (host memory)", STRAPS_GEMM_MULTI_MAX);
    GemmBatch batch;
    int gx = 0, gy = 0;
    for (int i = 0; i < n; ++i) {
        const straps_gemm_desc_t& d = descs_host[i];
        STRAPS_REQUIRE(d.a && d.b && d.c && d.m > 0 && d.n > 0 && d.k > 0, "straps_gemm_multi: problem %d: null operand or empty shape (m=%d n=%d k=%d)", i, d.m, d.n, d.k);
        batch.d[i] = d;
        gx = (d.n + 31) / 32 > gx ? (d.n + 31) / 32 : gx;
        gy = (d.m + 31) / 32 > gy ? (d.m + 31) / 32 : gy;
    }
    for (int i = n; i < STRAPS_GEMM_MULTI_MAX; ++i) batch.d[i] = descs_host[0];
    hipLaunchKernelGGL(gemm_multi_kernel, dim3(gx, gy, n), dim3(64 * LW), 0, (hipStream_t)stream, batch);
    STRAPS_CHECK_LAUNCH("gemm_multi_kernel");
    return STRAPS_OK;
}

extern "C" int straps_ief_pack(const float* fc1_w, const float* fc3_w, float* w1f, float* w1e, float* w3, int f, int p, int h1, int h2,
                               int ld_e, void* stream) {
    STRAPS_REQUIRE(fc1_w && fc3_w && w1f && w1e && w3, "straps_ief_pack: null pointer");
    STRAPS_REQUIRE(f > 0 && p > 0 && h1 > 0 && h2 > 0 && ld_e >= p, "straps_ief_pack: bad shape (f=%d p=%d h1=%d h2=%d ld_e=%d)", f, p, h1, h2, ld_e);
    const int p_pad = (p + 31) / 32 * 32;
    const long long total = (long long)h1 * f + (long long)h1 * ld_e + (long long)p_pad * h2;
    const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL(ief_pack_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, fc1_w, fc3_w, w1f, w1e, w3, f, p, h1, h2, ld_e, p_pad);
    STRAPS_CHECK_LAUNCH("ief_pack_kernel");
    return STRAPS_OK;
}

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

extern "C" int straps_broadcast_rows(const float* row, int cols, float* dst, int ld_dst, int m, void* stream) {
    STRAPS_REQUIRE(row && dst && cols > 0 && ld_dst >= cols && m > 0, "straps_broadcast_rows: bad arguments");
    const long long n = (long long)m * ld_dst;
    hipLaunchKernelGGL(broadcast_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, row, cols, dst, ld_dst, m);
    STRAPS_CHECK_LAUNCH("broadcast_rows_kernel");
    return STRAPS_OK;
}
