// metrics.hip -- on-device evaluation metrics: per-sample sums of point-wise L2 errors, raw, after
// scale+translation correction and after Procrustes (similarity) alignment -- what the reference computes per batch on
// the CPU with numpy + a per-sample SVD loop (metrics/train_loss_and_metrics_tracker.py:127-197,
// utils/eval_utils.py:7-85) after copying 4 x [B,6890,3] to the host.
//
// One workgroup per sample, two passes over its points: (1) fp64 sums for the means, variances and the 3x3 cross
// covariance K = X1 X2^T; one thread solves the orthogonal Procrustes problem (Jacobi eigen-decomposition of K^T K,
// R = V Z U^T with det(R) = +1, scale = tr(RK)/var1, t = mu2 - s R mu1); (2) the three error sums.  Fixed reduction order.
#include "common.h"

namespace {

__device__ void jacobi_eig3(double A[3][3], double V[3][3]) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) V[i][j] = i == j ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 16; ++sweep) {
        const double off = A[0][1] * A[0][1] + A[0][2] * A[0][2] + A[1][2] * A[1][2];
        if (off < 1e-300) break;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                if (fabs(A[p][q]) < 1e-300) continue;
                const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 3; ++k) {
                    const double akp = A[k][p], akq = A[k][q];
                    A[k][p] = c * akp - s * akq;
                    A[k][q] = s * akp + c * akq;
                }
                for (int k = 0; k < 3; ++k) {
                    const double apk = A[p][k], aqk = A[q][k];
                    A[p][k] = c * apk - s * aqk;
                    A[q][k] = s * apk + c * aqk;
                }
                for (int k = 0; k < 3; ++k) {
                    const double vkp = V[k][p], vkq = V[k][q];
                    V[k][p] = c * vkp - s * vkq;
                    V[k][q] = s * vkp + c * vkq;
                }
            }
    }
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__global__ __launch_bounds__(256) void point_metrics_kernel(const float* __restrict__ P, const float* __restrict__ T,
                                                            float* __restrict__ out, int N) {
    __shared__ double red[4][17];
    __shared__ double sol[20];          // mu1[3] mu2[3] R[9] scale t[3] sc_ratio
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* p0 = P + (long long)b * N * 3;
    const float* t0 = T + (long long)b * N * 3;
    double a[17];
#pragma unroll
    for (int i = 0; i < 17; ++i) a[i] = 0.0;
    for (int n = tid; n < N; n += 256) {
        const double x[3] = {p0[n * 3], p0[n * 3 + 1], p0[n * 3 + 2]};
        const double y[3] = {t0[n * 3], t0[n * 3 + 1], t0[n * 3 + 2]};
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            a[i] += x[i];
            a[3 + i] += y[i];
            a[6] += x[i] * x[i];
            a[7] += y[i] * y[i];
#pragma unroll
            for (int j = 0; j < 3; ++j) a[8 + i * 3 + j] += x[i] * y[j];
        }
    }
#pragma unroll
    for (int i = 0; i < 17; ++i) {
        const double s = wave_sum(a[i]);
        if (lane == 0) red[wave][i] = s;
    }
    __syncthreads();
    if (tid == 0) {
        double s[17];
        for (int i = 0; i < 17; ++i) s[i] = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
        const double n = (double)N;
        double mu1[3], mu2[3], K[3][3];
        for (int i = 0; i < 3; ++i) { mu1[i] = s[i] / n; mu2[i] = s[3 + i] / n; }
        const double var1 = s[6] - n * (mu1[0] * mu1[0] + mu1[1] * mu1[1] + mu1[2] * mu1[2]);
        const double var2 = s[7] - n * (mu2[0] * mu2[0] + mu2[1] * mu2[1] + mu2[2] * mu2[2]);
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) K[i][j] = s[8 + i * 3 + j] - n * mu1[i] * mu2[j];
        // K = U S V^T  ->  K^T K = V S^2 V^T
        double A[3][3], V[3][3];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) A[i][j] = K[0][i] * K[0][j] + K[1][i] * K[1][j] + K[2][i] * K[2][j];
        jacobi_eig3(A, V);
        int o0 = 0, o1 = 1, o2 = 2;   // order eigenvalues descending
        if (A[o0][o0] < A[o1][o1]) { int t = o0; o0 = o1; o1 = t; }
        if (A[o0][o0] < A[o2][o2]) { int t = o0; o0 = o2; o2 = t; }
        if (A[o1][o1] < A[o2][o2]) { int t = o1; o1 = o2; o2 = t; }
        double v1[3] = {V[0][o0], V[1][o0], V[2][o0]}, v2[3] = {V[0][o1], V[1][o1], V[2][o1]};
        double v3[3] = {v1[1] * v2[2] - v1[2] * v2[1], v1[2] * v2[0] - v1[0] * v2[2], v1[0] * v2[1] - v1[1] * v2[0]};
        double u1[3], u2[3];
        for (int i = 0; i < 3; ++i) {
            u1[i] = K[i][0] * v1[0] + K[i][1] * v1[1] + K[i][2] * v1[2];
            u2[i] = K[i][0] * v2[0] + K[i][1] * v2[1] + K[i][2] * v2[2];
        }
        double l1 = sqrt(u1[0] * u1[0] + u1[1] * u1[1] + u1[2] * u1[2]);
        for (int i = 0; i < 3; ++i) u1[i] /= (l1 > 0 ? l1 : 1.0);
        const double d12 = u1[0] * u2[0] + u1[1] * u2[1] + u1[2] * u2[2];
        for (int i = 0; i < 3; ++i) u2[i] -= d12 * u1[i];
        double l2 = sqrt(u2[0] * u2[0] + u2[1] * u2[1] + u2[2] * u2[2]);
        for (int i = 0; i < 3; ++i) u2[i] /= (l2 > 0 ? l2 : 1.0);
        const double u3[3] = {u1[1] * u2[2] - u1[2] * u2[1], u1[2] * u2[0] - u1[0] * u2[2], u1[0] * u2[1] - u1[1] * u2[0]};
        // R = V' U'^T with U' = [u1 u2 u1xu2], V' = [v1 v2 v1xv2]  (== V Z U^T of utils/eval_utils.py:38-42)
        double R[3][3];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) R[i][j] = v1[i] * u1[j] + v2[i] * u2[j] + v3[i] * u3[j];
        double trRK = 0.0;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) trRK += R[i][j] * K[j][i];
        const double scale = trRK / var1;
        for (int i = 0; i < 3; ++i) {
            sol[i] = mu1[i];
            sol[3 + i] = mu2[i];
            sol[16 + i] = mu2[i] - scale * (R[i][0] * mu1[0] + R[i][1] * mu1[1] + R[i][2] * mu1[2]);
            for (int j = 0; j < 3; ++j) sol[6 + i * 3 + j] = R[i][j];
        }
        sol[15] = scale;
        sol[19] = sqrt(var2 / n) / sqrt(var1 / n);       // T_scale / P_scale  (utils/eval_utils.py:75-83)
    }
    __syncthreads();
    double e0 = 0.0, e1 = 0.0, e2 = 0.0;
    const double sc = sol[19], s = sol[15];
    for (int n = tid; n < N; n += 256) {
        const double x[3] = {p0[n * 3], p0[n * 3 + 1], p0[n * 3 + 2]};
        const double y[3] = {t0[n * 3], t0[n * 3 + 1], t0[n * 3 + 2]};
        double d0 = 0.0, d1 = 0.0, d2 = 0.0;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const double r0 = x[i] - y[i];
            const double r1 = (x[i] - sol[i]) * sc + sol[3 + i] - y[i];
            const double r2 = s * (sol[6 + i * 3] * x[0] + sol[7 + i * 3] * x[1] + sol[8 + i * 3] * x[2]) + sol[16 + i] - y[i];
            d0 += r0 * r0; d1 += r1 * r1; d2 += r2 * r2;
        }
        e0 += sqrt(d0); e1 += sqrt(d1); e2 += sqrt(d2);
    }
    e0 = wave_sum(e0); e1 = wave_sum(e1); e2 = wave_sum(e2);
    __syncthreads();
    if (lane == 0) { red[wave][0] = e0; red[wave][1] = e1; red[wave][2] = e2; }
    __syncthreads();
    if (tid < 3) out[b * 3 + tid] = (float)((red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]));
}

}  // namespace

extern "C" int straps_point_metrics(const float* pred, const float* target, float* out3, long long batch, int npoints, void* stream) {
    STRAPS_REQUIRE(pred && target && out3 && batch > 0 && npoints >= 3, "straps_point_metrics: bad arguments");
    STRAPS_REQUIRE(batch < (1LL << 31), "straps_point_metrics: batch too large");
    hipLaunchKernelGGL(point_metrics_kernel, dim3((unsigned)batch), dim3(256), 0, (hipStream_t)stream, pred, target, out3, npoints);
    STRAPS_CHECK_LAUNCH("point_metrics_kernel");
    return STRAPS_OK;
}
