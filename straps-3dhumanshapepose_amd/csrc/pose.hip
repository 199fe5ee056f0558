// pose.hip -- rotation representations (utils/rigid_transform_utils.py:27-41; smplx batch_rodrigues)
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void rot6d_kernel(const float* __restrict__ x6, long long ld, int per_row,
                                                    float* __restrict__ R, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const long long row = i / per_row;
    const int j = (int)(i - row * per_row);
    const float* x = x6 + row * ld + j * 6;
    // interleaved 3x2: a1 = (x0,x2,x4), a2 = (x1,x3,x5)
    const float a1x = x[0], a2x = x[1], a1y = x[2], a2y = x[3], a1z = x[4], a2z = x[5];
    const float n1 = fmaxf(sqrtf(a1x * a1x + a1y * a1y + a1z * a1z), 1e-12f);
    const float b1x = a1x / n1, b1y = a1y / n1, b1z = a1z / n1;
    const float d = b1x * a2x + b1y * a2y + b1z * a2z;
    const float ux = a2x - d * b1x, uy = a2y - d * b1y, uz = a2z - d * b1z;
    const float n2 = fmaxf(sqrtf(ux * ux + uy * uy + uz * uz), 1e-12f);
    const float b2x = ux / n2, b2y = uy / n2, b2z = uz / n2;
    const float b3x = b1y * b2z - b1z * b2y, b3y = b1z * b2x - b1x * b2z, b3z = b1x * b2y - b1y * b2x;
    float* o = R + i * 9;   // columns (b1,b2,b3)
    o[0] = b1x; o[1] = b2x; o[2] = b3x;
    o[3] = b1y; o[4] = b2y; o[5] = b3y;
    o[6] = b1z; o[7] = b2z; o[8] = b3z;
}

__global__ __launch_bounds__(256) STRAPS_NO_PACKED_FP32 void rodrigues_kernel(const float* __restrict__ aa, float* __restrict__ R, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float rx = aa[i * 3 + 0], ry = aa[i * 3 + 1], rz = aa[i * 3 + 2];
    const float ex = rx + 1e-8f, ey = ry + 1e-8f, ez = rz + 1e-8f;
    const float angle = sqrtf(ex * ex + ey * ey + ez * ez);
    const float dx = rx / angle, dy = ry / angle, dz = rz / angle;
    const float s = sinf(angle), c1 = 1.0f - cosf(angle);
    // K = skew(d); R = I + s K + (1-c) K^2, K^2 = d d^T - |d|^2 I (|d| only ~1: keep the exact product)
    const float k01 = -dz, k02 = dy, k10 = dz, k12 = -dx, k20 = -dy, k21 = dx;
    const float q00 = k01 * k10 + k02 * k20, q01 = k02 * k21, q02 = k01 * k12;
    const float q10 = k12 * k20, q11 = k10 * k01 + k12 * k21, q12 = k10 * k02;
    const float q20 = k21 * k10, q21 = k20 * k01, q22 = k20 * k02 + k21 * k12;
    float* o = R + i * 9;
    o[0] = 1.0f + c1 * q00;           o[1] = s * k01 + c1 * q01;        o[2] = s * k02 + c1 * q02;
    o[3] = s * k10 + c1 * q10;        o[4] = 1.0f + c1 * q11;           o[5] = s * k12 + c1 * q12;
    o[6] = s * k20 + c1 * q20;        o[7] = s * k21 + c1 * q21;        o[8] = 1.0f + c1 * q22;
}

// ---- camera projections of the module-level helpers (utils/cam_utils.py:5-26, 40-71) ----
// scaled orthographic: (u, v) = s * (x + tx, y + ty); cam rows [s, tx, ty] with stride ld_cam
__global__ __launch_bounds__(256) void ortho_project_kernel(const float* __restrict__ pts, const float* __restrict__ cam, int ld_cam,
                                                            float* __restrict__ out, long long B, int N) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= B * N) return;
    const long long b = i / N;
    const float s = cam[b * ld_cam], tx = cam[b * ld_cam + 1], ty = cam[b * ld_cam + 2];
    out[i * 2 + 0] = s * (pts[i * 3 + 0] + tx);
    out[i * 2 + 1] = s * (pts[i * 3 + 1] + ty);
}

// its gradient: dpts = (s du, s dv, 0); dcam[b] = (sum du (x + tx) + dv (y + ty), s sum du, s sum dv) -- one workgroup per body, fixed
// summation order (lane-strided partial sums, wave shuffle tree, four waves through LDS)
__global__ __launch_bounds__(256) void ortho_project_bwd_kernel(const float* __restrict__ pts, const float* __restrict__ cam, int ld_cam,
                                                                const float* __restrict__ dout, float* __restrict__ dpts,
                                                                float* __restrict__ dcam, int N) {
    __shared__ float red[4][3];
    const long long b = blockIdx.x;
    const float s = cam[b * ld_cam], tx = cam[b * ld_cam + 1], ty = cam[b * ld_cam + 2];
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    for (int n = threadIdx.x; n < N; n += 256) {
        const long long i = b * N + n;
        const float du = dout[i * 2], dv = dout[i * 2 + 1];
        if (dpts) { dpts[i * 3] = s * du; dpts[i * 3 + 1] = s * dv; dpts[i * 3 + 2] = 0.f; }
        a0 += du * (pts[i * 3] + tx) + dv * (pts[i * 3 + 1] + ty);
        a1 += du;
        a2 += dv;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a0 += __shfl_xor(a0, o, 64); a1 += __shfl_xor(a1, o, 64); a2 += __shfl_xor(a2, o, 64); }
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = a0; red[threadIdx.x >> 6][1] = a1; red[threadIdx.x >> 6][2] = a2; }
    __syncthreads();
    if (threadIdx.x == 0 && dcam) {
        dcam[b * 3 + 0] = (red[0][0] + red[1][0]) + (red[2][0] + red[3][0]);
        dcam[b * 3 + 1] = s * ((red[0][1] + red[1][1]) + (red[2][1] + red[3][1]));
        dcam[b * 3 + 2] = s * ((red[0][2] + red[1][2]) + (red[2][2] + red[3][2]));
    }
}

// perspective: p = R x + t; p /= p_z; (u, v) = first two rows of K p.  K: one 3x3 (k_stride = 0) or one per body (k_stride = 9)
__global__ __launch_bounds__(256) STRAPS_NO_PACKED_FP32 void persp_project_kernel(const float* __restrict__ pts, const float* __restrict__ rot, const float* __restrict__ tr,
                                                            const float* __restrict__ K, int k_stride, float* __restrict__ out, long long B, int N) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= B * N) return;
    const long long b = i / N;
    const float* R = rot + b * 9;
    const float* Kb = K + b * k_stride;
    const float x = pts[i * 3], y = pts[i * 3 + 1], z = pts[i * 3 + 2];
    float p0 = (R[0] * x + R[1] * y) + R[2] * z + tr[b * 3];
    float p1 = (R[3] * x + R[4] * y) + R[5] * z + tr[b * 3 + 1];
    const float p2 = (R[6] * x + R[7] * y) + R[8] * z + tr[b * 3 + 2];
    p0 = p0 / p2;
    p1 = p1 / p2;
    const float one = p2 / p2;
    out[i * 2 + 0] = (Kb[0] * p0 + Kb[1] * p1) + Kb[2] * one;
    out[i * 2 + 1] = (Kb[3] * p0 + Kb[4] * p1) + Kb[5] * one;
}

}  // namespace

extern "C" int straps_orthographic_project(const float* points, const float* cam, int ld_cam, float* out, long long batch, int n, void* stream) {
    STRAPS_REQUIRE(points && cam && out && batch > 0 && n > 0 && ld_cam >= 3, "straps_orthographic_project: bad arguments");
    hipLaunchKernelGGL(ortho_project_kernel, dim3((unsigned)((batch * n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, points, cam, ld_cam, out, batch, n);
    STRAPS_CHECK_LAUNCH("ortho_project_kernel");
    return STRAPS_OK;
}

extern "C" int straps_orthographic_project_bwd(const float* points, const float* cam, int ld_cam, const float* dout, float* dpoints, float* dcam,
                                               long long batch, int n, void* stream) {
    STRAPS_REQUIRE(points && cam && dout && (dpoints || dcam) && batch > 0 && n > 0 && ld_cam >= 3, "straps_orthographic_project_bwd: bad arguments");
    hipLaunchKernelGGL(ortho_project_bwd_kernel, dim3((unsigned)batch), dim3(256), 0, (hipStream_t)stream, points, cam, ld_cam, dout, dpoints, dcam, n);
    STRAPS_CHECK_LAUNCH("ortho_project_bwd_kernel");
    return STRAPS_OK;
}

extern "C" int straps_perspective_project(const float* points, const float* rotation, const float* translation, const float* cam_k, int k_per_body,
                                          float* out, long long batch, int n, void* stream) {
    STRAPS_REQUIRE(points && rotation && translation && cam_k && out && batch > 0 && n > 0, "straps_perspective_project: bad arguments");
    hipLaunchKernelGGL(persp_project_kernel, dim3((unsigned)((batch * n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, points, rotation, translation,
                       cam_k, k_per_body ? 9 : 0, out, batch, n);
    STRAPS_CHECK_LAUNCH("persp_project_kernel");
    return STRAPS_OK;
}

extern "C" int straps_rot6d_fwd(const float* x6, long long ld, int per_row, float* rotmats, long long rows, void* stream) {
    STRAPS_REQUIRE(x6 && rotmats, "straps_rot6d_fwd: null pointer");
    STRAPS_REQUIRE(rows > 0 && per_row > 0 && ld >= 6LL * per_row, "straps_rot6d_fwd: bad shape rows=%lld per_row=%d ld=%lld", rows, per_row, ld);
    const long long n = rows * per_row;
    hipLaunchKernelGGL(rot6d_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x6, ld, per_row, rotmats, n);
    STRAPS_CHECK_LAUNCH("rot6d_kernel");
    return STRAPS_OK;
}

extern "C" int straps_rodrigues_fwd(const float* aa, float* rotmats, long long n, void* stream) {
    STRAPS_REQUIRE(aa && rotmats && n > 0, "straps_rodrigues_fwd: bad arguments");
    hipLaunchKernelGGL(rodrigues_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, aa, rotmats, n);
    STRAPS_CHECK_LAUNCH("rodrigues_kernel");
    return STRAPS_OK;
}
