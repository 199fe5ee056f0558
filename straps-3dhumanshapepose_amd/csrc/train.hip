// train.hip -- train-step glue kernels: network-input construction, prediction heads + multi-task loss (forward
// and backward fused), Adam.
#include <math.h>

#include "common.h"

namespace {

// COCO / H36M-LSP joint selections out of the 90-joint superset (reference config.py:27-32)
__constant__ int c_coco[17] = {24, 26, 25, 28, 27, 16, 17, 18, 19, 20, 21, 1, 2, 4, 5, 7, 8};
__constant__ int c_h36m14[14] = {73 + 6, 73 + 5, 73 + 4, 73 + 1, 73 + 2, 73 + 3, 73 + 16, 73 + 15, 73 + 14, 73 + 11, 73 + 12, 73 + 13, 73 + 8, 73 + 10};

// =====================================================================================================
// input construction: binary silhouette + Gaussian joint heat-maps (utils/label_conversions.py:48-55,90-127)
// =====================================================================================================
// one (body, channel) plane per blockIdx.y, 4 consecutive pixels of a row per thread (32-bit index math, float4
// stores: the kernel is a 302 MB write at B = 64 and should cost no more than that)
__device__ __forceinline__ float heat_value(int x, int y, int jx, int jy, int WH, int size, float step, float two_var) {
    // size = 2 std (the Gaussian is truncated two standard deviations from the joint, utils/label_conversions.py:102)
    if (!(jx > -size && jy > -size && jx < WH - 1 + size && jy < WH - 1 + size)) return 0.f;
    const int hx0 = max(0, jx - size), hx1 = min(WH - 1, jx + size);
    const int hy0 = max(0, jy - size), hy1 = min(WH - 1, jy + size);
    if (!(x >= hx0 && x < hx1 && y >= hy0 && y < hy1)) return 0.f;
    const int gx = x - hx0 + max(0, size - jx), gy = y - hy0 + max(0, size - jy);
    // torch.linspace(-size, size, 2 size): start + i*step in the first half, end - (2 size - 1 - i)*step in the second
    const float fs = (float)size;
    const float lx = gx < size ? -fs + step * gx : fs - step * (2 * size - 1 - gx);
    const float ly = gy < size ? -fs + step * gy : fs - step * (2 * size - 1 - gy);
    const float d = sqrtf(lx * lx + ly * ly);
    return expf(-(d * d / two_var));
}

__global__ __launch_bounds__(256) void build_proxy_kernel(const float* __restrict__ seg, const float* __restrict__ j2d,
                                                          float* __restrict__ out, int B, int NJ, int WH, int size, float step, float two_var) {
    const int plane = blockIdx.y;                       // b * (NJ + 1) + ch
    const int b = plane / (NJ + 1), ch = plane - b * (NJ + 1);
    const int npix = WH * WH;
    float* o = out + (long long)plane * npix;
    int jx = 0, jy = 0;
    if (ch > 0) {
        jx = (int)j2d[((long long)b * NJ + ch - 1) * 2 + 0];     // truncation toward zero == .int()
        jy = (int)j2d[((long long)b * NJ + ch - 1) * 2 + 1];
    }
    const float* s = seg + (long long)b * npix;
    const bool vec = (WH & 3) == 0;
    for (int p = (blockIdx.x * 256 + threadIdx.x) * 4; p < npix; p += gridDim.x * 1024) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int q = p + e;
            v[e] = 0.f;
            if (q < npix) {
                if (ch == 0) v[e] = s[q] != 0.f ? 1.f : 0.f;
                else {
                    const int y = q / WH, x = q - y * WH;
                    v[e] = heat_value(x, y, jx, jy, WH, size, step, two_var);
                }
            }
        }
        if (vec) *reinterpret_cast<f32x4*>(o + p) = f32x4{v[0], v[1], v[2], v[3]};
        else
            for (int e = 0; e < 4; ++e)
                if (p + e < npix) o[p + e] = v[e];
    }
}

// The same planes AND the stem's non-zero bit map in one pass (round 4): the layout of stem_nzmask_kernel (csrc/stem.hip) -- a wave owns one
// mask word = 4 rows x 256 columns of a plane, 32 cells of 4 x 8 -- with the values computed instead of loaded.  Lane l writes the four
// rows of half (l >> 5) of cell (l & 31): every store instruction of the wave is one contiguous KiB; a cell's bit is the OR of its two
// halves, which one 64-lane ballot delivers as (m | m >> 32).  Cells that cannot intersect the joint's window are written as zeros without
// evaluating anything -- all but 4-9 of a heat-map plane's 2048.  Bit-identical planes, and the map straps_stem_nzmask would read back
// from them (the 302 MB read of that pass -- 105-113 us at 64 bodies -- is gone).
__global__ __launch_bounds__(256) void build_proxy_nz_kernel(const float* __restrict__ seg, const float* __restrict__ j2d, float* __restrict__ out,
                                                             unsigned* __restrict__ mask, long long nwords, int NJ, int WH, int size, float step,
                                                             float two_var) {
    const int lane = threadIdx.x & 63;
    const long long word = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (word >= nwords) return;                         // (whole waves leave: the ballot below is among the lanes of one word)
    const int HC = (WH + 3) >> 2, WW = (WH + 255) >> 8;
    const int ww = (int)(word % WW);
    const long long t = word / WW;
    const int hc = (int)(t % HC);
    const int plane = (int)(t / HC);                    // b * (NJ + 1) + ch
    const int b = plane / (NJ + 1), ch = plane - b * (NJ + 1);
    const int x0 = ww * 256 + (lane & 31) * 8 + (lane >> 5) * 4, y0 = hc * 4;
    const int npix = WH * WH;
    float* o = out + (long long)plane * npix;
    bool nz = false;
    if (x0 < WH) {
        f32x4 v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (ch == 0) {
            const float* s = seg + (long long)b * npix;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (y0 + r < WH) {
                    const f32x4 a = *reinterpret_cast<const f32x4*>(s + (y0 + r) * WH + x0);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[r][e] = a[e] != 0.f ? 1.f : 0.f;
                }
        } else {
            const int jx = (int)j2d[((long long)b * NJ + ch - 1) * 2 + 0];     // truncation toward zero == .int()
            const int jy = (int)j2d[((long long)b * NJ + ch - 1) * 2 + 1];
            // heat_value is zero outside [j - size, j + size) in x and y: skip the cells that lie outside (a conservative test; the values
            // inside are evaluated by the same function as build_proxy_kernel's)
            if (x0 + 3 >= jx - size && x0 < jx + size && y0 + 3 >= jy - size && y0 < jy + size) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (y0 + r < WH)
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[r][e] = heat_value(x0 + e, y0 + r, jx, jy, WH, size, step, two_var);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (y0 + r < WH) {
                *reinterpret_cast<f32x4*>(o + (y0 + r) * WH + x0) = v[r];
                nz = nz || v[r][0] != 0.f || v[r][1] != 0.f || v[r][2] != 0.f || v[r][3] != 0.f;
            }
    }
    const unsigned long long m = __ballot(nz);
    if (lane == 0) mask[word] = (unsigned)(m | (m >> 32));
}

// =====================================================================================================
// heads + loss
// =====================================================================================================
// ws layout (floats): [0..1023] verts partial sums | [1024 ..) per-body sums [B][8] | then scalars
constexpr int NVB = 1024;

__global__ __launch_bounds__(256) void loss_verts_partial_kernel(const float* __restrict__ p, const float* __restrict__ t, long long n,
                                                                 float* __restrict__ part) {
    __shared__ float red[4];
    float s = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float d = p[i] - t[i];
        s = fmaf(d, d, s);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__device__ __forceinline__ bool joint_visible(float x, float y, float wh) { return !(x > wh || y > wh || x < 0.f || y < 0.f); }

__global__ __launch_bounds__(256) void loss_heads_kernel(const float* __restrict__ joints, const float* __restrict__ est, int ld_est,
                                                         const float* __restrict__ prot, const float* __restrict__ tj2d,
                                                         const float* __restrict__ tj3d, const float* __restrict__ tshape,
                                                         const float* __restrict__ trot, float* __restrict__ body_sums, long long B,
                                                         float wh) {
    const long long b = (long long)blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    const float* J = joints + b * 270;
    const float* e = est + b * ld_est;
    const float s = e[0], tx = e[1], ty = e[2];
    float sq2 = 0.f, nvis = 0.f, sq3 = 0.f, sqs = 0.f, sqr = 0.f;
    for (int k = 0; k < 17; ++k) {
        const float lx = tj2d[(b * 17 + k) * 2 + 0], ly = tj2d[(b * 17 + k) * 2 + 1];
        if (joint_visible(lx, ly, wh)) {
            const float* pj = J + c_coco[k] * 3;
            const float u = s * (pj[0] + tx), v = s * (pj[1] + ty);
            const float du = u - ((2.0f * lx) / wh - 1.0f), dv = v - ((2.0f * ly) / wh - 1.0f);
            sq2 += du * du + dv * dv;
            nvis += 1.f;
        }
    }
    for (int k = 0; k < 14; ++k) {
        const float* pj = J + c_h36m14[k] * 3;
        const float* tj = tj3d + (b * 14 + k) * 3;
        for (int c = 0; c < 3; ++c) { const float d = pj[c] - tj[c]; sq3 += d * d; }
    }
    for (int l = 0; l < 10; ++l) { const float d = e[147 + l] - tshape[b * 10 + l]; sqs += d * d; }
    for (int q = 0; q < 216; ++q) { const float d = prot[b * 216 + q] - trot[b * 216 + q]; sqr += d * d; }
    float* o = body_sums + b * 8;
    o[0] = sq2; o[1] = nvis; o[2] = sq3; o[3] = sqs; o[4] = sqr;
}

// single block: fixed-order fp64 sums -> MSEs, weighted losses, d/dlogvar, gradient coefficients
__global__ __launch_bounds__(256) void loss_finalize_kernel(const float* __restrict__ vpart, const float* __restrict__ body_sums,
                                                            const float* __restrict__ log_vars, float* __restrict__ loss_out,
                                                            float* __restrict__ dlogvar, float* __restrict__ coef, long long B,
                                                            const float* __restrict__ j2d_count, float count_scale) {
    __shared__ double red[256][6];
    double a[6] = {0, 0, 0, 0, 0, 0};
    for (int i = threadIdx.x; i < NVB; i += 256) a[0] += (double)vpart[i];
    for (long long b = threadIdx.x; b < B; b += 256) {
        const float* o = body_sums + b * 8;
        a[1] += o[0]; a[2] += o[1]; a[3] += o[2]; a[4] += o[3]; a[5] += o[4];
    }
    for (int q = 0; q < 6; ++q) red[threadIdx.x][q] = a[q];
    __syncthreads();
    if (threadIdx.x == 0) {
        double t[6] = {0, 0, 0, 0, 0, 0};
        for (int i = 0; i < 256; ++i)
            for (int q = 0; q < 6; ++q) t[q] += red[i][q];
        const double nvis = t[2];
        // element counts of the 'mean' reductions: verts B*6890*3, joints2D nvis*2, joints3D B*14*3, shape B*10, pose B*216.
        // Data parallel with the GLOBAL masked mean (straps_loss_fwd_bwd_gm): the joints2D denominator is the job's visible-joint count
        // over the world size, so that the average over ranks of this rank's value -- and of its gradients -- is the masked mean over
        // the global batch.
        const double n2 = j2d_count ? (double)j2d_count[0] * (double)count_scale : nvis;
        const double cnt[5] = {(double)B * 20670.0, n2 * 2.0, (double)B * 42.0, (double)B * 10.0, (double)B * 216.0};
        const double sq[5] = {t[0], t[1], t[3], t[4], t[5]};
        double total = 0.0;
        for (int k = 0; k < 5; ++k) {
            const double mse = sq[k] / cnt[k];            // 0/0 = NaN for no visible joint, like nn.MSELoss on an empty tensor
            const double s = (double)log_vars[k];
            const double w = exp(-s);
            loss_out[1 + k] = (float)(mse * w);
            loss_out[6 + k] = (float)mse;
            total += mse * w + s;
            dlogvar[k] = (float)(-mse * w + 1.0);
            coef[k] = (float)(2.0 * w / cnt[k]);
        }
        loss_out[0] = (float)total;
        loss_out[11] = (float)nvis;
    }
}

__global__ __launch_bounds__(256) void loss_grad_verts_kernel(const float* __restrict__ p, const float* __restrict__ t,
                                                              const float* __restrict__ coef, float* __restrict__ g, long long n) {
    const float c = coef[0];
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) g[i] = c * (p[i] - t[i]);
}

// one wave per body: zero-fill + scatter of the head gradients
__global__ __launch_bounds__(64) STRAPS_NO_PACKED_FP32 void loss_grad_heads_kernel(const float* __restrict__ joints, const float* __restrict__ est, int ld_est,
                                                             const float* __restrict__ prot, const float* __restrict__ tj2d,
                                                             const float* __restrict__ tj3d, const float* __restrict__ tshape,
                                                             const float* __restrict__ trot, const float* __restrict__ coef,
                                                             float* __restrict__ djoints, float* __restrict__ dest,
                                                             float* __restrict__ drot, long long B, float wh) {
    const long long b = blockIdx.x;
    const int lane = threadIdx.x;
    const float* J = joints + b * 270;
    const float* e = est + b * ld_est;
    float* dJ = djoints + b * 270;
    float* dE = dest + b * ld_est;
    const float c2 = coef[1], c3 = coef[2], cs = coef[3], cr = coef[4];
    for (int i = lane; i < 270; i += 64) dJ[i] = 0.f;
    for (int i = lane; i < ld_est; i += 64) dE[i] = (i >= 147 && i < 157) ? cs * (e[i] - tshape[b * 10 + i - 147]) : 0.f;
    for (int i = lane; i < 216; i += 64) drot[b * 216 + i] = cr * (prot[b * 216 + i] - trot[b * 216 + i]);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    const float s = e[0], tx = e[1], ty = e[2];
    float gs = 0.f, gtx = 0.f, gty = 0.f;
    if (lane < 17) {
        const float lx = tj2d[(b * 17 + lane) * 2 + 0], ly = tj2d[(b * 17 + lane) * 2 + 1];
        if (joint_visible(lx, ly, wh)) {
            const float* pj = J + c_coco[lane] * 3;
            const float u = s * (pj[0] + tx), v = s * (pj[1] + ty);
            const float du = c2 * (u - ((2.0f * lx) / wh - 1.0f)), dv = c2 * (v - ((2.0f * ly) / wh - 1.0f));
            dJ[c_coco[lane] * 3 + 0] = du * s;
            dJ[c_coco[lane] * 3 + 1] = dv * s;
            gs = du * (pj[0] + tx) + dv * (pj[1] + ty);
            gtx = du * s;
            gty = dv * s;
        }
    } else if (lane >= 32 && lane < 46) {
        const int k = lane - 32;
        const float* pj = J + c_h36m14[k] * 3;
        const float* tj = tj3d + (b * 14 + k) * 3;
        for (int c = 0; c < 3; ++c) dJ[c_h36m14[k] * 3 + c] = c3 * (pj[c] - tj[c]);
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) {
        gs += __shfl_xor(gs, o, 64);
        gtx += __shfl_xor(gtx, o, 64);
        gty += __shfl_xor(gty, o, 64);
    }
    if (lane == 0) { dE[0] = gs; dE[1] = gtx; dE[2] = gty; }
}

// =====================================================================================================
// Adam
// =====================================================================================================
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, long long n, float b1, float b2, float eps, float step_size,
                                                   float inv_sqrt_bc2, float gscale, float lr, const long long* __restrict__ step_dev) {
    if (step_dev) {   // step counter lives on the device (hipGraph replay): bias corrections computed here, in double
        const double t = (double)step_dev[0];
        step_size = (float)((double)lr / (1.0 - pow((double)b1, t)));
        inv_sqrt_bc2 = (float)(1.0 / sqrt(1.0 - pow((double)b2, t)));
    }
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float gi = g[i] * gscale;
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        p[i] -= step_size * mi / (sqrtf(vi) * inv_sqrt_bc2 + eps);
    }
}

inline unsigned capped_grid(long long n, int cap = 4096) {
    long long g = (n + 255) / 256;
    return (unsigned)(g > cap ? cap : (g < 1 ? 1 : g));
}

}  // namespace

extern "C" int straps_build_proxy_input_std(const float* seg, const float* joints2d, float* out_nchw, int batch, int nj, int wh, int std,
                                            void* stream) {
    STRAPS_REQUIRE(seg && joints2d && out_nchw && batch > 0 && nj > 0 && wh > 16, "straps_build_proxy_input: bad arguments");
    STRAPS_REQUIRE(std >= 1 && 4 * std <= wh, "straps_build_proxy_input: std must be an integer in [1, wh / 4] (got %d)", std);
    STRAPS_REQUIRE((long long)batch * (nj + 1) <= 65535 && wh <= 16384, "straps_build_proxy_input: batch*(nj+1) must be <= 65535 per call");
    const int gx = (wh * wh / 4 + 255) / 256;
    const int size = 2 * std;
    const float step = (float)(2 * size) / (float)(2 * size - 1);          // torch.linspace's fp32 step: (end - start) / (steps - 1)
    const float two_var = (float)(2.0 * std * std);
    hipLaunchKernelGGL(build_proxy_kernel, dim3(gx < 64 ? gx : 64, batch * (nj + 1)), dim3(256), 0, (hipStream_t)stream, seg, joints2d, out_nchw, batch, nj, wh,
                       size, step, two_var);
    STRAPS_CHECK_LAUNCH("build_proxy_kernel");
    return STRAPS_OK;
}

// straps_build_proxy_input_std that also writes the non-zero bit map of its output -- what straps_stem_nzmask(out_nchw, ...) would compute,
// [straps_stem_nzmask_words(batch, nj + 1, wh, wh)] words -- in the same pass.  wh must be a multiple of 8.
extern "C" int straps_build_proxy_input_nz(const float* seg, const float* joints2d, float* out_nchw, uint32_t* nzmask, int batch, int nj, int wh,
                                           int std, void* stream) {
    STRAPS_REQUIRE(seg && joints2d && out_nchw && nzmask && batch > 0 && nj > 0 && wh > 16, "straps_build_proxy_input_nz: bad arguments");
    STRAPS_REQUIRE(std >= 1 && 4 * std <= wh, "straps_build_proxy_input_nz: std must be an integer in [1, wh / 4] (got %d)", std);
    STRAPS_REQUIRE((wh & 7) == 0 && wh <= 16384, "straps_build_proxy_input_nz: wh must be a multiple of 8 (got %d)", wh);
    const long long nwords = (long long)batch * (nj + 1) * ((wh + 3) / 4) * ((wh + 255) / 256);
    STRAPS_REQUIRE((nwords + 3) / 4 < (1LL << 31) && (long long)batch * (nj + 1) * wh * wh < (1LL << 40), "straps_build_proxy_input_nz: input too large for one launch");
    const int size = 2 * std;
    const float step = (float)(2 * size) / (float)(2 * size - 1);          // torch.linspace's fp32 step: (end - start) / (steps - 1)
    const float two_var = (float)(2.0 * std * std);
    hipLaunchKernelGGL(build_proxy_nz_kernel, dim3((unsigned)((nwords + 3) / 4)), dim3(256), 0, (hipStream_t)stream, seg, joints2d, out_nchw, nzmask, nwords, nj,
                       wh, size, step, two_var);
    STRAPS_CHECK_LAUNCH("build_proxy_nz_kernel");
    return STRAPS_OK;
}

extern "C" int straps_build_proxy_input(const float* seg, const float* joints2d, float* out_nchw, int batch, int nj, int wh,
                                        void* stream) {
    return straps_build_proxy_input_std(seg, joints2d, out_nchw, batch, nj, wh, 4, stream);
}

extern "C" size_t straps_loss_workspace_bytes(long long batch) { return (size_t)(NVB + batch * 8 + 16) * sizeof(float); }

// visible target joints of a batch (utils/joints2d_utils.py:23-32 semantics), as a float: what every rank contributes to the job-wide
// count of the global masked mean
__global__ __launch_bounds__(256) void count_visible_kernel(const float* __restrict__ tj2d, float* __restrict__ out, long long n, float wh) {
    __shared__ int red[4];
    int c = 0;
    for (long long i = threadIdx.x; i < n; i += 256) c += joint_visible(tj2d[i * 2], tj2d[i * 2 + 1], wh) ? 1 : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = (float)(red[0] + red[1] + red[2] + red[3]);
}

extern "C" int straps_count_visible(const float* tgt_joints2d, float* out_count, long long batch, int nj, int img_wh, void* stream) {
    STRAPS_REQUIRE(tgt_joints2d && out_count && batch > 0 && nj > 0, "straps_count_visible: bad arguments");
    hipLaunchKernelGGL(count_visible_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, tgt_joints2d, out_count, batch * nj, (float)img_wh);
    STRAPS_CHECK_LAUNCH("count_visible_kernel");
    return STRAPS_OK;
}

extern "C" int straps_loss_fwd_bwd(const float* pred_verts, const float* pred_joints, const float* est, int ld_est, const float* pred_rot,
                                   const float* tgt_verts, const float* tgt_joints2d, const float* tgt_joints3d, const float* tgt_shape,
                                   const float* tgt_rot, const float* log_vars, float* loss_out, float* dverts, float* djoints,
                                   float* dest, float* drot, float* dlogvar, void* workspace, long long batch, int img_wh, void* stream) {
    return straps_loss_fwd_bwd_gm(pred_verts, pred_joints, est, ld_est, pred_rot, tgt_verts, tgt_joints2d, tgt_joints3d, tgt_shape, tgt_rot, log_vars, loss_out,
                                  dverts, djoints, dest, drot, dlogvar, workspace, batch, img_wh, nullptr, 1.0f, stream);
}

extern "C" int straps_loss_fwd_bwd_gm(const float* pred_verts, const float* pred_joints, const float* est, int ld_est, const float* pred_rot,
                                      const float* tgt_verts, const float* tgt_joints2d, const float* tgt_joints3d, const float* tgt_shape,
                                      const float* tgt_rot, const float* log_vars, float* loss_out, float* dverts, float* djoints,
                                      float* dest, float* drot, float* dlogvar, void* workspace, long long batch, int img_wh,
                                      const float* j2d_count_global, float count_scale, void* stream) {
    STRAPS_REQUIRE(pred_verts && pred_joints && est && pred_rot && tgt_verts && tgt_joints2d && tgt_joints3d && tgt_shape && tgt_rot &&
                       log_vars && loss_out && workspace,
                   "straps_loss_fwd_bwd: null pointer");
    STRAPS_REQUIRE(batch > 0 && ld_est >= 157, "straps_loss_fwd_bwd: bad shape batch=%lld ld_est=%d", batch, ld_est);
    const bool want_grad = dverts || djoints || dest || drot;
    STRAPS_REQUIRE(!want_grad || (dverts && djoints && dest && drot && dlogvar), "straps_loss_fwd_bwd: give all gradient outputs or none");
    hipStream_t st = (hipStream_t)stream;
    float* vpart = (float*)workspace;
    float* body = vpart + NVB;
    float* coef = body + batch * 8;
    float* dlv = dlogvar ? dlogvar : coef + 8;
    const long long nv = batch * 20670LL;
    hipLaunchKernelGGL(loss_verts_partial_kernel, dim3(NVB), dim3(256), 0, st, pred_verts, tgt_verts, nv, vpart);
    STRAPS_CHECK_LAUNCH("loss_verts_partial_kernel");
    hipLaunchKernelGGL(loss_heads_kernel, dim3((unsigned)((batch + 255) / 256)), dim3(256), 0, st, pred_joints, est, ld_est, pred_rot,
                       tgt_joints2d, tgt_joints3d, tgt_shape, tgt_rot, body, batch, (float)img_wh);
    STRAPS_CHECK_LAUNCH("loss_heads_kernel");
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(256), 0, st, vpart, body, log_vars, loss_out, dlv, coef, batch, j2d_count_global, count_scale);
    STRAPS_CHECK_LAUNCH("loss_finalize_kernel");
    if (want_grad) {
        hipLaunchKernelGGL(loss_grad_verts_kernel, dim3(capped_grid(nv)), dim3(256), 0, st, pred_verts, tgt_verts, coef, dverts, nv);
        STRAPS_CHECK_LAUNCH("loss_grad_verts_kernel");
        hipLaunchKernelGGL(loss_grad_heads_kernel, dim3((unsigned)batch), dim3(64), 0, st, pred_joints, est, ld_est, pred_rot, tgt_joints2d,
                           tgt_joints3d, tgt_shape, tgt_rot, coef, djoints, dest, drot, batch, (float)img_wh);
        STRAPS_CHECK_LAUNCH("loss_grad_heads_kernel");
    }
    return STRAPS_OK;
}

extern "C" int straps_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long long n, int step, float lr,
                                float beta1, float beta2, float eps, float grad_scale, const long long* step_dev, void* stream) {
    STRAPS_REQUIRE(params && grads && exp_avg && exp_avg_sq && n > 0 && (step >= 1 || step_dev), "straps_adam_step: bad arguments");
    if (step < 1) step = 1;
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    hipLaunchKernelGGL(adam_kernel, dim3(capped_grid(n)), dim3(256), 0, (hipStream_t)stream, params, grads, exp_avg, exp_avg_sq, n, beta1, beta2,
                       eps, (float)((double)lr / bc1), (float)(1.0 / sqrt(bc2)), grad_scale, lr, step_dev);
    STRAPS_CHECK_LAUNCH("adam_kernel");
    return STRAPS_OK;
}

// =====================================================================================================
// generic (row-masked) MSE used by the drop-in criterion module
// =====================================================================================================
namespace {

// out[0] = sum over kept rows of (p - (t*ts + tb))^2, out[1] = number of kept elements
__global__ __launch_bounds__(256) void mse_sum_kernel(const float* __restrict__ p, const float* __restrict__ t, const uint8_t* __restrict__ mask,
                                                      long long rows, int cols, float ts, float tb, float* __restrict__ part) {
    __shared__ float red[4][2];
    float s = 0.f, c = 0.f;
    const long long n = rows * cols;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        if (mask && !mask[i / cols]) continue;
        const float d = p[i] - (t[i] * ts + tb);
        s = fmaf(d, d, s);
        c += 1.f;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); c += __shfl_xor(c, o, 64); }
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = s; red[threadIdx.x >> 6][1] = c; }
    __syncthreads();
    if (threadIdx.x == 0) {
        part[blockIdx.x * 2 + 0] = (red[0][0] + red[1][0]) + (red[2][0] + red[3][0]);
        part[blockIdx.x * 2 + 1] = (red[0][1] + red[1][1]) + (red[2][1] + red[3][1]);
    }
}

__global__ __launch_bounds__(64) void mse_finalize_kernel(const float* __restrict__ part, int nblk, float* __restrict__ out) {
    double s = 0.0, c = 0.0;
    for (int i = threadIdx.x; i < nblk; i += 64) { s += part[i * 2]; c += part[i * 2 + 1]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); c += __shfl_xor(c, o, 64); }
    if (threadIdx.x == 0) { out[0] = (float)s; out[1] = (float)c; out[2] = (float)(s / c); }
}

// grad[i] = coef[0] * (p - (t*ts + tb)) on kept rows, 0 elsewhere
__global__ __launch_bounds__(256) void mse_grad_kernel(const float* __restrict__ p, const float* __restrict__ t, const uint8_t* __restrict__ mask,
                                                       long long rows, int cols, float ts, float tb, const float* __restrict__ coef,
                                                       float* __restrict__ g) {
    const long long n = rows * cols;
    const float c = coef[0];
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        g[i] = (mask && !mask[i / cols]) ? 0.f : c * (p[i] - (t[i] * ts + tb));
}

}  // namespace

extern "C" int straps_mse_fwd(const float* pred, const float* tgt, const uint8_t* row_mask, long long rows, int cols, float tgt_scale,
                              float tgt_shift, float* out3, void* workspace, void* stream) {
    STRAPS_REQUIRE(pred && tgt && out3 && workspace && rows > 0 && cols > 0, "straps_mse_fwd: bad arguments");
    float* part = (float*)workspace;     // 512 floats
    hipLaunchKernelGGL(mse_sum_kernel, dim3(256), dim3(256), 0, (hipStream_t)stream, pred, tgt, row_mask, rows, cols, tgt_scale, tgt_shift, part);
    STRAPS_CHECK_LAUNCH("mse_sum_kernel");
    hipLaunchKernelGGL(mse_finalize_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, part, 256, out3);
    STRAPS_CHECK_LAUNCH("mse_finalize_kernel");
    return STRAPS_OK;
}

extern "C" int straps_mse_bwd(const float* pred, const float* tgt, const uint8_t* row_mask, long long rows, int cols, float tgt_scale,
                              float tgt_shift, const float* coef, float* grad, void* stream) {
    STRAPS_REQUIRE(pred && tgt && coef && grad && rows > 0 && cols > 0, "straps_mse_bwd: bad arguments");
    hipLaunchKernelGGL(mse_grad_kernel, dim3(capped_grid(rows * cols)), dim3(256), 0, (hipStream_t)stream, pred, tgt, row_mask, rows, cols,
                       tgt_scale, tgt_shift, coef, grad);
    STRAPS_CHECK_LAUNCH("mse_grad_kernel");
    return STRAPS_OK;
}

namespace {
// target-side heads (train loop :138-143): H36M-LSP 3D joints and the perspective projection of the COCO joints
__global__ __launch_bounds__(256) void project_targets_kernel(const float* __restrict__ joints, const float* __restrict__ cam_t, float fx,
                                                              float fy, float cx, float cy, float* __restrict__ j2d,
                                                              float* __restrict__ j3d, long long B) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= B * 31) return;
    const long long b = i / 31;
    const int k = (int)(i - b * 31);
    const float* J = joints + b * 270;
    if (k < 17) {
        const float* p = J + c_coco[k] * 3;
        const float x = p[0] + cam_t[b * 3 + 0], y = p[1] + cam_t[b * 3 + 1], z = p[2] + cam_t[b * 3 + 2];
        j2d[(b * 17 + k) * 2 + 0] = fx * (x / z) + cx;      // K [p/p_z] with K = [[fx,0,cx],[0,fy,cy],[0,0,1]]
        j2d[(b * 17 + k) * 2 + 1] = fy * (y / z) + cy;
    } else {
        const float* p = J + c_h36m14[k - 17] * 3;
        float* o = j3d + (b * 14 + (k - 17)) * 3;
        o[0] = p[0]; o[1] = p[1]; o[2] = p[2];
    }
}
}  // namespace

extern "C" int straps_project_targets(const float* joints, const float* cam_t, float fx, float fy, float cx, float cy, float* joints2d,
                                      float* joints3d, long long batch, void* stream) {
    STRAPS_REQUIRE(joints && cam_t && joints2d && joints3d && batch > 0, "straps_project_targets: bad arguments");
    hipLaunchKernelGGL(project_targets_kernel, dim3((unsigned)((batch * 31 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, joints, cam_t, fx,
                       fy, cx, cy, joints2d, joints3d, batch);
    STRAPS_CHECK_LAUNCH("project_targets_kernel");
    return STRAPS_OK;
}
