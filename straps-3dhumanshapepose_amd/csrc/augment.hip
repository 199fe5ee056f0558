// augment.hip -- the random half of the synthetic training step, on the device:
//   * a counter-based generator (Philox4x32-10) that fills the step's uniform / normal draw buffers -- keyed by
//     (seed, sub-stream) and counted by a DEVICE-resident step counter, so a replayed hipGraph draws fresh numbers;
//   * G1 augment_smpl  (augmentation/smpl_augmentation.py:6-61): shape resampling + axis-angle -> rotation matrices;
//   * G2 augment_cam_t (augmentation/cam_augmentation.py:4-14);
//   * the joint jitter of G3 (augmentation/proxy_rep_augmentation.py:25-49).
// (The vertex noise of :5-22 is applied inside the rasteriser's projection kernel, raster.hip; body-part removal and box
//  occlusion live in train.hip::augment_seg_kernel.)
// The arithmetic on the draws is written unfused (fp contract off) in the order the reference's torch expressions
// evaluate, so that with the draws supplied the results equal oracle/straps_oracle.py bit for bit (sin/cos aside).
#include <math.h>

#include "common.h"

#pragma clang fp contract(off)

namespace {

struct u32x4 { uint32_t x, y, z, w; };

__device__ __forceinline__ u32x4 philox4x32_10(u32x4 c, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c.x), lo0 = 0xD2511F53u * c.x;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c.z), lo1 = 0xCD9E8D57u * c.z;
        c = u32x4{hi1 ^ c.y ^ k0, lo1, hi0 ^ c.w ^ k1, lo0};
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return c;
}

// one thread per group of 4 outputs.  kind 0: uniform [0,1) with 24 random bits ((x >> 8) * 2^-24);
// kind 1: standard normal, Box-Muller on the pairs (x,y) and (z,w): r = sqrt(-2 ln u1), u1 = ((x>>8)+1) 2^-24 in (0,1],
// theta = 2 pi (y>>8) 2^-24 -> (r cos theta, r sin theta).
__global__ __launch_bounds__(256) void philox_fill_kernel(unsigned long long seed, const long long* __restrict__ step_dev, long long step_host,
                                                          uint32_t substream, float* __restrict__ out, long long n, int kind) {
    const long long q = (long long)blockIdx.x * 256 + threadIdx.x;
    if (q * 4 >= n) return;
    const unsigned long long step = (unsigned long long)(step_dev ? step_dev[0] : step_host);
    const u32x4 ctr = {(uint32_t)q, (uint32_t)((unsigned long long)q >> 32), (uint32_t)step, substream + (uint32_t)(step >> 32) * 0x10000u};
    const u32x4 r = philox4x32_10(ctr, (uint32_t)seed, (uint32_t)(seed >> 32));
    float v[4];
    const float s24 = 5.9604644775390625e-08f;      // 2^-24
    if (kind == 0) {
        v[0] = (float)(r.x >> 8) * s24; v[1] = (float)(r.y >> 8) * s24; v[2] = (float)(r.z >> 8) * s24; v[3] = (float)(r.w >> 8) * s24;
    } else {
        const float u1a = (float)((r.x >> 8) + 1u) * s24, u1b = (float)((r.z >> 8) + 1u) * s24;
        const float tha = 6.283185307179586f * ((float)(r.y >> 8) * s24), thb = 6.283185307179586f * ((float)(r.w >> 8) * s24);
        const float ra = sqrtf(-2.0f * logf(u1a)), rb = sqrtf(-2.0f * logf(u1b));
        v[0] = ra * cosf(tha); v[1] = ra * sinf(tha); v[2] = rb * cosf(thb); v[3] = rb * sinf(thb);
    }
    if (q * 4 + 3 < n && (((uintptr_t)out & 15) == 0)) *reinterpret_cast<f32x4*>(out + q * 4) = f32x4{v[0], v[1], v[2], v[3]};
    else
        for (int e = 0; e < 4; ++e)
            if (q * 4 + e < n) out[q * 4 + e] = v[e];
}

__global__ void counter_add_kernel(long long* c, int n, long long delta) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i < n) c[i] += delta;
}

__global__ void gather_kernel(const float* __restrict__ src, const int* __restrict__ index, float* __restrict__ dst, int n) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i < n) dst[i] = src[index[i]];
}

// smplx batch_rodrigues (the formula of pose.hip::rodrigues_kernel)
__device__ __forceinline__ void rodrigues(float rx, float ry, float rz, float* o) {
    const float ex = rx + 1e-8f, ey = ry + 1e-8f, ez = rz + 1e-8f;
    const float angle = sqrtf(ex * ex + ey * ey + ez * ez);
    const float dx = rx / angle, dy = ry / angle, dz = rz / angle;
    const float s = sinf(angle), c1 = 1.0f - cosf(angle);
    const float k01 = -dz, k02 = dy, k10 = dz, k12 = -dx, k20 = -dy, k21 = dx;
    const float q00 = k01 * k10 + k02 * k20, q01 = k02 * k21, q02 = k01 * k12;
    const float q10 = k12 * k20, q11 = k10 * k01 + k12 * k21, q12 = k10 * k02;
    const float q20 = k21 * k10, q21 = k20 * k01, q22 = k20 * k02 + k21 * k12;
    o[0] = 1.0f + c1 * q00;           o[1] = s * k01 + c1 * q01;        o[2] = s * k02 + c1 * q02;
    o[3] = s * k10 + c1 * q10;        o[4] = 1.0f + c1 * q11;           o[5] = s * k12 + c1 * q12;
    o[6] = s * k20 + c1 * q20;        o[7] = s * k21 + c1 * q21;        o[8] = 1.0f + c1 * q22;
}

// one thread per (body, slot): slots 0..23 = joints (rotation matrices), 24..33 = the ten shape coefficients
__global__ __launch_bounds__(256) STRAPS_NO_PACKED_FP32 void augment_smpl_kernel(const float* __restrict__ pose_rows, long long n_rows, const float* __restrict__ u_index,
                                                           const float* __restrict__ orig_shape, const float* __restrict__ mean_shape,
                                                           const float* __restrict__ draws, int mode, const float* __restrict__ std_vector,
                                                           float range_lo, float range_scale, float* __restrict__ out_shape,
                                                           float* __restrict__ out_rot, float* __restrict__ out_pose, long long B) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= B * 34) return;
    const long long b = i / 34;
    const int s = (int)(i - b * 34);
    if (s < 24) {
        long long row = b;
        if (u_index) {                      // dataset stand-in: a uniformly drawn row of the resident pose pool
            row = (long long)(u_index[b] * (float)n_rows);
            row = row < 0 ? 0 : (row >= n_rows ? n_rows - 1 : row);
        }
        const float* p = pose_rows + row * 72 + s * 3;
        rodrigues(p[0], p[1], p[2], out_rot + (b * 24 + s) * 9);
        if (out_pose) { float* o = out_pose + b * 72 + s * 3; o[0] = p[0]; o[1] = p[1]; o[2] = p[2]; }
    } else {
        const int l = s - 24;
        float v;
        if (mode == 1) v = draws[b * 10 + l] * std_vector[l] + mean_shape[l];                    // randn * std + mean (:18-25)
        else if (mode == 2) v = (range_scale * draws[b * 10 + l] + range_lo) + mean_shape[l];    // (h-l) * rand + l + mean (:6-15)
        else v = orig_shape[b * 10 + l];
        out_shape[b * 10 + l] = v;
    }
}

__global__ __launch_bounds__(256) void augment_cam_kernel(const float* __restrict__ mean_cam_t, const float* __restrict__ normals_xy,
                                                          const float* __restrict__ uniform_z, float xy_std, float z_lo, float z_scale,
                                                          float* __restrict__ out, long long B) {
    const long long b = (long long)blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    out[b * 3 + 0] = mean_cam_t[b * 3 + 0] + normals_xy[b * 2 + 0] * xy_std;
    out[b * 3 + 1] = mean_cam_t[b * 3 + 1] + normals_xy[b * 2 + 1] * xy_std;
    out[b * 3 + 2] = mean_cam_t[b * 3 + 2] + (z_scale * uniform_z[b] + z_lo);
}

// 17 COCO joints: hips (11, 12) take their own range
__global__ __launch_bounds__(256) void deviate_joints_kernel(const float* __restrict__ j2d, const float* __restrict__ u, float lo, float scale,
                                                             float hip_lo, float hip_scale, float* __restrict__ out, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int j = (int)((i >> 1) % 17);
    const bool hip = j == 11 || j == 12;
    out[i] = j2d[i] + ((hip ? hip_scale : scale) * u[i] + (hip ? hip_lo : lo));
}

// random_remove_bodyparts + random_occlude (augmentation/proxy_rep_augmentation.py:52-101) in one pass: u[b][0..5] < prob[c]
// removes part class c+1 (fp32 comparison of the supplied draw); u[b][6] < occlude_prob zeroes a box whose centre is drawn
// from u[b][7], u[b][8] with the reference's own double arithmetic: x = (x_h - x_l) * rand + x_l, x_h = c - 0.3*wh/2,
// x_l = c + 0.3*wh/2, corners (x -/+ box/2).astype(int16); the first image axis is rows (seg[i, x1:x2, y1:y2] = 0).
__global__ __launch_bounds__(256) void augment_seg_kernel(const float* __restrict__ seg, const float* __restrict__ u,
                                                          const float* __restrict__ prob, float occl_prob, int box, float* __restrict__ out,
                                                          int B, int WH) {
    const long long n = (long long)B * WH * WH;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int x = (int)(i % WH);
        const int y = (int)((i / WH) % WH);
        const int b = (int)(i / ((long long)WH * WH));
        const float* ub = u + b * 9;
        float v = seg[i];
        const int cls = (int)v;
        if (cls >= 1 && cls <= 6 && ub[cls - 1] < prob[cls - 1]) v = 0.f;
        if (ub[6] < occl_prob) {
            const double c = (double)WH / 2.0, d = 0.3 * (double)WH / 2.0;
            const double hi = c - d, lo = c + d;
            const double cx = (hi - lo) * (double)ub[7] + lo, cy = (hi - lo) * (double)ub[8] + lo;
            const double hb = (double)box / 2.0;
            const int r1 = (int)(short)(cx - hb), r2 = (int)(short)(cx + hb), c1 = (int)(short)(cy - hb), c2 = (int)(short)(cy + hb);
            if (y >= r1 && y < r2 && x >= c1 && x < c2) v = 0.f;
        }
        out[i] = v;
    }
}

// vertices [n][3]: x,y += scale * u + lo (random_verts2D_deviation); z copied
__global__ __launch_bounds__(256) void deviate_verts_kernel(const float* __restrict__ v, const float* __restrict__ u, float lo, float scale,
                                                            float* __restrict__ out, long long n3) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n3) return;
    const long long vert = i / 3;
    const int c = (int)(i - vert * 3);
    out[i] = c < 2 ? v[i] + (scale * u[vert * 2 + c] + lo) : v[i];
}

}  // namespace

extern "C" int straps_philox_fill(unsigned long long seed, const long long* step_dev, long long step_host, unsigned substream, float* out,
                                  long long n, int kind, void* stream) {
    STRAPS_REQUIRE(out && n > 0 && (kind == 0 || kind == 1), "straps_philox_fill: bad arguments (n=%lld kind=%d)", n, kind);
    const long long quads = (n + 3) / 4;
    STRAPS_REQUIRE((quads + 255) / 256 < (1LL << 31), "straps_philox_fill: too many draws for one launch");
    hipLaunchKernelGGL(philox_fill_kernel, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, seed, step_dev, step_host,
                       (uint32_t)substream, out, n, kind);
    STRAPS_CHECK_LAUNCH("philox_fill_kernel");
    return STRAPS_OK;
}

extern "C" int straps_counter_add(long long* counters, int n, long long delta, void* stream) {
    STRAPS_REQUIRE(counters && n > 0, "straps_counter_add: bad arguments");
    hipLaunchKernelGGL(counter_add_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, (hipStream_t)stream, counters, n, delta);
    STRAPS_CHECK_LAUNCH("counter_add_kernel");
    return STRAPS_OK;
}

extern "C" int straps_gather_f32(const float* src, const int* index, float* dst, int n, void* stream) {
    STRAPS_REQUIRE(src && index && dst && n > 0, "straps_gather_f32: bad arguments");
    hipLaunchKernelGGL(gather_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, (hipStream_t)stream, src, index, dst, n);
    STRAPS_CHECK_LAUNCH("gather_kernel");
    return STRAPS_OK;
}

extern "C" int straps_augment_smpl(const float* pose_rows, long long n_rows, const float* u_index, const float* orig_shape,
                                   const float* mean_shape, const float* shape_draws, int shape_mode, const float* std_vector, double range_lo,
                                   double range_hi, float* out_shape, float* out_rotmats, float* out_pose, long long batch, void* stream) {
    STRAPS_REQUIRE(pose_rows && out_shape && out_rotmats && batch > 0 && n_rows > 0, "straps_augment_smpl: bad arguments");
    STRAPS_REQUIRE(u_index || n_rows >= batch, "straps_augment_smpl: without an index draw pose_rows must hold one row per body");
    STRAPS_REQUIRE(shape_mode >= 0 && shape_mode <= 2, "straps_augment_smpl: shape_mode must be 0 (keep), 1 (normal) or 2 (uniform)");
    if (shape_mode == 0) STRAPS_REQUIRE(orig_shape, "straps_augment_smpl: shape_mode 0 needs orig_shape");
    else STRAPS_REQUIRE(mean_shape && shape_draws && (shape_mode == 2 || std_vector), "straps_augment_smpl: shape resampling needs mean_shape, draws (and std_vector)");
    // ranges are doubles: (h - l) is formed in double like the reference's Python scalars, then rounded once to the fp32
    // scalar the tensor op uses (same in every entry point of this file)
    const float scale = (float)(range_hi - range_lo);
    hipLaunchKernelGGL(augment_smpl_kernel, dim3((unsigned)((batch * 34 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, pose_rows, n_rows,
                       u_index, orig_shape, mean_shape, shape_draws, shape_mode, std_vector, (float)range_lo, scale, out_shape, out_rotmats, out_pose,
                       batch);
    STRAPS_CHECK_LAUNCH("augment_smpl_kernel");
    return STRAPS_OK;
}

extern "C" int straps_augment_cam_t(const float* mean_cam_t, const float* normals_xy, const float* uniform_z, double xy_std, double z_lo,
                                    double z_hi, float* out_cam_t, long long batch, void* stream) {
    STRAPS_REQUIRE(mean_cam_t && normals_xy && uniform_z && out_cam_t && batch > 0, "straps_augment_cam_t: bad arguments");
    hipLaunchKernelGGL(augment_cam_kernel, dim3((unsigned)((batch + 255) / 256)), dim3(256), 0, (hipStream_t)stream, mean_cam_t, normals_xy,
                       uniform_z, (float)xy_std, (float)z_lo, (float)(z_hi - z_lo), out_cam_t, batch);
    STRAPS_CHECK_LAUNCH("augment_cam_kernel");
    return STRAPS_OK;
}

extern "C" int straps_deviate_joints2d(const float* joints2d, const float* uniforms, double lo, double hi, double hip_lo, double hip_hi,
                                       float* out, long long batch, void* stream) {
    STRAPS_REQUIRE(joints2d && uniforms && out && batch > 0, "straps_deviate_joints2d: bad arguments");
    const long long n = batch * 34;
    hipLaunchKernelGGL(deviate_joints_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, joints2d, uniforms, (float)lo,
                       (float)(hi - lo), (float)hip_lo, (float)(hip_hi - hip_lo), out, n);
    STRAPS_CHECK_LAUNCH("deviate_joints_kernel");
    return STRAPS_OK;
}

extern "C" int straps_augment_seg(const float* seg, const float* uniforms, const float* remove_prob, float occlude_prob, int box_dim,
                                  float* out, int batch, int wh, void* stream) {
    STRAPS_REQUIRE(seg && uniforms && remove_prob && out && batch > 0 && wh > 0, "straps_augment_seg: bad arguments");
    const long long n = (long long)batch * wh * wh;
    long long g = (n + 255) / 256;
    if (g > 8192) g = 8192;
    hipLaunchKernelGGL(augment_seg_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, seg, uniforms, remove_prob, occlude_prob,
                       box_dim, out, batch, wh);
    STRAPS_CHECK_LAUNCH("augment_seg_kernel");
    return STRAPS_OK;
}

extern "C" int straps_deviate_verts2d(const float* verts, const float* uniforms, double lo, double hi, float* out, long long nverts_total,
                                      void* stream) {
    STRAPS_REQUIRE(verts && uniforms && out && nverts_total > 0, "straps_deviate_verts2d: bad arguments");
    const long long n3 = nverts_total * 3;
    STRAPS_REQUIRE((n3 + 255) / 256 < (1LL << 31), "straps_deviate_verts2d: too many vertices for one launch");
    hipLaunchKernelGGL(deviate_verts_kernel, dim3((unsigned)((n3 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, verts, uniforms, (float)lo,
                       (float)(hi - lo), out, n3);
    STRAPS_CHECK_LAUNCH("deviate_verts_kernel");
    return STRAPS_OK;
}

// a fill KERNEL, not hipMemsetAsync: inside a captured hipGraph a memset node is not an ordinary member of the stream's kernel chain
// (round 3: with the batch generation and the step captured on ONE stream and sharing a memory pool, the runtime's memset nodes clobbered
// neighbouring pool memory -- tools/graph_bisect.py; as kernels the same graphs replay correctly)
__global__ __launch_bounds__(256) void fill_bytes_kernel(unsigned char* __restrict__ p, unsigned long long bytes, unsigned int word) {
    const unsigned long long head = (16 - (reinterpret_cast<unsigned long long>(p) & 15)) & 15;      // bytes in front of the first 16-byte boundary
    const unsigned long long h = head < bytes ? head : bytes;
    const unsigned long long n16 = (bytes - h) >> 4;
    uint4* q = reinterpret_cast<uint4*>(p + h);
    const uint4 v = make_uint4(word, word, word, word);
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (unsigned long long)gridDim.x * 256) q[i] = v;
    if (blockIdx.x == 0) {
        const unsigned char b = (unsigned char)(word & 0xff);
        for (unsigned long long i = threadIdx.x; i < h; i += 256) p[i] = b;
        const unsigned long long tail0 = h + (n16 << 4);
        for (unsigned long long i = tail0 + threadIdx.x; i < bytes; i += 256) p[i] = b;
    }
}

int straps_fill_bytes(void* ptr, size_t bytes, unsigned char value, hipStream_t st) {
    if (bytes == 0) return STRAPS_OK;
    const unsigned int word = 0x01010101u * value;
    unsigned long long blocks = ((bytes >> 4) + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(fill_bytes_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (unsigned char*)ptr, (unsigned long long)bytes, word);
    STRAPS_CHECK_LAUNCH("fill_bytes_kernel");
    return STRAPS_OK;
}

extern "C" int straps_memset_zero(void* ptr, size_t bytes, void* stream) {
    STRAPS_REQUIRE(ptr || bytes == 0, "straps_memset_zero: null pointer");
    return straps_fill_bytes(ptr, bytes, 0, (hipStream_t)stream);
}
