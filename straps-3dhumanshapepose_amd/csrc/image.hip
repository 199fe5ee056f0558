// image.hip -- on-device bounding-box crop + nearest-neighbour resize of the part segmentation and its 2-D joints
// (SURVEY 8f row f2): replaces the device->host hop, the per-sample numpy crop loop and the per-sample cv2.resize of
// train/train_synthetic_otf_rendering.py:161-170 (utils/image_utils.py:44-105).
//
//   crop_bbox_kernel   : one workgroup per sample: min/max row/col of the non-zero pixels, then exactly the box
//                        arithmetic of batch_crop_seg_to_bounding_box (:56-78): centre/height/width, scale and centre
//                        jitter from the supplied uniforms, int16 truncation of the corners, clamp of negatives to 0,
//                        numpy slice clamping at the far edge.  Emits box[b] = {r0, c0, r1, c1} (crop = seg[r0:r1, c0:c1]).
//   crop_resize_kernel : out[y][x] = crop[sy][sx], sx = min(floor(x * ifx), cw - 1), ifx = 1 / ((double)out / cw)  (cv2.INTER_NEAREST
//                        as OpenCV's resizeNN computes it) and joints' = (joints - [c0_orig, r0_orig]) * [out/cw, out/ch].
#include "common.h"

// the box arithmetic is the reference's double arithmetic, unfused, so that the int16 truncations agree with numpy's
#pragma clang fp contract(off)

namespace {

__global__ __launch_bounds__(1024) void crop_bbox_kernel(const float* __restrict__ seg, const float* __restrict__ u, double orig_scale,
                                                        double ds_lo, double ds_hi, double dc_lo, double dc_hi, int use_jitter,
                                                        int* __restrict__ box, int wh) {
    __shared__ int red[16][4];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* s = seg + (long long)b * wh * wh;
    int rmin = 1 << 30, rmax = -1, cmin = 1 << 30, cmax = -1;
    if ((wh & 3) == 0) {                                  // 16 waves x float4: the 256 KB image is one short sweep
        for (int i = tid * 4; i < wh * wh; i += 4096) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(s + i);
            const int r = i / wh, c = i - r * wh;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (v[e] != 0.f) { rmin = min(rmin, r); rmax = max(rmax, r); cmin = min(cmin, c + e); cmax = max(cmax, c + e); }
        }
    } else {
        for (int i = tid; i < wh * wh; i += 1024) {
            if (s[i] != 0.f) {
                const int r = i / wh, c = i - r * wh;
                rmin = min(rmin, r); rmax = max(rmax, r); cmin = min(cmin, c); cmax = max(cmax, c);
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        rmin = min(rmin, __shfl_xor(rmin, o, 64)); cmin = min(cmin, __shfl_xor(cmin, o, 64));
        rmax = max(rmax, __shfl_xor(rmax, o, 64)); cmax = max(cmax, __shfl_xor(cmax, o, 64));
    }
    if (lane == 0) { red[wave][0] = rmin; red[wave][1] = rmax; red[wave][2] = cmin; red[wave][3] = cmax; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 16; ++w) {
            rmin = min(rmin, red[w][0]); rmax = max(rmax, red[w][1]); cmin = min(cmin, red[w][2]); cmax = max(cmax, red[w][3]);
        }
        int* o = box + b * 6;
        if (rmax < 0) {   // empty silhouette: keep the whole frame (the reference would raise on np.amin of an empty array)
            o[0] = 0; o[1] = 0; o[2] = wh; o[3] = wh; o[4] = 0; o[5] = 0;
            return;
        }
        // utils/image_utils.py:22-41,56-78 in double, like numpy
        double cr = (rmin + rmax) / 2.0, cc = (cmin + cmax) / 2.0;
        const double height = rmax - rmin, width = cmax - cmin;
        double scale = orig_scale;
        if (use_jitter) {
            scale += (ds_hi - ds_lo) * (double)u[b * 3 + 0] + ds_lo;
            cr += (dc_hi - dc_lo) * (double)u[b * 3 + 1] + dc_lo;
            cc += (dc_hi - dc_lo) * (double)u[b * 3 + 2] + dc_lo;
        }
        const double side = (height > width ? height : width) * scale;
        int r0 = (int)(short)(cr - side / 2.0), c0 = (int)(short)(cc - side / 2.0);      // .astype(np.int16): truncation
        int r1 = (int)(short)(cr + side / 2.0), c1 = (int)(short)(cc + side / 2.0);
        if (r0 < 0) r0 = 0;
        if (c0 < 0) c0 = 0;
        if (r1 < 0) r1 = 0;
        if (c1 < 0) c1 = 0;
        o[4] = r0; o[5] = c0;                       // the joints are shifted by the (clamped) top-left corner (:73)
        o[0] = r0; o[1] = c0;
        o[2] = r1 > wh ? wh : r1;                   // numpy slicing clamps the far edge
        o[3] = c1 > wh ? wh : c1;
    }
}

__global__ __launch_bounds__(256) void crop_resize_kernel(const float* __restrict__ seg, const float* __restrict__ joints,
                                                          const int* __restrict__ box, float* __restrict__ out,
                                                          float* __restrict__ jout, int B, int wh, int owh, int nj) {
    const long long n = (long long)B * owh * owh;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int x = (int)(i % owh);
        const int y = (int)((i / owh) % owh);
        const int b = (int)(i / ((long long)owh * owh));
        const int* bx = box + b * 6;
        const int ch = bx[2] - bx[0], cw = bx[3] - bx[1];
        float v = 0.f;
        if (ch > 0 && cw > 0) {
            // OpenCV resizeNN: ifx = 1 / inv_scale_x with inv_scale_x = (double)dst / src; sx = min(cvFloor(x * ifx), src - 1)
            const double ify = 1.0 / ((double)owh / (double)ch), ifx = 1.0 / ((double)owh / (double)cw);
            int sy = (int)floor((double)y * ify), sx = (int)floor((double)x * ifx);
            sy = sy < ch - 1 ? sy : ch - 1;
            sx = sx < cw - 1 ? sx : cw - 1;
            v = seg[((long long)b * wh + bx[0] + sy) * wh + bx[1] + sx];
        }
        out[i] = v;
    }
    const long long nt = (long long)B * nj;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nt; i += (long long)gridDim.x * 256) {
        const int b = (int)(i / nj);
        const int* bx = box + b * 6;
        const int ch = bx[2] - bx[0], cw = bx[3] - bx[1];
        const double jx = (double)joints[i * 2 + 0] - bx[5], jy = (double)joints[i * 2 + 1] - bx[4];
        jout[i * 2 + 0] = (float)(jx * ((double)owh / (double)(cw > 0 ? cw : 1)));
        jout[i * 2 + 1] = (float)(jy * ((double)owh / (double)(ch > 0 ? ch : 1)));
    }
}

}  // namespace

extern "C" int straps_crop_resize(const float* seg, const float* joints2d, const float* uniforms, double orig_scale_factor,
                                  double delta_scale_lo, double delta_scale_hi, double delta_centre_lo, double delta_centre_hi,
                                  float* out_seg, float* out_joints2d, int* boxes, int batch, int wh, int out_wh, int nj, void* stream) {
    STRAPS_REQUIRE(seg && joints2d && out_seg && out_joints2d && boxes && batch > 0 && wh > 0 && out_wh > 0 && nj > 0,
                   "straps_crop_resize: bad arguments");
    STRAPS_REQUIRE(wh < 32768, "straps_crop_resize: image side must fit int16 like the reference's box arithmetic");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(crop_bbox_kernel, dim3(batch), dim3(1024), 0, st, seg, uniforms, orig_scale_factor, delta_scale_lo, delta_scale_hi,
                       delta_centre_lo, delta_centre_hi, uniforms != nullptr, boxes, wh);
    STRAPS_CHECK_LAUNCH("crop_bbox_kernel");
    long long g = ((long long)batch * out_wh * out_wh + 255) / 256;
    if (g > 8192) g = 8192;
    hipLaunchKernelGGL(crop_resize_kernel, dim3((unsigned)g), dim3(256), 0, st, seg, joints2d, boxes, out_seg, out_joints2d, batch, wh, out_wh, nj);
    STRAPS_CHECK_LAUNCH("crop_resize_kernel");
    return STRAPS_OK;
}
