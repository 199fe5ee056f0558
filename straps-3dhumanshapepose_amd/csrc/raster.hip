// raster.hip -- body-part segmentation rasteriser: replaces NMRRenderer.forward with
// rend_parts_seg=True (renderers/nmr_renderer.py:84-100), i.e. the third-party `neural_renderer` z-buffer pass
// + the get_parts look-up, for the on-the-fly training loop (train loop :155).
//
// HBM-bound integer/byte work, no MFMA.  Three launches per call:
//   1. project kernel  : one thread per (body, vertex): camera transform + pin-hole projection to NDC, exactly
//                        neural_renderer's `projection` camera mode.  The training loop's vertex noise
//                        (random_verts2D_deviation, train loop :146-151) is added here from a buffer of uniforms, so the
//                        noisy copy of the mesh is never materialised.
//   2. face kernel     : 16 lanes per (body, face) share the pixel-centre samples inside the face's bounding box (SMPL
//                        faces cover a few pixels, so that is 1-2 trips; a stretched face no longer serialises a wave),
//                        two-sided inside test, perspective-correct depth, 64-bit atomicMin of (depth bits << 32 | face id)
//                        into the per-body z-buffer.  min over (depth, id) pairs is order-independent -> deterministic,
//                        ties go to the lower face id like the sequential reference loop.
//   3. resolve kernel  : z-buffer -> part id per pixel through the per-face part table (the reference decodes a
//                        rendered texture through cube_parts; with one part per face that is this table), rows flipped
//                        like the renderer's final image flip.
// Every arithmetic step is written unfused (fp contract off) in the order oracle/straps_oracle.py::rasterize_parts
// uses, so the part maps agree bit for bit.
#include "common.h"

#pragma clang fp contract(off)

namespace {

__global__ __launch_bounds__(256) STRAPS_NO_PACKED_FP32 void raster_project_kernel(const float* __restrict__ verts, const float* __restrict__ K,
                                                             const float* __restrict__ R, const float* __restrict__ t,
                                                             float* __restrict__ ndc, long long n, int nverts, int cam_per_body,
                                                             float orig, const float* __restrict__ noise_u, float noise_lo,
                                                             float noise_scale) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const long long b = i / nverts;
    const float* Kb = K + (cam_per_body ? b * 9 : 0);
    const float* Rb = R + (cam_per_body ? b * 9 : 0);
    const float* tb = t + b * 3;
    float x = verts[i * 3 + 0], y = verts[i * 3 + 1];
    const float z = verts[i * 3 + 2];
    if (noise_u) {   // random_verts2D_deviation (augmentation/proxy_rep_augmentation.py:5-22): x,y += (h - l) * rand + l, rendering copy only
        x = x + (noise_scale * noise_u[i * 2 + 0] + noise_lo);
        y = y + (noise_scale * noise_u[i * 2 + 1] + noise_lo);
    }
    const float xc = ((Rb[0] * x + Rb[1] * y) + Rb[2] * z) + tb[0];
    const float yc = ((Rb[3] * x + Rb[4] * y) + Rb[5] * z) + tb[1];
    const float zc = ((Rb[6] * x + Rb[7] * y) + Rb[8] * z) + tb[2];
    const float den = zc + 1e-9f;
    const float xn = xc / den, yn = yc / den;
    const float u = (Kb[0] * xn + Kb[1] * yn) + Kb[2];
    float v = (Kb[3] * xn + Kb[4] * yn) + Kb[5];
    v = orig - v;
    const float half = orig / 2.f;
    ndc[i * 3 + 0] = 2.f * (u - half) / orig;
    ndc[i * 3 + 1] = 2.f * (v - half) / orig;
    ndc[i * 3 + 2] = zc;
}

__device__ __forceinline__ float edge_fn(float ax, float ay, float bx, float by, float px, float py) {
    return (px - ax) * (by - ay) - (py - ay) * (bx - ax);
}

#ifdef STRAPS_RASTER_CHECK_LOADS
__device__ unsigned g_raster_report[1 + 4 * 64];
#endif

template <int G>      // lanes per face: they take the samples of its bounding box round-robin (the z-buffer minimum does not depend on who visits what)
__global__ __launch_bounds__(256) STRAPS_NO_PACKED_FP32 void raster_face_kernel(const float* __restrict__ ndc, const int32_t* __restrict__ faces,
                                                          unsigned long long* __restrict__ zbuf, long long n, int nverts,
                                                          int nfaces, int wh, float near, float far) {
    // pixel-centre coordinates (2k + 1 - wh) / wh, computed per sample on purpose.  Rounds 2-4 read them from a table in LDS (filled per
    // workgroup, one barrier); round 4 found that form NOT bit-reproducible when -- and only when -- bf16x3 convolution kernels run on
    // another stream: a handful of z-buffer keys per launch differ, faces drawn a pixel off with other depths, although every table entry
    // read back equals the value written (tools/datagen_determinism_probe.py, PROBE_LOAD=conv; profiles/r04_raster_determinism.txt).  The
    // training step's data stream runs exactly there, and two replays of a resnet50 step graph differed in about one 60-step run of three.
    // Without LDS and barrier the kernel is reproducible under the same load (no key differs in 1500 launches, 16 of 16 long runs equal).
    const bool pow2 = (wh & (wh - 1)) == 0;
    const float inv_wh = 1.f / (float)wh;
#if defined(STRAPS_RASTER_WRITE_TABLE) && !defined(STRAPS_RASTER_LDS_TABLE)      // (reproducer variants: table written + barrier, never read)
    extern __shared__ float sample[];
    for (int k = threadIdx.x; k < wh; k += 256) sample[k] = (float)(2 * k + 1 - wh) / (float)wh;
    __syncthreads();
#endif
#ifdef STRAPS_RASTER_DUMMY_BARRIER
    __syncthreads();
#endif
#ifdef STRAPS_RASTER_LDS_TABLE      // (the form of rounds 2-4, kept for the reproducer: STRAPS_TOOLS_RASTER_FLAGS=-DSTRAPS_RASTER_LDS_TABLE builds it into the tools library)
    extern __shared__ float sample[];
    for (int k = threadIdx.x; k < wh; k += 256) sample[k] = (float)(2 * k + 1 - wh) / (float)wh;
    __syncthreads();
#endif
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long i = gid / G;                   // (body, face)
    const int sub = (int)(gid & (G - 1));          // lane within the face's group of G
    if (i >= n) return;
    const long long b = i / nfaces;
    const int f = (int)(i - b * nfaces);
    const int i0 = faces[f * 3 + 0], i1 = faces[f * 3 + 1], i2 = faces[f * 3 + 2];
    if ((unsigned)i0 >= (unsigned)nverts || (unsigned)i1 >= (unsigned)nverts || (unsigned)i2 >= (unsigned)nverts) return;
    const float* p0 = ndc + (b * nverts + i0) * 3;
    const float* p1 = ndc + (b * nverts + i1) * 3;
    const float* p2 = ndc + (b * nverts + i2) * 3;
    const float x0 = p0[0], y0 = p0[1], z0 = p0[2];
    const float x1 = p1[0], y1 = p1[1], z1 = p1[2];
    const float x2 = p2[0], y2 = p2[1], z2 = p2[2];
    const float area = edge_fn(x0, y0, x1, y1, x2, y2);
    if (!(fabsf(area) > 1e-12f)) return;                       // degenerate (or NaN) face
    // pixel-centre sample k sits at (2k + 1 - wh) / wh; conservative bounding box of sample indices
    const float fw = (float)wh;
    const float xmin = fminf(x0, fminf(x1, x2)), xmax = fmaxf(x0, fmaxf(x1, x2));
    const float ymin = fminf(y0, fminf(y1, y2)), ymax = fmaxf(y0, fmaxf(y1, y2));
    if (!(xmax >= -1.f && xmin <= 1.f && ymax >= -1.f && ymin <= 1.f)) return;
    int xa = (int)floorf((fmaxf(xmin, -1.f) * fw + fw - 1.f) * 0.5f), xb = (int)ceilf((fminf(xmax, 1.f) * fw + fw - 1.f) * 0.5f);
    int ya = (int)floorf((fmaxf(ymin, -1.f) * fw + fw - 1.f) * 0.5f), yb = (int)ceilf((fminf(ymax, 1.f) * fw + fw - 1.f) * 0.5f);
    xa = xa < 0 ? 0 : xa; ya = ya < 0 ? 0 : ya;
    xb = xb > wh - 1 ? wh - 1 : xb; yb = yb > wh - 1 ? wh - 1 : yb;
    if (xb < xa || yb < ya) return;
    unsigned long long* zb = zbuf + b * (long long)wh * wh;
    const int bw = xb - xa + 1;
    // the group's G lanes take the box samples round-robin in row-major order
    int xi = xa + sub, yi = ya;
    while (xi > xb) { xi -= bw; ++yi; }
    for (; yi <= yb;) {
        {
            float yp, xp;
#ifdef STRAPS_RASTER_LDS_TABLE
            if (wh > 0) {
                yp = sample[yi];
                xp = sample[xi];
            } else
#endif
            if (pow2) {        // (wh a power of two, 256 everywhere in the training step: the product with 1 / wh IS the quotient, bit for bit)
                yp = (float)(2 * yi + 1 - wh) * inv_wh;
                xp = (float)(2 * xi + 1 - wh) * inv_wh;
            } else {
                yp = (float)(2 * yi + 1 - wh) / (float)wh;
                xp = (float)(2 * xi + 1 - wh) / (float)wh;
            }
            const float e0 = edge_fn(x1, y1, x2, y2, xp, yp);     // weight of vertex 0
            const float e1 = edge_fn(x2, y2, x0, y0, xp, yp);
            const float e2 = edge_fn(x0, y0, x1, y1, xp, yp);
            const bool in = (e0 >= 0.f && e1 >= 0.f && e2 >= 0.f) || (e0 <= 0.f && e1 <= 0.f && e2 <= 0.f);
            if (in) {
            float w0 = e0 / area, w1 = e1 / area, w2 = e2 / area;
            w0 = fminf(fmaxf(w0, 0.f), 1.f); w1 = fminf(fmaxf(w1, 0.f), 1.f); w2 = fminf(fmaxf(w2, 0.f), 1.f);
            const float ws = (w0 + w1) + w2;
            w0 = w0 / ws; w1 = w1 / ws; w2 = w2 / ws;
            const float zp = 1.f / ((w0 / z0 + w1 / z1) + w2 / z2);
            if (zp > near && zp < far) {
                const unsigned long long key = ((unsigned long long)__float_as_uint(zp) << 32) | (unsigned)f;
                atomicMin(zb + (long long)yi * wh + xi, key);
            }
            }
        }
        xi += G;
        while (xi > xb) { xi -= bw; ++yi; }
    }
#ifdef STRAPS_RASTER_CHECK_LOADS
    // (reproducer builds only: did the nine coordinates this lane has been computing with come back from memory as they are stored there?)
    {
        const volatile float* q0 = p0; const volatile float* q1 = p1; const volatile float* q2 = p2;
        const float r[9] = {q0[0], q0[1], q0[2], q1[0], q1[1], q1[2], q2[0], q2[1], q2[2]};
        const float h[9] = {x0, y0, z0, x1, y1, z1, x2, y2, z2};
        for (int e = 0; e < 9; ++e)
            if (__float_as_uint(r[e]) != __float_as_uint(h[e])) {
                const unsigned k = atomicAdd(g_raster_report, 1u);
                if (k < 64) {
                    g_raster_report[1 + 4 * k + 0] = blockIdx.x; g_raster_report[1 + 4 * k + 1] = threadIdx.x | ((unsigned)e << 16) | ((unsigned)f << 20);
                    g_raster_report[1 + 4 * k + 2] = __float_as_uint(h[e]); g_raster_report[1 + 4 * k + 3] = __float_as_uint(r[e]);
                }
            }
    }
#endif
}

__global__ __launch_bounds__(256) void raster_resolve_kernel(const unsigned long long* __restrict__ zbuf,
                                                             const uint8_t* __restrict__ face_parts, float* __restrict__ parts,
                                                             float* __restrict__ depth, long long n, int wh, float far) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const long long b = i / ((long long)wh * wh);
    const int rem = (int)(i - b * (long long)wh * wh);
    const int row = rem / wh, col = rem - row * wh;
    const unsigned long long key = zbuf[b * (long long)wh * wh + (long long)(wh - 1 - row) * wh + col];   // final vertical flip
    const bool hit = key != ~0ULL;
    if (parts) parts[i] = hit ? (float)face_parts[(unsigned)(key & 0xffffffffULL)] : 0.f;
    if (depth) depth[i] = hit ? __uint_as_float((unsigned)(key >> 32)) : far;
}

}  // namespace

int straps_fill_bytes(void* ptr, size_t bytes, unsigned char value, hipStream_t st);      // csrc/augment.hip

extern "C" size_t straps_rasterize_workspace_bytes(long long batch, int nverts, int wh) {
    if (batch <= 0 || nverts <= 0 || wh <= 0) return 0;
    return (size_t)batch * ((size_t)wh * wh * sizeof(unsigned long long) + (size_t)nverts * 3 * sizeof(float));
}

extern "C" int straps_rasterize_parts(const float* verts, const int32_t* faces, const uint8_t* face_parts, const float* cam_K,
                                      const float* cam_R, const float* cam_t, float* parts, float* depth, void* workspace,
                                      long long batch, int nverts, int nfaces, int wh, int cam_per_body, float near, float far,
                                      const float* vert_noise_u, double noise_lo, double noise_hi, void* stream) {
    STRAPS_REQUIRE(verts && faces && face_parts && cam_K && cam_R && cam_t && workspace, "straps_rasterize_parts: null pointer");
    STRAPS_REQUIRE(parts || depth, "straps_rasterize_parts: neither parts nor depth requested");
    STRAPS_REQUIRE(batch > 0 && nverts > 0 && nfaces > 0 && wh > 0 && wh <= 4096, "straps_rasterize_parts: bad sizes (batch %lld, %d verts, %d faces, wh %d)",
                   batch, nverts, nfaces, wh);
    STRAPS_REQUIRE(far > near && near >= 0.f, "straps_rasterize_parts: need 0 <= near < far");
    hipStream_t st = (hipStream_t)stream;
    unsigned long long* zbuf = (unsigned long long*)workspace;
    float* ndc = (float*)(zbuf + batch * (long long)wh * wh);
    {   // z-buffer = all ones (a fill kernel: see straps_memset_zero on memset nodes in captured graphs)
        const int rc = straps_fill_bytes(zbuf, (size_t)batch * wh * wh * sizeof(unsigned long long), 0xff, st);
        if (rc != STRAPS_OK) return rc;
    }
    const long long nv = batch * nverts, nf = batch * nfaces, np = batch * (long long)wh * wh;
    STRAPS_REQUIRE((nv + 255) / 256 < (1LL << 31) && (nf * 16 + 255) / 256 < (1LL << 31) && (np + 255) / 256 < (1LL << 31), "straps_rasterize_parts: batch too large for one launch");
    hipLaunchKernelGGL(raster_project_kernel, dim3((unsigned)((nv + 255) / 256)), dim3(256), 0, st, verts, cam_K, cam_R, cam_t, ndc, nv, nverts,
                       cam_per_body, (float)wh, vert_noise_u, (float)noise_lo, (float)(noise_hi - noise_lo));
    STRAPS_CHECK_LAUNCH("raster_project_kernel");
    // lanes per face: a face of the 13 776-face mesh at 256 x 256 covers one or two samples and its box four to nine
    static const int lanes = STRAPS_TOOL_ENV_INT("STRAPS_RASTER_LANES", 16);      // (A/B switch of the tools build)
#if defined(STRAPS_RASTER_LDS_TABLE) || defined(STRAPS_RASTER_WRITE_TABLE) || defined(STRAPS_RASTER_DUMMY_LDS)
    // (round 5 reproducer switch: STRAPS_RASTER_LDS_EXTRA bytes of LDS the kernel never touches -- with 20 KB a workgroup no longer fits beside a
    //  convolution workgroup that holds 140-147 KB of a CU's 160 KB: co-residency on one CU switched off without touching the code)
    static const int lds_extra = STRAPS_TOOL_ENV_INT("STRAPS_RASTER_LDS_EXTRA", 0);
#define STRAPS_RASTER_LDS_BYTES ((size_t)wh * sizeof(float) + (size_t)lds_extra)
#else
#define STRAPS_RASTER_LDS_BYTES 0
#endif
#define STRAPS_RASTER_LAUNCH(G) hipLaunchKernelGGL(raster_face_kernel<G>, dim3((unsigned)((nf * G + 255) / 256)), dim3(256), STRAPS_RASTER_LDS_BYTES, st, ndc, faces, zbuf, nf, nverts, nfaces, wh, near, far)
    if (lanes == 1) STRAPS_RASTER_LAUNCH(1);
    else if (lanes == 2) STRAPS_RASTER_LAUNCH(2);
    else if (lanes == 4) STRAPS_RASTER_LAUNCH(4);
    else if (lanes == 8) STRAPS_RASTER_LAUNCH(8);
    else if (lanes == 32) STRAPS_RASTER_LAUNCH(32);
    else if (lanes == 64) STRAPS_RASTER_LAUNCH(64);
    else STRAPS_RASTER_LAUNCH(16);
#undef STRAPS_RASTER_LAUNCH
    STRAPS_CHECK_LAUNCH("raster_face_kernel");
    hipLaunchKernelGGL(raster_resolve_kernel, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, st, zbuf, face_parts, parts, depth, np, wh, far);
    STRAPS_CHECK_LAUNCH("raster_resolve_kernel");
    return STRAPS_OK;
}

#ifdef STRAPS_TOOLS
// tools build only (round 5, tools/datagen_determinism_probe.py PROBE_LOAD=occupy): workgroups that HOLD `lds_bytes` of LDS and sleep for about
// `microseconds` -- no LDS traffic, no matrix work, no memory traffic.  Beside them a small-LDS kernel is placed on the same CU, its allocation
// above theirs: does the victim of DESIGN section 1 need the convolution kernels' traffic, or only their LDS footprint?
namespace {
__global__ __launch_bounds__(256) void lds_occupier_kernel(int microseconds, unsigned* sink) {
    extern __shared__ unsigned occ[];
    if (threadIdx.x == 0) occ[0] = blockIdx.x;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();      // 100 MHz constant-rate counter
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)microseconds * 100ull) __builtin_amdgcn_s_sleep(32);
    if (threadIdx.x == 0 && occ[0] == 0xffffffffu) sink[0] = 1;          // (never true: keeps the LDS store alive)
}
}  // namespace
// ... and workgroups that do nothing but the convolution kernels' operand-fragment reads: LDS filled once, then ds_read_b128 of 64-byte swizzled rows
// (csrc/conv_x3.hip's addressing) feeding v_mfma_f32_32x32x16_bf16 (mfma != 0) or a checksum, one barrier per "chunk"; no memory traffic
namespace {
typedef __bf16 tool_bf16x8 __attribute__((ext_vector_type(8)));
template <int MFMA>
__device__ __forceinline__ void lds_frag_reader_body(unsigned* __restrict__ sink, int trips) {
    extern __shared__ __attribute__((aligned(16))) unsigned fr_lds[];      // 147 KB = 3 stages x 3 planes x (128 + 128) rows x 64 bytes
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 147 * 256; i += 256) fr_lds[i] = 0x3f803f80u ^ (unsigned)(i * 2654435761u >> 20);
    __syncthreads();
    const unsigned short* As = reinterpret_cast<const unsigned short*>(fr_lds);
    const int wm = wave >> 1, wn = wave & 1;
    int fo[2];
    for (int kk = 0; kk < 2; ++kk) fo[kk] = (lane & 31) * 32 + (((kk * 2 + (lane >> 5)) ^ (((lane & 31) >> 3) & 3)) << 3);
    f32x16 c[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) c[i][j][r] = 0.f;
    unsigned acc = 0;
    int stage = 0;
    // MFMA == 2: the convolution kernel's register footprint as well -- 200 more registers kept live across the loop (no instruction touches
    // them: the empty asm only pins them; the kernel may use all 512 registers), so that ONE wave per SIMD fits and registers spill over into AGPRs, as in the 128x128 kernel
    float keep[MFMA == 2 ? 230 : 1];
    for (int q = 0; q < (MFMA == 2 ? 230 : 1); ++q) keep[q] = (float)(lane + q);
    for (int t = 0; t < trips; ++t) {
        if (MFMA == 2) {
#pragma unroll
            for (int q = 0; q < (MFMA == 2 ? 230 : 1); ++q) asm volatile("" : "+v"(keep[q]));
        }
        const unsigned short* Ab = As + (stage * 3 * 256 + wm * 64) * 32;
        const unsigned short* Bb = As + (stage * 3 * 256 + 128 + wn * 64) * 32;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            tool_bf16x8 a[2][3], b[2][3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    a[i][pl] = *reinterpret_cast<const tool_bf16x8*>(Ab + (pl * 256 + i * 32) * 32 + fo[kk]);
                    b[i][pl] = *reinterpret_cast<const tool_bf16x8*>(Bb + (pl * 256 + i * 32) * 32 + fo[kk]);
                }
            if (MFMA) {
                constexpr int TA[6] = {1, 0, 2, 0, 1, 0}, TB[6] = {1, 2, 0, 1, 0, 0};
#pragma unroll
                for (int q = 0; q < 6; ++q)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) c[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][TA[q]], b[j][TB[q]], c[i][j], 0, 0, 0);
            } else {
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const uint4 ua = *reinterpret_cast<const uint4*>(&a[i][pl]), ub = *reinterpret_cast<const uint4*>(&b[i][pl]);
                        acc += (ua.x ^ ua.y ^ ua.z ^ ua.w) + (ub.x ^ ub.y ^ ub.z ^ ub.w);
                    }
            }
        }
        stage = stage == 2 ? 0 : stage + 1;
        __builtin_amdgcn_s_barrier();
    }
    if (MFMA) acc += (unsigned)(c[0][0][0] + c[0][1][5] + c[1][0][15] + c[1][1][7] == 12345.678f);
    if (MFMA == 2) { float sum = 0.f; for (int q = 0; q < 230; ++q) sum += keep[q]; acc += (unsigned)(sum == 12345.678f); }
    if (acc == 0x12345678u) sink[blockIdx.x] = acc;
}
template <int MFMA>
__global__ __launch_bounds__(256) void lds_frag_reader_kernel(unsigned* __restrict__ sink, int trips) { lds_frag_reader_body<MFMA>(sink, trips); }
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void lds_frag_reader_big_kernel(unsigned* __restrict__ sink, int trips) {
    lds_frag_reader_body<2>(sink, trips);
}
}  // namespace
extern "C" int straps_tool_lds_frag_reader(int mfma, int trips, int blocks, unsigned* sink, void* stream) {
    STRAPS_REQUIRE(trips > 0 && blocks > 0 && sink, "straps_tool_lds_frag_reader: bad arguments");
    if (mfma == 2) {
        STRAPS_RAISE_LDS(lds_frag_reader_big_kernel, 147 * 1024, "lds_frag_reader_big_kernel");
        hipLaunchKernelGGL(lds_frag_reader_big_kernel, dim3(blocks), dim3(256), 147 * 1024, (hipStream_t)stream, sink, trips);
    } else if (mfma) {
        STRAPS_RAISE_LDS(lds_frag_reader_kernel<1>, 147 * 1024, "lds_frag_reader_kernel");
        hipLaunchKernelGGL(lds_frag_reader_kernel<1>, dim3(blocks), dim3(256), 147 * 1024, (hipStream_t)stream, sink, trips);
    } else {
        STRAPS_RAISE_LDS(lds_frag_reader_kernel<0>, 147 * 1024, "lds_frag_reader_kernel");
        hipLaunchKernelGGL(lds_frag_reader_kernel<0>, dim3(blocks), dim3(256), 147 * 1024, (hipStream_t)stream, sink, trips);
    }
    STRAPS_CHECK_LAUNCH("lds_frag_reader_kernel");
    return STRAPS_OK;
}
extern "C" int straps_tool_lds_occupier(size_t lds_bytes, int microseconds, int blocks, unsigned* sink, void* stream) {
    STRAPS_REQUIRE(lds_bytes >= 4 && lds_bytes <= 160 * 1024 && blocks > 0 && sink, "straps_tool_lds_occupier: bad arguments");
    STRAPS_RAISE_LDS(lds_occupier_kernel, 160 * 1024, "lds_occupier_kernel");
    hipLaunchKernelGGL(lds_occupier_kernel, dim3(blocks), dim3(256), lds_bytes, (hipStream_t)stream, microseconds, sink);
    STRAPS_CHECK_LAUNCH("lds_occupier_kernel");
    return STRAPS_OK;
}
#endif

#ifdef STRAPS_RASTER_CHECK_LOADS
// reproducer builds only (-DSTRAPS_RASTER_CHECK_LOADS): copies out and clears what raster_face_kernel reported
extern "C" int straps_tool_raster_report(unsigned* host_out) {
    if (hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_raster_report), sizeof(unsigned) * (1 + 4 * 64)) != hipSuccess) return STRAPS_EHIP;
    unsigned zero[1 + 4 * 64] = {0};
    return hipMemcpyToSymbol(HIP_SYMBOL(g_raster_report), zero, sizeof(zero)) == hipSuccess ? STRAPS_OK : STRAPS_EHIP;
}
#endif
