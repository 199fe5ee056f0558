// tools/lds_read_hazard_repro.hip -- stand-alone form of the rasteriser finding of round 4 (profiles/r04_raster_determinism.txt):
// is a kernel that READS a small LDS table with data-dependent addresses inside a divergent loop bit-reproducible while, on another stream,
// workgroups stream global memory into LDS with LDS-DMA (global_load_lds_dwordx4) -- the operand path of the bf16x3 convolution kernels?
//
// Written at the end of round 4 and run twice with the round's last GPU seconds (150 launches per cell, profiles/r04_lds_read_hazard_repro.txt):
// 0 differing launches in all four cells, with aggressor mode 0 (LDS-DMA copies + a few LDS reads) and mode 3 (+ 12 bf16 MFMAs per wave and
// trip, eight waves) -- neither is sufficient; the library's convolution kernel is (tools/datagen_determinism_probe.py PROBE_LOAD=conv).
// Open: what else of that kernel matters (its 140 KB of LDS and one workgroup per CU, the swizzled ds_read_b128 fragment traffic, the padding
// source, the epilogue), or whether this victim -- synchronised with the host after every launch -- simply does not meet the aggressor the way
// the probe's back-to-back launches do.  No torch, no library of this repository.
//
//   victim    : raster_face_kernel's skeleton -- 256 threads fill a 256-entry float table in LDS, barrier, then every 16-lane group walks a
//               pseudo-random box of (row, column) pairs round-robin, reads table[row] and table[column], runs the three edge functions and the
//               depth interpolation of csrc/raster.hip on them, and combines the results with atomicMin into a 64-bit key buffer.
//               Variant 1 reads the table (the round 2-4 form); variant 0 computes the same two values arithmetically (the product form).
//   aggressor : workgroups of 256 threads with 128 KB of LDS that copy a large buffer into it, 8 KB per trip, with global_load_lds_dwordx4,
//               wait and barrier, and fold a few ds_read_b128 of what arrived into a checksum (so that nothing is optimised away); replayed
//               as a hipGraph of four launches to keep the chip full, like tools/datagen_determinism_probe.py's PROBE_LOAD=conv.
//   check     : the victim's key buffer of launch i against that of launch 0 (same inputs), on the host, for `reps` launches; with and
//               without the aggressor; for both variants.  Expected if the finding is what it looks like: differences only for
//               (variant 1, aggressor on).
//
//     hipcc --offload-arch=gfx950 -O3 tools/lds_read_hazard_repro.hip -o tools/bin/lds_read_hazard_repro && tools/bin/lds_read_hazard_repro [reps] [aggressor mode 0..3]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

#pragma clang fp contract(off)

constexpr int WH = 256, NFACE = 13776, NBODY = 4;

__device__ __forceinline__ float edge_fn(float ax, float ay, float bx, float by, float px, float py) { return (px - ax) * (by - ay) - (py - ay) * (bx - ax); }

template <int TABLE>
__global__ __launch_bounds__(256) void victim_kernel(const float* __restrict__ tri, unsigned long long* __restrict__ zbuf, long long n) {
    extern __shared__ float sample[];
    if (TABLE) {
        for (int k = threadIdx.x; k < WH; k += 256) sample[k] = (float)(2 * k + 1 - WH) / (float)WH;
        __syncthreads();
    }
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long i = gid >> 4;
    const int sub = (int)(gid & 15);
    if (i >= n) return;
    const long long b = i / NFACE;
    const int f = (int)(i - b * NFACE);
    const float* p = tri + i * 9;
    const float x0 = p[0], y0 = p[1], z0 = p[2], x1 = p[3], y1 = p[4], z1 = p[5], x2 = p[6], y2 = p[7], z2 = p[8];
    const float area = edge_fn(x0, y0, x1, y1, x2, y2);
    if (!(fabsf(area) > 1e-12f)) return;
    const float fw = (float)WH;
    const float xmin = fminf(x0, fminf(x1, x2)), xmax = fmaxf(x0, fmaxf(x1, x2));
    const float ymin = fminf(y0, fminf(y1, y2)), ymax = fmaxf(y0, fmaxf(y1, y2));
    int xa = (int)floorf((fmaxf(xmin, -1.f) * fw + fw - 1.f) * 0.5f), xb = (int)ceilf((fminf(xmax, 1.f) * fw + fw - 1.f) * 0.5f);
    int ya = (int)floorf((fmaxf(ymin, -1.f) * fw + fw - 1.f) * 0.5f), yb = (int)ceilf((fminf(ymax, 1.f) * fw + fw - 1.f) * 0.5f);
    xa = xa < 0 ? 0 : xa; ya = ya < 0 ? 0 : ya;
    xb = xb > WH - 1 ? WH - 1 : xb; yb = yb > WH - 1 ? WH - 1 : yb;
    if (xb < xa || yb < ya) return;
    unsigned long long* zb = zbuf + b * (long long)WH * WH;
    const int bw = xb - xa + 1;
    int xi = xa + sub, yi = ya;
    while (xi > xb) { xi -= bw; ++yi; }
    for (; yi <= yb;) {
        const float yp = TABLE ? sample[yi] : (float)(2 * yi + 1 - WH) * (1.f / (float)WH);
        const float xp = TABLE ? sample[xi] : (float)(2 * xi + 1 - WH) * (1.f / (float)WH);
        const float e0 = edge_fn(x1, y1, x2, y2, xp, yp), e1 = edge_fn(x2, y2, x0, y0, xp, yp), e2 = edge_fn(x0, y0, x1, y1, xp, yp);
        const bool in = (e0 >= 0.f && e1 >= 0.f && e2 >= 0.f) || (e0 <= 0.f && e1 <= 0.f && e2 <= 0.f);
        if (in) {
            float w0 = e0 / area, w1 = e1 / area, w2 = e2 / area;
            w0 = fminf(fmaxf(w0, 0.f), 1.f); w1 = fminf(fmaxf(w1, 0.f), 1.f); w2 = fminf(fmaxf(w2, 0.f), 1.f);
            const float ws = (w0 + w1) + w2;
            w0 = w0 / ws; w1 = w1 / ws; w2 = w2 / ws;
            const float zp = 1.f / ((w0 / z0 + w1 / z1) + w2 / z2);
            if (zp > 0.1f && zp < 100.f) atomicMin(zb + (long long)yi * WH + xi, ((unsigned long long)__float_as_uint(zp) << 32) | (unsigned)f);
        }
        xi += 16;
        while (xi > xb) { xi -= bw; ++yi; }
    }
}

typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8_t;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16_t;

// MODE bit 0: an MFMA block per trip on what arrived (12 x v_mfma_f32_32x32x16_bf16 per wave, operands by ds_read_b128: the convolution's inner loop
// in spirit); the host launches 256 or 512 threads (argv[2]) -- the steps towards "more like the convolution kernel" that the header names
template <int MODE>
__global__ __launch_bounds__(512) void aggressor_kernel(const unsigned* __restrict__ src, long long words, unsigned* __restrict__ sink, int trips) {
    extern __shared__ __attribute__((aligned(16))) unsigned lds[];          // 128 KB
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6) & 3;
    unsigned acc = 0;
    f32x16_t c = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int t = 0; t < trips; ++t) {
        const long long base = (((long long)blockIdx.x * trips + t) * 2048) % (words - 2048);           // 8 KB per trip
        unsigned* dst = lds + (t & 15) * 2048;
#pragma unroll
        for (int r = 0; r < 2; ++r)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + base + ((r * 4 + wave) * 64 + lane) * 4),
                                             (__attribute__((address_space(3))) void*)(dst + (r * 4 + wave) * 256), 16, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const uint4 v = *reinterpret_cast<const uint4*>(dst + ((tid * 4) & 2047));
        acc += v.x ^ v.y ^ v.z ^ v.w;
        if (MODE & 1) {
#pragma unroll
            for (int k = 0; k < 12; ++k) {
                const bf16x8_t a = *reinterpret_cast<const bf16x8_t*>(dst + ((lane * 4 + k * 256) & 2047));
                const bf16x8_t b = *reinterpret_cast<const bf16x8_t*>(dst + ((lane * 4 + k * 256 + 1024) & 2047));
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
            }
        }
        __builtin_amdgcn_s_barrier();
    }
    if (MODE & 1) acc += (unsigned)(c[0] + c[5] + c[15] == 12345.678f);
    if (acc == 0x12345678u) sink[blockIdx.x] = acc;          // (never true in practice: keeps the reads alive)
}

// Round 5 (tools/conv_family_probe2.sh, profiles/r05_conv_family_probe2.txt): inside the library the victim is disturbed by a CO-RESIDENT convolution
// workgroup exactly when that workgroup streams its operand FRAGMENTS out of LDS (ds_read_b128) -- with its LDS-DMA copies removed it still
// is, with the fragment reads removed (copies + barriers, or MFMAs + barriers) it is not, and a workgroup that merely holds the LDS is harmless.
// Aggressor modes 4 / 5 are that ingredient alone: one four-wave workgroup per CU (147 KB of LDS, so that nothing but a small-LDS kernel fits
// beside it), LDS filled once, then nothing but the convolution kernel's fragment reads -- 64-byte rows, the XOR swizzle of csrc/conv_x3.hip --
// feeding v_mfma_f32_32x32x16_bf16 (mode 4) or a checksum (mode 5), one barrier per "chunk".  No global memory traffic after the fill.
__device__ __forceinline__ int swz3(int r) { return (r >> 3) & 3; }
template <int MFMA>
__global__ __launch_bounds__(256) void frag_reader_kernel(unsigned* __restrict__ sink, int trips) {
    extern __shared__ __attribute__((aligned(16))) unsigned lds[];          // 147 KB = 3 stages x 3 planes x (128 + 128) rows x 64 bytes
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 147 * 256; i += 256) lds[i] = 0x3f803f80u ^ (unsigned)(i * 2654435761u >> 20);      // (bf16 values near 1)
    __syncthreads();
    const unsigned short* As = reinterpret_cast<const unsigned short*>(lds);
    const int wm = wave >> 1, wn = wave & 1;
    int fo[2];
    for (int kk = 0; kk < 2; ++kk) fo[kk] = (lane & 31) * 32 + (((kk * 2 + (lane >> 5)) ^ swz3(lane & 31)) << 3);
    f32x16_t c[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) c[i][j][r] = 0.f;
    unsigned acc = 0;
    int stage = 0;
    for (int t = 0; t < trips; ++t) {
        const unsigned short* Ab = As + (stage * 3 * 256 + wm * 64) * 32;
        const unsigned short* Bb = As + (stage * 3 * 256 + 128 + wn * 64) * 32;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8_t a[2][3], b[2][3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    a[i][pl] = *reinterpret_cast<const bf16x8_t*>(Ab + (pl * 256 + i * 32) * 32 + fo[kk]);
                    b[i][pl] = *reinterpret_cast<const bf16x8_t*>(Bb + (pl * 256 + i * 32) * 32 + fo[kk]);
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
    if (acc == 0x12345678u) sink[blockIdx.x] = acc;
}

int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 400;
    const long long n = (long long)NBODY * NFACE;
    // triangles: a few pixels wide, scattered over the middle of the image, depths around 42 (the training step's geometry in spirit)
    std::vector<float> tri(n * 9);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)(s >> 8) / 16777216.f; };
    for (long long i = 0; i < n; ++i) {
        const float cx = (rnd() - 0.5f) * 0.9f, cy = (rnd() - 0.5f) * 1.6f, z = 41.f + rnd() * 2.f;
        for (int v = 0; v < 3; ++v) { tri[i * 9 + v * 3 + 0] = cx + (rnd() - 0.5f) * 0.03f; tri[i * 9 + v * 3 + 1] = cy + (rnd() - 0.5f) * 0.03f; tri[i * 9 + v * 3 + 2] = z + rnd() * 0.1f; }
    }
    float* dtri; unsigned long long* dz; unsigned* dsrc; unsigned* dsink;
    const long long words = 1LL << 26;           // 256 MB of source for the aggressor
    const size_t zbytes = (size_t)NBODY * WH * WH * 8;
    CK(hipMalloc(&dtri, tri.size() * 4)); CK(hipMalloc(&dz, zbytes)); CK(hipMalloc(&dsrc, words * 4)); CK(hipMalloc(&dsink, 4096 * 4));
    CK(hipMemcpy(dtri, tri.data(), tri.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(dsrc, 0x5a, words * 4));
    const int amode = argc > 2 ? atoi(argv[2]) : 0;           // 0: LDS-DMA copies only; 1: + an MFMA block per trip; 2: 512 threads; 3: both; 4 / 5: fragment-read stream with / without MFMAs
    const int athreads = (amode & 2) ? 512 : 256;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(frag_reader_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 147 * 1024));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(frag_reader_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 147 * 1024));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(aggressor_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(aggressor_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    hipStream_t sv, sa;
    CK(hipStreamCreateWithFlags(&sv, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(sa, hipStreamCaptureModeThreadLocal));
    const int ftrips = argc > 3 ? atoi(argv[3]) : 400;
    for (int k = 0; k < 4; ++k) {
        if (amode == 4) hipLaunchKernelGGL(frag_reader_kernel<1>, dim3(256), dim3(256), 147 * 1024, sa, dsink, ftrips);
        else if (amode == 5) hipLaunchKernelGGL(frag_reader_kernel<0>, dim3(256), dim3(256), 147 * 1024, sa, dsink, ftrips);
        else if (amode & 1) hipLaunchKernelGGL(aggressor_kernel<1>, dim3(512), dim3(athreads), 128 * 1024, sa, dsrc, words, dsink, 64);
        else hipLaunchKernelGGL(aggressor_kernel<0>, dim3(512), dim3(athreads), 128 * 1024, sa, dsrc, words, dsink, 64);
    }
    if (amode >= 4) printf("aggressor mode %d: one 256-thread workgroup per CU holding 147 KB of LDS, fragment reads (ds_read_b128, 64-byte swizzled rows)%s, %d chunks per launch, no memory traffic\n", amode, amode == 4 ? " feeding 48 bf16 MFMAs per wave and chunk" : " into a checksum", ftrips);
    else printf("aggressor mode %d: %d threads per workgroup, 128 KB of LDS, LDS-DMA copies%s\n", amode, athreads, (amode & 1) ? " + 12 bf16 MFMAs per wave and trip" : "");
    CK(hipStreamEndCapture(sa, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    std::vector<unsigned long long> ref(zbytes / 8), cur(zbytes / 8);
    const unsigned grid = (unsigned)((n * 16 + 255) / 256);
    for (int table = 1; table >= 0; --table)
        for (int aggress = 0; aggress < 2; ++aggress) {
            long long differing = 0; int bad_launches = 0;
            for (int r = 0; r <= reps; ++r) {
                if (aggress) CK(hipGraphLaunch(ge, sa));
                CK(hipMemsetAsync(dz, 0xff, zbytes, sv));
                if (table) hipLaunchKernelGGL(victim_kernel<1>, dim3(grid), dim3(256), WH * 4, sv, dtri, dz, n);
                else hipLaunchKernelGGL(victim_kernel<0>, dim3(grid), dim3(256), 0, sv, dtri, dz, n);
                CK(hipMemcpyAsync(r == 0 ? ref.data() : cur.data(), dz, zbytes, hipMemcpyDeviceToHost, sv));
                CK(hipStreamSynchronize(sv));
                if (r) {
                    long long d = 0;
                    for (size_t k = 0; k < ref.size(); ++k) d += ref[k] != cur[k];
                    differing += d; bad_launches += d != 0;
                }
            }
            CK(hipStreamSynchronize(sa));
            printf("victim %s, LDS-DMA aggressor %s: %d of %d launches differ from the first (%lld keys in all)\n",
                   table ? "READS its LDS table" : "computes the coordinates", aggress ? "ON " : "off", bad_launches, reps, differing);
        }
    return 0;
}
