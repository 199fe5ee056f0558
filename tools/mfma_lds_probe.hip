// tools/mfma_lds_probe.hip -- what does a chunk of the bf16x3 implicit GEMM's inner loop cost, piece by piece, on one CU-filling workgroup per CU?
// One iteration = one K chunk of a 64x64 wave tile: 48 v_mfma_f32_32x32x16_bf16 on 4 accumulator blocks, fed by 24 ds_read_b128
// (1 KiB each, conflict-free), optionally a workgroup barrier, optionally LDS-DMA copies of 48 KiB per iteration into a ring (by the
// compute waves themselves or by four extra loader waves).  Variants (argv-less: all are run):
//   0  MFMA only
//   1  + fragment reads, each k step's 12 reads in front of its 24 MFMAs (what the compiler emits for the plain loop)
//   2  + fragment reads software-pipelined: the reads of the NEXT k step are issued between the MFMAs of the current one (two register sets)
//   3  as 1 + one s_barrier per iteration
//   4  as 2 + one s_barrier per iteration
//   5  as 3 + the 48 KiB of LDS-DMA copies per iteration issued by the compute waves (12 per wave, after the barrier, in front of the MFMAs)
//   6  as 3 + the copies issued by four extra loader waves (one per SIMD)
//   7  as 4 + loader waves
//   8  as 1 but 8 compute waves per workgroup (two per SIMD), each with its own 64x64 tile (twice the work per CU: time per MFMA is what counts)
//   9  as 3 with 8 compute waves
// Prints ns per iteration, cycles per MFMA at the measured shader clock, and the clock.  Standalone: hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ f32x16 mfma(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }

constexpr int STAGE_BYTES = 3 * (128 + 128) * 64;      // one K chunk of a 128x128 block tile: 48 KiB
constexpr int NST = 3;

__global__ void fill_kernel(unsigned* p, unsigned n) {
    for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) p[i] = 0x3c003c00u | ((i * 2654435761u) & 0x83ff83ffu);
}

template <int MODE>
__global__ __launch_bounds__(MODE == 6 || MODE == 7 || MODE >= 8 ? 512 : 256) void probe(const char* __restrict__ src, float* out, unsigned long long* clk, int iters) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr bool READS = MODE >= 1, PIPE = MODE == 2 || MODE == 4 || MODE == 7, BAR = MODE >= 3 && MODE != 8;
    constexpr bool SELF_COPY = MODE == 5, LOADERS = MODE == 6 || MODE == 7;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // operand-like LDS contents for every mode (power depends on the data: zeros clock higher)
    for (int i = tid; i < NST * STAGE_BYTES / 4; i += blockDim.x) {
        const unsigned h = (unsigned)(i * 2654435761u) ^ (unsigned)(blockIdx.x * 40503u);
        reinterpret_cast<unsigned*>(lds)[i] = 0x3c003c00u | (h & 0x83ff83ffu);
    }
    __syncthreads();
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), w0 = __builtin_amdgcn_s_memrealtime();
    if (LOADERS && wave >= 4) {
        // loader waves: 12 copies of 1 KiB per iteration each, ring of NST stages, two iterations ahead
        const int lw = wave - 4;
        const char* g = src + ((size_t)blockIdx.x * 4 + lw) * 12 * 1024 + lane * 16;
        for (int it = 0; it < iters; ++it) {
            g += 256 * 4 * 12 * 1024; if (g >= src + (60u << 20)) g -= (48u << 20);        // (fresh bytes every iteration, L2-resident footprint)
            char* st = lds + (it % NST) * STAGE_BYTES + lw * 12 * 1024;
#pragma unroll
            for (int k = 0; k < 12; ++k)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + k * 1024), (__attribute__((address_space(3))) void*)(st + k * 1024), 16, 0, 0);
            asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    // fragment addresses: 64-byte rows, the product kernels' swizzle (conflict-free)
    const int cw = MODE >= 8 ? (wave & 3) : wave;
    const int row = lane & 31, kh = lane >> 5;
    int fo[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) fo[kk] = row * 64 + ((((kk * 2 + kh) ^ ((row >> 3) & 3))) << 4);
    const char* Ab = lds + (cw >> 1) * 64 * 64;
    const char* Bb = lds + 3 * 128 * 64 + (cw & 1) * 64 * 64;
    bf16x8 fa[2][2][3], fb[2][2][3];       // [set][block][plane]
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {          // (mode 0 multiplies these: real values, not zeros)
                fa[s][i][pl] = *reinterpret_cast<const bf16x8*>(lds + ((s * 6 + i * 3 + pl) * 64 + lane) * 16);
                fb[s][i][pl] = *reinterpret_cast<const bf16x8*>(lds + ((12 + s * 6 + i * 3 + pl) * 64 + lane) * 16);
            }
    auto load = [&](int set, int stage, int kk) {
        const char* A = Ab + stage * STAGE_BYTES;
        const char* B = Bb + stage * STAGE_BYTES;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[set][i][pl] = *reinterpret_cast<const bf16x8*>(A + (pl * 128 + i * 32) * 64 + fo[kk]);
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[set][j][pl] = *reinterpret_cast<const bf16x8*>(B + (pl * 128 + j * 32) * 64 + fo[kk]);
        }
    };
    constexpr int TA[6] = {1, 0, 2, 0, 1, 0}, TB[6] = {1, 2, 0, 1, 0, 0};
    auto mm = [&](int set) {
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i * 2 + j] = mfma(fa[set][i][TA[t]], fb[set][j][TB[t]], acc[i * 2 + j]);
    };
    const char* g = src + ((size_t)blockIdx.x * 4 + (wave & 3)) * 12 * 1024 + lane * 16;
    if (PIPE) load(0, 0, 0);
    for (int it = 0; it < iters; ++it) {
        const int stage = it % NST;
        if (BAR) {
            if (SELF_COPY) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
        if (SELF_COPY) {
            g += 256 * 4 * 12 * 1024; if (g >= src + (60u << 20)) g -= (48u << 20);
            char* st = lds + ((it + 2) % NST) * STAGE_BYTES + wave * 12 * 1024;
#pragma unroll
            for (int k = 0; k < 12; ++k)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + k * 1024), (__attribute__((address_space(3))) void*)(st + k * 1024), 16, 0, 0);
        }
        if (!READS) {
            mm(0); mm(0);
        } else if (!PIPE) {
            load(0, stage, 0);
            mm(0);
            load(1, stage, 1);
            mm(1);
        } else {
            load(1, stage, 1);
            __builtin_amdgcn_sched_barrier(0);
            mm(0);
            __builtin_amdgcn_sched_barrier(0);
            load(0, (it + 1) % NST, 0);
            __builtin_amdgcn_sched_barrier(0);
            mm(1);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) t += acc[i][r];
    out[(size_t)blockIdx.x * blockDim.x + tid] = t;
    if (blockIdx.x == 0 && tid == 0) {
        clk[0] = __builtin_amdgcn_s_memtime() - c0;
        clk[1] = __builtin_amdgcn_s_memrealtime() - w0;
    }
}

template <int MODE>
void run(const char* src, float* out, unsigned long long* clk, int khz) {
    const int iters = 2000, blocks = 256;
    const int threads = (MODE == 6 || MODE == 7 || MODE >= 8) ? 512 : 256;
    const size_t lds = (size_t)NST * STAGE_BYTES;
    hipFuncSetAttribute((const void*)probe<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(a, 0);
        hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(threads), lds, 0, src, out, clk, iters);
        hipEventRecord(b, 0);
        hipEventSynchronize(b);
        hipEventElapsedTime(&ms, a, b);
    }
    unsigned long long h[2];
    hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    const double mhz = (double)h[0] / (double)h[1] * khz / 1e3;
    const double ns_it = ms * 1e6 / iters;
    const int mf_per_simd = (MODE >= 8 ? 2 : 1) * 48;
    printf("mode %d: %8.1f ns / iteration  = %6.1f cycles per MFMA per SIMD at %6.0f MHz  (%s)\n", MODE, ns_it, ns_it * mhz / 1e3 / mf_per_simd, mhz,
           hipGetErrorString(hipGetLastError()));
    fflush(stdout);
}

int main() {
    char* src; float* out; unsigned long long* clk;
    hipMalloc(&src, 64u << 20); hipMalloc(&out, 256 * 512 * 4); hipMalloc(&clk, 16);
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, reinterpret_cast<unsigned*>(src), (64u << 20) / 4);      // operand-like random bf16 pairs
    int khz = 0;
    hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, 0);
    run<0>(src, out, clk, khz); run<1>(src, out, clk, khz); run<2>(src, out, clk, khz); run<3>(src, out, clk, khz); run<4>(src, out, clk, khz);
    run<5>(src, out, clk, khz); run<6>(src, out, clk, khz); run<7>(src, out, clk, khz); run<8>(src, out, clk, khz); run<9>(src, out, clk, khz);
    return 0;
}
