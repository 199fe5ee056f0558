// tools/l2_line_probe.hip -- what does the L2 -> LDS path (global_load_lds_dwordx4) deliver when a wave instruction's 64 x 16 bytes are
//   0: sixteen 64-byte pieces at a 128-byte stride (HALF cache lines: a 32-channel chunk of one bf16 plane of a Cin = 64 tensor, the
//      implicit GEMM's A / B operand rows as laid out today),
//   1: the same at a 256-byte stride (Cin = 128),
//   2: one contiguous 1 KiB (a chunk-major plane layout: [Cin/32][pixel][32]),
//   3: eight full 128-byte lines at a 256-byte stride (64-channel chunks of a Cin = 128 plane),
//   4: mode 0 followed at once by the other 64-byte halves of the same lines (does the vector L1 keep the line?),
//   5: mode 0 over a 36 KiB tile, then the other halves of the whole tile (the second touch comes one K chunk later, as in the kernels),
//   6: as 0 at a 1152-byte stride (weight rows [cout][9][64] of one plane),
//   7: as 0 with dwordx4 replaced by plain global_load_dwordx4 into registers (is it the LDS-DMA path or the cache-line geometry?)
// from an L2-resident footprint.  Prints useful GB/s (bytes the kernel wanted) per mode.  Standalone: hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int SLOTS = 8;                       // 1 KiB LDS slots per wave

__device__ __forceinline__ void dma(const char* g, char* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

template <int MODE>
__global__ __launch_bounds__(256) void probe(const char* __restrict__ src, unsigned footprint, int iters, unsigned* sink) {
    __shared__ __attribute__((aligned(16))) char lds[4 * SLOTS * 1024];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    char* my = lds + wave * SLOTS * 1024;
    // per-lane offset inside the region one instruction covers, and the region's extent in bytes
    unsigned off, extent;
    if (MODE == 0 || MODE == 4 || MODE == 5 || MODE == 7) { off = (lane >> 2) * 128 + (lane & 3) * 16; extent = 16 * 128; }
    else if (MODE == 1) { off = (lane >> 2) * 256 + (lane & 3) * 16; extent = 16 * 256; }
    else if (MODE == 2) { off = lane * 16; extent = 1024; }
    else if (MODE == 3) { off = (lane >> 3) * 256 + (lane & 7) * 16; extent = 8 * 256; }
    else { off = (lane >> 2) * 1152 + (lane & 3) * 16; extent = 16 * 1152; }
    unsigned pos = ((blockIdx.x * 4 + wave) * 7919u * extent) % footprint;
    u32x4 acc = {0, 0, 0, 0};
    if (MODE == 5) {
        // tile = 9 instructions per wave (4 waves x 9 KiB = 36 KiB of useful bytes per half), halves one after the other
        for (int it = 0; it < iters / 18; ++it) {
            const unsigned tile = pos;
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int g = 0; g < 9; ++g) {
                    unsigned p = tile + g * extent;
                    if (p >= footprint) p -= footprint;
                    dma(src + p + off + h * 64, my + ((g + h) & (SLOTS - 1)) * 1024);
                }
                asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
            }
            pos += 9 * extent * 4;
            if (pos >= footprint) pos -= footprint;
        }
    } else {
        for (int it = 0; it < iters; ++it) {
            if (MODE == 7) {
                // plain loads, eight per lane in flight (the compiler places the waits)
                if ((it & 7) == 0) {
                    u32x4 v[8];
                    unsigned q = pos;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        v[j] = *reinterpret_cast<const u32x4*>(src + q + off);
                        q += extent * 4;
                        if (q >= footprint) q -= footprint;
                    }
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc += v[j];
                }
            } else {
                dma(src + pos + off, my + (it & (SLOTS - 1)) * 1024);
                if (MODE == 4) { dma(src + pos + off + 64, my + ((it + 1) & (SLOTS - 1)) * 1024); ++it; }
                if ((it & 3) == 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            }
            pos += extent * 4;                 // (the four waves of a workgroup walk interleaved regions)
            if (pos >= footprint) pos -= footprint;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (MODE == 7 && (acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[0] = 1;
    if (lds[tid] == 77 && sink[1] == 99) sink[2] = 1;      // keep the LDS writes observable
}

template <int MODE>
void run(const char* src, unsigned fp, unsigned* sink, int blocks, int iters) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(a, 0);
        hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(256), 0, 0, src, fp, iters, sink);
        hipEventRecord(b, 0);
        hipEventSynchronize(b);
        float ms = 0;
        hipEventElapsedTime(&ms, a, b);
        const int eff_iters = MODE == 5 ? (iters / 18) * 18 : iters;
        const double bytes = (double)blocks * 4 * eff_iters * 1024.0;
        if (rep == 2) printf("mode %d footprint %5.1f MB: %7.3f ms  %8.1f GB/s useful  (%s)\n", MODE, fp / 1048576.0, ms, bytes / ms * 1e-6, hipGetErrorString(hipGetLastError()));
        fflush(stdout);
    }
}

int main(int argc, char** argv) {
    const int blocks = 1024, iters = 4608;
    char* src; unsigned* sink;
    hipMalloc(&src, 256u << 20); hipMalloc(&sink, 64);
    hipMemset(src, 1, 256u << 20); hipMemset(sink, 0, 64);
    const unsigned fps[3] = {2u << 20, 24u << 20, 192u << 20};      // inside one XCD's L2 / inside the aggregate L2 + MALL / MALL + HBM
    for (unsigned fp : fps) {
        run<0>(src, fp, sink, blocks, iters); run<1>(src, fp, sink, blocks, iters); run<2>(src, fp, sink, blocks, iters); run<3>(src, fp, sink, blocks, iters);
        run<4>(src, fp, sink, blocks, iters); run<5>(src, fp, sink, blocks, iters); run<6>(src, fp, sink, blocks, iters); run<7>(src, fp, sink, blocks, iters);
    }
    return 0;
}
