// Is "s_waitcnt vmcnt(0); s_barrier" enough before another wave reads what global_load_lds_dwordx4 wrote?  (The question behind the
// single-patch-buffer halo experiment of round 2, tools/README: its reload was the only LDS-DMA copy read right after its wait.)
// Every workgroup (4 waves) copies a fresh 8 KB tile per iteration into LDS by LDS-DMA, waits, passes the barrier, and each wave at once
// reads with ds_read_b128 what the NEXT wave copied and compares with the source pattern; a second barrier frees the buffer.
// mode 0: wait + barrier only; mode 1: + s_sleep after the barrier; mode 2: + a second "s_waitcnt vmcnt(0)" + barrier pair.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int CH = 512, ITERS = 64;      // 16-byte chunks per tile, iterations per workgroup

__global__ void fill(u32x4* src, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) src[i] = u32x4{(unsigned)i, (unsigned)i ^ 0x5a5a5a5au, (unsigned)i * 2654435761u, (unsigned)(i >> 9)};
}

template <int MODE>
__global__ __launch_bounds__(256) void probe(const u32x4* __restrict__ src, unsigned long long* bad) {
    __shared__ __attribute__((aligned(16))) u32x4 lds[CH];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned long long mism = 0;
    for (int it = 0; it < ITERS; ++it) {
        const long long base = ((long long)blockIdx.x * ITERS + it) * CH;
#pragma unroll
        for (int r = 0; r < CH / 256; ++r)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + base + (r * 4 + wave) * 64 + lane),
                                             (__attribute__((address_space(3))) void*)(lds + (r * 4 + wave) * 64), 16, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (MODE == 1) __builtin_amdgcn_s_sleep(4);
        if (MODE == 2) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }
#pragma unroll
        for (int r = 0; r < CH / 256; ++r) {
            const int j = (r * 4 + ((wave + 1) & 3)) * 64 + (lane ^ 21);
            const u32x4 v = lds[j];
            const long long g = base + j;
            if (v[0] != (unsigned)g || v[1] != ((unsigned)g ^ 0x5a5a5a5au) || v[2] != (unsigned)g * 2654435761u || v[3] != (unsigned)(g >> 9)) ++mism;
        }
        __syncthreads();
    }
    if (mism) atomicAdd(bad, mism);
}

int main() {
    const int blocks = 1024;
    const long long n = (long long)blocks * ITERS * CH;
    u32x4* src; unsigned long long* bad;
    hipMalloc(&src, n * sizeof(u32x4)); hipMalloc(&bad, 8);
    hipLaunchKernelGGL(fill, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, src, n);
    for (int rep = 0; rep < 3; ++rep)
        for (int mode = 0; mode < 3; ++mode) {
            hipMemset(bad, 0, 8);
            if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(blocks), dim3(256), 0, 0, src, bad);
            else if (mode == 1) hipLaunchKernelGGL(probe<1>, dim3(blocks), dim3(256), 0, 0, src, bad);
            else hipLaunchKernelGGL(probe<2>, dim3(blocks), dim3(256), 0, 0, src, bad);
            unsigned long long h = 0;
            hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost);
            printf("rep %d mode %d: %llu mismatching 16-byte reads of %lld (%s)\n", rep, mode, h, (long long)blocks * ITERS * (CH / 256) * 256,
                   hipGetErrorString(hipGetLastError()));
        }
    return 0;
}
