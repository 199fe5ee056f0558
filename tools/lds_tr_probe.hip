// tools/lds_tr_probe.hip -- what does a wave-wide LDS read cost the CU's LDS port, by kind?  256 workgroups of 4 or 8 waves (one per CU), every wave
// issues READS reads per iteration of one kind at conflict-free addresses and folds them into a register:
//   kind 0  ds_read_b128 (16 B per lane, 1 KiB per instruction)
//   kind 1  ds_read_b64  ( 8 B per lane, 512 B)
//   kind 2  ds_read_b64_tr_b16 (the transpose read of the weight-gradient kernels: 8 B per lane, 512 B)
// Prints cycles of the CU's LDS port per instruction (= workgroup time x clock / (waves x reads)).  Standalone: hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef short short4w __attribute__((ext_vector_type(4)));
typedef int int4w __attribute__((ext_vector_type(4)));
typedef int int2w __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ __launch_bounds__(512) void probe(int* out, unsigned long long* clk, int iters) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 65536 / 4; i += blockDim.x) reinterpret_cast<int*>(lds)[i] = i * 2654435761u;
    __syncthreads();
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), w0 = __builtin_amdgcn_s_memrealtime();
    int acc = 0;
    const unsigned base = (unsigned)(size_t)(lds) + wave * 4096 + lane * (KIND == 0 ? 16 : 8);      // (LDS aperture: the low 32 bits are the LDS address)
    for (int it = 0; it < iters; ++it) {
        const unsigned a = base + (it & 1) * 32768;
        // 48 reads in flight in groups of 8 (register budget), inline assembly: the compiler can neither merge nor hoist them
#pragma unroll
        for (int g = 0; g < 6; ++g) {
            if (KIND == 0) {
                int4w v0, v1, v2, v3, v4, v5, v6, v7;
                asm volatile("ds_read_b128 %0, %8\n ds_read_b128 %1, %8 offset:1024\n ds_read_b128 %2, %8 offset:2048\n ds_read_b128 %3, %8 offset:3072\n"
                             "ds_read_b128 %4, %8\n ds_read_b128 %5, %8 offset:1024\n ds_read_b128 %6, %8 offset:2048\n ds_read_b128 %7, %8 offset:3072\n s_waitcnt lgkmcnt(0)"
                             : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3), "=v"(v4), "=v"(v5), "=v"(v6), "=v"(v7) : "v"(a) : "memory");
                acc ^= v0[0] ^ v1[1] ^ v2[2] ^ v3[3] ^ v4[0] ^ v5[1] ^ v6[2] ^ v7[3];
            } else if (KIND == 1) {
                int2w v0, v1, v2, v3, v4, v5, v6, v7;
                asm volatile("ds_read_b64 %0, %8\n ds_read_b64 %1, %8 offset:1024\n ds_read_b64 %2, %8 offset:2048\n ds_read_b64 %3, %8 offset:3072\n"
                             "ds_read_b64 %4, %8\n ds_read_b64 %5, %8 offset:1024\n ds_read_b64 %6, %8 offset:2048\n ds_read_b64 %7, %8 offset:3072\n s_waitcnt lgkmcnt(0)"
                             : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3), "=v"(v4), "=v"(v5), "=v"(v6), "=v"(v7) : "v"(a) : "memory");
                acc ^= v0[0] ^ v1[1] ^ v2[0] ^ v3[1] ^ v4[0] ^ v5[1] ^ v6[0] ^ v7[1];
            } else {
                int2w v0, v1, v2, v3, v4, v5, v6, v7;
                asm volatile("ds_read_b64_tr_b16 %0, %8\n ds_read_b64_tr_b16 %1, %8 offset:1024\n ds_read_b64_tr_b16 %2, %8 offset:2048\n ds_read_b64_tr_b16 %3, %8 offset:3072\n"
                             "ds_read_b64_tr_b16 %4, %8\n ds_read_b64_tr_b16 %5, %8 offset:1024\n ds_read_b64_tr_b16 %6, %8 offset:2048\n ds_read_b64_tr_b16 %7, %8 offset:3072\n s_waitcnt lgkmcnt(0)"
                             : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3), "=v"(v4), "=v"(v5), "=v"(v6), "=v"(v7) : "v"(a) : "memory");
                acc ^= v0[0] ^ v1[1] ^ v2[0] ^ v3[1] ^ v4[0] ^ v5[1] ^ v6[0] ^ v7[1];
            }
        }
    }
    out[blockIdx.x * blockDim.x + tid] = acc;
    if (blockIdx.x == 0 && tid == 0) {
        clk[0] = __builtin_amdgcn_s_memtime() - c0;
        clk[1] = __builtin_amdgcn_s_memrealtime() - w0;
    }
}

template <int KIND>
void run(int* out, unsigned long long* clk, int khz, int waves) {
    const int iters = 4000;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(a, 0);
        hipLaunchKernelGGL(probe<KIND>, dim3(256), dim3(64 * waves), 65536, 0, out, clk, iters);
        hipEventRecord(b, 0);
        hipEventSynchronize(b);
        hipEventElapsedTime(&ms, a, b);
    }
    unsigned long long h[2];
    hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    const double mhz = (double)h[0] / (double)h[1] * khz / 1e3;
    const double cyc = ms * 1e-3 * mhz * 1e6 / ((double)iters * 48 * waves);
    printf("kind %d, %d waves per CU: %6.2f LDS-port cycles per wave-instruction at %5.0f MHz (%s)   [kernel %.3f ms, workgroup 0: %llu shader ticks]\n", KIND, waves, cyc, mhz,
           hipGetErrorString(hipGetLastError()), ms, h[0]);
    fflush(stdout);
}

int main() {
    int* out; unsigned long long* clk;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&clk, 16);
    int khz = 0;
    hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, 0);
    hipFuncSetAttribute((const void*)probe<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute((const void*)probe<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute((const void*)probe<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    for (int waves = 4; waves <= 8; waves += 4) {
        run<0>(out, clk, khz, waves); run<1>(out, clk, khz, waves); run<2>(out, clk, khz, waves);
    }
    return 0;
}
