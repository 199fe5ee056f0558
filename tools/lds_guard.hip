// tools/lds_guard.hip -- does another kernel write into THIS kernel's LDS?   (round 4, the rasteriser's non-reproducible pixels)
//
// Workgroups of GUARD_THREADS (256) threads with `lds_bytes` of dynamic LDS fill it with a pattern, then sit on the CU re-reading it for `spins` rounds
// (s_sleep in between) and report the first words that changed: (workgroup, byte offset, value found, round) -> report[1 + 4 k ..], count in
// report[0].  Launched on one stream while the kernel under suspicion runs on another (tools/lds_guard_probe.py): a co-resident workgroup
// whose LDS writes -- ds_write or LDS-DMA -- run past its own allocation shows up here.
//     hipcc --offload-arch=gfx950 -O2 -shared -fPIC tools/lds_guard.hip -o tools/bin/liblds_guard.so
#include <hip/hip_runtime.h>
#include <stdlib.h>

__global__ __launch_bounds__(256) void lds_guard_kernel(unsigned* __restrict__ report, int lds_words, int spins, int max_reports) {
    extern __shared__ unsigned guard[];
    for (int i = threadIdx.x; i < lds_words; i += blockDim.x) guard[i] = 0xC0DE0000u ^ (unsigned)i;
    __syncthreads();
    for (int r = 0; r < spins; ++r) {
        for (int i = threadIdx.x; i < lds_words; i += blockDim.x) {
            const unsigned v = guard[i];
            if (v != (0xC0DE0000u ^ (unsigned)i)) {
                const unsigned k = atomicAdd(report, 1u);
                if ((int)k < max_reports) {
                    report[1 + 4 * k + 0] = blockIdx.x; report[1 + 4 * k + 1] = (unsigned)i * 4u; report[1 + 4 * k + 2] = v; report[1 + 4 * k + 3] = (unsigned)r;
                }
                guard[i] = 0xC0DE0000u ^ (unsigned)i;          // (re-arm)
            }
        }
        __builtin_amdgcn_s_sleep(64);
    }
}

extern "C" int lds_guard_launch(unsigned* report, int workgroups, int lds_bytes, int spins, int max_reports, void* stream) {
    static const int threads = getenv("GUARD_THREADS") ? atoi(getenv("GUARD_THREADS")) : 256;
    hipLaunchKernelGGL(lds_guard_kernel, dim3(workgroups), dim3(threads), (size_t)lds_bytes, (hipStream_t)stream, report, lds_bytes / 4, spins, max_reports);
    return (int)hipGetLastError();
}
