// tools/grid_barrier_probe.hip -- what would ONE fused IEF launch save over its chain of dependent launches?  
//
// The IEF head is a dependent chain of nine 64 x 1024 x 1024-class layers (models/ief_module.py:48-64; 3 iterations x fc1, fc2, fc3): each
// layer needs the WHOLE previous activation, its weights (7.6 MB in all) live in L2, and a layer is ~6 us of latency, not work.  A persistent
// fused kernel replaces the launch boundary between two layers by a device-wide barrier.  This probe measures exactly that exchange on the
// box at hand, with the layer's work reduced to its memory shape (every workgroup reads the previous layer's full 64 x 1024 fp32 activation
// -- 256 KB -- and writes its 32 x 32 slice of the next one):
//   (a) L dependent kernel launches, captured in a hipGraph and replayed (what the product does: straps_linear_fwd / straps_gemm_multi);
//   (b) one persistent launch with an agent-scope barrier (one atomic counter per layer, release-add + acquire-spin) between the L layers.
// It prints microseconds per layer for both, for the grid sizes the head uses (64 workgroups: one 32 x 32 tile each of a 64 x 1024 output)
// and for a full chip (256).
//
//     hipcc --offload-arch=gfx950 -O2 tools/grid_barrier_probe.hip -o tools/bin/grid_barrier_probe && tools/bin/grid_barrier_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

constexpr int ROWS = 64, COLS = 1024;

__device__ int g_light = 0;      // 1: the body reads 4 KB instead of the whole activation (isolates the boundary itself)

__device__ __forceinline__ void layer_body(const float* __restrict__ in, float* __restrict__ out, int wg, int nwg) {
    // read the whole previous activation (as every output tile of a dense layer must), write one 32 x 32 slice of the next
    float acc = 0.f;
    const int n4 = g_light ? 256 : ROWS * COLS / 4;
    for (int i = threadIdx.x; i < n4; i += blockDim.x) {
        const float4 v = reinterpret_cast<const float4*>(in)[i];
        acc += v.x + v.y + v.z + v.w;
    }
    const int per = ROWS * COLS / nwg;
    for (int i = threadIdx.x; i < per; i += blockDim.x) out[wg * per + i] = acc * 1e-9f + (float)i;
}

__global__ __launch_bounds__(256) void layer_kernel(const float* in, float* out) { layer_body(in, out, blockIdx.x, gridDim.x); }

__global__ __launch_bounds__(256) void fused_kernel(float* a, float* b, unsigned* counters, int layers) {
    const int nwg = gridDim.x;
    for (int l = 0; l < layers; ++l) {
        layer_body((l & 1) ? b : a, (l & 1) ? a : b, blockIdx.x, nwg);
        // device-wide barrier: every workgroup's stores of this layer visible to every workgroup's loads of the next
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(counters + l, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            while (__hip_atomic_load(counters + l, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)nwg) __builtin_amdgcn_s_sleep(1);
        }
        __syncthreads();
    }
}

int main() {
    const int L = 9, REPS = 200;
    float *a, *b;
    unsigned* counters;
    CK(hipMalloc(&a, ROWS * COLS * 4)); CK(hipMalloc(&b, ROWS * COLS * 4)); CK(hipMalloc(&counters, L * 4));
    CK(hipMemset(a, 0, ROWS * COLS * 4)); CK(hipMemset(b, 0, ROWS * COLS * 4));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int light = 0; light < 2; ++light)
    for (int nwg : {64, 256}) {
        CK(hipMemcpyToSymbol(HIP_SYMBOL(g_light), &light, sizeof(int)));
        printf("%s  ", light ? "[4 KB body]   " : "[256 KB body] ");
        // (a) chain of launches in a replayed graph
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int l = 0; l < L; ++l) hipLaunchKernelGGL(layer_kernel, dim3(nwg), dim3(256), 0, s, (l & 1) ? b : a, (l & 1) ? a : b);
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int i = 0; i < 20; ++i) CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < REPS; ++i) CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(e1, s));
        CK(hipStreamSynchronize(s));
        float ms_graph; CK(hipEventElapsedTime(&ms_graph, e0, e1));
        // (a') the same chain as plain back-to-back launches
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < REPS; ++i)
            for (int l = 0; l < L; ++l) hipLaunchKernelGGL(layer_kernel, dim3(nwg), dim3(256), 0, s, (l & 1) ? b : a, (l & 1) ? a : b);
        CK(hipEventRecord(e1, s));
        CK(hipStreamSynchronize(s));
        float ms_eager; CK(hipEventElapsedTime(&ms_eager, e0, e1));
        // (b) one persistent launch with L - 1 barriers (+ one at the end, as a consumer would need)
        float ms_fused = 0.f;
        for (int i = 0; i < 20 + REPS; ++i) {
            CK(hipMemsetAsync(counters, 0, L * 4, s));
            if (i == 20) CK(hipEventRecord(e0, s));
            hipLaunchKernelGGL(fused_kernel, dim3(nwg), dim3(256), 0, s, a, b, counters, L);
        }
        CK(hipEventRecord(e1, s));
        CK(hipStreamSynchronize(s));
        CK(hipEventElapsedTime(&ms_fused, e0, e1));
        printf("%3d workgroups, %d layers: graph-replayed chain %.2f us per layer (%.1f us per chain), eager chain %.2f us per layer, "
               "ONE persistent launch with agent-scope barriers %.2f us per layer (%.1f us per chain incl. its launch and the counter clear)\n",
               nwg, L, ms_graph * 1e3 / REPS / L, ms_graph * 1e3 / REPS, ms_eager * 1e3 / REPS / L, ms_fused * 1e3 / REPS / L, ms_fused * 1e3 / REPS);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    return 0;
}
