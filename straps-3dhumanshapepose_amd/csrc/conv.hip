// conv.hip -- implicit-GEMM convolution over NHWC fp32 on the exact-fp32 MFMA (v_mfma_f32_32x32x2_f32)
// (replaces nn.Conv2d 3x3 / 1x1 + the BN/ReLU/residual passes of models/resnet.py:61-77, 101-121, and the
// data-gradient half of their backward).
//
//   GEMM view:  M = output pixels,  N = output channels,  K = taps x Cin  (tap-major, channel-minor).
//   A operand = im2col rows gathered on the fly (each 32-channel K chunk of one tap is 128 contiguous bytes of one
//   source pixel -> 8 lanes x float4), B operand = weights repacked [n][tap][cin].
//   Both are copied global -> LDS by the LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write pass),
//   double-buffered with one barrier per K chunk, and read back as ds_read_b128 fragments; an XOR swizzle of the
//   16-byte K groups keeps the unpadded 128-byte rows bank-conflict free.  K is permuted inside each group of 8 (lane half h takes
//   k = 8g+4h..+3) so one b128 read feeds four MFMAs; A and B use the same permutation.
//   The taps are a small table (weight tap index, source-row offset, source-col offset), which lets ONE kernel run
//     * the forward conv (all R*S taps, source pixel = out*stride + tap - pad), and
//     * the data gradient of a stride-2 conv as four output-parity classes, each with only the taps that hit a
//       non-zero position of the (conceptually zero-dilated) dy -- no multiply-by-zero MFMA work.
//   Epilogue (C layout: lane = output channel, reg = pixel row): BN scale/shift, residual/addend, ReLU fused;
//   raw output + per-channel (sum, sumsq) partials in training mode.
//   Block ids are remapped so that consecutive tiles (same A rows / neighbouring halos) share an XCD's L2.
#include "conv_igemm.h"

namespace {


constexpr int LS = 36;   // LDS row stride in floats (32 data + 4 pad)

// diagnostic trace (tile_cfg bit 5 + straps_conv_trace_buffer): wave 0 of every workgroup accumulates shader-clock intervals of the
// four segments of its chunk loop -- [0] waiting for its own operand copies (s_waitcnt vmcnt(0)), [1] in the barrier, [2] issuing
// the next chunk's copies (+ the per-tap set-up), [3] fragment reads + MFMA burst -- plus [4] chunks, [5] whole kernel, [6] prologue,
// [7] epilogue, and writes them to trace[workgroup][8].  tools/igemm_trace.py.
__device__ long long* g_conv_trace = nullptr;
__device__ __forceinline__ long long clk() {
    long long t = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    return t;
}

__device__ __attribute__((aligned(16))) float k_zero16[4] = {0.f, 0.f, 0.f, 0.f};   // source of the padding pixels (LDS-DMA path)

// NS = 2 (default): operands copied global -> LDS directly (global_load_lds_dwordx4, two stages of unpadded 128-byte rows): no
//   staging registers, no ds_write pass, 32 KiB of LDS per 64x64 workgroup (5 resident per CU).  +3..10 % over NS = 0 on every
//   resnet layer shape, bit-identical results (tools/sweep_igemm_staging.py).  The DMA writes lane-linear (wave base + lane*16 B), so the bank-conflict-free layout is an XOR swizzle applied on BOTH sides: lane
//   (row r, slot c) fetches the 16-byte K group c ^ swz(r), and the fragment read of K group g at row r goes to slot g ^ swz(r).
//   Padding pixels copy from a 16-byte zero constant.
// NS = 0 (tile_cfg bit 4, kept for the A/B): operands staged global -> registers -> LDS (rows padded to 36 floats, two buffers).
// slot swizzle of row r (mod 32).  ds_read_b128 is serviced in four groups of 16 lanes -- {0-3,12-15,20-27}, {4-11,16-19,28-31}
// and the same +32 (MI355X_MICROARCH.md, LDS table) -- over 64 banks, i.e. two 128-byte rows: within a group the eight rows of
// each parity must land in eight different slots.  Bits 1,2 and 4 of the row number separate them in every group.
__device__ __forceinline__ int swz(int r) { return ((r >> 1) & 3) | ((r >> 2) & 4); }

template <int BM, int BN, int NS, bool TRACE = false>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvP p) {
    long long tr[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long t_start = 0, t_a = 0, t_b = 0;
    if constexpr (TRACE) t_start = clk();
    const ConvP::Class& c = p.cls[blockIdx.y];
    const int cMh = c.Mh, cMw = c.Mw, cM = c.M, cMT = c.MT, coah = c.oah, coaw = c.oaw, cntaps = c.ntaps;
    if ((int)blockIdx.x >= cMT * p.NT) return;                 // a smaller class of the same launch
    ClkSample clks;
    clk_begin(p, clks);
    constexpr int WTM = BM / 2, WTN = BN / 2, MI = WTM / 32, NI = WTN / 32;
    constexpr int AP = BM / 32, BP = BN / 32;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int RS_ = NS ? 32 : LS;            // LDS row stride
    constexpr int NBUF = NS ? NS : 2;
    float* As = smem;                    // [NBUF][BM][RS_]
    float* Bs = smem + NBUF * BM * RS_;  // [NBUF][BN][RS_]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int bid = xcd_remap(blockIdx.x, cMT * p.NT);
    const int nt = bid % p.NT, mt = bid / p.NT;
    const int m0 = mt * BM, n0 = nt * BN;
    const int lr = tid >> 3;
    const int lc = NS ? ((tid & 7) ^ swz(lr)) : (tid & 7);      // 16-byte K group this thread fetches

    // per-thread im2col row descriptors (AP rows of the A tile)
    // (32-bit element offsets: 64-bit integer multiplies in the per-chunk address math cost the kernel ~8 % -- the VALU
    //  work of the loads competes with the MFMA issue; the launcher checks that the tensors stay below 2^31 elements)
    int a_hi0[AP], a_wi0[AP];
    int a_base[AP];
    const int MhMw = cMh * cMw;
#pragma unroll
    for (int q = 0; q < AP; ++q) {
        const int m = m0 + lr + 32 * q;
        if (m < cM) {
            const int b = m / MhMw;
            const int rem = m - b * MhMw;
            const int ho = rem / cMw, wo = rem - ho * cMw;
            a_hi0[q] = ho * p.stride;
            a_wi0[q] = wo * p.stride;
            a_base[q] = ((b * p.H + a_hi0[q]) * p.W + a_wi0[q]) * p.Cin + lc * 4;   // element offset of the tap-(0,0) source pixel
        } else {
            a_hi0[q] = -(1 << 28);
            a_wi0[q] = 0;
            a_base[q] = 0;
        }
    }
    const float* wrow[BP];
#pragma unroll
    for (int q = 0; q < BP; ++q) wrow[q] = p.w + (long long)(n0 + lr + 32 * q) * p.wtaps * p.Cin + lc * 4;

    const int cchunks = p.Cin >> 5;
    const int nchunks = cntaps * cchunks;

    f32x4 ra[AP], rb[BP];
    auto load_tile = [&](int q) {
        const int tap = q / cchunks;
        const int c0 = (q - tap * cchunks) << 5;
        const int dh = c.tap_dh[tap], dw = c.tap_dw[tap], tw = c.tap_w[tap];
        const int toff = (dh * p.W + dw) * p.Cin + c0;            // wave-uniform: the per-lane part is one add
#pragma unroll
        for (int i = 0; i < AP; ++i) {
            const int hi = a_hi0[i] + dh, wi = a_wi0[i] + dw;
            const bool ok = (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (ok) v = *reinterpret_cast<const f32x4*>(p.x + (a_base[i] + toff));
            ra[i] = v;
        }
        const int woff = tw * p.Cin + c0;                       // wave-uniform
#pragma unroll
        for (int i = 0; i < BP; ++i) rb[i] = *reinterpret_cast<const f32x4*>(wrow[i] + woff);
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < AP; ++i) *reinterpret_cast<f32x4*>(As + (buf * BM + lr + 32 * i) * LS + lc * 4) = ra[i];
#pragma unroll
        for (int i = 0; i < BP; ++i) *reinterpret_cast<f32x4*>(Bs + (buf * BN + lr + 32 * i) * LS + lc * 4) = rb[i];
    };

    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const bool abl_a = TRACE && (((unsigned long long)g_conv_trace) & 1ull);
    // LDS-DMA copies of the NEXT chunk.  Chunks run tap-major: inside a tap the source pointers just move 32 channels on, so the
    // tap table, the halo test and the 64-bit address arithmetic are done once per tap, not once per chunk.
    const float* a_src[AP];
    int a_inc[AP];                 // 32 floats per chunk, 0 for a padding pixel (its source stays the zero constant)
    const float* b_src[BP];
    int n_tap = 0, n_cc = 0;
    // the tap table lives in three VGPRs (lane t = tap t) and is read with v_readlane: no scalar memory round trip in the loop
    const int tl = lane < 9 ? lane : 0;
    const int v_dh = c.tap_dh[tl], v_dw = c.tap_dw[tl], v_tw = c.tap_w[tl];
    const float* zsrc = k_zero16;
    asm volatile("" : "+s"(zsrc));          // (keep the constant's address in SGPRs instead of re-deriving it per tap)
    auto setup_tap = [&](int tap) {
        const int dh = __builtin_amdgcn_readlane(v_dh, tap), dw = __builtin_amdgcn_readlane(v_dw, tap), tw = __builtin_amdgcn_readlane(v_tw, tap);
        const int toff = (dh * p.W + dw) * p.Cin;
#pragma unroll
        for (int i = 0; i < AP; ++i) {
            const int hi = a_hi0[i] + dh, wi = a_wi0[i] + dw;
            const bool ok = (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
            a_src[i] = ok ? p.x + (a_base[i] + toff) : zsrc;
            a_inc[i] = ok ? 32 : 0;
        }
#pragma unroll
        for (int i = 0; i < BP; ++i) b_src[i] = wrow[i] + tw * p.Cin;
    };
    auto dma_next = [&](int stage) {
        // (TRACE build only, when the trace pointer's low bit is set: skip the A copies of two taps out of three -- wrong results,
        //  it prices what a halo-patch A operand, each input pixel copied once per channel chunk instead of once per tap, would buy)
        const bool skip_a = TRACE && abl_a && (n_tap % 3) != 0;
#pragma unroll
        for (int i = 0; i < AP; ++i) {
            if (skip_a) break;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)a_src[i],
                                             (__attribute__((address_space(3))) void*)(As + (stage * BM + 32 * i + 8 * wave_u) * 32), 16, 0, 0);
            a_src[i] += a_inc[i];
        }
#pragma unroll
        for (int i = 0; i < BP; ++i) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)b_src[i],
                                             (__attribute__((address_space(3))) void*)(Bs + (stage * BN + 32 * i + 8 * wave_u) * 32), 16, 0, 0);
            b_src[i] += 32;
        }
        if (++n_cc == cchunks) {
            n_cc = 0;
            if (++n_tap < cntaps) setup_tap(n_tap);
        }
    };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if constexpr (NS >= 2) {
        // two stages: chunk q lives in stage q & 1.  (Deeper rings that keep copies in flight across the barrier measured
        // slower -- 48 / 64 KiB of LDS leave 3 / 2 workgroups per CU instead of 5: tools/README.md.)
        if (nchunks > 0) { setup_tap(0); dma_next(0); }
        int fo[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) fo[kk] = (lane & 31) * 32 + (((kk * 2 + (lane >> 5)) ^ swz(lane & 31)) << 2);
        if constexpr (TRACE) { t_a = clk(); tr[6] = t_a - t_start; }
        for (int q = 0; q < nchunks; ++q) {
            const int stage = q & 1;
            // my copies of chunk q have landed, then everybody's have -- and every wave is done reading the other stage.
            // (raw s_barrier: __syncthreads() would be the same wait here, but the explicit count documents the protocol)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if constexpr (TRACE) { t_b = clk(); tr[0] += t_b - t_a; t_a = t_b; }
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if constexpr (TRACE) { t_b = clk(); tr[1] += t_b - t_a; t_a = t_b; }
            if (q + 1 < nchunks) dma_next(stage ^ 1);
            if constexpr (TRACE) { t_b = clk(); tr[2] += t_b - t_a; t_a = t_b; }
            const float* Ab = As + (stage * BM + wm * WTM) * 32;
            const float* Bb = Bs + (stage * BN + wn * WTN) * 32;
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                f32x4 a[MI], b[NI];
#pragma unroll
                for (int i = 0; i < MI; ++i) a[i] = *reinterpret_cast<const f32x4*>(Ab + i * 32 * 32 + fo[kk]);
#pragma unroll
                for (int j = 0; j < NI; ++j) b[j] = *reinterpret_cast<const f32x4*>(Bb + j * 32 * 32 + fo[kk]);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < MI; ++i)
#pragma unroll
                        for (int j = 0; j < NI; ++j) acc[i][j] = mfma32(a[i][e], b[j][e], acc[i][j]);
            }
            __builtin_amdgcn_s_setprio(0);
            if constexpr (TRACE) { asm volatile("s_nop 0" ::: "memory"); t_b = clk(); tr[3] += t_b - t_a; t_a = t_b; tr[4] += 1; }
        }
    } else {
    if (nchunks > 0) {
        load_tile(0);
        store_tile(0);
    }
    __syncthreads();

    const int frag_off = (lane & 31) * LS + (lane >> 5) * 4;
    for (int q = 0; q < nchunks; ++q) {
        const int buf = q & 1;
        if (q + 1 < nchunks) load_tile(q + 1);
        const float* Ab = As + (buf * BM + wm * WTM) * LS + frag_off;
        const float* Bb = Bs + (buf * BN + wn * WTN) * LS + frag_off;
        // raised priority inside the MFMA burst: the resident waves of a SIMD then finish their bursts one after another instead
        // of sharing the pipe evenly and all reaching their barrier / LDS phase together (+1.5 % measured on the step's convs)
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            f32x4 a[MI], b[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i) a[i] = *reinterpret_cast<const f32x4*>(Ab + i * 32 * LS + kk * 8);
#pragma unroll
            for (int j = 0; j < NI; ++j) b[j] = *reinterpret_cast<const f32x4*>(Bb + j * 32 * LS + kk * 8);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j) acc[i][j] = mfma32(a[i][e], b[j][e], acc[i][j]);
        }
        __builtin_amdgcn_s_setprio(0);
        if (q + 1 < nchunks) store_tile(buf ^ 1);
        __syncthreads();
    }
    }

    // ---------------- epilogue ----------------
    float s1[NI], s2[NI];
    igemm_store_rows<BM, BN>(p, c, acc, m0, n0, s1, s2);
    if constexpr (TRACE) {
        long long* trp = (long long*)(((unsigned long long)g_conv_trace) & ~1ull);
        if (trp && tid == 0 && blockIdx.y == 0) {
            const long long t_end = clk();
            tr[5] = t_end - t_start;
            tr[7] = t_end - t_a;
            for (int k = 0; k < 8; ++k) trp[(long long)blockIdx.x * 8 + k] = tr[k];
        }
    }
    igemm_store_stats<BM, BN>(p, s1, s2, mt, n0, smem);
    clk_end(p, clks);
}

// tile choice from the per-layer sweeps (tools/sweep_conv.py, tools/sweep_igemm_staging.py, B=64): the larger the tile the fewer
// operand bytes per flop go through L2 -> LDS, but a size only pays while it still yields >= 2 workgroups per CU and the
// reduction is long enough (K >= 512 / 1024) to amortise its heavier prologue / epilogue: 128x128 first (layer2: 124 vs 111 TFLOP/s),
// then 128x64 (layer3: 122 vs 113; layer1's 64-channel 3x3 layers: 98 vs 90), otherwise 64x64 whose 5 resident workgroups per CU hide
// each other's barrier / refill bubbles (layer4's 4096 pixels, the 1x1 down-sampling layers' short K).
inline void pick_tile(int cfg, long long M, int cout, int kdim, int& bm, int& bn) {
    cfg &= 15;               // bit 4 selects the register-staged operand path (A/B tools only), bit 5 the traced build
    const long long mt128 = (M + 127) / 128;
    if (cfg == 1) { bm = 128; bn = 128; }
    else if (cfg == 2) { bm = 128; bn = 64; }
    else if (cfg == 3) { bm = 64; bn = 64; }
    else if (cfg == 4) { bm = 256; bn = 64; }
    else if (cout % 128 == 0 && kdim >= 512 && mt128 * (cout / 128) >= 512) { bm = 128; bn = 128; }   // (>= 256 would give layer3 128x128 tiles: +3.5 % in isolation, -0.3 % inside the step)
    else if (kdim >= 1024 && mt128 * (cout / 64) >= 512) { bm = 128; bn = 64; }
    else if (cout == 64 && kdim >= 512 && mt128 >= 1024) { bm = 128; bn = 64; }   // layer1: only 64 output channels but 262 144 rows (round 2: 98 vs 90 TFLOP/s forward, 98 vs 94 data gradient, step -0.6 %)
    else { bm = 64; bn = 64; }
    if (cout % bn != 0) bn = 64;
}

template <int BM, int BN, int NS, bool TRACE = false>
int launch(const ConvP& p0, hipStream_t st) {
    ConvP p = p0;
    p.NT = p.Cout / BN;
    int maxblk = 0;
    for (int i = 0; i < p.ncls; ++i) {
        p.cls[i].MT = (p.cls[i].M + BM - 1) / BM;
        if (p.cls[i].MT * p.NT > maxblk) maxblk = p.cls[i].MT * p.NT;
    }
    const size_t lds = NS ? (size_t)NS * (BM + BN) * 32 * sizeof(float) : (size_t)2 * (BM + BN) * LS * sizeof(float);
    STRAPS_RAISE_LDS((conv_igemm_kernel<BM, BN, NS, TRACE>), lds, "conv_igemm_kernel");
    hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, NS, TRACE>), dim3(maxblk, p.ncls), dim3(256), lds, st, p);
    STRAPS_CHECK_LAUNCH("conv_igemm_kernel");
    return STRAPS_OK;
}

// all classes of one launch share the tile choice (made for their total size)
int dispatch(const ConvP& p, int tile_cfg, hipStream_t st) {
    int bm, bn, kdim = 0;
    long long M = 0;
    for (int i = 0; i < p.ncls; ++i) {
        M += p.cls[i].M;
        if (p.cls[i].ntaps * p.Cin > kdim) kdim = p.cls[i].ntaps * p.Cin;
    }
    pick_tile(tile_cfg, M, p.Cout, kdim, bm, bn);
    if (tile_cfg & 32) {        // diagnostic build of the LDS-DMA kernel with the shader-clock trace
        if (bm == 256) return launch<256, 64, 2, true>(p, st);
        if (bm == 128 && bn == 128) return launch<128, 128, 2, true>(p, st);
        if (bm == 128 && bn == 64) return launch<128, 64, 2, true>(p, st);
        return launch<64, 64, 2, true>(p, st);
    }
    if (tile_cfg & 16) {
        if (bm == 128 && bn == 128) return launch<128, 128, 0>(p, st);
        if (bm == 128 && bn == 64) return launch<128, 64, 0>(p, st);
        return launch<64, 64, 0>(p, st);
    }
    if (bm == 256) return launch<256, 64, 2>(p, st);
    if (bm == 128 && bn == 128) return launch<128, 128, 2>(p, st);
    if (bm == 128 && bn == 64) return launch<128, 64, 2>(p, st);
    return launch<64, 64, 2>(p, st);
}

}  // namespace

extern "C" int straps_conv_trace_buffer(long long* trace) {
    hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(g_conv_trace), &trace, sizeof(trace));
    if (e != hipSuccess) { straps_set_error("straps_conv_trace_buffer: %s", hipGetErrorString(e)); return STRAPS_EHIP; }
    return STRAPS_OK;
}

extern "C" int straps_conv_stat_blocks(int batch, int ho, int wo, int cout, int kdim, int tile_cfg) {
    int bm, bn;
    const long long M = (long long)batch * ho * wo;
    pick_tile(tile_cfg, M, cout, kdim, bm, bn);
    return (int)((M + bm - 1) / bm);
}

extern "C" int straps_conv_fwd(const float* x, const float* w, const float* scale, const float* shift, const float* residual, int relu,
                               float* y, float* stats_partial, int batch, int h, int wdt, int cin, int cout, int kh, int kw, int stride,
                               int pad, int tile_cfg, void* stream) {
    STRAPS_REQUIRE(x && w && y, "straps_conv_fwd: null pointer");
    STRAPS_REQUIRE(batch > 0 && h > 0 && wdt > 0, "straps_conv_fwd: empty input %dx%dx%d", batch, h, wdt);
    STRAPS_REQUIRE(cin % 32 == 0 && cout % 64 == 0, "straps_conv_fwd: need cin%%32==0 and cout%%64==0 (cin=%d cout=%d)", cin, cout);
    STRAPS_REQUIRE(kh >= 1 && kw >= 1 && kh * kw <= 9 && stride >= 1 && pad >= 0, "straps_conv_fwd: bad filter geometry");
    STRAPS_REQUIRE((scale == nullptr) == (shift == nullptr), "straps_conv_fwd: scale and shift must be given together");
    ConvP p;
    p.x = x; p.w = w;
    const int rc = conv_fwd_problem(p, scale, shift, residual, relu, y, stats_partial, batch, h, wdt, cin, cout, kh, kw, stride, pad);
    if (rc != STRAPS_OK) return rc;
    return dispatch(p, tile_cfg, (hipStream_t)stream);
}

// data gradient of straps_conv_fwd: dx[b][hi][wi][ci] = sum_{r,s,co} dy[b][ho][wo][co] * w[co][ci][r][s] over the
// (ho,wo) with ho*stride + r - pad == hi.  In terms of the rotated / transposed filters w'[ci][r'][s'][co]
// (r' = kh-1-r, straps_pack_conv_weight_dgrad) and pad' = kh-1-pad:  source row of tap r' is (hi - pad' + r') / stride
// when divisible.  stride 1: an ordinary conv over dy.  stride 2: one launch per output-parity class (hi%2, wi%2)
// carrying exactly the taps whose source position is integral.
extern "C" int straps_conv_dgrad(const float* dy, const float* w_crsk, const float* addend, float* dx, int batch, int h, int wdt,
                                 int cin, int cout, int kh, int kw, int stride, int pad, int tile_cfg, void* stream) {
    STRAPS_REQUIRE(dy && w_crsk && dx, "straps_conv_dgrad: null pointer");
    STRAPS_REQUIRE(cout % 32 == 0 && cin % 64 == 0, "straps_conv_dgrad: need cout%%32==0 and cin%%64==0 (cin=%d cout=%d)", cin, cout);
    STRAPS_REQUIRE(stride == 1 || stride == 2, "straps_conv_dgrad: stride must be 1 or 2");
    STRAPS_REQUIRE(kh * kw <= 9 && kh - 1 - pad >= 0 && kw - 1 - pad >= 0, "straps_conv_dgrad: unsupported filter geometry");
    hipStream_t st = (hipStream_t)stream;
    ConvP p;
    p.x = dy; p.w = w_crsk;
    const int rc = conv_dgrad_problem(p, addend, dx, batch, h, wdt, cin, cout, kh, kw, stride, pad);
    if (rc != STRAPS_OK) return rc;
    return p.ncls ? dispatch(p, tile_cfg, st) : STRAPS_OK;
}
