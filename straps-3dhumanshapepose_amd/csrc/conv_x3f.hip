// conv_x3f.hip -- the 1x1 convolutions of the bottleneck units (models/resnet.py:34-36 conv1x1, :80-121 Bottleneck) and their data gradients as a
// plain GEMM whose A operand is the fp32 TENSOR ITSELF (round 6).
//
//   The implicit-GEMM kernels of conv_x3.hip read every operand as three bf16 planes (6 B per element) that the producing kernel had to write
//   next to -- or instead of -- the fp32 tensor (10 or 6 B per element), and the BatchNorm + ReLU between two convolutions is a pass of its own
//   (read 4, write 6).  For the 1x1 layers of resnet50 that is most of the bytes they move, and they are byte-bound (DESIGN section 9.5:
//   0.06-0.28 of the matrix pipe).  Here the A operand is read ONCE from the fp32 tensor with 16-byte loads (8 lanes = one 128-byte line of a
//   pixel's 32-channel chunk), the producer's BatchNorm scale / shift (+ ReLU) is applied to it in registers (`a_scale`, `a_shift`: the
//   normalised activation is never materialised -- neither as fp32 nor as planes), the value is split into its three bf16 parts with the
//   hardware conversion (v_cvt_pk_bf16_f32: round to nearest even, the same planes as split3 of common.h) and written into the LDS image the
//   fragment reads of conv_x3.hip expect ([stage][plane][row][32 bf16], 16-byte K groups XOR-swizzled by row).  Everything behind that --
//   the six-product MFMA block, the weights' LDS-DMA ring (weights stay planes: they are split once per step by the batched pack), the
//   look-ahead epilogue with its fused statistics / BatchNorm-backward sums (conv_igemm.h) -- is the same code as the plane kernel's, in the
//   same order: on the same values the two kernels give the same bits (tests/test_gpu_conv_x3f.py).
//
//   Single-tap problems only (1x1 filters without padding, any stride; the live parity class of a stride-2 1x1 data gradient): a row of A is
//   one pixel, no halo, no zero rows.  The conversion of chunk q + 1 is done by the wave behind the MFMAs of chunk q (its loads are issued in
//   front of them); with two workgroups per CU the other workgroup's matrix work runs beside it.
#include "conv_igemm.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x16 mfma_bf16(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ int swz3(int r) { return (r >> 3) & 3; }      // (conv_x3.hip: slot swizzle of a 64-byte row)

// two fp32 values -> their three bf16 parts, packed (low half = first value).  v_cvt_pk_bf16_f32 rounds to nearest even like bf16_rn
// (common.h); the parts are expanded back with one shift / one mask per value; an infinite value keeps its leading part and zero residues
// (as split3 does: inf - inf would put a NaN into the low planes).
__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {
    const f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ void split3_pair(float a, float b, unsigned& q1, unsigned& q2, unsigned& q3) {
    q1 = cvt_pk_bf16(a, b);
    float ra = a - __uint_as_float(q1 << 16), rb = b - __uint_as_float(q1 & 0xffff0000u);
    ra = (__float_as_uint(a) & 0x7fffffffu) == 0x7f800000u ? 0.f : ra;
    rb = (__float_as_uint(b) & 0x7fffffffu) == 0x7f800000u ? 0.f : rb;
    q2 = cvt_pk_bf16(ra, rb);
    const float sa = ra - __uint_as_float(q2 << 16), sb = rb - __uint_as_float(q2 & 0xffff0000u);
    q3 = cvt_pk_bf16(sa, sb);
}

// LDS store of the converted operand through inline assembly: a store the compiler can see is ordered behind EVERY outstanding LDS-DMA copy
// (s_waitcnt vmcnt(0) in front of each ds_write: it cannot know that the weights' stages and the A image are different bytes), which would
// serialise the conversion behind the weight copies issued between the MFMAs before it.  The wave's own `s_waitcnt lgkmcnt(0)` in front of
// the next chunk's barrier covers these stores.
template <int OFF>
__device__ __forceinline__ void lds_write_b64(unsigned addr, const u32x2& v) {
    asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFF) : "memory");
}

// tools build: STRAPS_X3F_ABL = 1 no result stores, 2 no A loads (constants instead), 4 no MFMAs, 8 no statistics partials (any sum; wrong results)
inline void x3f_ablate(ConvP& p) {
    p.abl = STRAPS_TOOL_ENV_INT("STRAPS_X3F_ABL", 0);
    if (p.abl & 1) p.y = nullptr;
    if (p.abl & 8) p.stats = nullptr;
}

// ABN: the producer's BatchNorm (+ ReLU) in the operand path (p.a_scale / p.a_shift / p.a_relu)
template <int BM, int BN, int WGM, int WGN, int NST, bool ABN, int EPI>
__global__ __launch_bounds__(64 * WGM * WGN, 2) void conv_igemm_x3f_kernel(ConvP p) {
    const ConvP::Class& c = p.cls[blockIdx.y];
    const int cMh = c.Mh, cMw = c.Mw, cM = c.M, cMT = c.MT, cntaps = c.ntaps;
    if ((int)blockIdx.x >= cMT * p.NT) return;                 // a smaller class of the same launch
    ClkSample clk;
    clk_begin(p, clk);
    constexpr int NW = WGM * WGN, NTH = 64 * NW, RPP = 16 * NW;   // waves; threads; weight rows per copy pass (4 lanes x 16 bytes per 64-byte row)
    constexpr int WTM = BM / WGM, WTN = BN / WGN, MI = WTM / 32, NI = WTN / 32;
    constexpr int BP = BN / RPP;                               // weight copy passes per plane
    constexpr int ARP = NTH / 8, AP = BM / ARP;                // A rows per load pass (8 lanes x 16 bytes = the 128 bytes of a pixel's chunk); passes
    static_assert(BN % RPP == 0 && BM % ARP == 0 && WTM % 32 == 0 && WTN % 32 == 0 && ARP % 32 == 0, "tile / wave grid mismatch");
    constexpr int NPB = 3 * BP;                                // LDS-DMA instructions per thread and chunk (weights)
    constexpr int NMFMA = 2 * 6 * MI * NI;
    constexpr int GAP = NMFMA / NPB > 0 ? NMFMA / NPB : 1;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    u16* As = reinterpret_cast<u16*>(smem);       // [NST stages][3 planes][BM][32]
    u16* Bs = As + NST * 3 * BM * 32;             // [NST stages][3 planes][BN][32]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int bid = xcd_remap(blockIdx.x, cMT * p.NT);
    const int nt = bid % p.NT, mt = bid / p.NT;
    const int m0 = mt * BM, n0 = nt * BN;
    const u16* wg = reinterpret_cast<const u16*>(p.w);
    IgemmEpilogue<BM, BN, WGM, WGN> ep;
    ep.init(p, c, m0, n0);

    // ---- A: thread (ar, aj) owns channels 4 aj .. 4 aj + 3 of the chunk in rows ar + ARP i.  Element offset of the row's pixel (32-bit: the
    //      launcher checks the tensor sizes); rows behind the problem's last read its last pixel (their results are never stored)
    const int ar = tid >> 3, aj = tid & 7;
    int a_off[AP];
#pragma unroll
    for (int i = 0; i < AP; ++i) {
        int m = m0 + ar + ARP * i;
        m = m < cM ? m : cM - 1;
        const int MhMw = cMh * cMw;
        const int b = m / MhMw, rem = m - b * MhMw;
        const int ho = rem / cMw, wo = rem - ho * cMw;
        a_off[i] = ((b * p.H + ho * p.stride + c.tap_dh[0]) * p.W + wo * p.stride + c.tap_dw[0]) * p.Cin + aj * 4;
    }
    // LDS position of the thread's 8 bytes in a row of a plane: 16-byte K group aj >> 1 in slot (aj >> 1) ^ swz3(row), half aj & 1 (ARP % 32 == 0:
    // the swizzle of row ar + ARP i is the swizzle of ar)
    const int a_wo = ar * 32 + (((aj >> 1) ^ swz3(ar)) << 3) + (aj & 1) * 4;
    const float* xg = p.x;
    const bool arelu = p.a_relu != 0;
    static_assert((2 * BM + ARP * (AP - 1)) * 64 + 8 < 65536, "ds_write offsets");

    // ---- B: the weights' planes through the LDS-DMA, as in conv_igemm_x3_kernel
    const int lr = tid >> 2;                                   // row of the RPP-row pass this thread copies
    const int lc = (tid & 3) ^ swz3(lr);                       // 16-byte K group it fetches for its slot tid & 3
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int cchunks = p.Cin >> 5;
    const int nchunks = cntaps > 0 ? cchunks : 0;              // (single tap)
    const int b_cstep = p.Cout * 32;                           // one channel chunk on: elements
    const u16* b_src[BP];
#pragma unroll
    for (int q = 0; q < BP; ++q) b_src[q] = wg + (n0 + lr + RPP * q) * 32 + lc * 8 + (long long)(cntaps > 0 ? c.tap_w[0] : 0) * cchunks * b_cstep;
    auto piece = [&](int stage, int idx) {
        const int plane = idx / BP, r = idx % BP;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b_src[r] + plane * p.wps),
                                         (__attribute__((address_space(3))) void*)(Bs + ((stage * 3 + plane) * BN + RPP * r + 16 * wave_u) * 32), 16, 0, 0);
    };
    auto advance_b = [&]() {
#pragma unroll
        for (int i = 0; i < BP; ++i) b_src[i] += b_cstep;
    };

    f32x4 areg[AP], bsc, bsh;
    int a_cc = 0;                                               // chunk the next load_a fetches
    auto load_a = [&]() {
        if constexpr (ABN) {      // (in front of the tensor loads: the in-order counter retires them first)
            bsc = *reinterpret_cast<const f32x4*>(p.a_scale + a_cc * 32 + aj * 4);
            bsh = *reinterpret_cast<const f32x4*>(p.a_shift + a_cc * 32 + aj * 4);
        }
#ifdef STRAPS_TOOLS
        if (p.abl & 2) {
#pragma unroll
            for (int i = 0; i < AP; ++i) areg[i] = f32x4{1.f, 2.f, 3.f, 4.f};
        } else
#endif
#pragma unroll
        for (int i = 0; i < AP; ++i) areg[i] = *reinterpret_cast<const f32x4*>(xg + a_off[i] + a_cc * 32);
        ++a_cc;
        __builtin_amdgcn_sched_barrier(0);      // (the machine scheduler otherwise sinks the loads to their use, behind the matrix work)
    };
    auto convert_store = [&](int stage) {
        const unsigned dst = (unsigned)(stage * 3 * BM * 32 + a_wo) * 2;      // byte address in LDS (As is the base of the dynamic allocation)
        epi_static_for<AP>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            f32x4 v = areg[i];
            if constexpr (ABN) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = fmaf(v[e], bsc[e], bsh[e]);
                    v[e] = arelu ? fmaxf(v[e], 0.f) : v[e];
                }
            }
            unsigned l1, l2, l3, h1, h2, h3;
            split3_pair(v[0], v[1], l1, l2, l3);
            split3_pair(v[2], v[3], h1, h2, h3);
            const u32x2 q1 = {l1, h1}, q2 = {l2, h2}, q3 = {l3, h3};
            lds_write_b64<(0 * BM + ARP * i) * 64>(dst, q1);
            lds_write_b64<(1 * BM + ARP * i) * 64>(dst, q2);
            lds_write_b64<(2 * BM + ARP * i) * 64>(dst, q3);
        });
    };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // chunk q lives in stage q % NST; the weight copies run NST - 1 chunks ahead of the matrix work, the A conversion one chunk ahead.
    // Prologue: chunk 0's A loads first (the counter retires in order: they can then be awaited with the weight copies still in flight)
    if (nchunks > 0) load_a();
    int b_issued = 0;
#pragma unroll
    for (int s = 0; s < NST - 1; ++s)
        if (s < nchunks) {
#pragma unroll
            for (int idx = 0; idx < NPB; ++idx) piece(s, idx);
            advance_b();
            ++b_issued;
        }
    if (nchunks > 0) {
        if (NST == 2 || b_issued == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPB) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 1) * NPB) : "memory");
        convert_store(0);
    }
    int fo[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) fo[kk] = (lane & 31) * 32 + (((kk * 2 + (lane >> 5)) ^ swz3(lane & 31)) << 3);

    // plane pairs of the six products, smallest terms first
    constexpr int TA[6] = {1, 0, 2, 0, 1, 0};
    constexpr int TB[6] = {1, 2, 0, 1, 0, 0};

    // MORE_B: weight chunk q + NST - 1 exists and is issued between this chunk's MFMAs into the stage chunk q - 1 was read from.
    // MORE_A: chunk q + 1 exists: its A loads are issued in front of this chunk's MFMAs and converted into stage q + 1 behind them.
    // INFLIGHT: weight copies of younger chunks that may stay outstanding while this chunk's are awaited (the counter retires in order).
    auto chunk = [&](int stage, int nstage, int astage, auto more_b_c, auto more_a_c, auto inflight_c, auto last_c) {
        constexpr bool MORE_B = decltype(more_b_c)::value, MORE_A = decltype(more_a_c)::value;
        constexpr int INFLIGHT = decltype(inflight_c)::value;
        // my weight copies of this chunk have landed and my part of its A image is written, then everybody's -- and every wave is done
        // reading the stages refilled next
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(INFLIGHT) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        // the last chunk: no copy wait follows -- the epilogue's first operands are fetched under this chunk's matrix work (conv_igemm.h)
        if constexpr (decltype(last_c)::value && EPI == 0) ep.prefetch();
        if constexpr (MORE_A) load_a();
        const u16* Ab = As + (stage * 3 * BM + wm * WTM) * 32;
        const u16* Bb = Bs + (stage * 3 * BN + wn * WTN) * 32;
        int cnt = 0;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 a[MI][3], b[NI][3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
                for (int i = 0; i < MI; ++i) a[i][pl] = *reinterpret_cast<const bf16x8*>(Ab + (pl * BM + i * 32) * 32 + fo[kk]);
#pragma unroll
                for (int j = 0; j < NI; ++j) b[j][pl] = *reinterpret_cast<const bf16x8*>(Bb + (pl * BN + j * 32) * 32 + fo[kk]);
            }
#pragma unroll
            for (int t = 0; t < 6; ++t)
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j) {
#ifdef STRAPS_TOOLS
                        if (!(p.abl & 4))
#endif
                        acc[i][j] = mfma_bf16(a[i][TA[t]], b[j][TB[t]], acc[i][j]);
                        if (MORE_B && cnt % GAP == GAP - 1 && cnt / GAP < NPB) piece(nstage, cnt / GAP);
                        ++cnt;
                    }
            // (the conversion of the next chunk's A -- BatchNorm, split, LDS stores -- may be scheduled between the MFMAs of the second k step, not
            //  earlier: hoisted to the top of the chunk it would wait for its loads in front of all the matrix work)
            if (MORE_A && kk == 0) __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (MORE_B) {
#pragma unroll
            for (int idx = NMFMA / GAP; idx < NPB; ++idx) piece(nstage, idx);
            advance_b();
        }
        if constexpr (MORE_A) {
            // the A loads are older than this chunk's weight copies: they have landed once at most those are outstanding
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(MORE_B ? NPB : 0) : "memory");
            convert_store(astage);
        }
    };
    int stage = 0, nstage = NST - 1, astage = 1 % NST;
    auto next = [&]() {
        stage = stage + 1 == NST ? 0 : stage + 1;
        nstage = nstage + 1 == NST ? 0 : nstage + 1;
        astage = astage + 1 == NST ? 0 : astage + 1;
    };
    int q = 0;
    for (; q + NST - 1 < nchunks; ++q) {
        chunk(stage, nstage, astage, std::true_type{}, std::true_type{}, std::integral_constant<int, (NST - 2) * NPB>{}, std::false_type{});
        next();
    }
    if constexpr (NST == 3) {
        if (q + 1 < nchunks) { chunk(stage, nstage, astage, std::false_type{}, std::true_type{}, std::integral_constant<int, NPB>{}, std::false_type{}); next(); ++q; }
    }
    if (q < nchunks) chunk(stage, nstage, astage, std::false_type{}, std::false_type{}, std::integral_constant<int, 0>{}, std::true_type{});

    float s1[NI], s2[NI];
    double bd1[NI], bd2[NI];
    if constexpr (EPI == 1) {
        lean_epilogue_fwd<BM, BN, WGM, WGN>(p, acc, m0, n0, cM, s1, s2);
        igemm_store_stats<BM, BN, WGM, WGN>(p, s1, s2, mt, n0, smem);
    } else if constexpr (EPI == 2) {
        lean_epilogue_dgrad<BM, BN, WGM, WGN>(p, c, acc, m0, n0, bd1, bd2);
        igemm_store_bnr<BM, BN, WGM, WGN>(p, bd1, bd2, p.bnr_base[blockIdx.y] + mt, n0, smem);
    } else {
        ep.finish(p, c, acc, s1, s2, bd1, bd2);
        igemm_store_stats<BM, BN, WGM, WGN>(p, s1, s2, mt, n0, smem);
        igemm_store_bnr<BM, BN, WGM, WGN>(p, bd1, bd2, p.bnr_base[blockIdx.y] + mt, n0, smem);
    }
    clk_end(p, clk);
}

// ---- the STREAMING form for the byte-bound layers (a small reduction extent against a long pixel axis: layer1 / layer2 of resnet50) ----------
// What the tile kernel above costs on those: K = 64 is two chunks, so a workgroup lives for two exposed memory latencies (chunk 0's loads, chunk
// 1's loads), a few hundred MFMA cycles and its stores -- ~10 us per workgroup at ~2 us of work, with two or three of them per CU to overlap
// (tools/sweep_conv_x3f_cold.py: 82 us for the 64 -> 256 forward on 32 x 64 x 64 pixels, whose bytes are 34 us at 5 TB/s).
// Here a workgroup is PERSISTENT: it owns ONE tile of the output channels and walks over M tiles; its A operand is a continuous stream of
// (tile, chunk) steps that runs TWO steps ahead of the matrix work in two register sets -- the loads of step s + 2 are issued during step s and
// converted during step s + 1 -- straight across tile boundaries: the next tile's first chunks are in flight while a tile's epilogue stores
// its results, and after the first step no memory latency is exposed.
//   BRES: the weight tile -- ALL of K for BN output channels, K x BN <= 16 384: 96 KB of planes -- stays resident in LDS (one LDS-DMA sweep at the
//         start, no weight traffic afterwards; one workgroup per CU).  With BN = all output channels (64 -> 256) every A element is read and
//         converted exactly once.
//   else: the weights go through a two-stage ring (one chunk per step, copied under the previous step's matrix work); with 128 x 64 tiles two
//         workgroups share a CU.
// One class only (a stride-1 problem, or a forward 1x1 of any stride); K a multiple of 64 (the two register sets alternate with the chunk
// parity, and a tile's epilogue -- the tile kernel's, with the same partial blocks as its 128-row tiles -- is instantiated once, behind an odd chunk).
template <int BM, int BN, int WGM, int WGN, bool ABN, bool BRES, int EPI>
__global__ __launch_bounds__(64 * WGM * WGN, 2) void conv1x1_stream_kernel(ConvP p) {
    static_assert(EPI == 1 || EPI == 2, "the streaming kernel carries the lean epilogues only");
    const ConvP::Class& c = p.cls[0];
    const int cMh = c.Mh, cMw = c.Mw, cM = c.M, cMT = c.MT;
    ClkSample clk;
    clk_begin(p, clk);
    constexpr int NW = WGM * WGN, NTH = 64 * NW, RPP = 16 * NW;
    constexpr int WTM = BM / WGM, WTN = BN / WGN, MI = WTM / 32, NI = WTN / 32;
    constexpr int BP = BN / RPP;
    constexpr int ARP = NTH / 8, AP = BM / ARP;
    static_assert(BN % RPP == 0 && BM % ARP == 0 && WTM % 32 == 0 && WTN % 32 == 0 && ARP % 32 == 0 && AP >= 1, "tile / wave grid mismatch");
    static_assert((2 * BM + ARP * (AP - 1)) * 64 + 8 < 65536, "ds_write offsets");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int KC = p.Cin >> 5;                    // chunks of the reduction extent (even)
    u16* As = reinterpret_cast<u16*>(smem);       // [2 stages][3 planes][BM][32]
    u16* Ws = As + 2 * 3 * BM * 32;               // BRES: [3 planes][KC][BN][32]; else [3 planes][2 stages][BN][32]
    const int wchunks = BRES ? KC : 2;
    float* red = reinterpret_cast<float*>(Ws + 3 * wchunks * BN * 32);      // [WGM][BN][2] doubles: the epilogue's cross-wave sums (a region of its own: the operand images stay live)
    float* bnc = red + WGM * BN * 4;              // [2][Cin]: the operand-path BatchNorm's scale | shift

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int nt = blockIdx.x % p.NT, n0 = nt * BN;
    const int mstep = gridDim.x / p.NT;            // (the launcher makes the grid a multiple of NT)
    const int mt0 = blockIdx.x / p.NT;
    const u16* wg = reinterpret_cast<const u16*>(p.w);
    const float* xg = p.x;
    const bool arelu = p.a_relu != 0;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);

    // ---- the weights: thread's source row / K group as in the ring kernels; chunk cc of plane pl at + (cc * Cout) * 32 + pl * wps
    const int lr = tid >> 2, lc = (tid & 3) ^ swz3(lr);
    const u16* wsrc = wg + (n0 + lr) * 32 + lc * 8 + (long long)c.tap_w[0] * KC * p.Cout * 32;
    auto w_chunk = [&](int cc, int slot) {        // chunk cc of the weights -> LDS slot `slot` (BRES: slot = cc; ring: the stage)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int r = 0; r < BP; ++r)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc + ((long long)cc * p.Cout + RPP * r) * 32 + pl * p.wps),
                                                 (__attribute__((address_space(3))) void*)(Ws + ((pl * wchunks + slot) * BN + RPP * r + 16 * wave_u) * 32), 16, 0, 0);
    };
    if constexpr (BRES) {
        for (int cc = 0; cc < KC; ++cc) w_chunk(cc, cc);
    } else {
        w_chunk(0, 0);
    }
    if constexpr (ABN) {
        for (int k = tid; k < p.Cin; k += NTH) { bnc[k] = p.a_scale[k]; bnc[p.Cin + k] = p.a_shift[k]; }
    }

    // ---- the A stream: thread (ar, aj) owns channels 4 aj .. 4 aj + 3 of a chunk in rows ar + ARP i of a tile.  Requests behind the workgroup's
    //      last tile re-read its last chunk (the in-order counter then always sees the same number of loads per step: the waits below are counts)
    const int ar = tid >> 3, aj = tid & 7;
    const int a_wo = ar * 32 + (((aj >> 1) ^ swz3(ar)) << 3) + (aj & 1) * 4;
    int a_off[AP];
    int ld_mt = mt0, ld_cc = 0;
    auto set_rows = [&](int mt_) {
#pragma unroll
        for (int i = 0; i < AP; ++i) {
            int m = mt_ * BM + ar + ARP * i;
            m = m < cM ? m : cM - 1;
            const int MhMw = cMh * cMw;
            const int b = m / MhMw, rem = m - b * MhMw;
            const int ho = rem / cMw, wo = rem - ho * cMw;
            a_off[i] = ((b * p.H + ho * p.stride + c.tap_dh[0]) * p.W + wo * p.stride + c.tap_dw[0]) * p.Cin + aj * 4;
        }
    };
    set_rows(ld_mt < cMT ? ld_mt : cMT - 1);
    f32x4 areg[2][AP];
    auto load_a = [&](auto set_c) {
        constexpr int SET = decltype(set_c)::value;
#ifdef STRAPS_TOOLS
        if (p.abl & 2) {
#pragma unroll
            for (int i = 0; i < AP; ++i) areg[SET][i] = f32x4{1.f, 2.f, 3.f, 4.f};
        } else
#endif
#pragma unroll
        for (int i = 0; i < AP; ++i) areg[SET][i] = *reinterpret_cast<const f32x4*>(xg + a_off[i] + ld_cc * 32);
        if (ld_mt < cMT && ++ld_cc == KC) {
            ld_mt += mstep;
            if (ld_mt < cMT) { ld_cc = 0; set_rows(ld_mt); } else ld_cc = KC - 1;
        }
    };
    // conversion of the chunk held in register set SET (chunk number cc of its tile: the BatchNorm constants' index) into LDS stage SET
    auto convert_store = [&](auto set_c, int cc) {
        constexpr int SET = decltype(set_c)::value;
        const unsigned dst = (unsigned)(SET * 3 * BM * 32 + a_wo) * 2;
        f32x4 bsc, bsh;
        if constexpr (ABN) {
            bsc = *reinterpret_cast<const f32x4*>(bnc + cc * 32 + aj * 4);
            bsh = *reinterpret_cast<const f32x4*>(bnc + p.Cin + cc * 32 + aj * 4);
        }
        epi_static_for<AP>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            f32x4 v = areg[SET][i];
            if constexpr (ABN) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = fmaf(v[e], bsc[e], bsh[e]);
                    v[e] = arelu ? fmaxf(v[e], 0.f) : v[e];
                }
            }
            unsigned l1, l2, l3, h1, h2, h3;
            split3_pair(v[0], v[1], l1, l2, l3);
            split3_pair(v[2], v[3], h1, h2, h3);
            const u32x2 q1 = {l1, h1}, q2 = {l2, h2}, q3 = {l3, h3};
            lds_write_b64<(0 * BM + ARP * i) * 64>(dst, q1);
            lds_write_b64<(1 * BM + ARP * i) * 64>(dst, q2);
            lds_write_b64<(2 * BM + ARP * i) * 64>(dst, q3);
        });
    };

    f32x16 acc[MI][NI];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    };
    zero_acc();
    int fo[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) fo[kk] = (lane & 31) * 32 + (((kk * 2 + (lane >> 5)) ^ swz3(lane & 31)) << 3);
    constexpr int TA[6] = {1, 0, 2, 0, 1, 0};
    constexpr int TB[6] = {1, 2, 0, 1, 0, 0};

    // prologue: steps 0 and 1 requested, the weights (and the constants) landed, step 0 converted
    load_a(std::integral_constant<int, 0>{});
    load_a(std::integral_constant<int, 1>{});
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(AP) : "memory");      // (everything older than step 1's request: the weights, the constants' loads, step 0)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                 // bnc is complete
    asm volatile("" ::: "memory");
    convert_store(std::integral_constant<int, 0>{}, 0);

    // one step = the matrix work of chunk cc (parity P) of the current tile out of A stage P.  In front of it the barrier that publishes stage P;
    // behind its first k step: the conversion of the NEXT step's chunk (register set 1 - P, requested during the previous step) into stage 1 - P,
    // then the request of the step after that into set P (free: converted during the previous step).  Ring form: this step's weight chunk was copied
    // under the previous step; the next one's copy is issued first thing (in front of the A request: the counter retires in order, so the wait
    // for a step's weights may leave the younger A request outstanding).
    auto step = [&](auto par_c, int cc) {
        constexpr int P = decltype(par_c)::value;
        if constexpr (BRES) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(AP) : "memory");      // my copies of this step's weights have landed (the A request behind them may be in flight), my part of its A image is written ...
        __builtin_amdgcn_s_barrier();                              // ... everybody's are, and every wave is done reading the stages refilled next
        asm volatile("" ::: "memory");
        if constexpr (!BRES) w_chunk(cc + 1 == KC ? 0 : cc + 1, 1 - P);
        const u16* Ab = As + (P * 3 * BM + wm * WTM) * 32;
        const u16* Bb = Ws + ((BRES ? cc : P) * BN + wn * WTN) * 32;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 a[MI][3], b[NI][3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
                for (int i = 0; i < MI; ++i) a[i][pl] = *reinterpret_cast<const bf16x8*>(Ab + (pl * BM + i * 32) * 32 + fo[kk]);
#pragma unroll
                for (int j = 0; j < NI; ++j) b[j][pl] = *reinterpret_cast<const bf16x8*>(Bb + (pl * wchunks * BN + j * 32) * 32 + fo[kk]);
            }
#pragma unroll
            for (int t = 0; t < 6; ++t)
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j) {
#ifdef STRAPS_TOOLS
                        if (!(p.abl & 4))
#endif
                        acc[i][j] = mfma_bf16(a[i][TA[t]], b[j][TB[t]], acc[i][j]);
                    }
            if (kk == 0) {
                __builtin_amdgcn_sched_barrier(0);
                convert_store(std::integral_constant<int, 1 - P>{}, cc + 1 == KC ? 0 : cc + 1);
                load_a(std::integral_constant<int, P>{});
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    for (int mt = mt0; mt < cMT; mt += mstep) {
        for (int cc = 0; cc < KC; cc += 2) {
            step(std::integral_constant<int, 0>{}, cc);
            step(std::integral_constant<int, 1>{}, cc + 1);
        }
        if constexpr (EPI == 1) {
            float s1[NI], s2[NI];
            lean_epilogue_fwd<BM, BN, WGM, WGN>(p, acc, mt * BM, n0, cM, s1, s2);
            igemm_store_stats<BM, BN, WGM, WGN>(p, s1, s2, mt, n0, red);
        } else {
            double bd1[NI], bd2[NI];
            lean_epilogue_dgrad<BM, BN, WGM, WGN>(p, c, acc, mt * BM, n0, bd1, bd2);
            igemm_store_bnr<BM, BN, WGM, WGN>(p, bd1, bd2, p.bnr_base[0] + mt, n0, red);
        }
        zero_acc();
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // (the requests and copies issued ahead of the end)
    clk_end(p, clk);
}

// which epilogue a problem takes (conv_igemm.h: lean_epilogue_choice); the tools build can switch the lean forms off for the A/B
inline int x3f_epilogue(const ConvP& p) {
    if (STRAPS_TOOL_ENV_INT("STRAPS_X3F_LEAN", 1) == 0) return 0;
    if (p.cls[0].ntaps != 1) return 0;
    return lean_epilogue_choice(p);
}

// LDS of the streaming kernel for a reduction extent of cin channels
template <int BM, int BN, int WGM, bool BRES>
size_t stream_lds_bytes(int cin) {
    return (size_t)2 * 3 * BM * 32 * 2 + (size_t)3 * (BRES ? cin / 32 : 2) * BN * 32 * 2 + (size_t)WGM * BN * 4 * 4 + (size_t)2 * cin * 4;
}

template <int BM, int BN, int WGM, int WGN, bool ABN, bool BRES, int EPI>
int launch_stream_abn(const ConvP& p0, hipStream_t st) {
    ConvP p = p0;
    p.NT = p.Cout / BN;
    p.cls[0].MT = (p.cls[0].M + BM - 1) / BM;
    p.bnr_base[0] = 0;
    const size_t lds = stream_lds_bytes<BM, BN, WGM, BRES>(p.Cin);
    STRAPS_REQUIRE(lds <= 160 * 1024, "conv1x1_stream_kernel: the operand images do not fit the LDS (cin=%d, BN=%d)", p.Cin, BN);
    // as many workgroups as stay resident (LDS decides), at most one per M tile; a multiple of the N tiles (a workgroup keeps ONE weight tile)
    const int per_cu = (int)((160 * 1024) / lds) > 2 ? 2 : (int)((160 * 1024) / lds);
    int wgs = 256 * per_cu / p.NT;
    {
        const int f = STRAPS_TOOL_ENV_INT("STRAPS_X3F_STREAM_WGS", 0);      // (tools: workgroups per N tile; a large number = one tile per workgroup, nothing persistent)
        if (f > 0) wgs = f;
    }
    if (wgs > p.cls[0].MT) wgs = p.cls[0].MT;
    if (wgs < 1) wgs = 1;
    x3f_ablate(p);
    STRAPS_RAISE_LDS((conv1x1_stream_kernel<BM, BN, WGM, WGN, ABN, BRES, EPI>), lds, "conv1x1_stream_kernel");
    hipLaunchKernelGGL((conv1x1_stream_kernel<BM, BN, WGM, WGN, ABN, BRES, EPI>), dim3(wgs * p.NT), dim3(64 * WGM * WGN), lds, st, p);
    STRAPS_CHECK_LAUNCH("conv1x1_stream_kernel");
    return STRAPS_OK;
}
template <int BM, int BN, int WGM, int WGN, bool BRES>
int launch_stream(const ConvP& p, hipStream_t st) {
    if (x3f_epilogue(p) == 2) return launch_stream_abn<BM, BN, WGM, WGN, false, BRES, 2>(p, st);
    return p.a_scale ? launch_stream_abn<BM, BN, WGM, WGN, true, BRES, 1>(p, st) : launch_stream_abn<BM, BN, WGM, WGN, false, BRES, 1>(p, st);
}

// width of the streaming kernel's output-channel tile for this problem, 0 = not eligible (several classes, a reduction extent that is not a
// multiple of 64).  256 wide with resident weights where all of K fits beside it (K = 64), else 128 / 64 wide with the weight ring.
inline int stream_bn(const ConvP& p) {
    if (p.ncls != 1 || p.cls[0].ntaps != 1 || p.Cin % 64 != 0 || x3f_epilogue(p) == 0) return 0;
    if (p.Cout % 256 == 0 && p.Cin <= 64) return 256;
    return p.Cout % 128 == 0 && p.Cin < p.Cout ? 128 : 64;
}

template <int BM, int BN, int WGM, int WGN, int NST, bool ABN, int EPI>
int launch_x3f_abn(const ConvP& p0, hipStream_t st) {
    ConvP p = p0;
    p.NT = p.Cout / BN;
    int maxblk = 0;
    int base = 0;
    for (int i = 0; i < p.ncls; ++i) {
        p.cls[i].MT = (p.cls[i].M + BM - 1) / BM;
        if (p.cls[i].MT * p.NT > maxblk) maxblk = p.cls[i].MT * p.NT;
        p.bnr_base[i] = base;                   // (BatchNorm-backward partials: one block per M tile, classes one after the other)
        base += p.cls[i].MT;
    }
    const size_t lds = (size_t)NST * 3 * (BM + BN) * 32 * sizeof(u16);
    x3f_ablate(p);
    STRAPS_RAISE_LDS((conv_igemm_x3f_kernel<BM, BN, WGM, WGN, NST, ABN, EPI>), lds, "conv_igemm_x3f_kernel");
    hipLaunchKernelGGL((conv_igemm_x3f_kernel<BM, BN, WGM, WGN, NST, ABN, EPI>), dim3(maxblk, p.ncls), dim3(64 * WGM * WGN), lds, st, p);
    STRAPS_CHECK_LAUNCH("conv_igemm_x3f_kernel");
    return STRAPS_OK;
}
template <int BM, int BN, int WGM, int WGN, int NST>
int launch_x3f(const ConvP& p, hipStream_t st) {
    switch (x3f_epilogue(p)) {
        case 1: return p.a_scale ? launch_x3f_abn<BM, BN, WGM, WGN, NST, true, 1>(p, st) : launch_x3f_abn<BM, BN, WGM, WGN, NST, false, 1>(p, st);
        case 2: return launch_x3f_abn<BM, BN, WGM, WGN, NST, false, 2>(p, st);
        default: return p.a_scale ? launch_x3f_abn<BM, BN, WGM, WGN, NST, true, 0>(p, st) : launch_x3f_abn<BM, BN, WGM, WGN, NST, false, 0>(p, st);
    }
}

// tile_cfg & 15: 0 = auto, 1 = 128x64 (4 waves, 2 stages: 72 KB of LDS, two workgroups per CU), 2 = 64x64 (4 waves, 2 stages: 48 KB, three per CU),
// 5 = the streaming kernel (persistent workgroups, 128-row
// tiles, resident weights) where the problem is eligible (stream_bn), else as 0.
// The 1x1 layers this kernel exists for are byte-bound: what counts is that a CU always has a workgroup in its load phase beside one in its
// store phase, i.e. SEVERAL workgroups per CU, not a large tile (rule below from tools/sweep_conv_x3f_cold.py, profiles/r06_x3f_cold_sweep.txt)
inline int pick_tile_x3f(int cfg, long long M, int cout, int ncls, int& bm, int& bn, int sbn = 0) {
    cfg &= 15;
    // the streaming kernel: explicitly (5), or by rule for long problems -- at least four 128-row tiles per workgroup
    if ((cfg == 5 || (cfg == 0 && M >= 4 * 128 * (256 / (cout / (sbn ? sbn : cout))))) && sbn) { bm = sbn == 256 ? 64 : 128; bn = sbn; return 5; }
    if (cfg == 5) cfg = 0;
    if (cfg == 0) {
        const long long t64 = ((M / ncls + 127) / 128) * (cout / 64);       // 128x64 tiles of a class
        cfg = t64 < 512 ? 2 : 1;
    }
    if (cfg < 1 || cfg > 2) cfg = 1;
    bm = cfg == 2 ? 64 : 128;
    bn = 64;
    return cfg;
}

int dispatch_x3f(const ConvP& p, int tile_cfg, hipStream_t st) {
    int bm, bn;
    long long M = 0;
    for (int i = 0; i < p.ncls; ++i) M += p.cls[i].M;
    switch (pick_tile_x3f(tile_cfg, M, p.Cout, p.ncls, bm, bn, stream_bn(p))) {
        case 5:
            if (bn == 256) return launch_stream<64, 256, 2, 4, true>(p, st);
            if (bn == 128) return launch_stream<128, 128, 2, 4, false>(p, st);
            return launch_stream<128, 64, 2, 2, false>(p, st);
        case 2: return launch_x3f<64, 64, 2, 2, 2>(p, st);
        default: return launch_x3f<128, 64, 2, 2, 2>(p, st);
    }
}

int x3f_blocks(const ConvP& p, int tile_cfg) {
    int bm, bn;
    long long M = 0;
    for (int i = 0; i < p.ncls; ++i) M += p.cls[i].M;
    pick_tile_x3f(tile_cfg, M, p.Cout, p.ncls, bm, bn, stream_bn(p));
    int blocks = 0;
    for (int i = 0; i < p.ncls; ++i) blocks += (p.cls[i].M + bm - 1) / bm;
    return blocks;
}

// single-tap geometry only: a 1x1 filter without padding
inline bool x3f_geometry_ok(int kh, int kw, int pad) { return kh == 1 && kw == 1 && pad == 0; }

}  // namespace

// does straps_conv_fwd_x3f / straps_conv_dgrad_x3f cover this geometry?  (hosts route the other layers through the plane kernels)
extern "C" int straps_conv_x3f_supported(int cin, int cout, int kh, int kw, int stride, int pad) {
    return x3f_geometry_ok(kh, kw, pad) && (stride == 1 || stride == 2) && cin % 64 == 0 && cout % 64 == 0;
}

// 1x1 convolution on the bf16 matrix pipe with the A operand read from the fp32 NHWC tensor x [batch][h][w][cin] (see the head of this
// file).  a_scale / a_shift (both or neither; [cin]): the producer's BatchNorm, applied to x in the operand path, followed by ReLU if
// a_relu -- x is then the RAW output of the previous convolution.  Everything else as straps_conv_fwd_x3.
extern "C" int straps_conv_fwd_x3f(const float* x, const float* a_scale, const float* a_shift, int a_relu, const unsigned short* w3, long long w_plane_stride,
                                   const float* scale, const float* shift, const float* residual, int relu, float* y, float* stats_partial,
                                   int batch, int h, int wdt, int cin, int cout, int kh, int kw, int stride, int pad, int tile_cfg, void* stream) {
    STRAPS_REQUIRE(x && w3 && y, "straps_conv_fwd_x3f: null pointer");
    STRAPS_REQUIRE(batch > 0 && h > 0 && wdt > 0, "straps_conv_fwd_x3f: empty input %dx%dx%d", batch, h, wdt);
    STRAPS_REQUIRE(x3f_geometry_ok(kh, kw, pad) && stride >= 1, "straps_conv_fwd_x3f: 1x1 filters without padding only (kh=%d kw=%d pad=%d)", kh, kw, pad);
    STRAPS_REQUIRE(cin % 32 == 0 && cout % 64 == 0, "straps_conv_fwd_x3f: need cin%%32==0 and cout%%64==0 (cin=%d cout=%d)", cin, cout);
    STRAPS_REQUIRE((scale == nullptr) == (shift == nullptr), "straps_conv_fwd_x3f: scale and shift must be given together");
    STRAPS_REQUIRE((a_scale == nullptr) == (a_shift == nullptr), "straps_conv_fwd_x3f: a_scale and a_shift must be given together");
    STRAPS_REQUIRE(w_plane_stride % 8 == 0, "straps_conv_fwd_x3f: the plane stride must be a multiple of 8 elements");
    STRAPS_REQUIRE((reinterpret_cast<uintptr_t>(x) & 15) == 0 && (!a_scale || ((reinterpret_cast<uintptr_t>(a_scale) | reinterpret_cast<uintptr_t>(a_shift)) & 15) == 0),
                   "straps_conv_fwd_x3f: x, a_scale and a_shift must be 16-byte aligned");
    ConvP p;
    p.x = x; p.w = reinterpret_cast<const float*>(w3);
    p.xps = 0; p.wps = w_plane_stride;
    const int rc = conv_fwd_problem(p, scale, shift, residual, relu, y, stats_partial, batch, h, wdt, cin, cout, kh, kw, stride, pad);
    if (rc != STRAPS_OK) return rc;
    p.a_scale = a_scale; p.a_shift = a_shift; p.a_relu = a_relu;
    return dispatch_x3f(p, tile_cfg, (hipStream_t)stream);
}

// number of [cout][2] statistics partials straps_conv_fwd_x3f writes for this geometry (= its M tiles)
extern "C" int straps_conv_x3f_stat_blocks(int batch, int h, int w, int cin, int cout, int kh, int kw, int stride, int pad, int tile_cfg) {
    ConvP p;
    p.x = nullptr; p.w = nullptr; p.xps = p.wps = 0;
    if (!x3f_geometry_ok(kh, kw, pad) || conv_fwd_problem(p, nullptr, nullptr, nullptr, 0, nullptr, nullptr, batch, h, w, cin, cout, kh, kw, stride, pad) != STRAPS_OK) return -1;
    p.y = reinterpret_cast<float*>(16);      // (stand-in, as above: a training forward)
    return x3f_blocks(p, tile_cfg);
}

// data gradient of a 1x1 convolution with the gradient dy read as the fp32 tensor [batch][ho][wo][cout] (no planes of it need exist): every
// optional operand of the plane entry points in one signature -- addend (+ addend_bits: the addend is the unmasked gradient of a residual
// unit's output, bit = its ReLU decision), and the fused BatchNorm-backward sums of the BatchNorm whose output the convolution read (bn_raw
// != NULL: bn_mean, bn_invstd, bn_partials required, and bn_out_bits or bn_mask_scale / bn_mask_shift as the ReLU mask).
extern "C" int straps_conv_dgrad_x3f(const float* dy, const unsigned short* w3_crsk, long long w_plane_stride, const float* addend, const unsigned* addend_bits,
                                     float* dx, int batch, int h, int wdt, int cin, int cout, int kh, int kw, int stride, int pad, int tile_cfg,
                                     const float* bn_raw, const unsigned* bn_out_bits, const float* bn_mask_scale, const float* bn_mask_shift,
                                     const float* bn_mean, const float* bn_invstd, double* bn_partials, void* stream) {
    STRAPS_REQUIRE(dy && w3_crsk && dx, "straps_conv_dgrad_x3f: null pointer");
    STRAPS_REQUIRE(x3f_geometry_ok(kh, kw, pad) && (stride == 1 || stride == 2), "straps_conv_dgrad_x3f: 1x1 filters without padding, stride 1 or 2 only");
    STRAPS_REQUIRE(cout % 32 == 0 && cin % 64 == 0, "straps_conv_dgrad_x3f: need cout%%32==0 and cin%%64==0 (cin=%d cout=%d)", cin, cout);
    STRAPS_REQUIRE(!addend_bits || addend, "straps_conv_dgrad_x3f: the ReLU bits mask an addend");
    STRAPS_REQUIRE(!bn_raw || (bn_mean && bn_invstd && bn_partials && (bn_out_bits || (bn_mask_scale && bn_mask_shift))),
                   "straps_conv_dgrad_x3f: the BatchNorm sums need raw, mean, invstd, partials, and the output's bits or mask scale / shift");
    STRAPS_REQUIRE(w_plane_stride % 8 == 0 && (reinterpret_cast<uintptr_t>(dy) & 15) == 0, "straps_conv_dgrad_x3f: plane stride %% 8, dy 16-byte aligned");
    ConvP p;
    p.x = dy; p.w = reinterpret_cast<const float*>(w3_crsk);
    p.xps = 0; p.wps = w_plane_stride;
    const int rc = conv_dgrad_problem(p, addend, dx, batch, h, wdt, cin, cout, kh, kw, stride, pad);
    if (rc != STRAPS_OK) return rc;
    p.a_scale = p.a_shift = nullptr; p.a_relu = 0;
    p.res_bits = addend_bits;
    if (bn_raw) {
        p.bnr_raw = bn_raw; p.bnr_out = nullptr; p.bnr_sc = bn_mask_scale; p.bnr_sh = bn_mask_shift; p.bnr_mean = bn_mean; p.bnr_invstd = bn_invstd;
        p.bnr_part = bn_partials; p.bnr_bits = bn_out_bits;
    }
    return p.ncls ? dispatch_x3f(p, tile_cfg, (hipStream_t)stream) : STRAPS_OK;
}

// M tiles (= BatchNorm-backward partial blocks) of straps_conv_dgrad_x3f for this geometry, all parity classes
extern "C" int straps_conv_dgrad_x3f_bn_blocks(int batch, int h, int w, int cin, int cout, int kh, int kw, int stride, int pad, int tile_cfg) {
    ConvP p;
    p.x = nullptr; p.w = nullptr; p.xps = p.wps = 0;
    if (!x3f_geometry_ok(kh, kw, pad) || !(stride == 1 || stride == 2)) return -1;
    if (conv_dgrad_problem(p, nullptr, nullptr, batch, h, w, cin, cout, kh, kw, stride, pad) != STRAPS_OK) return -1;
    p.y = reinterpret_cast<float*>(16); p.bnr_raw = reinterpret_cast<const float*>(16);      // (stand-ins: the tile rule looks at WHICH operands a launch has -- x3f_epilogue)
    return x3f_blocks(p, tile_cfg);
}
