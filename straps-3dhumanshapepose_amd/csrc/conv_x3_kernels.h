// conv_x3_kernels.h -- the kernels of conv_x3.hip (bf16x3 implicit GEMM on planes: the im2col ring kernel and the halo-patch kernel) and their launch
// templates, in a header since round 6: conv_x3.hip instantiates them with the shared epilogue of conv_igemm.h (EPI = 0: every fused form),
// conv_x3_lean.hip with the lean epilogues (EPI = 1: raw result + statistics, a training step's forward; EPI = 2: a data gradient with addend / ReLU
// bits / fused BatchNorm sums) -- two translation units, compiled in parallel.  See conv_x3.hip for the arithmetic and the layouts.
#pragma once
#include "conv_igemm.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ f32x16 mfma_bf16(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }

__device__ __attribute__((aligned(16))) float k_zero16_x3[4] = {0.f, 0.f, 0.f, 0.f};   // source of the padding pixels

// slot swizzle of a 64-byte row (four 16-byte K groups): ds_read_b128 is serviced in groups of 16 lanes -- rows {0-3,12-15,20-27},
// {4-11,16-19,28-31} (MI355X_MICROARCH.md, LDS table) -- over 64 banks = four rows: the four rows of a group that share r & 3 must
// use four different slots; bits 3 and 4 of the row number separate them in both groups.
__device__ __forceinline__ int swz3(int r) { return (r >> 3) & 3; }

// ABL (tools only, wrong results): 1 = no MFMAs and no fragment reads (prices the operand copies alone), 2 = no operand copies
//   (prices the matrix work + fragment reads alone), 3 = neither copies nor fragment reads (the MFMA stream + barriers alone)
// PIPE: software-pipelined chunk loop.  The fragments of the two 16-wide k steps of a chunk live in two register sets; the barrier that
//   publishes chunk q+1 sits BETWEEN the two MFMA blocks of chunk q: k step 1 of chunk q is read before it, k step 0 of chunk q+1
//   right after it, each half a chunk ahead of its MFMAs (no LDS latency in front of an MFMA block), and the stage of chunk q --
//   dead once every wave is past that barrier -- is refilled with chunk q+NST during the second block (copies run NST chunks ahead).
template <int BM, int BN, int WGM, int WGN, int NST, int ABL = 0, bool PIPE = false, int EPI = 0>
__global__ __launch_bounds__(64 * WGM * WGN) void conv_igemm_x3_kernel(ConvP p) {
    const ConvP::Class& c = p.cls[blockIdx.y];
    const int cMh = c.Mh, cMw = c.Mw, cM = c.M, cMT = c.MT, cntaps = c.ntaps;
    if ((int)blockIdx.x >= cMT * p.NT) return;                 // a smaller class of the same launch
    ClkSample clk;
    clk_begin(p, clk);
    constexpr int NW = WGM * WGN, RPP = 16 * NW;               // waves; rows per copy pass (4 lanes x 16 bytes per 64-byte row)
    constexpr int WTM = BM / WGM, WTN = BN / WGN, MI = WTM / 32, NI = WTN / 32;
    constexpr int AP = BM / RPP, BP = BN / RPP;                // copy passes
    static_assert(BM % RPP == 0 && BN % RPP == 0 && WTM % 32 == 0 && WTN % 32 == 0, "tile / wave grid mismatch");
    constexpr int NPIECE = 3 * (AP + BP);                      // LDS-DMA instructions per thread and chunk
    constexpr int NMFMA = 2 * 6 * MI * NI;
    constexpr int GAP = NMFMA / NPIECE > 0 ? NMFMA / NPIECE : 1;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    u16* As = reinterpret_cast<u16*>(smem);       // [NST stages][3 planes][BM][32]
    u16* Bs = As + NST * 3 * BM * 32;             // [NST stages][3 planes][BN][32]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int bid = xcd_remap(blockIdx.x, cMT * p.NT);
    const int nt = bid % p.NT, mt = bid / p.NT;
    const int m0 = mt * BM, n0 = nt * BN;
    const int lr = tid >> 2;                                   // row of the RPP-row pass this thread copies
    const int lc = (tid & 3) ^ swz3(lr);                       // 16-byte K group it fetches for its slot tid & 3
    const u16* xg = reinterpret_cast<const u16*>(p.x);
    const u16* wg = reinterpret_cast<const u16*>(p.w);
    X3Epilogue<BM, BN, WGM, WGN, EPI> ep;
    ep.init(p, c, m0, n0);

    int a_hi0[AP], a_wi0[AP], a_base[AP];
    const int MhMw = cMh * cMw;
#pragma unroll
    for (int q = 0; q < AP; ++q) {
        const int m = m0 + lr + RPP * q;
        if (m < cM) {
            const int b = m / MhMw;
            const int rem = m - b * MhMw;
            const int ho = rem / cMw, wo = rem - ho * cMw;
            a_hi0[q] = ho * p.stride;
            a_wi0[q] = wo * p.stride;
            a_base[q] = ((b * p.H + a_hi0[q]) * p.W + a_wi0[q]) * 32 + lc * 8;       // chunk-major planes: 64 bytes per (pixel, chunk)
        } else {
            a_hi0[q] = -(1 << 28);
            a_wi0[q] = 0;
            a_base[q] = 0;
        }
    }
    const u16* wrow[BP];
#pragma unroll
    for (int q = 0; q < BP; ++q) wrow[q] = wg + (n0 + lr + RPP * q) * 32 + lc * 8;      // + ((tap * cchunks + chunk) * Cout) * 32

    const int cchunks = p.Cin >> 5;
    const int nchunks = cntaps * cchunks;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int a_cstep = p.xrows * 32, b_cstep = p.Cout * 32;      // one channel chunk on: elements

    const u16* a_src[AP];
    long long a_ps[AP];            // plane stride, 0 for a padding pixel (all three planes read the zero constant)
    int a_inc[AP];
    const u16* b_src[BP];
    int n_tap = 0, n_cc = 0;
    const int tl = lane < 9 ? lane : 0;
    const int v_dh = c.tap_dh[tl], v_dw = c.tap_dw[tl], v_tw = c.tap_w[tl];
    const u16* zsrc = reinterpret_cast<const u16*>(k_zero16_x3);
    asm volatile("" : "+s"(zsrc));
    auto setup_tap = [&](int tap) {
        const int dh = __builtin_amdgcn_readlane(v_dh, tap), dw = __builtin_amdgcn_readlane(v_dw, tap), tw = __builtin_amdgcn_readlane(v_tw, tap);
        const int toff = (dh * p.W + dw) * 32;
#pragma unroll
        for (int i = 0; i < AP; ++i) {
            const int hi = a_hi0[i] + dh, wi = a_wi0[i] + dw;
            const bool ok = (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
            a_src[i] = ok ? xg + (a_base[i] + toff) : zsrc;
            a_ps[i] = ok ? p.xps : 0;
            a_inc[i] = ok ? a_cstep : 0;
        }
#pragma unroll
        for (int i = 0; i < BP; ++i) b_src[i] = wrow[i] + (long long)tw * cchunks * b_cstep;
    };
    // copy piece `idx` (compile-time after unrolling) of the next chunk: planes outermost, A passes then B passes
    auto piece = [&](int stage, int idx) {
        if constexpr (ABL == 2 || ABL == 3) return;
        const int plane = idx / (AP + BP), r = idx % (AP + BP);
        if (r < AP) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_src[r] + plane * a_ps[r]),
                                             (__attribute__((address_space(3))) void*)(As + ((stage * 3 + plane) * BM + RPP * r + 16 * wave_u) * 32), 16, 0, 0);
        } else {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b_src[r - AP] + plane * p.wps),
                                             (__attribute__((address_space(3))) void*)(Bs + ((stage * 3 + plane) * BN + RPP * (r - AP) + 16 * wave_u) * 32), 16, 0, 0);
        }
    };
    auto advance = [&]() {
#pragma unroll
        for (int i = 0; i < AP; ++i) a_src[i] += a_inc[i];
#pragma unroll
        for (int i = 0; i < BP; ++i) b_src[i] += b_cstep;
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

    // chunk q lives in stage q % NST; the copies run NST - 1 (PIPE: NST) chunks ahead of the matrix work
    if (nchunks > 0) setup_tap(0);
#pragma unroll
    for (int s = 0; s < (PIPE ? NST : NST - 1); ++s)
        if (s < nchunks) {
#pragma unroll
            for (int idx = 0; idx < NPIECE; ++idx) piece(s, idx);
            advance();
        }
    int fo[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) fo[kk] = (lane & 31) * 32 + (((kk * 2 + (lane >> 5)) ^ swz3(lane & 31)) << 3);

    // plane pairs of the six products, smallest terms first
    constexpr int TA[6] = {1, 0, 2, 0, 1, 0};
    constexpr int TB[6] = {1, 2, 0, 1, 0, 0};

    // MORE: chunk q + NST - 1 exists and is issued between this chunk's MFMAs into the stage that chunk q - 1 was read from.
    // INFLIGHT: copies of younger chunks that may stay outstanding while this chunk's are awaited (the counter retires in order).
    auto chunk = [&](int stage, int nstage, auto more_c, auto inflight_c, auto last_c) {
        constexpr bool MORE = decltype(more_c)::value;
        constexpr int INFLIGHT = decltype(inflight_c)::value;
        // my copies of this chunk have landed, then everybody's have -- and every wave is done reading the stage refilled next
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(INFLIGHT) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        // the last chunk: no copy wait follows -- the epilogue's first operands are fetched under this chunk's matrix work (conv_igemm.h)
        if constexpr (decltype(last_c)::value) ep.prefetch();
        const u16* Ab = As + (stage * 3 * BM + wm * WTM) * 32;
        const u16* Bb = Bs + (stage * 3 * BN + wn * WTN) * 32;
        int cnt = 0;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 a[MI][3], b[NI][3];
            if constexpr (ABL == 3) {      // no fragment reads either: the matrix work on whatever the registers hold
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
                    for (int i = 0; i < MI; ++i) asm volatile("" : "=v"(a[i][pl]));
#pragma unroll
                    for (int j = 0; j < NI; ++j) asm volatile("" : "=v"(b[j][pl]));
                }
            }
#pragma unroll
            for (int pl = 0; pl < 3 && ABL != 3; ++pl) {
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
                        if constexpr (ABL != 1) acc[i][j] = mfma_bf16(a[i][TA[t]], b[j][TB[t]], acc[i][j]);
                        if (MORE && cnt % GAP == GAP - 1 && cnt / GAP < NPIECE) piece(nstage, cnt / GAP);
                        ++cnt;
                    }
        }
        if (MORE) {
#pragma unroll
            for (int idx = NMFMA / GAP; idx < NPIECE; ++idx) piece(nstage, idx);
            advance();
        }
    };
    if constexpr (PIPE) {
        bf16x8 fa[2][MI][3], fb[2][NI][3];
        auto load_frags = [&](int stage, int kk, int set) {
            const u16* Ab = As + (stage * 3 * BM + wm * WTM) * 32;
            const u16* Bb = Bs + (stage * 3 * BN + wn * WTN) * 32;
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
                for (int i = 0; i < MI; ++i) fa[set][i][pl] = *reinterpret_cast<const bf16x8*>(Ab + (pl * BM + i * 32) * 32 + fo[kk]);
#pragma unroll
                for (int j = 0; j < NI; ++j) fb[set][j][pl] = *reinterpret_cast<const bf16x8*>(Bb + (pl * BN + j * 32) * 32 + fo[kk]);
            }
        };
        constexpr int HALF = NMFMA / 2, GAP2 = HALF / NPIECE > 0 ? HALF / NPIECE : 1;
        auto mfma_block = [&](int set, int nstage, auto more_c) {
            constexpr bool more = decltype(more_c)::value;      // (compile-time: a run-time test per copy splits the block at every MFMA)
            int cnt = 0;
#pragma unroll
            for (int t = 0; t < 6; ++t)
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j) {
                        acc[i][j] = mfma_bf16(fa[set][i][TA[t]], fb[set][j][TB[t]], acc[i][j]);
                        if (set == 1 && cnt % GAP2 == GAP2 - 1 && cnt / GAP2 < NPIECE) { if (more) piece(nstage, cnt / GAP2); }
                        ++cnt;
                    }
            if (set == 1 && more) {
#pragma unroll
                for (int idx = HALF / GAP2; idx < NPIECE; ++idx) piece(nstage, idx);
                advance();
            }
        };
        if (nchunks > 0) {
            // chunk 0 has landed (up to NST - 1 younger chunks may stay in flight), then everybody's has
            if (nchunks >= NST) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 1) * NPIECE) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            load_frags(0, 0, 0);
        }
        int stage = 0;
        auto body = [&](int q, auto more_c, auto last_c) {
            const int nxt = stage + 1 == NST ? 0 : stage + 1;
            load_frags(stage, 1, 1);
            __builtin_amdgcn_sched_barrier(0);
            mfma_block(0, 0, std::false_type{});
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (!decltype(last_c)::value) {
                // chunk q+1 (issued NST chunks ago) has landed -- the NST - 2 chunks after it may still be in flight
                if (q + NST - 1 < nchunks) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * NPIECE) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // (my reads of this chunk's stage are done before anyone may refill it)
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                load_frags(nxt, 0, 0);
            } else {
                // the last chunk, first fragment set consumed: no copy wait follows -- the epilogue's first operands are fetched under the
                // second MFMA block (conv_igemm.h)
                ep.prefetch();
            }
            __builtin_amdgcn_sched_barrier(0);
            mfma_block(1, stage, more_c);                     // refill this chunk's stage (chunk q + NST): no wave reads it any more
            __builtin_amdgcn_sched_barrier(0);
            stage = nxt;
        };
        int q = 0;
        for (; q + NST < nchunks; ++q) body(q, std::true_type{}, std::false_type{});
        for (; q + 1 < nchunks; ++q) body(q, std::false_type{}, std::false_type{});
        if (q < nchunks) body(q, std::false_type{}, std::true_type{});      // (peeled: the prefetched unit's registers are live from here on only)
    } else {
    int stage = 0, nstage = NST - 1;
    auto next = [&]() {
        stage = stage + 1 == NST ? 0 : stage + 1;
        nstage = nstage + 1 == NST ? 0 : nstage + 1;
    };
    int q = 0;
    for (; q + NST - 1 < nchunks; ++q) { chunk(stage, nstage, std::true_type{}, std::integral_constant<int, (NST - 2) * NPIECE>{}, std::false_type{}); next(); }
    if constexpr (NST == 3) {
        if (q + 1 < nchunks) { chunk(stage, nstage, std::false_type{}, std::integral_constant<int, NPIECE>{}, std::false_type{}); next(); ++q; }
    }
    if (q < nchunks) chunk(stage, nstage, std::false_type{}, std::integral_constant<int, 0>{}, std::true_type{});
    }

    float s1[NI], s2[NI];
    double bd1[NI], bd2[NI];
    if constexpr (EPI == 1) {
        lean_epilogue_fwd<BM, BN, WGM, WGN>(p, acc, m0, n0, c.M, s1, s2);
        igemm_store_stats<BM, BN, WGM, WGN>(p, s1, s2, mt, n0, smem);
    } else if constexpr (EPI == 2) {
        ep.finish(p, c, acc, s1, s2, bd1, bd2);      // (LeanDgradEpilogue: its first two units were requested under the last chunk)
        igemm_store_bnr<BM, BN, WGM, WGN>(p, bd1, bd2, p.bnr_base[blockIdx.y] + mt, n0, smem);
    } else {
        ep.finish(p, c, acc, s1, s2, bd1, bd2);
        igemm_store_stats<BM, BN, WGM, WGN>(p, s1, s2, mt, n0, smem);
        igemm_store_bnr<BM, BN, WGM, WGN>(p, bd1, bd2, p.bnr_base[blockIdx.y] + mt, n0, smem);
    }
    clk_end(p, clk);
}

// ---- halo-patch variant for the 3x3 / stride-1 layers (forward, and the data gradient, which is the same convolution over dy) ----
// The im2col A operand above copies every input pixel nine times per channel chunk (once per tap) from L2; here a tile's input patch
// -- its BM output pixels are whole image rows (or whole images), so the patch is (rows + 2) x (W + 2) pixels per image -- is copied
// ONCE per 32-channel chunk into a double-buffered LDS image and the nine taps are nine shifted fragment addresses into it.  The
// reduction runs channel-chunk-major / tap-minor; the weights keep their NST-stage ring (one (tap, chunk) slice per step).
// L2 -> LDS bytes per step: (BM + BN) x 192  ->  (BN + patch / 9) x 192: 2.0x fewer for layer1 (128x64), 1.7-2.2x for the others.
// PS = patch slot capacity.  Slot swizzle: 16-byte group g of slot s sits at g ^ ((s >> 2) & 3), conflict-free for the 16-lane
// groups of ds_read_b128 over consecutive slots at any tap shift.
// PBUF = 1 (the 64-channel outputs: layer1 and its data gradients): ONE patch buffer and a two-stage weight ring -- 76 KB of LDS, two
// workgroups per CU.  The next chunk's patch is then copied at the chunk boundary itself (barrier: every wave is done with the old patch;
// copy; wait; barrier) instead of a chunk ahead: the co-resident workgroup computes meanwhile.  123 us against 137 us for the im2col tile
// on the layer1 shape (tools/x3d_probe.py): the 64-channel layers are bound by the L2 -> CU operand stream, and this form fetches every
// input pixel once per channel chunk instead of once per tap without giving up the second workgroup.
template <int BM, int BN, int WGM, int WGN, int NST, int PS, int PBUF = 2, int EPI = 0>
__global__ __launch_bounds__(64 * WGM * WGN) void conv_igemm_x3h_kernel(ConvP p) {
    const ConvP::Class& c = p.cls[0];
    const int cntaps = c.ntaps;
    ClkSample clk;
    clk_begin(p, clk);
    constexpr int NW = WGM * WGN, NTH = 64 * NW, RPP = 16 * NW;
    constexpr int WTM = BM / WGM, WTN = BN / WGN, MI = WTM / 32, NI = WTN / 32;
    constexpr int BP = BN / RPP;
    static_assert(BN % RPP == 0 && WTM % 32 == 0 && WTN % 32 == 0 && PS % 16 == 0, "tile / wave grid mismatch");
    constexpr int NPB = 3 * BP;                                // weight copies per thread and step
    constexpr int NPA = (PS * 4 + NTH - 1) / NTH;              // patch copy rounds per plane (the last one may cover only some waves)
    constexpr int NMFMA = 2 * 6 * MI * NI;
    constexpr int GAP = NMFMA / NPB > 0 ? NMFMA / NPB : 1;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    static_assert(PBUF == 1 || PBUF == 2, "one or two patch buffers");
    u16* As = reinterpret_cast<u16*>(smem);       // [PBUF patch buffers][3 planes][PS slots][32]
    u16* Bs = As + PBUF * 3 * PS * 32;            // [NST stages][3 planes][BN][32]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int bid = xcd_remap(blockIdx.x, c.MT * p.NT);
    const int nt = bid % p.NT, mt = bid / p.NT;
    const int m0 = mt * BM, n0 = nt * BN;
    const u16* xg = reinterpret_cast<const u16*>(p.x);
    const u16* wg = reinterpret_cast<const u16*>(p.w);
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    X3Epilogue<BM, BN, WGM, WGN, EPI> ep;
    ep.init(p, c, m0, n0);

    // ---- tile geometry: nimg images x rows_t rows x W columns; patch (rows_t + 2) x (W + 2) slots per image
    const int HW = p.H * p.W, PW = p.W + 2;
    const int nimg = BM >= HW ? BM / HW : 1;
    const int rows_t = BM >= HW ? p.H : BM / p.W;
    const int pslots = (rows_t + 2) * PW;                      // per image
    const int nslots = nimg * pslots;
    const int b0 = m0 / HW, row0 = (m0 - b0 * HW) / p.W;

    // ---- patch copies: round pi covers slots pi*NTH/4 + (tid >> 2), 16-byte slot tid & 3 <- channel group (tid & 3) ^ ((slot >> 2) & 3)
    int poff[NPA];
#pragma unroll
    for (int pi = 0; pi < NPA; ++pi) {
        const int slot = pi * (NTH / 4) + (tid >> 2);
        const int im = slot / pslots, rem = slot - im * pslots;
        const int py = rem / PW, px = rem - py * PW;
        const int sr = row0 + py - 1, sc = px - 1;
        const bool ok = slot < nslots && (unsigned)sr < (unsigned)p.H && (unsigned)sc < (unsigned)p.W;
        poff[pi] = ok ? (((b0 + im) * p.H + sr) * p.W + sc) * 32 + (((tid & 3) ^ ((slot >> 2) & 3)) << 3) : -1;     // chunk-major planes
    }
    const u16* zsrc = reinterpret_cast<const u16*>(k_zero16_x3);
    asm volatile("" : "+s"(zsrc));
    auto patch_dma = [&](int cc) {
        u16* dst = As + (PBUF == 2 ? (cc & 1) : 0) * (3 * PS * 32);
#pragma unroll
        for (int pi = 0; pi < NPA; ++pi) {
            if (pi * (NTH / 4) + 16 * wave_u >= nslots) break;                 // wave-uniform: nothing of this round lies inside the patch
            if (pi * (NTH / 4) + 16 * wave_u >= PS) break;
            const bool ok = poff[pi] >= 0;
            const u16* src = ok ? xg + (poff[pi] + (long long)cc * p.xrows * 32) : zsrc;
            const long long ps = ok ? p.xps : 0;
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + pl * ps),
                                                 (__attribute__((address_space(3))) void*)(dst + (pl * PS + pi * (NTH / 4) + 16 * wave_u) * 32), 16, 0, 0);
        }
    };

    // ---- weight copies (as in the im2col kernel), issue stream runs NST - 1 steps ahead: step = (chunk cc, tap), tap fastest
    const int lr = tid >> 2;
    const int lc = (tid & 3) ^ swz3(lr);
    const u16* wrow[BP];
#pragma unroll
    for (int q = 0; q < BP; ++q) wrow[q] = wg + (n0 + lr + RPP * q) * 32 + lc * 8;      // + ((tap * cchunks + chunk) * Cout) * 32
    const int cchunks = p.Cin >> 5;
    const int nsteps = cntaps * cchunks;
    const int tl = lane < 9 ? lane : 0;
    const int v_tw = c.tap_w[tl];
    const int v_sh = (c.tap_dh[tl] + 1) * PW + c.tap_dw[tl] + 1;              // slot shift of the tap
    int i_tap = 0, i_cc = 0;                                                   // issue stream position
    auto piece = [&](int stage, int idx) {
        const int plane = idx / BP, r = idx % BP;
        const int tw = __builtin_amdgcn_readlane(v_tw, i_tap);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wrow[r] + (long long)(tw * cchunks + i_cc) * p.Cout * 32 + plane * p.wps),
                                         (__attribute__((address_space(3))) void*)(Bs + ((stage * 3 + plane) * BN + RPP * r + 16 * wave_u) * 32), 16, 0, 0);
    };
    auto advance = [&]() {
        if (++i_tap == cntaps) { i_tap = 0; ++i_cc; }
    };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- fragment addresses
    int a_slot[MI];                                             // patch slot of the lane's output pixel at tap shift 0
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int il = wm * WTM + i * 32 + (lane & 31);
        const int im = il / (rows_t * p.W), rem = il - im * (rows_t * p.W);
        const int y = rem / p.W, x = rem - y * p.W;
        a_slot[i] = im * pslots + y * PW + x;
    }
    int fo[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) fo[kk] = (lane & 31) * 32 + (((kk * 2 + (lane >> 5)) ^ swz3(lane & 31)) << 3);
    const int kh2 = lane >> 5;

    if (nsteps > 0) patch_dma(0);
#pragma unroll
    for (int s = 0; s < NST - 1; ++s)
        if (s < nsteps) {
#pragma unroll
            for (int idx = 0; idx < NPB; ++idx) piece(s, idx);
            advance();
        }
    constexpr int TA[6] = {1, 0, 2, 0, 1, 0};
    constexpr int TB[6] = {1, 2, 0, 1, 0, 0};
    int c_tap = 0, c_cc = 0;                                    // compute stream position
    auto step = [&](int stage, int nstage, auto more_c, auto inflight_c, auto last_c) {
        constexpr bool MORE = decltype(more_c)::value;
        constexpr int INFLIGHT = decltype(inflight_c)::value;
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(INFLIGHT) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        // first tap of a chunk: every wave is past the previous chunk, its patch buffer is free -> the next chunk's patch is copied
        // into it (issued before this step's weight copies: the in-order counter then retires it before any later weights)
        if constexpr (PBUF == 2) {
            if (c_tap == 0 && c_cc + 1 < cchunks) patch_dma(c_cc + 1);
        } else {
            // one buffer: every wave is past the barrier above, i.e. done with the previous chunk's patch -- this chunk's is copied now and
            // awaited (the weights of later steps that are in flight land with it: the in-order counter is at zero afterwards)
            if (c_tap == 0 && c_cc > 0) {
                patch_dma(c_cc);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
        }
        // the last step: no copy wait follows -- the epilogue's first operands are fetched under this step's matrix work (conv_igemm.h)
        if constexpr (decltype(last_c)::value) ep.prefetch();
        const u16* Ap = As + (PBUF == 2 ? (c_cc & 1) : 0) * (3 * PS * 32);
        const u16* Bb = Bs + (stage * 3 * BN + wn * WTN) * 32;
        const int tsh = __builtin_amdgcn_readlane(v_sh, c_tap);
        int ao[MI][2];
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int sl = a_slot[i] + tsh;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) ao[i][kk] = sl * 32 + (((kk * 2 + kh2) ^ ((sl >> 2) & 3)) << 3);
        }
        int cnt = 0;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 a[MI][3], b[NI][3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
                for (int i = 0; i < MI; ++i) a[i][pl] = *reinterpret_cast<const bf16x8*>(Ap + pl * PS * 32 + ao[i][kk]);
#pragma unroll
                for (int j = 0; j < NI; ++j) b[j][pl] = *reinterpret_cast<const bf16x8*>(Bb + (pl * BN + j * 32) * 32 + fo[kk]);
            }
#pragma unroll
            for (int t = 0; t < 6; ++t)
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j) {
                        acc[i][j] = mfma_bf16(a[i][TA[t]], b[j][TB[t]], acc[i][j]);
                        if (MORE && cnt % GAP == GAP - 1 && cnt / GAP < NPB) piece(nstage, cnt / GAP);
                        ++cnt;
                    }
        }
        if (MORE) {
#pragma unroll
            for (int idx = NMFMA / GAP; idx < NPB; ++idx) piece(nstage, idx);
            advance();
        }
        if (++c_tap == cntaps) { c_tap = 0; ++c_cc; }
    };
    int stage = 0, nstage = NST - 1;
    auto next = [&]() {
        stage = stage + 1 == NST ? 0 : stage + 1;
        nstage = nstage + 1 == NST ? 0 : nstage + 1;
    };
    int q = 0;
    for (; q + NST - 1 < nsteps; ++q) { step(stage, nstage, std::true_type{}, std::integral_constant<int, (NST - 2) * NPB>{}, std::false_type{}); next(); }
    if constexpr (NST == 3) {
        if (q + 1 < nsteps) { step(stage, nstage, std::false_type{}, std::integral_constant<int, NPB>{}, std::false_type{}); next(); ++q; }
    }
    if (q < nsteps) step(stage, nstage, std::false_type{}, std::integral_constant<int, 0>{}, std::true_type{});

    float s1[NI], s2[NI];
    double bd1[NI], bd2[NI];
    if constexpr (EPI == 1) {
        lean_epilogue_fwd<BM, BN, WGM, WGN>(p, acc, m0, n0, c.M, s1, s2);
        igemm_store_stats<BM, BN, WGM, WGN>(p, s1, s2, mt, n0, smem);
    } else if constexpr (EPI == 2) {
        ep.finish(p, c, acc, s1, s2, bd1, bd2);      // (LeanDgradEpilogue: its first two units were requested under the last chunk)
        igemm_store_bnr<BM, BN, WGM, WGN>(p, bd1, bd2, p.bnr_base[blockIdx.y] + mt, n0, smem);
    } else {
        ep.finish(p, c, acc, s1, s2, bd1, bd2);
        igemm_store_stats<BM, BN, WGM, WGN>(p, s1, s2, mt, n0, smem);
        igemm_store_bnr<BM, BN, WGM, WGN>(p, bd1, bd2, p.bnr_base[blockIdx.y] + mt, n0, smem);
    }
    clk_end(p, clk);
}

template <int BM, int BN, int WGM, int WGN, int NST, int PS, int PBUF = 2, int EPI = 0>
int launch_x3h(const ConvP& p0, hipStream_t st) {
    ConvP p = p0;
    p.NT = p.Cout / BN;
    p.cls[0].MT = p.cls[0].M / BM;
    p.bnr_base[0] = 0;
    const size_t lds = ((size_t)PBUF * 3 * PS + (size_t)NST * 3 * BN) * 32 * sizeof(u16);
    STRAPS_RAISE_LDS((conv_igemm_x3h_kernel<BM, BN, WGM, WGN, NST, PS, PBUF, EPI>), lds, "conv_igemm_x3h_kernel");
    hipLaunchKernelGGL((conv_igemm_x3h_kernel<BM, BN, WGM, WGN, NST, PS, PBUF, EPI>), dim3(p.cls[0].MT * p.NT, 1), dim3(64 * WGM * WGN), lds, st, p);
    STRAPS_CHECK_LAUNCH("conv_igemm_x3h_kernel");
    return STRAPS_OK;
}

// does the halo-patch kernel (BM = 128) cover this problem?  3x3 / stride 1 / pad 1 as ONE class over the full map, tiles of whole rows
// inside one image or of whole images, no ragged tile, patch within the slot capacity
inline int halo_patch_slots(const ConvP& p) {
    constexpr int BM = 128;
    if (p.ncls != 1 || p.stride != 1 || p.omul != 1) return 0;
    const ConvP::Class& c = p.cls[0];
    if (c.ntaps != 9 || c.Mh != p.H || c.Mw != p.W || p.OH != p.H || p.OW != p.W || c.oah != 0 || c.oaw != 0 || c.M % BM != 0) return 0;
    for (int t = 0; t < 9; ++t)
        if (c.tap_dh[t] < -1 || c.tap_dh[t] > 1 || c.tap_dw[t] < -1 || c.tap_dw[t] > 1) return 0;
    const int HW = p.H * p.W;
    if (BM >= HW) { if (BM % HW != 0) return 0; return (BM / HW) * (p.H + 2) * (p.W + 2); }
    if (BM % p.W != 0 || HW % BM != 0) return 0;
    return (BM / p.W + 2) * (p.W + 2);
}

template <int BM, int BN, int WGM, int WGN, int NST, int ABL = 0, bool PIPE = false, int EPI = 0>
int launch_x3(const ConvP& p0, hipStream_t st) {
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
    STRAPS_RAISE_LDS((conv_igemm_x3_kernel<BM, BN, WGM, WGN, NST, ABL, PIPE, EPI>), lds, "conv_igemm_x3_kernel");
    hipLaunchKernelGGL((conv_igemm_x3_kernel<BM, BN, WGM, WGN, NST, ABL, PIPE, EPI>), dim3(maxblk, p.ncls), dim3(64 * WGM * WGN), lds, st, p);
    STRAPS_CHECK_LAUNCH("conv_igemm_x3_kernel");
    return STRAPS_OK;
}

}  // namespace
