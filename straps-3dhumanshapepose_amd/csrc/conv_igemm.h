// conv_igemm.h -- what the implicit-GEMM convolution kernels share: the problem descriptor (tap table, output-parity classes),
// the host-side geometry set-up of the forward convolution and of its data gradient, and the epilogue of the 2x2-wave block tile.
// Included by conv.hip (exact-fp32 MFMA) and conv_x3.hip (three-plane bf16 operands, six products per term).
#pragma once
#include "common.h"
#include <type_traits>
#include <utility>

namespace {

struct ConvP {
    const float* x;
    const float* w;
    const float* scale;
    const float* shift;
    const float* res;
    float* y;
    float* stats;
    int H, W, Cin, Cout, relu;      // source tensor [B][H][W][Cin]
    int stride;                     // source pixel of logical (ho,wo) before the tap offset: (ho*stride, wo*stride)
    int OH, OW, omul;               // physical output pixel = (b, ho*omul + oah, wo*omul + oaw) in [B][OH][OW][Cout]
    int NT, wtaps;                  // N tiles; taps stored per output channel in w
    long long xps, wps;             // three-plane bf16 operands (conv_x3.hip): elements between the planes of x / of w
    int xrows, yrows;               // pixels of the source / of the physical output tensor (chunk-major plane addressing, common.h)
    // BatchNorm-backward sums fused into a data gradient's epilogue (conv_x3.hip, straps_conv_dgrad_x3_bn): the tensor this launch
    // writes is the gradient dy entering the BatchNorm (+ ReLU) that produced the convolution's input; its two backward sums
    // S1 = sum mask*dy, S2 = invstd * sum mask*dy*(raw - mean) per channel are accumulated here (double) as one partial per M tile
    // instead of by a pass of their own over (dy, raw).  mask = out > 0 if bnr_out, else fma(raw, bnr_sc, bnr_sh) > 0.
    const float* bnr_raw;
    const float* bnr_out;
    const float* bnr_sc;
    const float* bnr_sh;
    const float* bnr_mean;
    const float* bnr_invstd;
    double* bnr_part;               // [blocks][Cout][2]
    // ReLU decisions as bits instead of fp32 tensors (round 4; word [pixel][Cout / 32], bit c & 31 = "activation of channel c > 0", written by
    // straps_bn_apply_bits_x3): bnr_bits replaces bnr_out as the mask of the sums above, res_bits masks the addend (`res` is then the
    // UNMASKED gradient of a residual unit's output, of which the skip connection receives the part the unit's ReLU let through)
    const unsigned* bnr_bits;
    const unsigned* res_bits;
    int bnr_base[4];                // first partial block of each class
    // eval-mode forward on the bf16x3 route: the epilogue's result also as three bf16 planes (the next convolution's operand),
    // y itself may then be NULL when nothing reads the fp32 tensor
    unsigned short* yplanes;
    long long yps;
    // measurement aid (straps_set_clock_accumulator; NULL in library use): workgroup 0 of every launch adds the shader-clock and the
    // constant-rate wall-clock ticks it lived for to clk[0] / clk[1] -- their ratio is the clock the kernel really ran at
    unsigned long long* clk;
    // 1 = the look-ahead epilogue (IgemmEpilogue below: memory operands fetched ahead of their use), 0 = the row-by-row form of rounds 1-4
    // (A/B switch of the tools build, STRAPS_EPI; the product library always takes 1)
    int epi;
    // Up to four independent sub-problems per launch (blockIdx.y): the output-parity classes of a stride-2 data gradient
    // are GEMMs over a quarter of the pixels each with their own tap subset -- launched together they fill the chip
    // instead of queueing as four small grids.  A forward conv / stride-1 gradient is the single class 0.
    struct Class {
        int Mh, Mw;                 // logical output grid enumerated by M = B*Mh*Mw
        int M, MT;
        int oah, oaw;
        int ntaps;                  // taps used
        int tap_w[9], tap_dh[9], tap_dw[9];
    } cls[4];
    int ncls;
};

struct ClkSample { unsigned long long c0, w0; };
__device__ __forceinline__ void clk_begin(const ConvP& p, ClkSample& s) {
    s.c0 = 0; s.w0 = 0;
    if (p.clk) { s.c0 = __builtin_amdgcn_s_memtime(); s.w0 = __builtin_amdgcn_s_memrealtime(); }
}
__device__ __forceinline__ void clk_end(const ConvP& p, const ClkSample& s) {
    if (p.clk && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
        atomicAdd(p.clk, (unsigned long long)__builtin_amdgcn_s_memtime() - s.c0);
        atomicAdd(p.clk + 1, (unsigned long long)__builtin_amdgcn_s_memrealtime() - s.w0);
    }
}

// Epilogue of a BM x BN block tile held as 32x32 accumulator blocks by 2x2 waves (C layout: lane = output channel, reg = pixel
// row): BN scale/shift, residual/addend, ReLU fused; returns the per-lane (sum, sum of squares) of the raw values for the
// training-mode batch statistics.
// (BNR: also accumulate the BatchNorm-backward sums described at ConvP::bnr_raw into d1 / d2 -- register arrays of the caller)
template <int BM, int BN, int WGM, int WGN, bool BNR>
__device__ __forceinline__ void igemm_store_rows_impl(const ConvP& p, const ConvP::Class& c, const f32x16 (&acc)[BM / WGM / 32][BN / WGN / 32], int m0,
                                                      int n0, float (&s1)[BN / WGN / 32], float (&s2)[BN / WGN / 32], double (&d1)[BN / WGN / 32],
                                                      double (&d2)[BN / WGN / 32]) {
    constexpr int WTM = BM / WGM, WTN = BN / WGN, MI = WTM / 32, NI = WTN / 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int cMh = c.Mh, cMw = c.Mw, cM = c.M, coah = c.oah, coaw = c.oaw;
    const int MhMw = cMh * cMw;
    const bool remap = p.omul != 1 || coah != 0 || coaw != 0 || p.OH != cMh || p.OW != cMw;
    float sc[NI], sh[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int n = n0 + wn * WTN + j * 32 + (lane & 31);
        sc[j] = p.scale ? p.scale[n] : 1.f;
        sh[j] = p.shift ? p.shift[n] : 0.f;
        s1[j] = 0.f;
        s2[j] = 0.f;
    }
    [[maybe_unused]] float bsc[NI], bsh[NI], bmu[NI];
    [[maybe_unused]] const bool bnr = BNR && p.bnr_raw != nullptr;
    if constexpr (BNR) {
        if (bnr) {
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const int n = n0 + wn * WTN + j * 32 + (lane & 31);
                bsc[j] = (p.bnr_out || p.bnr_bits) ? 0.f : p.bnr_sc[n];
                bsh[j] = (p.bnr_out || p.bnr_bits) ? 0.f : p.bnr_sh[n];
                bmu[j] = p.bnr_mean[n];
                d1[j] = 0.0;
                d2[j] = 0.0;
            }
        }
    }
    // (32-bit element offsets, see the launcher's size check; an M tile that lies inside the problem skips the per-row test)
    const bool full = m0 + BM <= cM;
    auto rows = [&](auto full_c) {
        constexpr bool FULL = decltype(full_c)::value;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            // the lane's 16 rows are m = mb + (r & 3) + 8 * (r >> 2): the physical pixel of a remapped output (a parity class of
            // a stride-2 data gradient) is found by division once and then walked row by row
            const int mb = m0 + wm * WTM + i * 32 + 4 * (lane >> 5);
            int b_ = 0, ho_ = 0, wo_ = 0;
            if (remap) {
                b_ = mb / MhMw;
                const int rem = mb - b_ * MhMw;
                ho_ = rem / cMw;
                wo_ = rem - ho_ * cMw;
            }
            int pixr[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                pixr[r] = mb + (r & 3) + 8 * (r >> 2);
                if (remap) {
                    pixr[r] = (b_ * p.OH + ho_ * p.omul + coah) * p.OW + wo_ * p.omul + coaw;
                    wo_ += (r & 3) == 3 ? 5 : 1;
                    while (wo_ >= cMw) {
                        wo_ -= cMw;
                        if (++ho_ == cMh) { ho_ = 0; ++b_; }
                    }
                }
            }
            // the residual / skip-gradient values of all 16 rows are fetched before the first store: p.res may alias p.y (in
            // place), so the compiler would otherwise serialise load -> store -> load through 16 memory round trips
            float rv[16][NI];
            if (p.res) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
#pragma unroll
                    for (int j = 0; j < NI; ++j) {
                        const int m = mb + (r & 3) + 8 * (r >> 2);
                        rv[r][j] = (FULL || m < cM) ? p.res[pixr[r] * p.Cout + n0 + wn * WTN + j * 32 + (lane & 31)] : 0.f;
                    }
                if (p.res_bits) {      // (one word per row and 32-channel group, the same for the 32 lanes of a row: a broadcast load)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
#pragma unroll
                        for (int j = 0; j < NI; ++j) {
                            const int m = mb + (r & 3) + 8 * (r >> 2);
                            const unsigned wd = (FULL || m < cM) ? p.res_bits[pixr[r] * (p.Cout >> 5) + ((n0 + wn * WTN + j * 32) >> 5)] : 0u;
                            rv[r][j] = ((wd >> (lane & 31)) & 1u) ? rv[r][j] : 0.f;
                        }
                }
            }
            // (BNR: raw -- and, for a residual unit's last BatchNorm, the activation -- of eight rows at a time: fetched together like rv;
            //  two halves keep the 256x128 / 8-wave tile inside its 256 registers)
#pragma unroll
            for (int hb = 0; hb < 2; ++hb) {
                [[maybe_unused]] float xr[8][NI], yo[8][NI];
                if constexpr (BNR) {
                    if (bnr) {
#pragma unroll
                        for (int q = 0; q < 8; ++q)
#pragma unroll
                            for (int j = 0; j < NI; ++j) {
                                const int r = hb * 8 + q;
                                const int m = mb + (r & 3) + 8 * (r >> 2);
                                const int o = pixr[r] * p.Cout + n0 + wn * WTN + j * 32 + (lane & 31);
                                xr[q][j] = (FULL || m < cM) ? p.bnr_raw[o] : 0.f;
                                if (p.bnr_bits)      // (the word rides in the register the activation would occupy)
                                    yo[q][j] = (FULL || m < cM) ? __uint_as_float(p.bnr_bits[pixr[r] * (p.Cout >> 5) + ((n0 + wn * WTN + j * 32) >> 5)]) : 0.f;
                                else
                                    yo[q][j] = (p.bnr_out && (FULL || m < cM)) ? p.bnr_out[o] : 0.f;
                            }
                    }
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int r = hb * 8 + q;
                    const int m = mb + (r & 3) + 8 * (r >> 2);
                    if (FULL || m < cM) {
#pragma unroll
                        for (int j = 0; j < NI; ++j) {
                            const int o = pixr[r] * p.Cout + n0 + wn * WTN + j * 32 + (lane & 31);
                            float v = acc[i][j][r];
                            s1[j] += v;
                            s2[j] = fmaf(v, v, s2[j]);
                            if (p.scale) v = fmaf(v, sc[j], sh[j]);
                            if (p.res) v += rv[r][j];
                            if (p.relu) v = fmaxf(v, 0.f);
                            if (!BNR || p.y) p.y[o] = v;
                            if constexpr (BNR) {
                                if (p.yplanes) {
                                    u16 b1, b2, b3;
                                    split3(v, b1, b2, b3);
                                    const long long oc = cm_index(pixr[r], n0 + wn * WTN + j * 32 + (lane & 31), p.yrows);
                                    p.yplanes[oc] = b1;
                                    p.yplanes[p.yps + oc] = b2;
                                    p.yplanes[2 * p.yps + oc] = b3;
                                }
                                if (bnr) {
                                    const bool on = p.bnr_bits ? (((__float_as_uint(yo[q][j]) >> (lane & 31)) & 1u) != 0)
                                                               : (p.bnr_out ? yo[q][j] > 0.f : fmaf(xr[q][j], bsc[j], bsh[j]) > 0.f);
                                    const float g = on ? v : 0.f;
                                    d1[j] += (double)g;
                                    d2[j] += (double)g * ((double)xr[q][j] - (double)bmu[j]);
                                }
                            }
                        }
                    }
                }
            }
        }
    };
    if (full) rows(std::true_type{}); else rows(std::false_type{});
}

template <int BM, int BN, int WGM = 2, int WGN = 2>
__device__ __forceinline__ void igemm_store_bnr(const ConvP& p, const double (&bd1)[BN / WGN / 32], const double (&bd2)[BN / WGN / 32], int blk, int n0, float* smem);
template <int BM, int BN, int WGM = 2, int WGN = 2>
__device__ __forceinline__ void igemm_store_stats(const ConvP& p, const float (&s1)[BN / WGN / 32], const float (&s2)[BN / WGN / 32], int mt, int n0, float* smem);

// ---- round 5: the look-ahead ROW epilogue ---------------------------------------------------------------------------------------------
// What the row-by-row epilogue above costs when it has memory OPERANDS (a data gradient's addend; the BatchNorm input `raw` and the ReLU
// bits of the fused BatchNorm-backward sums): per 32-row block it fetches the addend rows, waits, then twice (fetch eight raw rows + eight
// bit words, wait, compute, store) -- and because loads and stores retire through ONE in-order counter on this part, every wait behind a
// batch of stores also waits for those stores to be acknowledged by memory.  Six dependent memory round trips per workgroup of a 128-row
// tile, 1.5-2.5 us each under load, at the end of a main loop of 7-14 us, with two workgroups per CU to hide them: `dgrad+bn` launches
// ran 16-54 us behind the forward convolution of the same FLOPs (190 vs 122 us on layer1's 64 -> 64 3x3, profiles/r04_bench_train_b64.json).
// And every access of that form is 4 bytes per lane (the accumulator layout: lane = channel): the launches that are ALL epilogue -- resnet50's
// 1x1 layers, K = 2-8 chunks -- moved their 200-450 MB at 2.4-3.0 TB/s where the streaming kernels of this library reach 5.
//
// Two changes (the first measured alone: profiles/r05_epilogue_ab.txt, resnet18 step 8.57 -> 8.28 ms, resnet50 14.88 -> 14.24 ms):
//  * LOOK-AHEAD.  A UNIT is one 32x32 accumulator block (i, j) of a wave and its operands are fetched AHEAD of their use: unit 0 before the
//    matrix work of the last K chunk (`prefetch()`, called by the kernels where no copy wait follows any more), the others -- up to DEPTH
//    in flight -- before unit 0 is computed and stored.  No load ever waits behind a store (a younger store does not hold back an older load
//    in the in-order counter), and one round trip, partly under the last chunk's MFMAs, is exposed instead of six.
//  * ROWS.  The unit's 32x32 values go through a per-wave LDS slice once (16 ds_write_b32 in the accumulator layout, 4 ds_read_b128 back):
//    afterwards lane l holds FOUR CONSECUTIVE CHANNELS (l & 7) * 4 .. + 3 of rows (l >> 3) + 8 t, t = 0..3, and every global access of the
//    epilogue -- addend, raw, the fp32 result -- is 16 bytes per lane, eight whole 128-byte lines per wave instruction; a row's word of ReLU
//    bits is one broadcast load for its eight lanes.  The batch-statistics sums and the BatchNorm-backward sums are kept per lane for its four
//    channels and combined over the eight row lanes and the M waves through LDS in a fixed order (igemm_row_stats / igemm_row_bnr below).
// Every training-mode launch takes this path on every tile that lies inside the problem; a ragged last M tile, the eval-mode forms (folded
// scale / shift, plane output) and the fp32-activation mask of the fused sums -- the A/B reference of the bit form -- keep the row-by-row form.
// The sums are the same terms in another order than there (per lane over its rows, then lanes, then waves): fixed, so every launch of a shape
// is bit-reproducible, but the double partials of the two forms agree to rounding (1e-13), not bit for bit; the gradient itself is bit-identical
// (tests/test_gpu_conv_x3.py::test_relu_bits_forms_equal_the_fp32_mask_forms_bit_for_bit: the row form against the row-by-row form on every tile
// configuration).
template <typename F, int... Rs>
__device__ __forceinline__ void epi_for16(F&& f, std::integer_sequence<int, Rs...>) { (f(std::integral_constant<int, Rs>{}), ...); }
// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>): expanded by the front end (a `#pragma unroll` loop over units is NOT reliable
// here -- one unit's body is a few thousand instructions, and beyond the pragma-unroll threshold the loop stays a loop, its unit and accumulator
// arrays get indexed at run time and land in scratch memory)
template <int N, typename F>
__device__ __forceinline__ void epi_static_for(F&& f) { epi_for16(f, std::make_integer_sequence<int, N>{}); }

struct EpiUnit {
    f32x4 rv[4];            // addend (p.res): rows (l >> 3) + 8 t, channels (l & 7) * 4 .. + 3
    f32x4 xr[4];            // BatchNorm input (p.bnr_raw)
    unsigned bw[4], rbw[4]; // the rows' words of ReLU bits (p.bnr_bits / p.res_bits)
};
constexpr int EPI_TLD = 36;                          // floats per row of the transposition slice (32 + 4: 144-byte rows, 16-byte aligned)
constexpr int EPI_TSLICE = 32 * EPI_TLD;             // floats per wave

// workgroup barrier for LDS hand-offs ONLY: __syncthreads() also drains the vector-memory counter (its release fence), i.e. it would wait
// for the operands requested ahead and for every result store to be acknowledged -- exactly the round trips this epilogue exists to avoid
__device__ __forceinline__ void epi_lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// DEPTH: units whose operands may be in flight at once (40 registers each).  init() copies the handful of wave-uniform values the epilogue needs
// out of the kernel argument block (scalar registers): a reference to the 800-byte ConvP kept in a member, or captured by the per-row
// lambdas, makes the compiler materialise the whole block in scratch.
// (eight-wave workgroups share a SIMD's 512 registers between two waves: their four-unit tiles keep two units in flight)
// PF: how many of them are requested under the last chunk's matrix work (the fragments of that chunk are still live there: one unit for the
// eight-wave four-unit tile, all of them elsewhere)
template <int BM, int BN, int WGM, int WGN, int DEPTH = (WGM * WGN >= 8 && (BM / WGM / 32) * (BN / WGN / 32) > 2) ? 2 : (BM / WGM / 32) * (BN / WGN / 32),
          int PF = (WGM * WGN >= 8 && (BM / WGM / 32) * (BN / WGN / 32) > 2) ? 1 : DEPTH>
struct IgemmEpilogue {
    static constexpr int WTM = BM / WGM, WTN = BN / WGN, MI = WTM / 32, NI = WTN / 32, NU = MI * NI;
    static_assert(DEPTH >= 1 && DEPTH <= NU && PF >= 1 && PF <= DEPTH, "look-ahead depth");
    static_assert(WGM * WGN * EPI_TSLICE * 4 <= 40 * 1024 && WGM * 8 * BN * 2 * 8 <= 72 * 1024, "epilogue LDS use exceeds the smallest kernel's allocation");
    int m0, n0, lane, wave, wm, wn, rq, cq;
    bool look, full, remap, bnr;
    EpiUnit u[DEPTH];
    // wave-uniform copies (see above)
    const float *g_raw, *g_res, *g_bsc, *g_bsh, *g_mean;
    const unsigned *g_bits, *g_rbits;
    float* g_y;
    int Cout, relu, OH, OW, omul, oah, oaw, cMh, cMw, cM;

    __device__ __forceinline__ void init(const ConvP& p, const ConvP::Class& c, int m0_, int n0_) {
        const int tid = threadIdx.x;
        wave = tid >> 6;
        m0 = m0_; n0 = n0_;
        lane = tid & 63; wm = wave / WGN; wn = wave % WGN;
        rq = lane >> 3; cq = (lane & 7) * 4;
        bnr = p.bnr_raw != nullptr;
        full = m0 + BM <= c.M;
        // (the eval-mode forms -- folded scale / shift, plane output -- and the fp32-activation mask of the fused sums, the A/B reference of the
        //  bit form, keep the row-by-row epilogue: every conditional load of this one sits in issue(), in straight-line code)
        // (and a class without taps -- the dead parity classes of a 1x1 / stride-2 data gradient, dx = addend there -- which has no last chunk to
        //  prefetch under: a second prefetch site behind the loop would merge two definitions of unit 0 and wait for the loads where they meet)
        look = p.epi != 0 && p.yplanes == nullptr && p.y != nullptr && full && p.scale == nullptr && p.bnr_out == nullptr && c.ntaps > 0;
        remap = p.omul != 1 || c.oah != 0 || c.oaw != 0 || p.OH != c.Mh || p.OW != c.Mw;
        g_raw = p.bnr_raw; g_res = p.res; g_bsc = p.bnr_sc; g_bsh = p.bnr_sh; g_mean = p.bnr_mean;
        g_bits = p.bnr_bits; g_rbits = p.res_bits; g_y = p.y;
        Cout = p.Cout; relu = p.relu; OH = p.OH; OW = p.OW; omul = p.omul; oah = c.oah; oaw = c.oaw; cMh = c.Mh; cMw = c.Mw; cM = c.M;
    }

    // physical output pixels of the lane's four rows (rq + 8 t) of block row i: a remapped class -- a parity class of a stride-2 data gradient --
    // finds its first pixel by division and walks on eight logical rows at a time
    __device__ __forceinline__ void pixels(int i, int (&pix)[4]) const {
        const int mb = m0 + wm * WTM + i * 32 + rq;
        if (!remap) {
#pragma unroll
            for (int t = 0; t < 4; ++t) pix[t] = mb + 8 * t;
        } else {
            const int MhMw = cMh * cMw;
            int b_ = mb / MhMw;
            const int rem = mb - b_ * MhMw;
            int ho_ = rem / cMw, wo_ = rem - ho_ * cMw;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                pix[t] = (b_ * OH + ho_ * omul + oah) * OW + wo_ * omul + oaw;
                wo_ += 8;
                while (wo_ >= cMw) {
                    wo_ -= cMw;
                    if (++ho_ == cMh) { ho_ = 0; ++b_; }
                }
            }
        }
    }
    // the unit's operands: first the row offsets (integer work), then the loads, in ONE straight-line block: loads issued inside the two arms
    // of a branch would be awaited where the arms meet (their destination registers are merged there)
    __device__ __forceinline__ void issue(int i, int j, EpiUnit& un) const {
        const int nb = n0 + wn * WTN + j * 32 + cq;
        int pix[4];
        pixels(i, pix);
        const int wcol = (n0 + wn * WTN + j * 32) >> 5, wld = Cout >> 5;
        if (bnr) {
#pragma unroll
            for (int t = 0; t < 4; ++t) un.xr[t] = *reinterpret_cast<const f32x4*>(g_raw + pix[t] * Cout + nb);
        }
        if (g_res) {
#pragma unroll
            for (int t = 0; t < 4; ++t) un.rv[t] = *reinterpret_cast<const f32x4*>(g_res + pix[t] * Cout + nb);
        }
        if (g_bits) {
#pragma unroll
            for (int t = 0; t < 4; ++t) un.bw[t] = g_bits[pix[t] * wld + wcol];
        }
        if (g_rbits) {
#pragma unroll
            for (int t = 0; t < 4; ++t) un.rbw[t] = g_rbits[pix[t] * wld + wcol];
        }
    }
    // the operands of the first DEPTH units, ahead of the last chunk's matrix work.  The kernels call this exactly once, where no copy wait
    // follows (every later s_waitcnt vmcnt of the main loop would wait for these loads too), on a path peeled out of the chunk loop -- inside
    // the loop the units' registers would be live across every iteration.  ALL of them here, not unit 0 alone: behind the loop the compiler
    // drains the vector-memory counter once (it cannot see the hand-written waits that retired the LDS-DMA copies and protects the first LDS
    // access of the epilogue), so whatever is requested later starts a second round trip.
    __device__ __forceinline__ void prefetch() {
        if (!look) return;
        epi_static_for<PF>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            issue(k / NI, k % NI, u[k]);
        });
    }

    // the unit's 32x32 values from the accumulator layout (lane = channel, register r = row (r & 3) + 8 (r >> 2) + 4 (lane >> 5)) into the row
    // layout, through this wave's LDS slice.  LDS operations of one wave execute in issue order: the compiler only has to keep them in
    // program order (no fence: a fence would also drain the vector-memory counter, see epi_lds_barrier)
    __device__ __forceinline__ void transpose(const f32x16& acc, float* T, f32x4 (&v)[4]) const {
        asm volatile("" ::: "memory");
#pragma unroll
        for (int r = 0; r < 16; ++r) T[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * EPI_TLD + (lane & 31)] = acc[r];
        asm volatile("" ::: "memory");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int t = 0; t < 4; ++t) v[t] = *reinterpret_cast<const f32x4*>(T + (rq + 8 * t) * EPI_TLD + cq);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // (the slice is rewritten by the next unit: its reads have returned)
        __builtin_amdgcn_wave_barrier();
    }

    __device__ __forceinline__ void consume(int i, int j, const EpiUnit& un, const f32x16& acc, float* T, f32x4& s1, f32x4& s2, double (&d1)[4], double (&d2)[4],
                                            const f32x4& bsc, const f32x4& bsh, const f32x4& bmu) const {
        f32x4 v[4];
        transpose(acc, T, v);
        const int nb = n0 + wn * WTN + j * 32 + cq;
        int pix[4];
        pixels(i, pix);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float x = v[t][e];
                s1[e] += x;
                s2[e] = fmaf(x, x, s2[e]);
                if (g_res) {
                    float a = un.rv[t][e];
                    if (g_rbits) a = ((un.rbw[t] >> (cq + e)) & 1u) ? a : 0.f;
                    x += a;
                }
                if (relu) x = fmaxf(x, 0.f);
                if (bnr) {
                    const bool on = g_bits ? (((un.bw[t] >> (cq + e)) & 1u) != 0) : fmaf(un.xr[t][e], bsc[e], bsh[e]) > 0.f;
                    const float g = on ? x : 0.f;
                    d1[e] += (double)g;
                    d2[e] += (double)g * ((double)un.xr[t][e] - (double)bmu[e]);
                }
                o[e] = x;
            }
            *reinterpret_cast<f32x4*>(g_y + pix[t] * Cout + nb) = o;
        }
    }

    // PRE: prefetch() has run (units 0 .. DEPTH - 1 are in flight)
    template <bool PRE>
    __device__ __forceinline__ void run(const f32x16 (&acc)[MI][NI], float* smem, f32x4 (&s1)[NI], f32x4 (&s2)[NI], double (&d1)[NI][4], double (&d2)[NI][4]) {
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            s1[j] = f32x4{0.f, 0.f, 0.f, 0.f}; s2[j] = s1[j];
#pragma unroll
            for (int e = 0; e < 4; ++e) { d1[j][e] = 0.0; d2[j][e] = 0.0; }
        }
        // per-channel constants of the fused sums (mean; scale / shift of the recomputed mask): loaded here, in front of everything that waits
        f32x4 bsc[NI], bsh[NI], bmu[NI];
        if (bnr) {
#pragma unroll
            for (int j = 0; j < NI; ++j) bmu[j] = *reinterpret_cast<const f32x4*>(g_mean + n0 + wn * WTN + j * 32 + cq);
            if (!g_bits) {
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    bsc[j] = *reinterpret_cast<const f32x4*>(g_bsc + n0 + wn * WTN + j * 32 + cq);
                    bsh[j] = *reinterpret_cast<const f32x4*>(g_bsh + n0 + wn * WTN + j * 32 + cq);
                }
            }
        }
        // units 0 .. DEPTH - 1 are in flight; unit k + DEPTH is requested as soon as unit k's registers are free (slot k % DEPTH): every load
        // is issued in front of the stores of the units before it, none behind a store it would have to wait for
        epi_static_for<DEPTH - (PRE ? PF : 0)>([&](auto kc) {
            constexpr int k = decltype(kc)::value + (PRE ? PF : 0);
            issue(k / NI, k % NI, u[k]);
        });
        epi_lds_barrier();                                 // every wave is done with the last chunk's fragment reads: LDS is free
        float* T = smem + wave * EPI_TSLICE;
        epi_static_for<NU>([&](auto kc) {
            constexpr int k = decltype(kc)::value, i = k / NI, j = k % NI;
            consume(i, j, u[k % DEPTH], acc[i][j], T, s1[j], s2[j], d1[j], d2[j], bsc[j], bsh[j], bmu[j]);
            if constexpr (k + DEPTH < NU) issue((k + DEPTH) / NI, (k + DEPTH) % NI, u[k % DEPTH]);
        });
    }

    // per-channel (sum, sum of squares) partials of this M tile -> stats[mt][Cout][2]: lane (rq, cq) holds them for its four channels; the eight
    // row lanes and the M waves are added in a fixed order by thread c < BN
    __device__ __forceinline__ void row_stats(float* stats, int mt, float* smem, const f32x4 (&s1)[NI], const f32x4 (&s2)[NI]) const {
        if (!stats) return;
        epi_lds_barrier();                                 // (the transposition slices are dead)
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int c = wn * WTN + j * 32 + cq;
            *reinterpret_cast<f32x4*>(smem + ((0 * WGM + wm) * 8 + rq) * BN + c) = s1[j];
            *reinterpret_cast<f32x4*>(smem + ((1 * WGM + wm) * 8 + rq) * BN + c) = s2[j];
        }
        epi_lds_barrier();
        const int tid = threadIdx.x;
        if (tid < BN) {
            float t1 = 0.f, t2 = 0.f;
#pragma unroll
            for (int w = 0; w < WGM * 8; ++w) { t1 += smem[w * BN + tid]; t2 += smem[(WGM * 8 + w) * BN + tid]; }
            float* o = stats + ((long long)mt * Cout + n0 + tid) * 2;
            o[0] = t1;
            o[1] = t2;
        }
    }
    // BatchNorm-backward partial of this M tile -> part[blk][Cout][2] = (S1, invstd * S2), the same way in double
    __device__ __forceinline__ void row_bnr(double* part, const float* invstd, int blk, float* smem, const double (&d1)[NI][4], const double (&d2)[NI][4]) const {
        if (!bnr) return;
        epi_lds_barrier();
        double* red = reinterpret_cast<double*>(smem);
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int c = wn * WTN + j * 32 + cq + e;
                red[((0 * WGM + wm) * 8 + rq) * BN + c] = d1[j][e];
                red[((1 * WGM + wm) * 8 + rq) * BN + c] = d2[j][e];
            }
        epi_lds_barrier();
        const int tid = threadIdx.x;
        if (tid < BN) {
            double t1 = 0.0, t2 = 0.0;
#pragma unroll
            for (int w = 0; w < WGM * 8; ++w) { t1 += red[w * BN + tid]; t2 += red[(WGM * 8 + w) * BN + tid]; }
            double* o = part + ((long long)blk * Cout + n0 + tid) * 2;
            o[0] = t1;
            o[1] = t2 * (double)invstd[n0 + tid];
        }
    }

    // the whole epilogue of the tile: rows, batch-statistics partial, BatchNorm-backward partial -- look-ahead row form where it applies, else
    // the row-by-row form of rounds 1-4
    template <bool PRE = true>
    __device__ __forceinline__ void finish(const ConvP& p, const ConvP::Class& c, const f32x16 (&acc)[MI][NI], int mt, int bnr_blk, float* smem) {
        if (!look) {
            float s1[NI], s2[NI];
            double bd1[NI], bd2[NI];
            igemm_store_rows_impl<BM, BN, WGM, WGN, true>(p, c, acc, m0, n0, s1, s2, bd1, bd2);
            igemm_store_stats<BM, BN, WGM, WGN>(p, s1, s2, mt, n0, smem);
            igemm_store_bnr<BM, BN, WGM, WGN>(p, bd1, bd2, bnr_blk, n0, smem);
            return;
        }
        f32x4 s1[NI], s2[NI];
        double d1[NI][4], d2[NI][4];
        run<PRE>(acc, smem, s1, s2, d1, d2);
        row_stats(p.stats, mt, smem, s1, s2);
        row_bnr(p.bnr_part, p.bnr_invstd, bnr_blk, smem, d1, d2);
    }
};

template <int BM, int BN, int WGM = 2, int WGN = 2>
__device__ __forceinline__ void igemm_store_rows(const ConvP& p, const ConvP::Class& c, const f32x16 (&acc)[BM / WGM / 32][BN / WGN / 32], int m0, int n0,
                                                 float (&s1)[BN / WGN / 32], float (&s2)[BN / WGN / 32]) {
    double d1[BN / WGN / 32], d2[BN / WGN / 32];          // (unused)
    igemm_store_rows_impl<BM, BN, WGM, WGN, false>(p, c, acc, m0, n0, s1, s2, d1, d2);
}

// BatchNorm-backward partial of one M tile -> bnr_part[blk][Cout][2] = (S1, invstd * S2), summed in a fixed order
template <int BM, int BN, int WGM, int WGN>
__device__ __forceinline__ void igemm_store_bnr(const ConvP& p, const double (&bd1)[BN / WGN / 32], const double (&bd2)[BN / WGN / 32], int blk, int n0,
                                                float* smem) {
    constexpr int WTN = BN / WGN, NI = WTN / 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    if (!p.bnr_raw) return;
    __syncthreads();   // all fragment reads of the last chunk are done: LDS is free
    double* red = reinterpret_cast<double*>(smem);   // [WGM (wm)][BN][2]
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const double u1 = bd1[j] + __shfl_xor(bd1[j], 32, 64);
        const double u2 = bd2[j] + __shfl_xor(bd2[j], 32, 64);
        if (lane < 32) {
            const int c = wn * WTN + j * 32 + lane;
            red[(wm * BN + c) * 2 + 0] = u1;
            red[(wm * BN + c) * 2 + 1] = u2;
        }
    }
    __syncthreads();
    if (tid < BN) {
        double t1 = red[tid * 2 + 0], t2 = red[tid * 2 + 1];
#pragma unroll
        for (int w = 1; w < WGM; ++w) { t1 += red[(w * BN + tid) * 2 + 0]; t2 += red[(w * BN + tid) * 2 + 1]; }
        double* o = p.bnr_part + ((long long)blk * p.Cout + n0 + tid) * 2;
        o[0] = t1;
        o[1] = t2 * (double)p.bnr_invstd[n0 + tid];
    }
}

// per-channel (sum, sum of squares) partials of one M tile -> stats[mt][Cout][2]
template <int BM, int BN, int WGM, int WGN>
__device__ __forceinline__ void igemm_store_stats(const ConvP& p, const float (&s1)[BN / WGN / 32], const float (&s2)[BN / WGN / 32], int mt, int n0, float* smem) {
    constexpr int WTN = BN / WGN, NI = WTN / 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    if (p.stats) {
        // lanes l and l+32 hold the same channel; the M-waves are combined through LDS
        __syncthreads();   // all fragment reads of the last chunk are done: LDS is free
        float* red = smem;   // [WGM (wm)][BN][2]
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const float u1 = s1[j] + __shfl_xor(s1[j], 32, 64);
            const float u2 = s2[j] + __shfl_xor(s2[j], 32, 64);
            if (lane < 32) {
                const int c = wn * WTN + j * 32 + lane;
                red[(wm * BN + c) * 2 + 0] = u1;
                red[(wm * BN + c) * 2 + 1] = u2;
            }
        }
        __syncthreads();
        if (tid < BN) {
            float* o = p.stats + ((long long)mt * p.Cout + n0 + tid) * 2;
            float t1 = red[tid * 2 + 0], t2 = red[tid * 2 + 1];
#pragma unroll
            for (int w = 1; w < WGM; ++w) { t1 += red[(w * BN + tid) * 2 + 0]; t2 += red[(w * BN + tid) * 2 + 1]; }
            o[0] = t1;
            o[1] = t2;
        }
    }
}

// geometry of the forward convolution (the operand pointers p.x / p.w are set by the caller)
inline int conv_fwd_problem(ConvP& p, const float* scale, const float* shift, const float* residual, int relu, float* y, float* stats_partial,
                            int batch, int h, int wdt, int cin, int cout, int kh, int kw, int stride, int pad) {
    p.scale = scale; p.shift = shift; p.res = residual; p.y = y; p.stats = stats_partial;
    p.bnr_raw = p.bnr_out = p.bnr_sc = p.bnr_sh = p.bnr_mean = p.bnr_invstd = nullptr; p.bnr_part = nullptr; p.bnr_bits = p.res_bits = nullptr;
    p.yplanes = nullptr; p.yps = 0; p.clk = straps_clk_acc_current(); p.epi = STRAPS_TOOL_ENV_INT("STRAPS_EPI", 1);
    p.H = h; p.W = wdt; p.Cin = cin; p.Cout = cout; p.relu = relu; p.stride = stride;
    ConvP::Class& c = p.cls[0];
    p.ncls = 1;
    c.Mh = (h + 2 * pad - kh) / stride + 1;
    c.Mw = (wdt + 2 * pad - kw) / stride + 1;
    p.OH = c.Mh; p.OW = c.Mw; p.omul = 1; c.oah = 0; c.oaw = 0;
    c.ntaps = p.wtaps = kh * kw;
    for (int r = 0; r < kh; ++r)
        for (int s = 0; s < kw; ++s) { c.tap_w[r * kw + s] = r * kw + s; c.tap_dh[r * kw + s] = r - pad; c.tap_dw[r * kw + s] = s - pad; }
    const long long M = (long long)batch * c.Mh * c.Mw;
    STRAPS_REQUIRE(M < (1LL << 31) && (long long)batch * h * wdt * cin < (1LL << 31) && M * cout < (1LL << 31),
                   "straps_conv_fwd: tensors must stay below 2^31 elements (32-bit offsets)");
    c.M = (int)M;
    p.xrows = batch * h * wdt;
    p.yrows = (int)M;
    return STRAPS_OK;
}

// geometry of the data gradient: an ordinary convolution over dy for stride 1, one class per output parity for stride 2, each with
// exactly the taps whose source position is integral (see straps_conv_dgrad)
inline int conv_dgrad_problem(ConvP& p, const float* addend, float* dx, int batch, int h, int wdt, int cin, int cout, int kh, int kw,
                              int stride, int pad) {
    const int ho = (h + 2 * pad - kh) / stride + 1, wo = (wdt + 2 * pad - kw) / stride + 1;
    const int padh = kh - 1 - pad, padw = kw - 1 - pad;
    p.scale = nullptr; p.shift = nullptr; p.res = addend; p.y = dx; p.stats = nullptr;
    p.bnr_raw = p.bnr_out = p.bnr_sc = p.bnr_sh = p.bnr_mean = p.bnr_invstd = nullptr; p.bnr_part = nullptr; p.bnr_bits = p.res_bits = nullptr;
    p.yplanes = nullptr; p.yps = 0; p.clk = straps_clk_acc_current(); p.epi = STRAPS_TOOL_ENV_INT("STRAPS_EPI", 1);
    p.H = ho; p.W = wo; p.Cin = cout; p.Cout = cin; p.relu = 0; p.stride = 1;
    p.OH = h; p.OW = wdt; p.wtaps = kh * kw;
    p.omul = stride;
    p.xrows = batch * ho * wo;
    p.yrows = batch * h * wdt;
    p.ncls = 0;
    for (int ph = 0; ph < stride; ++ph) {
        for (int pw = 0; pw < stride; ++pw) {
            ConvP::Class& c = p.cls[p.ncls];
            c.Mh = (h - ph + stride - 1) / stride;
            c.Mw = (wdt - pw + stride - 1) / stride;
            if (c.Mh <= 0 || c.Mw <= 0) continue;
            c.oah = ph; c.oaw = pw;
            c.ntaps = 0;
            for (int r = 0; r < kh; ++r) {
                const int nh = ph - padh + r;                 // source row numerator of logical row 0
                if (((nh % stride) + stride) % stride) continue;
                for (int s = 0; s < kw; ++s) {
                    const int nw = pw - padw + s;
                    if (((nw % stride) + stride) % stride) continue;
                    c.tap_w[c.ntaps] = r * kw + s;
                    c.tap_dh[c.ntaps] = (nh - (((nh % stride) + stride) % stride)) / stride;   // exact: nh divisible
                    c.tap_dw[c.ntaps] = (nw - (((nw % stride) + stride) % stride)) / stride;
                    // floor division for negative numerators
                    if (nh < 0) c.tap_dh[c.ntaps] = -((-nh) / stride);
                    if (nw < 0) c.tap_dw[c.ntaps] = -((-nw) / stride);
                    ++c.ntaps;
                }
            }
            const long long M = (long long)batch * c.Mh * c.Mw;
            STRAPS_REQUIRE(M < (1LL << 31) && (long long)batch * ho * wo * cout < (1LL << 31) && (long long)batch * h * wdt * cin < (1LL << 31),
                           "straps_conv_dgrad: tensors must stay below 2^31 elements (32-bit offsets)");
            c.M = (int)M;
            ++p.ncls;
        }
    }
    // heaviest class first (blockIdx.y = 0 is dispatched first): the 4-tap class of a 3x3/s2 gradient runs four times as long per
    // workgroup as the 1-tap one -- started last it would be the launch's tail
    for (int a = 1; a < p.ncls; ++a)
        for (int b = a; b > 0 && (long long)p.cls[b].ntaps * p.cls[b].M > (long long)p.cls[b - 1].ntaps * p.cls[b - 1].M; --b) {
            const ConvP::Class t = p.cls[b];
            p.cls[b] = p.cls[b - 1];
            p.cls[b - 1] = t;
        }
    return STRAPS_OK;
}

}  // namespace
