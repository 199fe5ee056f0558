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
    // fp32 A operand (conv_x3f.hip, round 6): x is the fp32 tensor itself; a_scale / a_shift (NULL: none) = the producer's BatchNorm applied to it in
    // the operand path, then ReLU if a_relu.  The plane kernels ignore the three fields.
    const float* a_scale;
    const float* a_shift;
    int a_relu;
    int abl;                        // tools build only (STRAPS_X3F_ABL: ablations of conv_x3f.hip's kernels, wrong results by design); 0 in the product
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

// workgroup barrier for LDS hand-offs ONLY: __syncthreads() also drains the vector-memory counter (its release fence), i.e. it would wait for
// loads requested ahead of their use and for every result store to be acknowledged by memory
__device__ __forceinline__ void epi_lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
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

// ---- round 5: the look-ahead epilogue ------------------------------------------------------------------------------------------------
// What the row-by-row epilogue above costs when it has memory OPERANDS (a data gradient's addend; the BatchNorm input `raw` and the ReLU
// bits of the fused BatchNorm-backward sums): per 32-row block it fetches the addend rows, waits, then twice (fetch eight raw rows + eight
// bit words, wait, compute, store) -- and because loads and stores retire through ONE in-order counter on this part, every wait behind a
// batch of stores also waits for those stores to be acknowledged by memory.  Six dependent memory round trips per workgroup of a 128-row
// tile, 1.5-2.5 us each under load, at the end of a main loop of 7-14 us, with two workgroups per CU to hide them: `dgrad+bn` launches
// ran 16-54 us behind the forward convolution of the same FLOPs (190 vs 122 us on layer1's 64 -> 64 3x3, profiles/r04_bench_train_b64.json).
// Here a UNIT is one 32x32 accumulator block (i, j) of a wave -- 16 elements per lane -- and its operands (16 addend values, 16 raw values,
// one word of ReLU bits per operand tensor) are fetched AHEAD of their use: the first PF units before the matrix work of the last K chunk
// (`prefetch()`, called by the kernels where no copy wait follows any more), unit k + DEPTH as soon as unit k is computed and stored.  No load
// ever waits behind a store (a younger store does not hold back an older load in the in-order counter), and one round trip -- under the last
// chunk's MFMAs -- is exposed instead of six.  Three details decide whether that works (each found in the ISA, round 5):
//   * the loads of a unit sit in ONE straight-line block behind the integer work that finds its pixels: loads issued in the two arms of a
//     branch (remapped class or not) are awaited where the arms meet, because their destination registers are merged there;
//   * ONE prefetch site: a second one behind the loop (for classes without taps) merges two definitions of the same registers -- same wait;
//   * the workgroup barriers of the epilogue are LDS-only (epi_lds_barrier): __syncthreads() drains the vector-memory counter too, i.e. it waits
//     for the operands requested ahead and for every result store to be acknowledged.  Same arithmetic in the same order per lane as the row-by-row form: results are bit-identical
// (tests/test_gpu_conv_x3.py::test_relu_bits_forms_equal_the_fp32_mask_forms_bit_for_bit runs the fp32-mask form -- still row by row -- against it
// on every tile configuration).
//
// ReLU bits: the word of (pixel row, 32-channel group) is the same for the 32 lanes of a row.  The row-by-row form loads it once per (lane,
// row): 16 broadcast loads and 16 registers per unit and operand.  Here lane q = l & 15 of each 16-lane DPP row loads the word of the block
// row its HALF-wave will need as its q-th -- (q & 3) + 8 (q >> 2) + 4 (l >> 5), the accumulator layout's row order -- and element r reads it
// with `v_mov_b32_dpp row_newbcast:r`: one load and one register per unit and operand.
template <int R>
__device__ __forceinline__ unsigned epi_row_word(unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x150 + R, 0xf, 0xf, false); }
template <typename F, int... Rs>
__device__ __forceinline__ void epi_for16(F&& f, std::integer_sequence<int, Rs...>) { (f(std::integral_constant<int, Rs>{}), ...); }
// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>): expanded by the front end (a `#pragma unroll` loop over units is NOT reliable
// here -- one unit's body is a few thousand instructions, and beyond the pragma-unroll threshold the loop stays a loop, its unit and accumulator
// arrays get indexed at run time and land in scratch memory)
template <int N, typename F>
__device__ __forceinline__ void epi_static_for(F&& f) { epi_for16(f, std::make_integer_sequence<int, N>{}); }

__device__ __attribute__((aligned(16))) const float k_epi_consts[2] = {0.f, 1.f};      // stand-ins for absent per-channel tensors (IgemmEpilogue::constants)

#ifndef STRAPS_EPI_PF8
#define STRAPS_EPI_PF8 1
#endif
struct EpiUnit {
    float rv[16];       // addend (p.res) of the lane's 16 rows
    float xr[16];       // BatchNorm input (p.bnr_raw)
    unsigned bw, rbw;   // ReLU-bit words (p.bnr_bits / p.res_bits), one block row per lane of a DPP row (see above)
};

// DEPTH: units whose operands may be in flight at once (34 registers each); PF: how many of them are requested under the last chunk's matrix
// work (the fragments of that chunk are still live there).  init() copies the handful of wave-uniform values the epilogue needs out of the
// kernel argument block (scalar registers): a reference to the 800-byte ConvP kept in a member, or captured by the per-row lambdas, makes the
// compiler materialise the whole block in scratch.
// (eight-wave workgroups share a SIMD's 512 registers between two waves: their four-unit tiles keep two units in flight, one of them early)
template <int BM, int BN, int WGM, int WGN, int DEPTH = (WGM * WGN >= 8 && (BM / WGM / 32) * (BN / WGN / 32) > 2) ? 2 : (BM / WGM / 32) * (BN / WGN / 32),
          int PF = (WGM * WGN >= 8 && (BM / WGM / 32) * (BN / WGN / 32) > 2) ? STRAPS_EPI_PF8 : DEPTH>
struct IgemmEpilogue {
    static constexpr int WTM = BM / WGM, WTN = BN / WGN, MI = WTM / 32, NI = WTN / 32, NU = MI * NI;
    static_assert(DEPTH >= 1 && DEPTH <= NU && PF >= 1 && PF <= DEPTH, "look-ahead depth");
    int m0, n0, lane, wm, wn;
    bool look, full, remap, bnr;
    EpiUnit u[DEPTH];
    float k_sc[NI], k_sh[NI], k_bsc[NI], k_bsh[NI], k_bmu[NI];      // per-channel constants of the lane's NI columns (constants())
    // wave-uniform copies (see above)
    const float *g_raw, *g_res, *g_scale, *g_shift, *g_bsc, *g_bsh, *g_mean;
    const unsigned *g_bits, *g_rbits;
    float* g_y;
    int Cout, relu, OH, OW, omul, oah, oaw, cMh, cMw, cM;

    __device__ __forceinline__ void init(const ConvP& p, const ConvP::Class& c, int m0_, int n0_) {
        const int tid = threadIdx.x, wave = tid >> 6;
        m0 = m0_; n0 = n0_;
        lane = tid & 63; wm = wave / WGN; wn = wave % WGN;
        bnr = p.bnr_raw != nullptr;
        // the look-ahead form covers every launch that HAS memory operands, except: the fp32-activation mask of the BatchNorm sums (bnr_out: the
        // A/B reference of the bit form) and the eval-mode plane output (yplanes), kept row by row; a ragged last M tile (one workgroup row of
        // a launch at most), which keeps the row-by-row form with its per-row range tests; and a class without taps (the dead parity classes of a
        // 1x1 / stride-2 data gradient: dx = addend there), which has no last chunk to prefetch under
        full = m0 + BM <= c.M;
        look = p.epi != 0 && (bnr || p.res != nullptr) && p.bnr_out == nullptr && p.yplanes == nullptr && p.y != nullptr && full && c.ntaps > 0;
        remap = p.omul != 1 || c.oah != 0 || c.oaw != 0 || p.OH != c.Mh || p.OW != c.Mw;
        g_raw = p.bnr_raw; g_res = p.res; g_scale = p.scale; g_shift = p.shift; g_bsc = p.bnr_sc; g_bsh = p.bnr_sh; g_mean = p.bnr_mean;
        g_bits = p.bnr_bits; g_rbits = p.res_bits; g_y = p.y;
        Cout = p.Cout; relu = p.relu; OH = p.OH; OW = p.OW; omul = p.omul; oah = c.oah; oaw = c.oaw; cMh = c.Mh; cMw = c.Mw; cM = c.M;
    }

    // per-channel constants (folded scale / shift of the eval forms; mean and the recomputed mask's scale / shift of the fused sums).  Loads
    // whose result would be merged with a literal where a flag is off are awaited at the merge: the POINTER is selected instead (a pair of
    // constants stands in for the absent tensor) and every load is unconditional
    __device__ __forceinline__ void constants() {
        const float* one = k_epi_consts + 1;
        const float* zero = k_epi_consts;
        asm volatile("" : "+s"(one), "+s"(zero));          // (opaque: otherwise the selects fold back into branches around the loads)
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int n = n0 + wn * WTN + j * 32 + (lane & 31);
            k_sc[j] = *(g_scale ? g_scale + n : one);
            k_sh[j] = *(g_shift ? g_shift + n : zero);
            k_bsc[j] = *((bnr && !g_bits) ? g_bsc + n : zero);
            k_bsh[j] = *((bnr && !g_bits) ? g_bsh + n : zero);
            k_bmu[j] = *(bnr ? g_mean + n : zero);
        }
    }

    // physical output pixels of the lane's 16 rows of block row i (rows m = mb + (r & 3) + 8 (r >> 2)).  A remapped class (a parity class of a
    // stride-2 data gradient) finds the first pixel by division and walks on, as igemm_store_rows_impl does.  Integer work only.
    __device__ __forceinline__ void pixels(int i, int (&pix)[16]) const {
        const int mb = m0 + wm * WTM + i * 32 + 4 * (lane >> 5);
        if (!remap) {
#pragma unroll
            for (int r = 0; r < 16; ++r) pix[r] = mb + (r & 3) + 8 * (r >> 2);
        } else {
            const int MhMw = cMh * cMw;
            int b_ = mb / MhMw;
            const int rem = mb - b_ * MhMw;
            int ho_ = rem / cMw, wo_ = rem - ho_ * cMw;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                pix[r] = (b_ * OH + ho_ * omul + oah) * OW + wo_ * omul + oaw;
                wo_ += (r & 3) == 3 ? 5 : 1;
                while (wo_ >= cMw) {
                    wo_ -= cMw;
                    if (++ho_ == cMh) { ho_ = 0; ++b_; }
                }
            }
        }
    }
    // the unit's operands (full tiles only): the pixels first, then every load in one straight-line block
    __device__ __forceinline__ void issue(int i, int j, EpiUnit& un) const {
        const int n = n0 + wn * WTN + j * 32 + (lane & 31);
        int pix[16];
        pixels(i, pix);
        int pixl = 0;
        if (g_bits || g_rbits) {
            // the block row whose bit words this lane fetches: the q-th row of its half-wave, q = lane & 15
            const int q = lane & 15;
            const int ml = m0 + wm * WTM + i * 32 + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
            pixl = ml;
            if (remap) {
                const int MhMw = cMh * cMw;
                const int b_ = ml / MhMw, rem = ml - b_ * MhMw;
                const int ho_ = rem / cMw, wo_ = rem - ho_ * cMw;
                pixl = (b_ * OH + ho_ * omul + oah) * OW + wo_ * omul + oaw;
            }
        }
        if (bnr) {
#pragma unroll
            for (int r = 0; r < 16; ++r) un.xr[r] = g_raw[pix[r] * Cout + n];
        }
        if (g_res) {
#pragma unroll
            for (int r = 0; r < 16; ++r) un.rv[r] = g_res[pix[r] * Cout + n];
        }
        const int wi = pixl * (Cout >> 5) + ((n0 + wn * WTN + j * 32) >> 5);
        if (g_bits) un.bw = g_bits[wi];
        if (g_rbits) un.rbw = g_rbits[wi];
    }
    // the operands of the first PF units, ahead of the last chunk's matrix work.  The kernels call this exactly once, where no copy wait follows
    // (every later s_waitcnt vmcnt of the main loop would wait for these loads too), on a path peeled out of the chunk loop -- inside the loop
    // the units' registers would be live across every iteration.
    __device__ __forceinline__ void prefetch() {
        if (!look) return;
        if constexpr (PF == DEPTH) constants();      // (the eight-wave four-unit tile has no registers to spare under its last chunk: it loads them behind the loop)
        epi_static_for<PF>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            issue(k / NI, k % NI, u[k]);
        });
    }

    __device__ __forceinline__ void consume(int i, int j, const EpiUnit& un, const f32x16& acc, float& s1, float& s2, double& d1, double& d2,
                                            float sc, float sh, float bsc, float bsh, float bmu) const {
        const int cl = lane & 31;
        const int n = n0 + wn * WTN + j * 32 + cl;
        int pix[16];
        pixels(i, pix);
        epi_static_for<16>([&](auto rc) {
            constexpr int r = decltype(rc)::value;
            float v = acc[r];
            s1 += v;
            s2 = fmaf(v, v, s2);
            if (g_scale) v = fmaf(v, sc, sh);
            if (g_res) {
                float a = un.rv[r];
                if (g_rbits) a = ((epi_row_word<r>(un.rbw) >> cl) & 1u) ? a : 0.f;
                v += a;
            }
            if (relu) v = fmaxf(v, 0.f);
            if (bnr) {
                const bool on = g_bits ? (((epi_row_word<r>(un.bw) >> cl) & 1u) != 0) : fmaf(un.xr[r], bsc, bsh) > 0.f;
                const float g = on ? v : 0.f;
                d1 += (double)g;
                d2 += (double)g * ((double)un.xr[r] - (double)bmu);
            }
            g_y[pix[r] * Cout + n] = v;
        });
    }

    // PRE: prefetch() has run (units 0 .. PF - 1 are in flight)
    template <bool PRE>
    __device__ __forceinline__ void run(const f32x16 (&acc)[MI][NI], float (&s1)[NI], float (&s2)[NI], double (&d1)[NI], double (&d2)[NI]) {
        // the rest of the first DEPTH units; afterwards unit k + DEPTH is requested as soon as unit k's registers are free (slot k % DEPTH): every
        // load is issued in front of the stores of the units before it, none behind a store it would have to wait for
        epi_static_for<DEPTH - (PRE ? PF : 0)>([&](auto kc) {
            constexpr int k = decltype(kc)::value + (PRE ? PF : 0);
            issue(k / NI, k % NI, u[k]);
        });
        if constexpr (!PRE || PF != DEPTH) constants();
#pragma unroll
        for (int j = 0; j < NI; ++j) { s1[j] = 0.f; s2[j] = 0.f; d1[j] = 0.0; d2[j] = 0.0; }
        epi_static_for<NU>([&](auto kc) {
            constexpr int k = decltype(kc)::value, i = k / NI, j = k % NI;
            consume(i, j, u[k % DEPTH], acc[i][j], s1[j], s2[j], d1[j], d2[j], k_sc[j], k_sh[j], k_bsc[j], k_bsh[j], k_bmu[j]);
            if constexpr (k + DEPTH < NU) issue((k + DEPTH) / NI, (k + DEPTH) % NI, u[k % DEPTH]);
        });
    }

    // the whole epilogue of the tile's rows: look-ahead form where it applies, else the row-by-row form
    template <bool PRE = true>
    __device__ __forceinline__ void finish(const ConvP& p, const ConvP::Class& c, const f32x16 (&acc)[MI][NI], float (&s1)[NI], float (&s2)[NI],
                                           double (&d1)[NI], double (&d2)[NI]) {
        if (!look) {
            igemm_store_rows_impl<BM, BN, WGM, WGN, true>(p, c, acc, m0, n0, s1, s2, d1, d2);
            return;
        }
        run<PRE>(acc, s1, s2, d1, d2);
    }
};

// ---- lean epilogues (round 6) -------------------------------------------------------------------------------------------------------
// The shared epilogue above is one code path for every fused form (eval scale / shift / residual / ReLU, plane outputs, addend with fp32 or bit
// masks, BatchNorm sums with three mask forms), selected by wave-uniform RUN-TIME flags per value: 40-60 instructions per 64 results.  Behind a
// long reduction that is a few per cent; behind the K = 64 ... 256 reductions of resnet50's 1x1 layers it was the kernel (counters of the 64 -> 256
// forward at layer1's size, tools/r06_gpu_5.sh / profiles/r06_x3f_pmc.txt: ~1 300 instructions per wave and 64-row tile against 48 MFMAs, two
// waves per SIMD issuing one instruction per 4 cycles each -- with every memory access and every MFMA ablated the launch still took 48 of its
// 75 us).  The two forms a TRAINING step uses are therefore written out with compile-time structure:
//   EPI = 1  forward: raw result + per-channel (sum, sum of squares) partials -- per value one add, one fma, one store whose address is a
//            wave-uniform row base (scalar registers, scalar ALU) + one per-lane 32-bit offset;
//   EPI = 2  data gradient: + addend (optionally masked by ReLU bits) and the fused BatchNorm-backward sums (mask from bits or re-derived from
//            raw).  A row's bit word is wave-uniform per half-wave: the two words of a register's rows are read into scalar registers
//            (v_readlane of one coalesced load per 32 rows) and used directly as the LANE MASK of a v_cndmask; the two sums are accumulated in
//            fp32 over the 16 values of a unit and added to the lane's double accumulators once per unit (2 + 3 fp64 operations per unit
//            instead of 5 per value: the double accumulation exists for the cancellation across ~10^5 values of a channel, not across 16);
//            every operand of UG units is requested before the first result store (loads and stores retire through one in-order counter).
// Same per-lane summation order for EPI = 1 as the shared epilogue (bit-identical statistics); EPI = 2's sums differ from it by rounding only.
// Full tiles take the unpredicated form; the one ragged tile of a launch predicates per lane; remapped outputs (the parity classes of a
// stride-2 data gradient) find their physical pixels per lane as the shared epilogue does.  Everything else (eval forms, fp32 masks, plane
// outputs) stays on EPI = 0.
template <int BM, int BN, int WGM, int WGN>
__device__ __forceinline__ void lean_epilogue_fwd(const ConvP& p, const f32x16 (&acc)[BM / WGM / 32][BN / WGN / 32], int m0, int n0, int M,
                                                  float (&s1)[BN / WGN / 32], float (&s2)[BN / WGN / 32]) {
    constexpr int WTM = BM / WGM, WTN = BN / WGN, MI = WTM / 32, NI = WTN / 32;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = __builtin_amdgcn_readfirstlane(wave / WGN), wn = __builtin_amdgcn_readfirstlane(wave % WGN);
    const int Cout = p.Cout;
    const int half4 = 4 * (lane >> 5);
    const int loff = half4 * Cout + (lane & 31);                 // the lane's part of an element offset (32-bit)
    float* const yb = p.y + (n0 + wn * WTN);
    const bool full = m0 + BM <= M;
#pragma unroll
    for (int j = 0; j < NI; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
    if (full) {
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int mrow = m0 + wm * WTM + i * 32;               // wave-uniform
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float* const rowp = yb + (long long)(mrow + (r & 3) + 8 * (r >> 2)) * Cout;      // wave-uniform base
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    const float v = acc[i][j][r];
                    s1[j] += v;
                    s2[j] = fmaf(v, v, s2[j]);
                    rowp[loff + j * 32] = v;
                }
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int mrow = m0 + wm * WTM + i * 32;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rr = (r & 3) + 8 * (r >> 2);
                float* const rowp = yb + (long long)(mrow + rr) * Cout;
                const bool ok = mrow + rr + half4 < M;
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    const float v = acc[i][j][r];
                    if (ok) {
                        s1[j] += v;
                        s2[j] = fmaf(v, v, s2[j]);
                        rowp[loff + j * 32] = v;
                    }
                }
            }
        }
    }
}

// v = mask bit of this lane ? a : 0, the 64-bit lane mask in scalar registers: v_cndmask_b32 with the mask as its select operand.  Through the
// builtin, not inline assembly: gfx950 needs two wait states between a VALU write of a scalar register (the v_readlane that fetched the mask)
// and a VALU read of it -- the compiler's hazard recogniser inserts them, assembly text is invisible to it (the first form of this function
// was an asm statement and selected with stale masks).
__device__ __forceinline__ float lane_masked(float a, unsigned long long mask) { return __builtin_amdgcn_inverse_ballot_w64(mask) ? a : 0.f; }

// UG: units (32 x 32 accumulator blocks: 16 + 16 operand values per lane) whose operands are in flight together; default two (all of a two-unit wave tile)
template <int BM, int BN, int WGM, int WGN, int UG = ((BM / WGM / 32) * (BN / WGN / 32) > 2 ? 2 : (BM / WGM / 32) * (BN / WGN / 32))>
__device__ __forceinline__ void lean_epilogue_dgrad(const ConvP& p, const ConvP::Class& c, const f32x16 (&acc)[BM / WGM / 32][BN / WGN / 32], int m0, int n0,
                                                    double (&d1)[BN / WGN / 32], double (&d2)[BN / WGN / 32]) {
    constexpr int WTM = BM / WGM, WTN = BN / WGN, MI = WTM / 32, NI = WTN / 32, NU = MI * NI;
    static_assert(UG >= 1 && NU % UG == 0, "unit groups");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = __builtin_amdgcn_readfirstlane(wave / WGN), wn = __builtin_amdgcn_readfirstlane(wave % WGN);
    const int M = c.M, Cout = p.Cout, CW = Cout >> 5;
    const int half4 = 4 * (lane >> 5);
    const int cb = n0 + wn * WTN;                                  // first channel of the wave's columns (wave-uniform)
    const bool full = m0 + BM <= M;
    const bool remap = p.omul != 1 || c.oah != 0 || c.oaw != 0 || p.OH != c.Mh || p.OW != c.Mw;
    const bool has_add = p.res != nullptr, has_abits = p.res_bits != nullptr, bnr = p.bnr_raw != nullptr, has_obits = p.bnr_bits != nullptr;
    float bmu[NI], bsc[NI], bsh[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int n = cb + j * 32 + (lane & 31);
        bmu[j] = bnr ? p.bnr_mean[n] : 0.f;
        bsc[j] = (bnr && !has_obits) ? p.bnr_sc[n] : 0.f;
        bsh[j] = (bnr && !has_obits) ? p.bnr_sh[n] : 0.f;
        d1[j] = 0.0;
        d2[j] = 0.0;
    }
    // physical pixel of logical row m (a parity class of a stride-2 data gradient lands on every other pixel of the output; else the row itself)
    auto pixel = [&](int m) {
        m = m < M ? m : M - 1;                                     // (ragged tile: rows behind the end re-read a valid one; their results are masked out)
        if (!remap) return m;
        const int MhMw = c.Mh * c.Mw;
        const int b_ = m / MhMw, rem = m - b_ * MhMw;
        const int ho_ = rem / c.Mw, wo_ = rem - ho_ * c.Mw;
        return (b_ * p.OH + ho_ * p.omul + c.oah) * p.OW + wo_ * p.omul + c.oaw;
    };
#pragma unroll
    for (int g = 0; g < NU / UG; ++g) {
        float rv[UG][16], xr[UG][16];
        unsigned aw[UG], ow[UG];                                   // lane k (and k + 32): the bit word of row mrow + k of the unit's column group
        int po[UG][16];                                            // element offset (32-bit) of the lane's 16 results of a unit
#pragma unroll
        for (int u = 0; u < UG; ++u) {
            const int i = (g * UG + u) / NI, j = (g * UG + u) % NI;
            const int mrow = m0 + wm * WTM + i * 32;               // wave-uniform
            aw[u] = 0xffffffffu; ow[u] = 0xffffffffu;
            {
                const long long wi = (long long)pixel(mrow + (lane & 31)) * CW + ((cb + j * 32) >> 5);
                if (has_abits) aw[u] = p.res_bits[wi];
                if (has_obits) ow[u] = p.bnr_bits[wi];
            }
            const int coff = cb + j * 32 + (lane & 31);
            if (!remap && full) {
#pragma unroll
                for (int r = 0; r < 16; ++r) po[u][r] = (mrow + (r & 3) + 8 * (r >> 2) + half4) * Cout + coff;
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) po[u][r] = pixel(mrow + (r & 3) + 8 * (r >> 2) + half4) * Cout + coff;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                rv[u][r] = has_add ? p.res[po[u][r]] : 0.f;
                xr[u][r] = bnr ? p.bnr_raw[po[u][r]] : 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < UG; ++u) {
            const int i = (g * UG + u) / NI, j = (g * UG + u) % NI;
            const int mrow = m0 + wm * WTM + i * 32;
            float sg = 0.f, sgx = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rr = (r & 3) + 8 * (r >> 2);
                // lane masks of this register's two rows (lanes 0-31: row rr, lanes 32-63: row rr + 4)
                const unsigned long long am = (unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)aw[u], rr) |
                                              ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)aw[u], rr + 4) << 32);
                const float v = acc[i][j][r] + lane_masked(rv[u][r], am);
                const bool ok = full || mrow + rr + half4 < M;
                if (bnr) {
                    float gq;
                    if (has_obits) {
                        const unsigned long long om = (unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)ow[u], rr) |
                                                      ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)ow[u], rr + 4) << 32);
                        gq = lane_masked(v, om);
                    } else {
                        gq = fmaf(xr[u][r], bsc[j], bsh[j]) > 0.f ? v : 0.f;
                    }
                    if (!ok) gq = 0.f;
                    sg += gq;
                    sgx = fmaf(gq, xr[u][r] - bmu[j], sgx);
                }
                if (ok) p.y[po[u][r]] = v;
            }
            d1[j] += (double)sg;
            d2[j] += (double)sgx;
        }
    }
}

// The lean data-gradient epilogue with the shared epilogue's LOOK-AHEAD (plane kernels, conv_x3_lean.hip): two units' operands in flight, the first two
// requested under the last K chunk's matrix work (prefetch(), called where the shared epilogue's is), unit k + 2 as soon as unit k is stored.  PRESUM =
// false: the two BatchNorm sums are accumulated per value in double, in the shared epilogue's order -- bit-identical partials (the plane kernels'
// bits-against-fp32-mask tests compare them bit for bit); true: fp32 over a unit's 16 values first (lean_epilogue_dgrad's arithmetic).
template <int BM, int BN, int WGM, int WGN, bool PRESUM = false>
struct LeanDgradEpilogue {
    static constexpr int WTM = BM / WGM, WTN = BN / WGN, MI = WTM / 32, NI = WTN / 32, NU = MI * NI, DEPTH = NU > 1 ? 2 : 1;
    static constexpr int PF = (WGM * WGN >= 8 && NU > 2) ? 1 : DEPTH;      // (eight-wave tiles: one unit under the last chunk, as the shared epilogue)
    struct Unit { float rv[16], xr[16]; unsigned aw, ow; };
    Unit u[DEPTH];
    int m0, n0, lane, wm, wn, M, Cout, half4, cb;
    bool full, remap, has_add, has_abits, bnr, has_obits, pre;
    const float *g_res, *g_raw;
    const unsigned *g_abits, *g_obits;
    float* g_y;
    int cMh, cMw, OH, OW, omul, oah, oaw;

    __device__ __forceinline__ void init(const ConvP& p, const ConvP::Class& c, int m0_, int n0_) {
        const int wave = threadIdx.x >> 6;
        m0 = m0_; n0 = n0_; lane = threadIdx.x & 63;
        wm = __builtin_amdgcn_readfirstlane(wave / WGN); wn = __builtin_amdgcn_readfirstlane(wave % WGN);
        M = c.M; Cout = p.Cout; half4 = 4 * (lane >> 5); cb = n0 + wn * WTN;
        full = m0 + BM <= M;
        remap = p.omul != 1 || c.oah != 0 || c.oaw != 0 || p.OH != c.Mh || p.OW != c.Mw;
        has_add = p.res != nullptr; has_abits = p.res_bits != nullptr; bnr = p.bnr_raw != nullptr; has_obits = p.bnr_bits != nullptr;
        g_res = p.res; g_raw = p.bnr_raw; g_abits = p.res_bits; g_obits = p.bnr_bits; g_y = p.y;
        cMh = c.Mh; cMw = c.Mw; OH = p.OH; OW = p.OW; omul = p.omul; oah = c.oah; oaw = c.oaw;
        pre = false;
    }
    __device__ __forceinline__ int pixel(int m) const {
        m = m < M ? m : M - 1;
        if (!remap) return m;
        const int MhMw = cMh * cMw;
        const int b_ = m / MhMw, rem = m - b_ * MhMw;
        const int ho_ = rem / cMw, wo_ = rem - ho_ * cMw;
        return (b_ * OH + ho_ * omul + oah) * OW + wo_ * omul + oaw;
    }
    __device__ __forceinline__ void issue(int k, Unit& un) const {
        const int i = k / NI, j = k % NI;
        const int mrow = m0 + wm * WTM + i * 32;
        un.aw = 0xffffffffu; un.ow = 0xffffffffu;
        {
            const long long wi = (long long)pixel(mrow + (lane & 31)) * (Cout >> 5) + ((cb + j * 32) >> 5);
            if (has_abits) un.aw = g_abits[wi];
            if (has_obits) un.ow = g_obits[wi];
        }
        const int coff = cb + j * 32 + (lane & 31);
        int pix[16];
        pixels(mrow, pix);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            un.rv[r] = has_add ? g_res[pix[r] * Cout + coff] : 0.f;
            un.xr[r] = bnr ? g_raw[pix[r] * Cout + coff] : 0.f;
        }
    }
    // physical pixels of the lane's 16 rows of the block row at mrow (rows mrow + half4 + (r & 3) + 8 (r >> 2)); remapped classes: one division, then a walk
    __device__ __forceinline__ void pixels(int mrow, int (&pix)[16]) const {
        const int mb = mrow + half4;
        if (!remap) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { const int m = mb + (r & 3) + 8 * (r >> 2); pix[r] = (full || m < M) ? m : M - 1; }
        } else {
            const int MhMw = cMh * cMw;
            int mm = mb < M ? mb : M - 1;
            int b_ = mm / MhMw;
            const int rem = mm - b_ * MhMw;
            int ho_ = rem / cMw, wo_ = rem - ho_ * cMw;
            // ragged last tile: a row behind the end re-reads the class's LAST pixel (its result is masked out) -- the walk below would carry it past the
            // tensor, and an operand read there (whatever the memory holds: a NaN times a zero gradient is a NaN in the BatchNorm sums) is not masked
            const int plast = full ? 0 : pixel(M - 1);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                pix[r] = (full || mb + (r & 3) + 8 * (r >> 2) < M) ? (b_ * OH + ho_ * omul + oah) * OW + wo_ * omul + oaw : plast;
                wo_ += (r & 3) == 3 ? 5 : 1;
                while (wo_ >= cMw) {
                    wo_ -= cMw;
                    if (++ho_ == cMh) { ho_ = 0; ++b_; }
                }
            }
        }
    }
    // under the last chunk's matrix work (the kernels call this exactly where they call the shared epilogue's prefetch)
    __device__ __forceinline__ void prefetch() {
        epi_static_for<PF>([&](auto kc) { issue(decltype(kc)::value, u[decltype(kc)::value]); });
        pre = true;
    }
    template <bool PRE = true>
    __device__ __forceinline__ void finish(const ConvP& p, const ConvP::Class&, const f32x16 (&acc)[MI][NI], float (&)[NI], float (&)[NI], double (&d1)[NI],
                                           double (&d2)[NI]) {
        float bmu[NI], bsc[NI], bsh[NI];
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int n = cb + j * 32 + (lane & 31);
            bmu[j] = bnr ? p.bnr_mean[n] : 0.f;
            bsc[j] = (bnr && !has_obits) ? p.bnr_sc[n] : 0.f;
            bsh[j] = (bnr && !has_obits) ? p.bnr_sh[n] : 0.f;
            d1[j] = 0.0;
            d2[j] = 0.0;
        }
        if (!pre) epi_static_for<PF>([&](auto kc) { issue(decltype(kc)::value, u[decltype(kc)::value]); });      // (a class without taps never reaches a last chunk)
        epi_static_for<DEPTH - PF>([&](auto kc) { issue(decltype(kc)::value + PF, u[decltype(kc)::value + PF]); });
        epi_static_for<NU>([&](auto kc) {
            constexpr int k = decltype(kc)::value, i = k / NI, j = k % NI;
            Unit& un = u[k % DEPTH];
            const int mrow = m0 + wm * WTM + i * 32;
            const int coff = cb + j * 32 + (lane & 31);
            int pix[16];
            pixels(mrow, pix);
            float sg = 0.f, sgx = 0.f;
            epi_static_for<16>([&](auto rc) {
                constexpr int r = decltype(rc)::value, rr = (r & 3) + 8 * (r >> 2);
                const unsigned long long am = (unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)un.aw, rr) |
                                              ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)un.aw, rr + 4) << 32);
                const float v = acc[i][j][r] + lane_masked(un.rv[r], am);
                const bool ok = full || mrow + rr + half4 < M;
                if (bnr) {
                    float gq;
                    if (has_obits) {
                        const unsigned long long om = (unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)un.ow, rr) |
                                                      ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)un.ow, rr + 4) << 32);
                        gq = lane_masked(v, om);
                    } else {
                        gq = fmaf(un.xr[r], bsc[j], bsh[j]) > 0.f ? v : 0.f;
                    }
                    if (!ok) gq = 0.f;
                    if constexpr (PRESUM) {
                        sg += gq;
                        sgx = fmaf(gq, un.xr[r] - bmu[j], sgx);
                    } else {
                        d1[j] += (double)gq;
                        d2[j] += (double)gq * ((double)un.xr[r] - (double)bmu[j]);
                    }
                }
                if (ok) g_y[pix[r] * Cout + coff] = v;
            });
            if constexpr (PRESUM) { d1[j] += (double)sg; d2[j] += (double)sgx; }
            if constexpr (k + DEPTH < NU) issue(k + DEPTH, un);
        });
    }
};

// which epilogue a problem takes: 1 = the lean forward form (raw result + statistics: a training step's forward; one class, no remap), 2 = the lean
// data-gradient form (addend with optional ReLU bits, BatchNorm sums with bits or the re-derived mask; any class structure), 0 = the shared epilogue
inline int lean_epilogue_choice(const ConvP& p) {
    if (p.yplanes || p.bnr_out || !p.y || p.scale || p.relu) return 0;
    const ConvP::Class& c = p.cls[0];
    const bool remap = p.omul != 1 || c.oah != 0 || c.oaw != 0 || p.OH != c.Mh || p.OW != c.Mw;
    if (!p.res && !p.bnr_raw && !p.res_bits) return (p.ncls == 1 && !remap) ? 1 : 0;
    if (!p.stats && (p.res || p.bnr_raw) && (!p.res_bits || p.res)) {
        for (int i = 0; i < p.ncls; ++i)
            if (p.cls[i].ntaps == 0) return 0;      // (the dead parity classes of a 1x1 / stride-2 gradient: dx = addend there -- the shared epilogue's row form)
        return 2;
    }
    return 0;
}

// the epilogue object of the plane kernels (conv_x3_kernels.h): the look-ahead epilogue for EPI = 0, nothing for the lean forms
template <int BM, int BN, int WGM, int WGN, int EPI>
struct X3Epilogue {
    __device__ __forceinline__ void init(const ConvP&, const ConvP::Class&, int, int) {}
    __device__ __forceinline__ void prefetch() {}
    template <bool PRE = true>
    __device__ __forceinline__ void finish(const ConvP&, const ConvP::Class&, const f32x16 (&)[BM / WGM / 32][BN / WGN / 32], float (&)[BN / WGN / 32],
                                           float (&)[BN / WGN / 32], double (&)[BN / WGN / 32], double (&)[BN / WGN / 32]) {}
};
template <int BM, int BN, int WGM, int WGN>
struct X3Epilogue<BM, BN, WGM, WGN, 0> : IgemmEpilogue<BM, BN, WGM, WGN> {};
template <int BM, int BN, int WGM, int WGN>
struct X3Epilogue<BM, BN, WGM, WGN, 2> : LeanDgradEpilogue<BM, BN, WGM, WGN, false> {};

template <int BM, int BN, int WGM = 2, int WGN = 2>
__device__ __forceinline__ void igemm_store_rows(const ConvP& p, const ConvP::Class& c, const f32x16 (&acc)[BM / WGM / 32][BN / WGN / 32], int m0, int n0,
                                                 float (&s1)[BN / WGN / 32], float (&s2)[BN / WGN / 32]) {
    double d1[BN / WGN / 32], d2[BN / WGN / 32];          // (unused)
    igemm_store_rows_impl<BM, BN, WGM, WGN, false>(p, c, acc, m0, n0, s1, s2, d1, d2);
}

// BatchNorm-backward partial of one M tile -> bnr_part[blk][Cout][2] = (S1, invstd * S2), summed in a fixed order
template <int BM, int BN, int WGM = 2, int WGN = 2>
__device__ __forceinline__ void igemm_store_bnr(const ConvP& p, const double (&bd1)[BN / WGN / 32], const double (&bd2)[BN / WGN / 32], int blk, int n0,
                                                float* smem) {
    constexpr int WTN = BN / WGN, NI = WTN / 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    if (!p.bnr_raw) return;
    epi_lds_barrier();   // all fragment reads of the last chunk are done: LDS is free
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
    epi_lds_barrier();
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
template <int BM, int BN, int WGM = 2, int WGN = 2>
__device__ __forceinline__ void igemm_store_stats(const ConvP& p, const float (&s1)[BN / WGN / 32], const float (&s2)[BN / WGN / 32], int mt, int n0, float* smem) {
    constexpr int WTN = BN / WGN, NI = WTN / 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    if (p.stats) {
        // lanes l and l+32 hold the same channel; the M-waves are combined through LDS
        epi_lds_barrier();   // all fragment reads of the last chunk are done: LDS is free
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
        epi_lds_barrier();
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
    p.a_scale = p.a_shift = nullptr; p.a_relu = 0; p.abl = 0;
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
    p.a_scale = p.a_shift = nullptr; p.a_relu = 0; p.abl = 0;
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
