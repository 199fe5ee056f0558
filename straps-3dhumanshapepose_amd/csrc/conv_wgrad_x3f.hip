// conv_wgrad_x3f.hip -- weight gradient of the 1x1 convolutions (models/resnet.py:34-36; their backward in the bottleneck units :80-121) with BOTH
// operands read from the fp32 tensors (round 6; the forward / data-gradient twin is conv_x3f.hip).
//
//   dW[co][ci] = sum over pixels m of dy[m][co] * act(x)[m][ci],   act(x) = x, or relu(x * a_scale + a_shift): the producer's BatchNorm (+ ReLU) in
//   the operand path -- x is then the RAW output of the previous convolution, and the normalised activation the forward pass never materialised
//   (straps_conv_fwd_x3f with a_scale) is not materialised for the backward either.
//
//   The per-tap kernel on planes (backward.hip, conv_wgrad_x3_kernel) reads 6 B per operand element that the BatchNorm kernels had to write next to
//   or instead of the fp32 tensors, once per tile of the OTHER channel dimension.  Here a workgroup is persistent over a split of the pixels: 32
//   pixels per step, both operand tiles fetched with 16-byte loads (a pixel's BC channels are BC * 4 contiguous bytes), split into their three bf16
//   parts in registers (hardware conversion, as conv_x3f.hip) and written into the LDS image the transpose reads of the plane kernel expect ([32
//   pixels][BC channels] bf16 rows per plane, 64-byte segments permuted by the pixel number); requests run two steps ahead of the matrix work in
//   two register sets.  Channel blocks are chosen so that the SMALLER channel dimension is covered whole wherever it fits (64 <-> 256: every
//   operand element is read exactly once).  Same six-product arithmetic, same split-K partial layout and fixed-order reduction as the plane kernel.
#include "common.h"
#include <utility>

// fixed-order reduction of split-K partials [split][co][tap][ci] into OIHW (backward.hip)
int straps_internal_wgrad_reduce(const float* part, float* dw_oihw, int splits, int cout, int cin, int taps, int accumulate, hipStream_t st);

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef short short4w __attribute__((ext_vector_type(4)));
typedef short short8w __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

struct WgradFP {
    const float* x;
    const float* dy;
    const float* a_scale;
    const float* a_shift;
    int a_relu;
    float* part;
    int H, W, Cin, Cout, stride, Ho, Wo;      // x: [B][H][W][Cin], dy: [B][Ho][Wo][Cout]
    int M, rows_per_split, ct, it;
};

template <int BC>
__device__ __forceinline__ int seg_swz(int px) { return BC == 64 ? ((px >> 1) & 1) : (px & 3); }

__device__ __forceinline__ bf16x8 tr_frag(const u16* lo, const u16* hi) {      // eight pixels of one channel: two transpose reads of four pixels each
    const short4w a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4w*)lo);
    const short4w b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4w*)hi);
    const short8w v = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}

__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {
    const f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
// (conv_x3f.hip: the three bf16 parts of two fp32 values, packed; an infinite value keeps its leading part and zero residues)
__device__ __forceinline__ void split3_pair(float a, float b, unsigned& q1, unsigned& q2, unsigned& q3) {
    q1 = cvt_pk_bf16(a, b);
    float ra = a - __uint_as_float(q1 << 16), rb = b - __uint_as_float(q1 & 0xffff0000u);
    ra = (__float_as_uint(a) & 0x7fffffffu) == 0x7f800000u ? 0.f : ra;
    rb = (__float_as_uint(b) & 0x7fffffffu) == 0x7f800000u ? 0.f : rb;
    q2 = cvt_pk_bf16(ra, rb);
    const float sa = ra - __uint_as_float(q2 << 16), sb = rb - __uint_as_float(q2 & 0xffff0000u);
    q3 = cvt_pk_bf16(sa, sb);
}
// (conv_x3f.hip: LDS stores the compiler cannot order behind outstanding loads it cannot tell apart; covered by the wave's lgkmcnt(0) before the barrier)
__device__ __forceinline__ void lds_write_b64(unsigned addr, const u32x2& v) { asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(v) : "memory"); }

template <typename F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

// BCO x BCI channel block, WGO x WGI waves (eight; four for the 64 x 64 block); ABN: the operand-path BatchNorm on x
template <int BCO, int BCI, int WGO, int WGI, bool ABN>
__global__ __launch_bounds__(64 * WGO * WGI, 2) void conv_wgrad_x3f_kernel(WgradFP p) {
    constexpr int NTH = 64 * WGO * WGI;
    constexpr int PLD = 32 * BCO, PLX = 32 * BCI;                 // u16 elements per plane of a stage
    constexpr int STG = 3 * (PLD + PLX);
    constexpr int TPO = BCO / 4, TPX = BCI / 4;                   // threads per pixel row of the dy / x tile (16 bytes each)
    constexpr int DPT = 32 * TPO / NTH > 0 ? 32 * TPO / NTH : 1;  // 16-byte loads per thread and step
    constexpr int XPT = 32 * TPX / NTH > 0 ? 32 * TPX / NTH : 1;
    constexpr bool DY_ALL = 32 * TPO >= NTH, X_ALL = 32 * TPX >= NTH;      // (a 64-channel tile is 512 loads: every thread one; never fewer)
    static_assert(DY_ALL && X_ALL, "tiles of at least 64 channels");
    constexpr int WTO = BCO / WGO, WTI = BCI / WGI, MI = WTO / 32, NI = WTI / 32;
    static_assert(WTO % 32 == 0 && WTI % 32 == 0, "wave tile");
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    u16* smem = reinterpret_cast<u16*>(smem_f);                   // [2 stages][dy: 3 planes][32 px][BCO] | [x: 3 planes][32 px][BCI]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wo = wave / WGI, wi = wave % WGI;
    const int itile = blockIdx.x % p.it, ctile = blockIdx.x / p.it;
    const int co0 = ctile * BCO, ci0 = itile * BCI;
    const int mbeg = blockIdx.y * p.rows_per_split;
    const int mend = min(mbeg + p.rows_per_split, p.M);
    const int nsteps = (mend - mbeg + 31) >> 5;

    // ---- load slots: thread t owns 16 bytes (4 channels) of pixel row t / TP (+ NTH / TP per further load) of a step's tile
    const int co4 = (tid % TPO) * 4, pxo = tid / TPO;             // channel offset inside the block, first pixel row
    const int ci4 = (tid % TPX) * 4, pxx = tid / TPX;
    constexpr int PPO = NTH / TPO, PPX = NTH / TPX;               // pixel rows per load pass
    // LDS element (u16) offsets of the thread's 8 bytes in a plane: row px, 64-byte segment (c >> 5) ^ swz(px) (PPO, PPX are multiples of 4, or there
    // is a single pass: the permutation of row pxo + PPO q is that of pxo)
    const int wod = pxo * BCO + (((co4 >> 5) ^ seg_swz<BCO>(pxo)) << 5) + (co4 & 31);
    const int wox = pxx * BCI + (((ci4 >> 5) ^ seg_swz<BCI>(pxx)) << 5) + (ci4 & 31);
    static_assert((PPO % 4 == 0 || DPT == 1) && (PPX % 4 == 0 || XPT == 1), "segment permutation per pass");
    const float* dyp = p.dy + co0 + co4;
    const float* xp = p.x + ci0 + ci4;
    f32x4 bsc, bsh;
    if constexpr (ABN) {
        bsc = *reinterpret_cast<const f32x4*>(p.a_scale + ci0 + ci4);
        bsh = *reinterpret_cast<const f32x4*>(p.a_shift + ci0 + ci4);
    }
    const bool arelu = p.a_relu != 0;
    const bool pointwise = p.stride == 1;
    const int HoWo = p.Ho * p.Wo;
    // element offset of pixel m's row in x (stride 1: m itself)
    auto x_row = [&](int m) -> long long {
        if (pointwise) return (long long)m * p.Cin;
        const int b = m / HoWo, rem = m - b * HoWo;
        const int ho = rem / p.Wo, wo_ = rem - ho * p.Wo;
        return (((long long)b * p.H + ho * p.stride) * p.W + wo_ * p.stride) * p.Cin;
    };
    f32x4 dreg[2][DPT], xreg[2][XPT];
    int ld_s = 0;                                                 // step the next request fetches
    auto load_step = [&](auto set_c) {
        constexpr int SET = decltype(set_c)::value;
        const int m0 = mbeg + ld_s * 32;
#pragma unroll
        for (int q = 0; q < DPT; ++q) {
            int m = m0 + pxo + PPO * q;
            m = m < mend ? m : mend - 1;                          // (rows behind the split's end re-read its last pixel; they are zeroed at the conversion)
            dreg[SET][q] = *reinterpret_cast<const f32x4*>(dyp + (long long)m * p.Cout);
        }
#pragma unroll
        for (int q = 0; q < XPT; ++q) {
            int m = m0 + pxx + PPX * q;
            m = m < mend ? m : mend - 1;
            xreg[SET][q] = *reinterpret_cast<const f32x4*>(xp + x_row(m));
        }
        if (ld_s + 1 < nsteps) ++ld_s;                            // (requests behind the last step repeat it: the same number of loads every step)
    };
    auto convert_step = [&](auto set_c, int s) {                  // register set SET = step s -> LDS stage SET
        constexpr int SET = decltype(set_c)::value;
        const int m0 = mbeg + s * 32;
        const unsigned D = (unsigned)(SET * STG) * 2, X = D + 3 * PLD * 2;
        static_for<DPT>([&](auto qc) {
            constexpr int q = decltype(qc)::value;
            f32x4 v = dreg[SET][q];
            if (m0 + pxo + PPO * q >= mend) v = f32x4{0.f, 0.f, 0.f, 0.f};
            unsigned l1, l2, l3, h1, h2, h3;
            split3_pair(v[0], v[1], l1, l2, l3);
            split3_pair(v[2], v[3], h1, h2, h3);
            const unsigned a = D + (unsigned)(wod + PPO * q * BCO) * 2;
            lds_write_b64(a, u32x2{l1, h1});
            lds_write_b64(a + PLD * 2, u32x2{l2, h2});
            lds_write_b64(a + 2 * PLD * 2, u32x2{l3, h3});
        });
        static_for<XPT>([&](auto qc) {
            constexpr int q = decltype(qc)::value;
            f32x4 v = xreg[SET][q];
            if constexpr (ABN) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = fmaf(v[e], bsc[e], bsh[e]);
                    v[e] = arelu ? fmaxf(v[e], 0.f) : v[e];
                }
            }
            if (m0 + pxx + PPX * q >= mend) v = f32x4{0.f, 0.f, 0.f, 0.f};
            unsigned l1, l2, l3, h1, h2, h3;
            split3_pair(v[0], v[1], l1, l2, l3);
            split3_pair(v[2], v[3], h1, h2, h3);
            const unsigned a = X + (unsigned)(wox + PPX * q * BCI) * 2;
            lds_write_b64(a, u32x2{l1, h1});
            lds_write_b64(a + PLX * 2, u32x2{l2, h2});
            lds_write_b64(a + 2 * PLX * 2, u32x2{l3, h3});
        });
    };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int a = 0; a < MI; ++a)
#pragma unroll
        for (int c = 0; c < NI; ++c)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[a][c][q] = 0.f;
    // fragment addresses (backward.hip, conv_wgrad_x3_kernel): lane t = lane & 15 names row t >> 2 of a 4-pixel group and channel quad t & 3 of its 16-channel half
    const int tt = lane & 15, ch16 = (lane >> 4) & 1, kh = lane >> 5;
    int fd[2][2][MI], fx[2][2][NI];                               // [k step][read][32-channel block]
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int rd = 0; rd < 2; ++rd) {
            const int px = ks * 16 + kh * 8 + rd * 4 + (tt >> 2);
#pragma unroll
            for (int a = 0; a < MI; ++a) {
                const int c4 = wo * WTO + a * 32 + ch16 * 16 + (tt & 3) * 4;
                fd[ks][rd][a] = px * BCO + (((c4 >> 5) ^ seg_swz<BCO>(px)) << 5) + (c4 & 31);
            }
#pragma unroll
            for (int c = 0; c < NI; ++c) {
                const int c4 = wi * WTI + c * 32 + ch16 * 16 + (tt & 3) * 4;
                fx[ks][rd][c] = px * BCI + (((c4 >> 5) ^ seg_swz<BCI>(px)) << 5) + (c4 & 31);
            }
        }
    constexpr int TA[6] = {1, 0, 2, 0, 1, 0};
    constexpr int TB[6] = {1, 2, 0, 1, 0, 0};

    if (nsteps > 0) {
        load_step(std::integral_constant<int, 0>{});
        load_step(std::integral_constant<int, 1>{});
        convert_step(std::integral_constant<int, 0>{}, 0);
    }
    // one step: the matrix work of step s out of stage P; behind its first k step the conversion of step s + 1 (set 1 - P, requested during step
    // s - 1) into stage 1 - P, then the request of step s + 2 into set P
    auto step = [&](auto par_c, int s) {
        constexpr int P = decltype(par_c)::value;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const u16* D = smem + P * STG;
        const u16* X = D + 3 * PLD;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 av[MI][3], bv[NI][3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
                for (int a = 0; a < MI; ++a) av[a][pl] = tr_frag(D + pl * PLD + fd[ks][0][a], D + pl * PLD + fd[ks][1][a]);
#pragma unroll
                for (int c = 0; c < NI; ++c) bv[c][pl] = tr_frag(X + pl * PLX + fx[ks][0][c], X + pl * PLX + fx[ks][1][c]);
            }
#pragma unroll
            for (int e = 0; e < 6; ++e)
#pragma unroll
                for (int a = 0; a < MI; ++a)
#pragma unroll
                    for (int c = 0; c < NI; ++c)
                        acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[a][TA[e]], bv[c][TB[e]], acc[a][c], 0, 0, 0);
            if (ks == 0) {
                __builtin_amdgcn_sched_barrier(0);
                if (s + 1 < nsteps) convert_step(std::integral_constant<int, 1 - P>{}, s + 1);
                load_step(std::integral_constant<int, P>{});
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    for (int s = 0; s < nsteps; s += 2) {
        step(std::integral_constant<int, 0>{}, s);
        if (s + 1 < nsteps) step(std::integral_constant<int, 1>{}, s + 1);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    // C layout: lane -> ci (column), register -> co (row).  partial[split][co][ci]
    const int i = lane & 31;
    float* o = p.part + (long long)blockIdx.y * p.Cout * p.Cin;
#pragma unroll
    for (int a = 0; a < MI; ++a)
#pragma unroll
        for (int c = 0; c < NI; ++c)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int co = co0 + wo * WTO + a * 32 + mfma_row(q, lane);
                o[(long long)co * p.Cin + ci0 + wi * WTI + c * 32 + i] = acc[a][c][q];
            }
}

// channel block of a layer: the smaller channel dimension whole where it fits beside 256 of the other (every element read once), else 256 x 128 /
// 128 x 256 (the wide operand once, the narrow one twice), 64 x 64 for the 64 <-> 64 layer
inline void wgrad_x3f_block(int cin, int cout, int* bco, int* bci) {
    if (cout % 256 == 0 && cin == 64) { *bco = 256; *bci = 64; }
    else if (cin % 256 == 0 && cout == 64) { *bco = 64; *bci = 256; }
    else if (cout % 256 == 0 && cin % 128 == 0 && cout >= cin) { *bco = 256; *bci = 128; }
    else if (cin % 256 == 0 && cout % 128 == 0) { *bco = 128; *bci = 256; }
    else if (cout % 128 == 0 && cin % 128 == 0) { *bco = 128; *bci = 128; }
    else { *bco = 64; *bci = 64; }
}

inline int wgrad_x3f_splits(long long M, int tiles, bool tiles_are_small = false) {
    int s = ((tiles_are_small ? 512 : 256) + tiles - 1) / tiles;  // one eight-wave workgroup per CU (two four-wave ones)
    const long long max_s = (M + 63) / 64;                        // at least two steps per split
    if (s > max_s) s = (int)max_s;
    return s < 1 ? 1 : s;
}

template <int BCO, int BCI, int WGO, int WGI>
int launch_wgrad_x3f(const WgradFP& q, int tiles, int splits, hipStream_t st) {
    const size_t lds = (size_t)2 * 3 * 32 * (BCO + BCI) * sizeof(u16);
    if (q.a_scale) {
        STRAPS_RAISE_LDS((conv_wgrad_x3f_kernel<BCO, BCI, WGO, WGI, true>), lds, "conv_wgrad_x3f_kernel");
        hipLaunchKernelGGL((conv_wgrad_x3f_kernel<BCO, BCI, WGO, WGI, true>), dim3(tiles, splits), dim3(64 * WGO * WGI), lds, st, q);
    } else {
        STRAPS_RAISE_LDS((conv_wgrad_x3f_kernel<BCO, BCI, WGO, WGI, false>), lds, "conv_wgrad_x3f_kernel");
        hipLaunchKernelGGL((conv_wgrad_x3f_kernel<BCO, BCI, WGO, WGI, false>), dim3(tiles, splits), dim3(64 * WGO * WGI), lds, st, q);
    }
    STRAPS_CHECK_LAUNCH("conv_wgrad_x3f_kernel");
    return STRAPS_OK;
}

}  // namespace

extern "C" size_t straps_conv_wgrad_x3f_workspace_bytes(int batch, int h, int w, int cin, int cout, int kh, int kw, int stride, int pad) {
    if (kh != 1 || kw != 1 || pad != 0 || stride < 1 || cin % 64 || cout % 64) return 0;
    const long long M = (long long)batch * ((h - 1) / stride + 1) * ((w - 1) / stride + 1);
    int bco, bci;
    wgrad_x3f_block(cin, cout, &bco, &bci);
    return (size_t)wgrad_x3f_splits(M, (cout / bco) * (cin / bci), bco == 64 && bci == 64) * cout * cin * sizeof(float);
}

// dW (OIHW [cout][cin][1][1]) of a 1x1 convolution from the fp32 tensors: x [batch][h][w][cin] (a_scale / a_shift / a_relu as in
// straps_conv_fwd_x3f: x is then the raw output the producer's BatchNorm is applied to in the operand path), dy [batch][ho][wo][cout].
extern "C" int straps_conv_wgrad_x3f(const float* x, const float* a_scale, const float* a_shift, int a_relu, const float* dy, float* dw_oihw, void* workspace,
                                     int batch, int h, int w, int cin, int cout, int kh, int kw, int stride, int pad, int accumulate, void* stream) {
    STRAPS_REQUIRE(x && dy && dw_oihw && workspace, "straps_conv_wgrad_x3f: null pointer");
    STRAPS_REQUIRE(kh == 1 && kw == 1 && pad == 0 && (stride == 1 || stride == 2), "straps_conv_wgrad_x3f: 1x1 filters without padding, stride 1 or 2 only");
    STRAPS_REQUIRE(cin % 64 == 0 && cout % 64 == 0, "straps_conv_wgrad_x3f: need cin%%64==0 and cout%%64==0 (cin=%d cout=%d)", cin, cout);
    STRAPS_REQUIRE((a_scale == nullptr) == (a_shift == nullptr), "straps_conv_wgrad_x3f: a_scale and a_shift must be given together");
    STRAPS_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(a_scale) | reinterpret_cast<uintptr_t>(a_shift)) & 15) == 0,
                   "straps_conv_wgrad_x3f: x, dy, a_scale and a_shift must be 16-byte aligned");
    WgradFP q;
    q.x = x; q.dy = dy; q.a_scale = a_scale; q.a_shift = a_shift; q.a_relu = a_relu; q.part = (float*)workspace;
    q.H = h; q.W = w; q.Cin = cin; q.Cout = cout; q.stride = stride;
    q.Ho = (h - 1) / stride + 1; q.Wo = (w - 1) / stride + 1;
    const long long M = (long long)batch * q.Ho * q.Wo;
    STRAPS_REQUIRE(M < (1LL << 31) && M > 0, "straps_conv_wgrad_x3f: bad problem size");
    q.M = (int)M;
    int bco, bci;
    wgrad_x3f_block(cin, cout, &bco, &bci);
    q.ct = cout / bco; q.it = cin / bci;
    const int tiles = q.ct * q.it;
    const int splits = wgrad_x3f_splits(M, tiles, bco == 64 && bci == 64);
    q.rows_per_split = (int)(((M + splits - 1) / splits + 31) / 32 * 32);
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if (bco == 256 && bci == 64) rc = launch_wgrad_x3f<256, 64, 4, 2>(q, tiles, splits, st);
    else if (bco == 64 && bci == 256) rc = launch_wgrad_x3f<64, 256, 2, 4>(q, tiles, splits, st);
    else if (bco == 256 && bci == 128) rc = launch_wgrad_x3f<256, 128, 4, 2>(q, tiles, splits, st);
    else if (bco == 128 && bci == 256) rc = launch_wgrad_x3f<128, 256, 2, 4>(q, tiles, splits, st);
    else if (bco == 128 && bci == 128) rc = launch_wgrad_x3f<128, 128, 4, 2>(q, tiles, splits, st);
    else rc = launch_wgrad_x3f<64, 64, 2, 2>(q, tiles, splits, st);
    if (rc != STRAPS_OK) return rc;
    return straps_internal_wgrad_reduce(q.part, dw_oihw, splits, cout, cin, 1, accumulate, st);
}
