// conv_x3.hip -- the implicit-GEMM convolution of conv.hip on the bf16 matrix pipe at fp32 accuracy.
//
//   Every fp32 operand value is carried as three bf16 planes  x = x1 + x2 + x3  (round-to-nearest splits: the sum is EXACT, 3 x 8
//   significand bits cover fp32's 24; bf16 has fp32's exponent, so no scaling and no range analysis is needed; only below 2^-110 do
//   the last bits fall under bf16's smallest subnormal: absolute error <= 2^-133 there; above bf16's largest finite value, 3.39e38,
//   the leading plane rounds to infinity and the value is lost: an fp32 activation that close to overflow is already a failed run), and a product
//   a*b is evaluated as the six bf16 products of weight >= 2^-16
//        a1*b1 + a1*b2 + a2*b1 + a1*b3 + a3*b1 + a2*b2          (dropped: a2*b3 + a3*b2 + a3*b3 <= 2^-23 |a*b|)
//   each exact in the fp32 accumulator of v_mfma_f32_32x32x16_bf16.  Six 32-cycle instructions do the work of eight 64-cycle
//   v_mfma_f32_32x32x2_f32: 2.67x the exact-fp32 pipe at the same accuracy class (measured error vs float64: tests/test_gpu_conv_x3.py).
//   The planes are separate [pixels][C] bf16 tensors (straps_split3_bf16 writes them; 6 bytes per element instead of 4).
//
//   Same structure as conv.hip: tap table, LDS-DMA operand copies (a 32-channel K chunk of one plane is 64 contiguous bytes of one
//   pixel -> 4 lanes x 16 bytes), two stages, one barrier per chunk, XOR-swizzled 64-byte rows read back as one ds_read_b128 per
//   (32-row block, plane, 16-wide k step), the shared epilogue of conv_igemm.h.  The copies of chunk q+1 are issued between the
//   MFMAs of chunk q (a chunk's matrix work is only 6 x 2 x MI x NI x 32 cycles: issued in front of it they would cost as much
//   as the burst itself).
#include "conv_x3_kernels.h"

// conv_x3_lean.hip: the same kernels with the lean epilogues (p: a ConvP of this translation unit -- the header gives both the same layout)
int straps_internal_dispatch_x3_lean(const void* p, int halo, int cfg, int epi, hipStream_t st);

namespace {

// tile_cfg & 15: 0 = auto, 1 = 128x128 (8 waves, 3 stages), 2 = 128x64 (4 waves, 2 stages, two workgroups per CU), 3 = 64x64 (4 waves, 3 stages),
// 4 = 256x128 (8 waves, 2 stages), 5 = 128x128 (4 waves, 3 stages), 6 = alias of 4 (a four-wave 256x128 tile until round 5), 7 = 128x64 (4 waves, 3 stages);
// software-pipelined loop (PIPE): 8 = as 5, 9 = as 1, 10 = as 7, 11 = as 2, 12 = as 4
// auto rule from tools/sweep_conv_x3.py (resnet18 shapes, B = 64): 64-channel outputs take 128x64 tiles, two workgroups per CU; otherwise
// the largest tile that still gives every CU a workgroup: 256x128 from 512 128x128-tiles on (layer2: 63 vs 68 us), 128x128 from 256
// (layer3: 97 vs 108), else 128x64 with the three-stage ring (layer4's 4096 pixels: 113 vs 150).
// Stride-2 data gradients (ncls = 4 output-parity classes, each a quarter of the pixels with 1-4 of the taps -- a 1x1 filter has ONE live
// class) are sized by the tiles of a class, not of the launch (round 3, tools/sweep_conv_x3.py on the resnet18 and resnet50 shapes):
// resnet50's 1024 -> 2048 shortcut 124 -> 54 us, 256 -> 256 3x3 77 -> 59, 512 -> 1024 shortcut 71 -> 59; resnet18's 128 -> 256 74 -> 58,
// 256 -> 512 75 -> 64.
inline int pick_tile_x3(int cfg, long long M, int cout, int kdim, int& bm, int& bn, int ncls = 1, bool one_tap = false) {
    (void)kdim;
    cfg &= 15;
    if (cfg == 6) cfg = 4;      // (the four-wave 256x128 tile spilled registers from round 3 on and no rule ever chose it: retired in round 5, an alias of 4)
    if (cfg == 0) {
        const long long t128 = ((M + 127) / 128) * (cout / 128);
        if (cout % 128 != 0) cfg = 11;
        // fewer than 128 128x128-tile equivalents per class: 128x64 tiles would leave half the CUs without a workgroup -- 64x64 (round 4, from the
        // COLD sweep tools/sweep_conv_x3_cold.py: resnet50's layer4 at 32 bodies, 2 048 pixels x 512 channels: 3x3 96 -> 74 us, 1x1 2048 -> 512
        // 55 -> 41 us; resnet18's 256 -> 512 stride-2 data gradient 80 -> 66 us)
        else if (t128 / ncls < STRAPS_TOOL_ENV_INT("STRAPS_X3_SMALL_T128", 128)) cfg = 3;      // (tools: the threshold of the 64x64 rule, for A/B runs)
        else if (ncls > 1) cfg = t128 / ncls >= 512 ? 12 : t128 / ncls >= 256 ? (one_tap ? 9 : 12) : 11;
        else cfg = t128 >= 512 ? 12 : t128 >= 256 ? 5 : 7;       // (11 / 12: the pipelined loop pays with two-stage rings: -7 %)
        {   // (tools: one configuration for every class the three size rules above decide -- A/B runs of the rule itself with the lean epilogues in place)
            const int f = STRAPS_TOOL_ENV_INT("STRAPS_X3_RULE_CFG", 0);
            if (f > 0 && cout % 128 == 0 && t128 / ncls >= 128) cfg = f;
            const int lo = STRAPS_TOOL_ENV_INT("STRAPS_X3_LOW_CFG", 0);      // (... and for the smallest of the three buckets alone: 128 <= tiles < 256)
            if (lo > 0 && ncls == 1 && cout % 128 == 0 && t128 >= 128 && t128 < 256) cfg = lo;
        }
    }
    if (cout % 128 != 0 && cfg != 3 && cfg != 7 && cfg != 10 && cfg != 11) cfg = 2;
    bm = (cfg == 4 || cfg == 6 || cfg == 12) ? 256 : cfg == 3 ? 64 : 128;
    bn = (cfg == 2 || cfg == 3 || cfg == 7 || cfg == 10 || cfg == 11) ? 64 : 128;
    return cfg;
}

// auto tile choice only (tile_cfg & 15 == 0; bit 8 = im2col kernel only, bit 9 = halo kernel wherever it applies: A/B tools; bit 10 = the
// single-patch-buffer halo kernel for 64-channel outputs, explicit only):
// 1 = halo kernel 128x128, 2 = halo kernel 128x64, 3 = halo kernel 128x64 with one patch buffer, 0 = no.  Measured (tools/sweep_conv_x3.py, B = 64): halving the L2 -> LDS bytes
// buys only 4-5 % where the grid still fills the chip with 128x128 tiles (layer2: 104 vs 108 us -- 98 with the pipelined 256x128 tile,
// which is what layer2 uses --, layer3: 96 vs 101) and loses against the smaller / two-per-CU tiles of layer4 (157 vs 111) and, in
// its two-buffer form (140 KB of LDS, one workgroup per CU), of layer1 (154 vs 135).  With ONE patch buffer and a two-stage weight
// ring (76 KB, two workgroups per CU) the 64-channel layers do gain: 123 vs 137 us -- that form is the rule for them.
inline int halo_choice(const ConvP& p, int tile_cfg) {
    if ((tile_cfg & 15) != 0 || (tile_cfg & 256)) return 0;
    const int slots = halo_patch_slots(p);
    const bool all = (tile_cfg & 512) != 0;
    const long long t128 = (long long)(p.cls[0].M / 128) * (p.Cout / 128);
    if (slots > 0 && p.Cout % 128 != 0 && slots <= 272 && (!all || (tile_cfg & 1024))) return 3;      // single patch buffer, two workgroups per CU: layer1 123 vs 137 us
    if (slots > 0 && p.Cout % 128 == 0 && slots <= 208 && (all || (t128 >= 256 && t128 < 512))) return 1;
    if (slots > 0 && p.Cout % 128 != 0 && slots <= 272 && all) return 2;
    return 0;
}

template <int ABL>
int dispatch_x3_abl(const ConvP& p, int cfg, hipStream_t st) {
    switch (cfg) {
        case 1: return launch_x3<128, 128, 4, 2, 3, ABL>(p, st);
        case 2: return launch_x3<128, 64, 2, 2, 2, ABL>(p, st);
        case 3: return launch_x3<64, 64, 2, 2, 3, ABL>(p, st);
        case 4: return launch_x3<256, 128, 4, 2, 2, ABL>(p, st);
        case 5: return launch_x3<128, 128, 2, 2, 3, ABL>(p, st);
        case 8: return launch_x3<128, 128, 2, 2, 3, 0, true>(p, st);
        case 9: return launch_x3<128, 128, 4, 2, 3, 0, true>(p, st);
        case 10: return launch_x3<128, 64, 2, 2, 3, 0, true>(p, st);
        case 11: return launch_x3<128, 64, 2, 2, 2, 0, true>(p, st);
        case 12: return launch_x3<256, 128, 4, 2, 2, 0, true>(p, st);
        default: return launch_x3<128, 64, 2, 2, 3, ABL>(p, st);
    }
}

int dispatch_x3(const ConvP& p, int tile_cfg, hipStream_t st) {
    int bm, bn, kdim = 0;
    long long M = 0;
    for (int i = 0; i < p.ncls; ++i) {
        M += p.cls[i].M;
        if (p.cls[i].ntaps * p.Cin > kdim) kdim = p.cls[i].ntaps * p.Cin;
    }
    const int halo = halo_choice(p, tile_cfg);
    // the lean epilogues (conv_igemm.h; conv_x3_lean.hip holds the instantiations) on the automatic tiles: a training step's forward -- raw result +
    // statistics: +3.2 % / +2.8 % on the resnet18 / resnet50 step -- and its data gradients (LeanDgradEpilogue: the same look-ahead as the shared
    // epilogue -- two units' operands requested under the last chunk's matrix work -- with the lean arithmetic, per-value double sums in the shared
    // epilogue's order: bit-identical results, +0.2 ... +1 % on the steps; without the look-ahead the lean form was a wash behind these kernels' long
    // reductions: profiles/r06_lean_epilogue_ab.txt)
    if ((tile_cfg & 15) == 0 && !(tile_cfg & (64 | 128)) && STRAPS_TOOL_ENV_INT("STRAPS_X3_LEAN", 1)) {
        int epi = lean_epilogue_choice(p);
        if (epi == 2 && !STRAPS_TOOL_ENV_INT("STRAPS_X3_LEAN_DGRAD", 1)) epi = 0;      // (tools: A/B of the data-gradient form alone)
        if (epi) {
            const int cfg = halo ? 0 : pick_tile_x3(tile_cfg, M, p.Cout, kdim, bm, bn, p.ncls, kdim == p.Cin);
            if (halo == 1 || halo == 3 || (!halo && (cfg == 3 || cfg == 5 || cfg == 7 || cfg == 9 || cfg == 11 || cfg == 12)))
                return straps_internal_dispatch_x3_lean(&p, halo, cfg, epi, st);
        }
    }
    if (halo == 1) return launch_x3h<128, 128, 2, 2, 3, 208>(p, st);
    if (halo == 2) return launch_x3h<128, 64, 2, 2, 3, 272>(p, st);
    if (halo == 3) return launch_x3h<128, 64, 2, 2, 2, 272, 1>(p, st);
    const int cfg = pick_tile_x3(tile_cfg, M, p.Cout, kdim, bm, bn, p.ncls, kdim == p.Cin);
#ifdef STRAPS_TOOLS
    // ablation instantiations (wrong results by design: tools/x3_ablate.py) exist only in the tools build; the product library ignores the bits
    if ((tile_cfg & 192) == 192) return dispatch_x3_abl<3>(p, cfg, st);
    if (tile_cfg & 64) return dispatch_x3_abl<1>(p, cfg, st);
    if (tile_cfg & 128) return dispatch_x3_abl<2>(p, cfg, st);
#endif
    return dispatch_x3_abl<0>(p, cfg, st);
}

__global__ __launch_bounds__(256) void split3_kernel(const float* __restrict__ x, u16* __restrict__ o, long long n, long long ps) {
    const long long i4 = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i4 + 3 < n) {
        store_planes4(o, ps, i4, *reinterpret_cast<const f32x4*>(x + i4));
    } else {
        for (long long i = i4; i < n; ++i) {
            u16 b1, b2, b3;
            split3(x[i], b1, b2, b3);
            o[i] = b1; o[ps + i] = b2; o[2 * ps + i] = b3;
        }
    }
}

// the same split into the chunk-major layout the convolution kernels read (common.h: cm_index): x is a [rows][C] fp32 tensor
__global__ __launch_bounds__(256) void split3_cm_kernel(const float* __restrict__ x, u16* __restrict__ o, long long rows, int C, long long ps) {
    const int C4 = C >> 2;
    const long long n4 = rows * C4;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const long long r = i / C4;
        const int c4 = (int)(i - r * C4);
        store_planes4_cm(o, ps, r, c4 * 4, rows, *reinterpret_cast<const f32x4*>(x + i * 4));
    }
}

}  // namespace

extern "C" int straps_split3_bf16_cm(const float* x, unsigned short* planes, long long rows, int c, long long plane_stride, void* stream) {
    STRAPS_REQUIRE(x && planes, "straps_split3_bf16_cm: null pointer");
    STRAPS_REQUIRE(rows > 0 && c > 0 && c % 32 == 0, "straps_split3_bf16_cm: need rows > 0 and c %% 32 == 0 (rows=%lld c=%d)", rows, c);
    STRAPS_REQUIRE(plane_stride >= rows * c && plane_stride % 8 == 0, "straps_split3_bf16_cm: plane_stride must be >= rows*c and a multiple of 8");
    STRAPS_REQUIRE((reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(planes) & 15) == 0, "straps_split3_bf16_cm: pointers must be 16-byte aligned");
    const long long n4 = rows * (c >> 2);
    const long long g = (n4 + 255) / 256;
    hipLaunchKernelGGL(split3_cm_kernel, dim3((unsigned)(g > 16384 ? 16384 : g)), dim3(256), 0, (hipStream_t)stream, x, planes, rows, c, plane_stride);
    STRAPS_CHECK_LAUNCH("split3_cm_kernel");
    return STRAPS_OK;
}

extern "C" int straps_split3_bf16(const float* x, unsigned short* planes, long long n, long long plane_stride, void* stream) {
    STRAPS_REQUIRE(x && planes, "straps_split3_bf16: null pointer");
    STRAPS_REQUIRE(n >= 0 && plane_stride >= n && plane_stride % 8 == 0, "straps_split3_bf16: plane_stride must be >= n and a multiple of 8");
    if (n == 0) return STRAPS_OK;
    STRAPS_REQUIRE((reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(planes) & 15) == 0, "straps_split3_bf16: pointers must be 16-byte aligned");
    hipLaunchKernelGGL(split3_kernel, dim3(cdiv(n, 1024)), dim3(256), 0, (hipStream_t)stream, x, planes, n, plane_stride);
    STRAPS_CHECK_LAUNCH("split3_kernel");
    return STRAPS_OK;
}

extern "C" int straps_conv_fwd_x3(const unsigned short* x3, long long x_plane_stride, const unsigned short* w3, long long w_plane_stride,
                                  const float* scale, const float* shift, const float* residual, int relu, float* y, float* stats_partial,
                                  int batch, int h, int wdt, int cin, int cout, int kh, int kw, int stride, int pad, int tile_cfg, void* stream) {
    STRAPS_REQUIRE(x3 && w3 && y, "straps_conv_fwd_x3: null pointer");
    STRAPS_REQUIRE(batch > 0 && h > 0 && wdt > 0, "straps_conv_fwd_x3: empty input %dx%dx%d", batch, h, wdt);
    STRAPS_REQUIRE(cin % 32 == 0 && cout % 64 == 0, "straps_conv_fwd_x3: need cin%%32==0 and cout%%64==0 (cin=%d cout=%d)", cin, cout);
    STRAPS_REQUIRE(kh >= 1 && kw >= 1 && kh * kw <= 9 && stride >= 1 && pad >= 0, "straps_conv_fwd_x3: bad filter geometry");
    STRAPS_REQUIRE((scale == nullptr) == (shift == nullptr), "straps_conv_fwd_x3: scale and shift must be given together");
    STRAPS_REQUIRE(x_plane_stride % 8 == 0 && w_plane_stride % 8 == 0, "straps_conv_fwd_x3: plane strides must be multiples of 8 elements");
    ConvP p;
    p.x = reinterpret_cast<const float*>(x3); p.w = reinterpret_cast<const float*>(w3);
    p.xps = x_plane_stride; p.wps = w_plane_stride;
    const int rc = conv_fwd_problem(p, scale, shift, residual, relu, y, stats_partial, batch, h, wdt, cin, cout, kh, kw, stride, pad);
    if (rc != STRAPS_OK) return rc;
    return dispatch_x3(p, tile_cfg, (hipStream_t)stream);
}

// straps_conv_fwd_x3 whose epilogue also writes the result's three bf16 planes (eval-mode chains: the next convolution's operand without
// a split pass); y may be NULL when only the planes are consumed.
extern "C" int straps_conv_fwd_x3p(const unsigned short* x3, long long x_plane_stride, const unsigned short* w3, long long w_plane_stride,
                                   const float* scale, const float* shift, const float* residual, int relu, float* y, unsigned short* y_planes,
                                   long long y_plane_stride, int batch, int h, int wdt, int cin, int cout, int kh, int kw, int stride, int pad,
                                   int tile_cfg, void* stream) {
    STRAPS_REQUIRE(x3 && w3 && y_planes, "straps_conv_fwd_x3p: null pointer");
    STRAPS_REQUIRE(batch > 0 && h > 0 && wdt > 0, "straps_conv_fwd_x3p: empty input %dx%dx%d", batch, h, wdt);
    STRAPS_REQUIRE(cin % 32 == 0 && cout % 64 == 0, "straps_conv_fwd_x3p: need cin%%32==0 and cout%%64==0 (cin=%d cout=%d)", cin, cout);
    STRAPS_REQUIRE(kh >= 1 && kw >= 1 && kh * kw <= 9 && stride >= 1 && pad >= 0, "straps_conv_fwd_x3p: bad filter geometry");
    STRAPS_REQUIRE((scale == nullptr) == (shift == nullptr), "straps_conv_fwd_x3p: scale and shift must be given together");
    STRAPS_REQUIRE(x_plane_stride % 8 == 0 && w_plane_stride % 8 == 0 && y_plane_stride % 8 == 0, "straps_conv_fwd_x3p: plane strides must be multiples of 8 elements");
    ConvP p;
    p.x = reinterpret_cast<const float*>(x3); p.w = reinterpret_cast<const float*>(w3);
    p.xps = x_plane_stride; p.wps = w_plane_stride;
    const int rc = conv_fwd_problem(p, scale, shift, residual, relu, y, nullptr, batch, h, wdt, cin, cout, kh, kw, stride, pad);
    if (rc != STRAPS_OK) return rc;
    STRAPS_REQUIRE(y_plane_stride >= (long long)p.cls[0].M * cout, "straps_conv_fwd_x3p: y_plane_stride smaller than the output");
    p.yplanes = y_planes; p.yps = y_plane_stride;
    return dispatch_x3(p, tile_cfg, (hipStream_t)stream);
}

static int conv_dgrad_x3_impl(const unsigned short* dy3, long long dy_plane_stride, const unsigned short* w3_crsk, long long w_plane_stride,
                              const float* addend, const unsigned* addend_bits, float* dx, int batch, int h, int wdt, int cin, int cout, int kh, int kw,
                              int stride, int pad, int tile_cfg, void* stream) {
    STRAPS_REQUIRE(dy3 && w3_crsk && dx, "straps_conv_dgrad_x3: null pointer");
    STRAPS_REQUIRE(!addend_bits || (addend && cin % 32 == 0), "straps_conv_dgrad_x3_bits: the ReLU bits mask an addend (and need cin %% 32 == 0)");
    STRAPS_REQUIRE(cout % 32 == 0 && cin % 64 == 0, "straps_conv_dgrad_x3: need cout%%32==0 and cin%%64==0 (cin=%d cout=%d)", cin, cout);
    STRAPS_REQUIRE(stride == 1 || stride == 2, "straps_conv_dgrad_x3: stride must be 1 or 2");
    STRAPS_REQUIRE(kh * kw <= 9 && kh - 1 - pad >= 0 && kw - 1 - pad >= 0, "straps_conv_dgrad_x3: unsupported filter geometry");
    STRAPS_REQUIRE(dy_plane_stride % 8 == 0 && w_plane_stride % 8 == 0, "straps_conv_dgrad_x3: plane strides must be multiples of 8 elements");
    ConvP p;
    p.x = reinterpret_cast<const float*>(dy3); p.w = reinterpret_cast<const float*>(w3_crsk);
    p.xps = dy_plane_stride; p.wps = w_plane_stride;
    const int rc = conv_dgrad_problem(p, addend, dx, batch, h, wdt, cin, cout, kh, kw, stride, pad);
    if (rc != STRAPS_OK) return rc;
    p.res_bits = addend_bits;
    return p.ncls ? dispatch_x3(p, tile_cfg, (hipStream_t)stream) : STRAPS_OK;
}

extern "C" int straps_conv_dgrad_x3(const unsigned short* dy3, long long dy_plane_stride, const unsigned short* w3_crsk, long long w_plane_stride,
                                    const float* addend, float* dx, int batch, int h, int wdt, int cin, int cout, int kh, int kw, int stride,
                                    int pad, int tile_cfg, void* stream) {
    return conv_dgrad_x3_impl(dy3, dy_plane_stride, w3_crsk, w_plane_stride, addend, nullptr, dx, batch, h, wdt, cin, cout, kh, kw, stride, pad, tile_cfg,
                              stream);
}

// straps_conv_dgrad_x3 whose addend is the UNMASKED gradient of a residual unit's output: dx = dgrad + (bit ? addend : 0), with the unit's ReLU
// decisions as bits (straps_bn_apply_bits_x3: word [pixel][cin / 32], bit c & 31) -- the masked copy `dz` of that gradient is never written
extern "C" int straps_conv_dgrad_x3_bits(const unsigned short* dy3, long long dy_plane_stride, const unsigned short* w3_crsk, long long w_plane_stride,
                                         const float* addend, float* dx, int batch, int h, int wdt, int cin, int cout, int kh, int kw, int stride,
                                         int pad, int tile_cfg, const unsigned* addend_bits, void* stream) {
    STRAPS_REQUIRE(addend_bits, "straps_conv_dgrad_x3_bits: null bit mask");
    return conv_dgrad_x3_impl(dy3, dy_plane_stride, w3_crsk, w_plane_stride, addend, addend_bits, dx, batch, h, wdt, cin, cout, kh, kw, stride, pad,
                              tile_cfg, stream);
}

// number of [cout][2] statistics partials straps_conv_fwd_x3 writes for this geometry (= its M tiles)
// M tiles (= BatchNorm-backward partial blocks) of straps_conv_dgrad_x3[_bn] for this geometry, all parity classes
static int dgrad_x3_blocks(const ConvP& p, int tile_cfg) {
    if (halo_choice(p, tile_cfg)) return p.cls[0].M / 128;
    int bm, bn, kdim = 0;
    long long M = 0;
    for (int i = 0; i < p.ncls; ++i) {
        M += p.cls[i].M;
        if (p.cls[i].ntaps * p.Cin > kdim) kdim = p.cls[i].ntaps * p.Cin;
    }
    pick_tile_x3(tile_cfg, M, p.Cout, kdim, bm, bn, p.ncls, kdim == p.Cin);
    int blocks = 0;
    for (int i = 0; i < p.ncls; ++i) blocks += (p.cls[i].M + bm - 1) / bm;
    return blocks;
}

extern "C" int straps_conv_dgrad_x3_bn_blocks(int batch, int h, int w, int cin, int cout, int kh, int kw, int stride, int pad, int tile_cfg) {
    ConvP p;
    p.x = nullptr; p.w = nullptr; p.xps = p.wps = 0;
    if (!(stride == 1 || stride == 2) || kh * kw > 9 || kh - 1 - pad < 0 || kw - 1 - pad < 0) return -1;
    if (conv_dgrad_problem(p, nullptr, nullptr, batch, h, w, cin, cout, kh, kw, stride, pad) != STRAPS_OK) return -1;
    return dgrad_x3_blocks(p, tile_cfg);
}

static int conv_dgrad_x3_bn_impl(const unsigned short* dy3, long long dy_plane_stride, const unsigned short* w3_crsk, long long w_plane_stride,
                                 const float* addend, const unsigned* addend_bits, float* dx, int batch, int h, int wdt, int cin, int cout, int kh, int kw,
                                 int stride, int pad, int tile_cfg, const float* bn_raw, const float* bn_out, const unsigned* bn_out_bits,
                                 const float* bn_mask_scale, const float* bn_mask_shift, const float* bn_mean, const float* bn_invstd,
                                 double* bn_partials, void* stream) {
    STRAPS_REQUIRE(dy3 && w3_crsk && dx, "straps_conv_dgrad_x3_bn: null pointer");
    STRAPS_REQUIRE(bn_raw && bn_mean && bn_invstd && bn_partials && (bn_out || bn_out_bits || (bn_mask_scale && bn_mask_shift)),
                   "straps_conv_dgrad_x3_bn: the BatchNorm tensors (raw, mean, invstd, partials, and out / its bits or mask scale / shift) are required");
    STRAPS_REQUIRE(!addend_bits || addend, "straps_conv_dgrad_x3_bn_bits: the ReLU bits mask an addend");
    STRAPS_REQUIRE(!(addend_bits || bn_out_bits) || cin % 32 == 0, "straps_conv_dgrad_x3_bn_bits: bit masks need cin %% 32 == 0 (cin=%d)", cin);
    STRAPS_REQUIRE(cout % 32 == 0 && cin % 64 == 0, "straps_conv_dgrad_x3_bn: need cout%%32==0 and cin%%64==0 (cin=%d cout=%d)", cin, cout);
    STRAPS_REQUIRE(stride == 1 || stride == 2, "straps_conv_dgrad_x3_bn: stride must be 1 or 2");
    STRAPS_REQUIRE(kh * kw <= 9 && kh - 1 - pad >= 0 && kw - 1 - pad >= 0, "straps_conv_dgrad_x3_bn: unsupported filter geometry");
    STRAPS_REQUIRE(dy_plane_stride % 8 == 0 && w_plane_stride % 8 == 0, "straps_conv_dgrad_x3_bn: plane strides must be multiples of 8 elements");
    ConvP p;
    p.x = reinterpret_cast<const float*>(dy3); p.w = reinterpret_cast<const float*>(w3_crsk);
    p.xps = dy_plane_stride; p.wps = w_plane_stride;
    const int rc = conv_dgrad_problem(p, addend, dx, batch, h, wdt, cin, cout, kh, kw, stride, pad);
    if (rc != STRAPS_OK) return rc;
    p.bnr_raw = bn_raw; p.bnr_out = bn_out; p.bnr_sc = bn_mask_scale; p.bnr_sh = bn_mask_shift; p.bnr_mean = bn_mean; p.bnr_invstd = bn_invstd;
    p.bnr_part = bn_partials;
    p.bnr_bits = bn_out_bits; p.res_bits = addend_bits;
    return p.ncls ? dispatch_x3(p, tile_cfg, (hipStream_t)stream) : STRAPS_OK;
}

extern "C" int straps_conv_dgrad_x3_bn(const unsigned short* dy3, long long dy_plane_stride, const unsigned short* w3_crsk, long long w_plane_stride,
                                       const float* addend, float* dx, int batch, int h, int wdt, int cin, int cout, int kh, int kw, int stride,
                                       int pad, int tile_cfg, const float* bn_raw, const float* bn_out, const float* bn_mask_scale,
                                       const float* bn_mask_shift, const float* bn_mean, const float* bn_invstd, double* bn_partials, void* stream) {
    return conv_dgrad_x3_bn_impl(dy3, dy_plane_stride, w3_crsk, w_plane_stride, addend, nullptr, dx, batch, h, wdt, cin, cout, kh, kw, stride, pad, tile_cfg,
                                 bn_raw, bn_out, nullptr, bn_mask_scale, bn_mask_shift, bn_mean, bn_invstd, bn_partials, stream);
}

// straps_conv_dgrad_x3_bn with ReLU decisions as bits (word [pixel][cin / 32], bit c & 31; straps_bn_apply_bits_x3) wherever the fp32 form reads
// a whole activation tensor for its sign: bn_out_bits replaces bn_out as the mask of the BatchNorm sums (4 B per element less), addend_bits
// masks the addend, which is then the unmasked gradient of the later unit's output (either may be NULL: that operand is used as before)
extern "C" int straps_conv_dgrad_x3_bn_bits(const unsigned short* dy3, long long dy_plane_stride, const unsigned short* w3_crsk, long long w_plane_stride,
                                            const float* addend, float* dx, int batch, int h, int wdt, int cin, int cout, int kh, int kw, int stride,
                                            int pad, int tile_cfg, const float* bn_raw, const float* bn_out, const float* bn_mask_scale,
                                            const float* bn_mask_shift, const float* bn_mean, const float* bn_invstd, double* bn_partials,
                                            const unsigned* addend_bits, const unsigned* bn_out_bits, void* stream) {
    return conv_dgrad_x3_bn_impl(dy3, dy_plane_stride, w3_crsk, w_plane_stride, addend, addend_bits, dx, batch, h, wdt, cin, cout, kh, kw, stride, pad,
                                 tile_cfg, bn_raw, bn_out, bn_out_bits, bn_mask_scale, bn_mask_shift, bn_mean, bn_invstd, bn_partials, stream);
}

extern "C" int straps_conv_x3_stat_blocks(int batch, int h, int w, int cin, int cout, int kh, int kw, int stride, int pad, int tile_cfg) {
    ConvP p;
    p.x = nullptr; p.w = nullptr; p.xps = p.wps = 0;
    if (conv_fwd_problem(p, nullptr, nullptr, nullptr, 0, nullptr, nullptr, batch, h, w, cin, cout, kh, kw, stride, pad) != STRAPS_OK) return -1;
    const long long M = p.cls[0].M;
    if (halo_choice(p, tile_cfg)) return (int)(M / 128);
    int bm, bn;
    pick_tile_x3(tile_cfg, M, cout, kh * kw * cin, bm, bn);
    return (int)((M + bm - 1) / bm);
}
