// backward.hip -- gradient kernels of the encoder / IEF / rot6d (what autograd + cuDNN/cuBLAS do for the
// reference's loss.backward(), train/train_synthetic_otf_rendering.py:232).
//
//   conv_wgrad_kernel      dW[co][r][s][ci] = sum_m dy[m][co] * x[m@(r,s)][ci]   -- fp32 MFMA, contraction over the
//                          B*Ho*Wo pixels, split over many workgroups (deterministic partials + fixed-order reduce
//                          that also transposes KRSC -> OIHW, the layout of the parameter's .grad)
//   stem_wgrad_kernel      same for the 7x7/s2 stem straight from the NCHW input (LDS halo patch, like stem.hip)
//   bn_bwd_*               training-mode BatchNorm backward with the ReLU mask fused (two passes: reduce, apply)
//   maxpool / gap backward
//   rot6d_bwd_kernel       Gram-Schmidt backward
#include "common.h"
#include <stdlib.h>

namespace {

// =====================================================================================================
// conv weight gradient
// =====================================================================================================
struct WgradP {
    const float* x;    // [B][H][W][Cin]
    const float* dy;   // [B][Ho][Wo][Cout]
    float* part;       // [splits][Cout][R*S][Cin]
    int B, H, W, Cin, Cout, R, S, stride, pad, Ho, Wo;
    int M, rows_per_split, ct, it;   // ct = Cout/64 tiles, it = Cin/64 tiles
};

__device__ __attribute__((aligned(16))) float k_zero16w[4] = {0.f, 0.f, 0.f, 0.f};   // source of padding / out-of-range pixels

// One workgroup = one filter tap x a BCO(co) x BCI(ci) block x one split of the B*Ho*Wo pixels, 32 pixels per step.  Both operand
// tiles ([32 px][BCO | BCI channels], unpadded rows) go global -> LDS through the LDS-DMA (one wave-instruction = 1 KiB = 4 or 2
// pixels), two stages, one barrier per step; the fragment reads are ds_read_b32 over 32 consecutive channels (conflict-free without
// padding).  The pixel coordinates of a thread's copy slots advance incrementally (no division in the loop).
// The operands of this contraction stream from HBM (every pixel row is read once per tile of the OTHER channel dimension), so the
// tile sets the roofline: 64x64 = 16 flop/B (~80 TFLOP/s at 5 TB/s), 128x128 = 32 flop/B.
template <int BCO, int BCI>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradP p) {
    constexpr int STG = 32 * (BCO + BCI);                         // floats per stage: [32 px][BCO] then [32 px][BCI]
    constexpr int LPD = BCO / 4, PWD = 64 / LPD, RDD = 8 / PWD;   // dy: lanes per pixel, pixels per wave-copy, copy rounds per wave
    constexpr int LPX = BCI / 4, PWX = 64 / LPX, RDX = 8 / PWX;
    constexpr int MI = BCO / 64, NI = BCI / 64;
    extern __shared__ __attribute__((aligned(16))) float smem[];  // [2][STG]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int wm = wave >> 1, wn = wave & 1;
    int t = blockIdx.x;
    const int itile = t % p.it; t /= p.it;
    const int ctile = t % p.ct; t /= p.ct;
    const int tap = t;
    const int r = tap / p.S, s = tap - r * p.S;
    const int split = blockIdx.y;
    const int co0 = ctile * BCO, ci0 = itile * BCI;
    const int mbeg = split * p.rows_per_split;
    const int mend = min(mbeg + p.rows_per_split, p.M);
    const int nsteps = (mend - mbeg + 31) >> 5;
    const bool pointwise = p.R == 1 && p.S == 1 && p.stride == 1 && p.pad == 0;
    const float* zsrc = k_zero16w;
    asm volatile("" : "+s"(zsrc));          // the constant's address stays in SGPRs (else: a GOT load + wait per copy, inside the loop)

    // copy slots at step 0: round q of this wave copies pixels (q*4 + wave) * PW .. of the tile
    int md[RDD];                                                  // dy rows need only the linear pixel index
    int mx[RDX], wo_[RDX], ho_[RDX], b_[RDX];
    const int cd = (lane % LPD) * 4, cx = (lane % LPX) * 4;
#pragma unroll
    for (int q = 0; q < RDD; ++q) md[q] = mbeg + (q * 4 + wave) * PWD + lane / LPD;
#pragma unroll
    for (int q = 0; q < RDX; ++q) {
        const int m = mbeg + (q * 4 + wave) * PWX + lane / LPX;
        mx[q] = m;
        const int HoWo = p.Ho * p.Wo;
        b_[q] = m / HoWo;
        const int rem = m - b_[q] * HoWo;
        ho_[q] = rem / p.Wo;
        wo_[q] = rem - ho_[q] * p.Wo;
    }
    const int adv_h = 32 / p.Wo, adv_w = 32 - adv_h * p.Wo;      // a step moves every slot 32 pixels on
    auto dma_step = [&](int stage) {
        float* D = smem + stage * STG;
        float* X = D + 32 * BCO;
#pragma unroll
        for (int q = 0; q < RDD; ++q) {
            const float* src = md[q] < mend ? p.dy + (long long)md[q] * p.Cout + co0 + cd : zsrc;
            md[q] += 32;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(D + (q * 4 + wave_u) * PWD * BCO), 16, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < RDX; ++q) {
            const bool in = mx[q] < mend;
            const float* src = zsrc;
            if (pointwise) {
                if (in) src = p.x + (long long)mx[q] * p.Cin + ci0 + cx;
            } else {
                const int hi = ho_[q] * p.stride - p.pad + r, wi = wo_[q] * p.stride - p.pad + s;
                if (in && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W)
                    src = p.x + (((long long)b_[q] * p.H + hi) * p.W + wi) * p.Cin + ci0 + cx;
                wo_[q] += adv_w; ho_[q] += adv_h;
                if (wo_[q] >= p.Wo) { wo_[q] -= p.Wo; ++ho_[q]; }
                if (ho_[q] >= p.Ho) { const int k = ho_[q] / p.Ho; ho_[q] -= k * p.Ho; b_[q] += k; }
            }
            mx[q] += 32;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(X + (q * 4 + wave_u) * PWX * BCI), 16, 0, 0);
        }
    };
    f32x16 acc[MI][NI];
#pragma unroll
    for (int a = 0; a < MI; ++a)
#pragma unroll
        for (int c = 0; c < NI; ++c)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[a][c][q] = 0.f;
    if (nsteps > 0) dma_step(0);
    const int i = lane & 31, h = lane >> 5;
    const int fa = h * 4 * BCO + wm * (BCO / 2) + i, fb = 32 * BCO + h * 4 * BCI + wn * (BCI / 2) + i;
    for (int st = 0; st < nsteps; ++st) {
        const int stage = st & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // my copies of step st have landed ...
        __builtin_amdgcn_s_barrier();                         // ... everybody's have, and the other stage is no longer being read
        asm volatile("" ::: "memory");
        if (st + 1 < nsteps) dma_step(stage ^ 1);
        const float* S = smem + stage * STG;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k = kk * 8 + e;
                float av[MI], bv[NI];
#pragma unroll
                for (int a = 0; a < MI; ++a) av[a] = S[fa + k * BCO + a * 32];
#pragma unroll
                for (int c = 0; c < NI; ++c) bv[c] = S[fb + k * BCI + c * 32];
#pragma unroll
                for (int a = 0; a < MI; ++a)
#pragma unroll
                    for (int c = 0; c < NI; ++c) acc[a][c] = mfma32(av[a], bv[c], acc[a][c]);
            }
        }
        __builtin_amdgcn_s_setprio(0);
    }
    // C layout: lane -> ci (col), reg -> co (row).  partial[split][co][tap][ci]
    const long long RS = (long long)p.R * p.S;
    float* o = p.part + (long long)split * p.Cout * RS * p.Cin;
#pragma unroll
    for (int a = 0; a < MI; ++a)
#pragma unroll
        for (int c = 0; c < NI; ++c)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int co = co0 + wm * (BCO / 2) + a * 32 + mfma_row(q, lane);
                o[((long long)co * RS + tap) * p.Cin + ci0 + wn * (BCI / 2) + c * 32 + i] = acc[a][c][q];
            }
}

// ---- 3x3 / stride 1 / pad 1 weight gradient with an LDS-staged input halo patch ----------------------------------
// All nine taps of a 64(co) x 64(ci) block are accumulated by ONE workgroup: per 32-pixel chunk (rpc output rows x cw
// columns, cw = min(Wo,32)) it stages the dy tile [32][64] and the (rpc+2) x (cw+2) x 64 input patch once, then runs
// 9 taps x 16 = 144 MFMAs per wave between barriers; the nine shifted views are just LDS addresses.  Versus the
// per-tap kernel above: 1/9 of the dy traffic, ~1/3 of the x traffic, 9x fewer barriers per MFMA.
struct Wgrad3P {
    const float* x;
    const float* dy;
    float* part;
    int H, W, Cin, Cout;       // stride 1, pad 1: Ho = H, Wo = W
    int cw, cw_log2, rpc;      // chunk geometry
    int chunks_per_row, chunk_rows_per_img, nchunks, chunks_per_split, it;
};

constexpr int W3PX = 32 + 112;   // pixels per LDS stage: dy tile + input patch rounded up to whole 16-pixel copy rounds

// Staging: both operands go global -> LDS through the LDS-DMA (global_load_lds_dwordx4; one wave-instruction = 4 pixels x 256 B,
// rows unpadded: the fragment reads are ds_read_b32 over 32 consecutive channels, conflict-free as they are), two stages, one
// barrier per chunk.  Operands: the four pixels a lane half handles per 8-pixel group sit side by side in one row, so their
// nine-tap windows overlap: 3 x 6 patch values + 4 dy values feed 36 MFMAs (instead of 36 + 4), and the next group's 22 values
// are read while the current 36 MFMAs issue.
// NG = 2: two 4-wave groups per workgroup take alternate chunks of the split into their own LDS stages and add their accumulators
// through LDS at the end -- one partial per CU instead of two: half the partial-sum traffic (write here, read in the reduce pass).
template <int NG>
__global__ __launch_bounds__(256 * NG) void conv_wgrad3x3_kernel(Wgrad3P p) {
    extern __shared__ __attribute__((aligned(16))) float smem_all[];
    const int pw = p.cw + 2;
    const int npatch = (p.rpc + 2) * pw;                 // <= 102 pixels
    const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;              // position inside the 4-wave group
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int grp = NG > 1 ? __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 8) : 0;
    float* smem = smem_all + grp * (2 * W3PX * 64);
    const int wm = wave >> 1, wn = wave & 1;
    const int itile = blockIdx.x % p.it, ctile = blockIdx.x / p.it;
    const int co0 = ctile * 64, ci0 = itile * 64;
    const int cbeg = blockIdx.y * p.chunks_per_split;
    const int cend = min(cbeg + p.chunks_per_split, p.nchunks);
    const int lp = tid >> 4, lc = tid & 15;              // staging: pixel slot, float4 column
    const float* zsrc = k_zero16w;
    asm volatile("" : "+s"(zsrc));          // the constant's address stays in SGPRs (else: a GOT load + wait per copy, inside the loop)

    // chunk-independent part of the copy addresses: dy pixel (row, col) inside the chunk, patch pixel (row, col) inside the patch
    int d_off[2], x_pr[7], x_pc[7];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int px = lp + 16 * q;
        d_off[q] = ((px >> p.cw_log2) * p.W + (px & (p.cw - 1))) * p.Cout + co0 + lc * 4;
    }
#pragma unroll
    for (int q = 0; q < 7; ++q) {
        const int pp = lp + 16 * q;
        x_pr[q] = pp / pw - 1;
        x_pc[q] = pp - (x_pr[q] + 1) * pw - 1;
    }
    auto dma_chunk = [&](int c, int stage) {
        const int per_img = p.chunk_rows_per_img * p.chunks_per_row;
        const int b = c / per_img;
        const int rem = c - b * per_img;
        const int cr = rem / p.chunks_per_row, cc = rem - cr * p.chunks_per_row;
        const int ho0 = cr * p.rpc, wo0 = cc * p.cw;
        float* D = smem + stage * (W3PX * 64);
        float* X = D + 32 * 64;
        const float* dsrc = p.dy + ((long long)(b * p.H + ho0) * p.W + wo0) * p.Cout;
        const float* xsrc = p.x + ((long long)(b * p.H + ho0) * p.W + wo0) * p.Cin + ci0 + lc * 4;
#pragma unroll
        for (int q = 0; q < 2; ++q)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(dsrc + d_off[q]),
                                             (__attribute__((address_space(3))) void*)(D + (4 * wave_u + 16 * q) * 64), 16, 0, 0);
#pragma unroll
        for (int q = 0; q < 7; ++q) {
            if (q * 16 >= npatch) break;                 // wave-uniform
            const int hi = ho0 + x_pr[q], wi = wo0 + x_pc[q];
            const bool ok = (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;     // (slots past the patch read anything)
            const float* src = ok ? xsrc + (x_pr[q] * p.W + x_pc[q]) * p.Cin : zsrc;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(X + (4 * wave_u + 16 * q) * 64), 16, 0, 0);
        }
    };
    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[t][q] = 0.f;
    if (cbeg + grp < cend) dma_chunk(cbeg + grp, 0);
    const int i = lane & 31, h = lane >> 5;
    // LDS float offsets of the lane's four 4-pixel groups (group kk = pixels kk*8 + h*4 .. +3 of the chunk)
    int g_d[4], g_x[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        const int px = kk * 8 + h * 4;
        g_d[kk] = px * 64 + wm * 32 + i;
        g_x[kk] = 32 * 64 + ((px >> p.cw_log2) * pw + (px & (p.cw - 1))) * 64 + wn * 32 + i;
    }
    const int pw64 = pw * 64;
    const int nj = (cend - cbeg + NG - 1) / NG;               // both groups run the same number of rounds (the barrier is shared)
    for (int j = 0; j < nj; ++j) {
        const int c = cbeg + grp + NG * j;
        const int stage = j & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // my copies of chunk c have landed ...
        __builtin_amdgcn_s_barrier();                         // ... everybody's have, and the other stage is no longer being read
        asm volatile("" ::: "memory");
        if (c + NG < cend) dma_chunk(c + NG, stage ^ 1);
        if (NG > 1 && c >= cend) continue;                    // (odd chunk count: the second group idles in the last round)
        const float* S = smem + stage * (W3PX * 64);
        float a[2][4], w[2][18];
        auto read_group = [&](int kk, float* av, float* wv) {
            const float* D = S + g_d[kk];
            const float* X = S + g_x[kk];
#pragma unroll
            for (int e = 0; e < 4; ++e) av[e] = D[e * 64];
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int cidx = 0; cidx < 6; ++cidx) wv[r * 6 + cidx] = X[r * pw64 + cidx * 64];
        };
        read_group(0, a[0], w[0]);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            if (kk + 1 < 4) read_group(kk + 1, a[(kk + 1) & 1], w[(kk + 1) & 1]);
            const float* av = a[kk & 1];
            const float* wv = w[kk & 1];
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int s2 = 0; s2 < 3; ++s2) acc[r * 3 + s2] = mfma32(av[e], wv[r * 6 + e + s2], acc[r * 3 + s2]);
        }
        __builtin_amdgcn_s_setprio(0);
    }
    if constexpr (NG > 1) {
        // group 1 -> LDS -> group 0, three taps per round (48 KB), fixed order: acc(group 0) + acc(group 1)
        float* R = smem_all + ((wave * 3) * 16) * 64 + lane;
#pragma unroll
        for (int rd = 0; rd < 3; ++rd) {
            __syncthreads();                                  // the stages (round 0) / the previous round's values are no longer needed
            if (grp == 1) {
#pragma unroll
                for (int t = 0; t < 3; ++t)
#pragma unroll
                    for (int q = 0; q < 16; ++q) R[(t * 16 + q) * 64] = acc[rd * 3 + t][q];
            }
            __syncthreads();
            if (grp == 0) {
#pragma unroll
                for (int t = 0; t < 3; ++t)
#pragma unroll
                    for (int q = 0; q < 16; ++q) acc[rd * 3 + t][q] += R[(t * 16 + q) * 64];
            }
        }
        if (grp != 0) return;
    }
    float* o = p.part + (long long)blockIdx.y * p.Cout * 9 * p.Cin;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int co = co0 + wm * 32 + mfma_row(q, lane);
            o[((long long)co * 9 + t) * p.Cin + ci0 + wn * 32 + i] = acc[t][q];
        }
}

// ---- the same weight gradient on the bf16 matrix pipe (bf16x3 route, see conv_x3.hip) -------------------------------------------
// Operands are the three bf16 planes of x and dy (x = x1 + x2 + x3 exactly); a term dy * x is the six bf16 products of weight >= 2^-16
// in the fp32 accumulator of v_mfma_f32_32x32x16_bf16.  The reduction index of this GEMM is the PIXEL, but both operands are stored
// pixel-major / channel-minor, so a lane's eight consecutive k values are eight different LDS rows: they are gathered by the
// hardware transpose read ds_read_b64_tr_b16 -- in a 16-lane group, lanes 4j .. 4j+3 each name 4 contiguous channels of pixel j
// and lane t receives channel t of pixels 0..3 (tools/tr_probe.hip prints the lane map) -- two reads per plane and operand.
// LDS image per stage and plane: dy tile [32 pixels][64 channels] and input patch [128 pixel slots][64 channels], rows of 128 bytes
// copied by the LDS-DMA; the two 64-byte halves of a row are swapped when bit 1 of the pixel (slot) number is set, so four
// consecutive rows -- whatever the tap shift -- touch all 64 banks once per 32-lane read group.
struct Wgrad3XP {
    const u16* x3;
    const u16* dy3;
    long long xps, dps;
    float* part;
    int H, W, Cin, Cout;
    int cw, cw_log2, rpc;
    int chunks_per_row, chunk_rows_per_img, nchunks, chunks_per_split, it;
    long long rows;                  // B * H * W: pixels of x and of dy (stride 1), the chunk stride of their chunk-major planes
};

typedef __bf16 bf16x8w __attribute__((ext_vector_type(8)));
typedef short short4w __attribute__((ext_vector_type(4)));
typedef short short8w __attribute__((ext_vector_type(8)));

__device__ __attribute__((aligned(16))) float k_zero16x[4] = {0.f, 0.f, 0.f, 0.f};

constexpr int W3X_TG = 2;          // taps whose MFMA chains are interleaved

// eight k values (pixels) of one channel: two transpose reads of four pixels each
__device__ __forceinline__ bf16x8w tr_frag(const u16* lo, const u16* hi) {
    const short4w a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4w*)lo);
    const short4w b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4w*)hi);
    const short8w v = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8w, v);
}

// per-tap weight gradient (stride-2 / 1x1 / any shape the halo plan does not take) on the planes: conv_wgrad_kernel's structure -- one tap
// x a BCO x BCI channel block x one split of the pixels per workgroup, 32 pixels per step, LDS-DMA copies, two stages -- with the
// operand gathers of the kernel below (ds_read_b64_tr_b16).  Rows are BCO (BCI) bf16 channels; their 64-byte segments are permuted
// by the pixel number so four consecutive pixel rows cover all 64 banks (2 segments per row: swap on bit 1; 4 or more: xor with
// the low two bits).
struct WgradXP {
    const u16* x3;
    const u16* dy3;
    long long xps, dps;
    float* part;
    int B, H, W, Cin, Cout, R, S, stride, pad, Ho, Wo;
    int M, rows_per_split, ct, it;
};

template <int BC>
__device__ __forceinline__ int seg_swz(int px) { return BC == 64 ? ((px >> 1) & 1) : (px & 3); }

// NST stages: two in production.  Round 3 tried three and four (copies of NST - 1 steps in flight, s_waitcnt vmcnt(DPS * (NST - 2))), on
// the reasoning that a 32-pixel step is only 0.1-0.4 us of matrix work: no gain on any resnet18 / resnet50 shape and 10-60 % slower on
// the 3x3 / stride-2 layers (tools/sweep_wgrad_x3.py with STRAPS_WGRAD_TAP_NST=3|4: the LDS the extra stages take costs a resident
// workgroup) -- the kernel is bound by the operand bytes it streams from L2, not by their latency.
template <int BCO, int BCI, int NST = 2>
__global__ __launch_bounds__(256) void conv_wgrad_x3_kernel(WgradXP p) {
    constexpr int PLD = 32 * BCO, PLX = 32 * BCI;                 // u16 elements per plane of a stage: [32 px][BCO], [32 px][BCI]
    constexpr int STG = 3 * (PLD + PLX);
    constexpr int LPD = BCO / 8, PWD = 64 / LPD, RDD = 8 / PWD;   // dy: lanes per pixel, pixels per wave-copy, copy rounds per wave
    constexpr int LPX = BCI / 8, PWX = 64 / LPX, RDX = 8 / PWX;
    constexpr int MI = BCO / 64, NI = BCI / 64;
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    u16* smem = reinterpret_cast<u16*>(smem_f);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int wm = wave >> 1, wn = wave & 1;
    int t = blockIdx.x;
    const int itile = t % p.it; t /= p.it;
    const int ctile = t % p.ct; t /= p.ct;
    const int tap = t;
    const int r = tap / p.S, s = tap - r * p.S;
    const int split = blockIdx.y;
    const int co0 = ctile * BCO, ci0 = itile * BCI;
    const int mbeg = split * p.rows_per_split;
    const int mend = min(mbeg + p.rows_per_split, p.M);
    const int nsteps = (mend - mbeg + 31) >> 5;
    const bool pointwise = p.R == 1 && p.S == 1 && p.stride == 1 && p.pad == 0;
    const u16* zsrc = reinterpret_cast<const u16*>(k_zero16w);
    asm volatile("" : "+s"(zsrc));

    // copy slots at step 0: round q of this wave copies pixels (q*4 + wave) * PW .. of the tile; a lane's 16-byte slot j = lane % LP
    // of pixel row px receives channel group ((j >> 2) ^ swz(px)) * 4 + (j & 3)
    // (chunk-major planes, common.h: channel c of pixel row m sits at ((c >> 5) * rows + m) * 32 + (c & 31); gd / gx = the chunk's
    //  base + the 16-byte piece inside it, fixed per lane)
    int md[RDD];
    long long gd[RDD], gx[RDX];
    int mx[RDX], wo_[RDX], ho_[RDX], b_[RDX];
    const long long xrows = (long long)p.B * p.H * p.W;
#pragma unroll
    for (int q = 0; q < RDD; ++q) {
        const int px = (q * 4 + wave) * PWD + lane / LPD, j = lane % LPD;
        md[q] = mbeg + px;
        const int c = co0 + ((((j >> 2) ^ seg_swz<BCO>(px)) << 2) | (j & 3)) * 8;
        gd[q] = (long long)(c >> 5) * p.M * 32 + (c & 31);
    }
#pragma unroll
    for (int q = 0; q < RDX; ++q) {
        const int px = (q * 4 + wave) * PWX + lane / LPX, j = lane % LPX;
        const int m = mbeg + px;
        mx[q] = m;
        const int c = ci0 + ((((j >> 2) ^ seg_swz<BCI>(px)) << 2) | (j & 3)) * 8;
        gx[q] = (long long)(c >> 5) * xrows * 32 + (c & 31);
        const int HoWo = p.Ho * p.Wo;
        b_[q] = m / HoWo;
        const int rem = m - b_[q] * HoWo;
        ho_[q] = rem / p.Wo;
        wo_[q] = rem - ho_[q] * p.Wo;
    }
    const int adv_h = 32 / p.Wo, adv_w = 32 - adv_h * p.Wo;      // a step moves every slot 32 pixels on
    auto dma_step = [&](int stage) {
        u16* D = smem + stage * STG;
        u16* X = D + 3 * PLD;
#pragma unroll
        for (int q = 0; q < RDD; ++q) {
            const bool in = md[q] < mend;
            const u16* src = in ? p.dy3 + (long long)md[q] * 32 + gd[q] : zsrc;
            const long long ps = in ? p.dps : 0;
            md[q] += 32;
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + pl * ps),
                                                 (__attribute__((address_space(3))) void*)(D + pl * PLD + (q * 4 + wave_u) * PWD * BCO), 16, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < RDX; ++q) {
            bool in = mx[q] < mend;
            const u16* src = zsrc;
            if (pointwise) {
                if (in) src = p.x3 + (long long)mx[q] * 32 + gx[q];
            } else {
                const int hi = ho_[q] * p.stride - p.pad + r, wi = wo_[q] * p.stride - p.pad + s;
                in = in && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
                if (in) src = p.x3 + (((long long)b_[q] * p.H + hi) * p.W + wi) * 32 + gx[q];
                wo_[q] += adv_w; ho_[q] += adv_h;
                if (wo_[q] >= p.Wo) { wo_[q] -= p.Wo; ++ho_[q]; }
                if (ho_[q] >= p.Ho) { const int k = ho_[q] / p.Ho; ho_[q] -= k * p.Ho; b_[q] += k; }
            }
            const long long ps = in ? p.xps : 0;
            mx[q] += 32;
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + pl * ps),
                                                 (__attribute__((address_space(3))) void*)(X + pl * PLX + (q * 4 + wave_u) * PWX * BCI), 16, 0, 0);
        }
    };
    f32x16 acc[MI][NI];
#pragma unroll
    for (int a = 0; a < MI; ++a)
#pragma unroll
        for (int c = 0; c < NI; ++c)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[a][c][q] = 0.f;
#pragma unroll
    for (int s0 = 0; s0 < NST - 1; ++s0)
        if (s0 < nsteps) dma_step(s0);
    // fragment addresses: lane t = lane & 15 names row t >> 2 of a 4-pixel group and channel quad t & 3 of its 16-channel half
    const int tt = lane & 15, ch16 = (lane >> 4) & 1, kh = lane >> 5;
    int fd[2][2][MI], fx[2][2][NI];                               // [k step][read][32-channel block]
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int rd = 0; rd < 2; ++rd) {
            const int px = ks * 16 + kh * 8 + rd * 4 + (tt >> 2);
#pragma unroll
            for (int a = 0; a < MI; ++a) {
                const int c4 = wm * (BCO / 2) + a * 32 + ch16 * 16 + (tt & 3) * 4;
                fd[ks][rd][a] = px * BCO + (((c4 >> 5) ^ seg_swz<BCO>(px)) << 5) + (c4 & 31);
            }
#pragma unroll
            for (int c = 0; c < NI; ++c) {
                const int c4 = wn * (BCI / 2) + c * 32 + ch16 * 16 + (tt & 3) * 4;
                fx[ks][rd][c] = px * BCI + (((c4 >> 5) ^ seg_swz<BCI>(px)) << 5) + (c4 & 31);
            }
        }
    constexpr int TA[6] = {1, 0, 2, 0, 1, 0};
    constexpr int TB[6] = {1, 2, 0, 1, 0, 0};
    constexpr int DPS = 3 * (RDD + RDX);                      // LDS-DMA instructions per step and wave
    static_assert(DPS * (NST - 2) <= 63, "vmcnt range");
    int stage = 0;
    for (int st = 0; st < nsteps; ++st) {
        // my copies of step st have landed (the younger steps' may still be in flight) ...
        const int ahead = nsteps - 1 - st;
        if (NST >= 4 && ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DPS * (NST >= 4 ? 2 : 0)) : "memory");
        else if (NST >= 3 && ahead >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DPS * (NST >= 3 ? 1 : 0)) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                         // ... everybody's have, and the stage about to be refilled is no longer being read
        asm volatile("" ::: "memory");
        if (st + NST - 1 < nsteps) dma_step(stage == 0 ? NST - 1 : stage - 1);
        const u16* D = smem + stage * STG;
        const u16* X = D + 3 * PLD;
        stage = stage + 1 == NST ? 0 : stage + 1;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8w av[MI][3], bv[NI][3];
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
        }
        __builtin_amdgcn_s_setprio(0);
    }
    // C layout: lane -> ci (col), reg -> co (row).  partial[split][co][tap][ci]
    const int i = lane & 31;
    const long long RS = (long long)p.R * p.S;
    float* o = p.part + (long long)split * p.Cout * RS * p.Cin;
#pragma unroll
    for (int a = 0; a < MI; ++a)
#pragma unroll
        for (int c = 0; c < NI; ++c)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int co = co0 + wm * (BCO / 2) + a * 32 + mfma_row(q, lane);
                o[((long long)co * RS + tap) * p.Cin + ci0 + wn * (BCI / 2) + c * 32 + i] = acc[a][c][q];
            }
}

// NG = 2: two 4-wave groups share every stage; group g takes the k steps g, g + 2, .. (16 pixels each) of each chunk and the two accumulator
// sets are added through LDS at the end (fixed order) -- two waves per SIMD cover each other's LDS latency and barrier waits.
// CP = pixels per chunk (32 or 64; the plan's cw x rpc): 64 halves the barriers per MFMA and the halo overhead of the patch
// ((rpc + 2) x (cw + 2) slots per chunk: 3.2 -> 2.1 per pixel at cw = 32); a group then runs two k steps per chunk.
// ABL: compile-time measurement switches (tools, STRAPS_WGRAD3_ABL; 0 in production): 1 = no operand copies after the first chunk,
// 2 = no MFMAs (the fragments are folded into one accumulator on the VALU), 4 = a tenth of the fragment reads (every tap uses tap 0's)
template <int NG, int CP, int ABL = 0>
__global__ __launch_bounds__(256 * NG) void conv_wgrad3x3_x3_kernel(Wgrad3XP p) {
    constexpr int DPX = CP, XPX = CP == 32 ? 128 : 136;          // pixel rows per stage and plane: dy tile, input patch slots
    constexpr int STAGE = 3 * (DPX + XPX) * 64;                   // u16 elements per stage
    constexpr int DR = CP / 32, XR = (XPX + 31) / 32;             // copy rounds of 32 pixel slots: dy tile, patch
    constexpr int KSG = CP / 16 / NG;                             // k steps per group and chunk
    static_assert(CP % (16 * NG) == 0, "chunk / group mismatch");
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    u16* smem = reinterpret_cast<u16*>(smem_f);
    const int pw = p.cw + 2;
    const int npatch = (p.rpc + 2) * pw;                 // <= 102 (CP = 32) / 136 (CP = 64) pixels
    const int tid = threadIdx.x, lane = tid & 63, wave = (tid >> 6) & 3;           // wave inside its 4-wave group
    const int grp = NG > 1 ? __builtin_amdgcn_readfirstlane(tid >> 8) : 0;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int wm = wave >> 1, wn = wave & 1;
    const int itile = blockIdx.x % p.it, ctile = blockIdx.x / p.it;
    const int co0 = ctile * 64, ci0 = itile * 64;
    const int cbeg = blockIdx.y * p.chunks_per_split;
    const int cend = min(cbeg + p.chunks_per_split, p.nchunks);
    const u16* zsrc = reinterpret_cast<const u16*>(k_zero16x);
    asm volatile("" : "+s"(zsrc));

    // ---- copies: thread -> (pixel slot of a 32-slot round, 16-byte slot tid & 7 of its row); it fetches channel group slot ^ swap.
    // The five rounds of a chunk (dy tile, four patch rounds) alternate between the groups when NG = 2.
    const int lp = (tid & 255) >> 3, ls = tid & 7;
    const int d_g = ls ^ (((lp >> 1) & 1) << 2);           // (bit 1 of lp + 32 q is bit 1 of lp)
    // (chunk-major planes, common.h: a 64-channel row is two 64-byte pieces, one in each of two 32-channel chunks)
    long long d_off[DR];
#pragma unroll
    for (int q = 0; q < DR; ++q) {
        const int px = lp + 32 * q;
        d_off[q] = ((long long)((co0 >> 5) + (d_g >> 2)) * p.rows + (px >> p.cw_log2) * p.W + (px & (p.cw - 1))) * 32 + (d_g & 3) * 8;
    }
    int x_pr[XR], x_pc[XR];
    long long x_g[XR];
#pragma unroll
    for (int q = 0; q < XR; ++q) {
        const int pp = lp + 32 * q;
        x_pr[q] = pp / pw - 1;
        x_pc[q] = pp - (x_pr[q] + 1) * pw - 1;
        const int g = ls ^ (((pp >> 1) & 1) << 2);
        x_g[q] = (long long)((ci0 >> 5) + (g >> 2)) * p.rows * 32 + (g & 3) * 8;
    }
    auto dma_chunk = [&](int c, int stage) {
        const int per_img = p.chunk_rows_per_img * p.chunks_per_row;
        const int b = c / per_img;
        const int rem = c - b * per_img;
        const int cr = rem / p.chunks_per_row, cc = rem - cr * p.chunks_per_row;
        const int ho0 = cr * p.rpc, wo0 = cc * p.cw;
        u16* D = smem + stage * STAGE;
        u16* X = D + 3 * DPX * 64;
        const u16* dbase = p.dy3 + ((long long)(b * p.H + ho0) * p.W + wo0) * 32;
        const u16* xsrc = p.x3 + ((long long)(b * p.H + ho0) * p.W + wo0) * 32;
#pragma unroll
        for (int q = 0; q < DR; ++q) {
            if (NG > 1 && (q & 1) != grp) continue;          // the copy rounds of a chunk (dy rounds, then patch rounds) alternate between the groups
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(dbase + d_off[q] + pl * p.dps),
                                                 (__attribute__((address_space(3))) void*)(D + (pl * DPX + 32 * q + 8 * wave_u) * 64), 16, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < XR; ++q) {
            if (q * 32 + 8 * wave_u >= npatch || q * 32 + 8 * wave_u >= XPX) break;     // wave-uniform: nothing of this round lies inside the patch
            if (NG > 1 && ((q + DR) & 1) != grp) continue;
            const int hi = ho0 + x_pr[q], wi = wo0 + x_pc[q];
            const bool ok = (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;     // (slots past the patch read anything)
            const u16* src = ok ? xsrc + (x_pr[q] * p.W + x_pc[q]) * 32 + x_g[q] : zsrc;
            const long long ps = ok ? p.xps : 0;
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + pl * ps),
                                                 (__attribute__((address_space(3))) void*)(X + (pl * XPX + 32 * q + 8 * wave_u) * 64), 16, 0, 0);
        }
    };

    // ---- fragment addresses (u16 element offsets inside a stage, plane 0), fixed for the whole kernel.
    // lane: t = lane & 15 (row j = t >> 2 of the 4-pixel group, channel quad t & 3), 16-channel half (lane >> 4) & 1, k half lane >> 5.
    const int t = lane & 15, ch16 = (lane >> 4) & 1, kh = lane >> 5;
    auto row_off = [&](int slot_px, int c4) {            // element offset of 4 contiguous channels c4 .. c4+3 (of 64) in pixel slot slot_px
        const int g = (c4 >> 3) ^ (((slot_px >> 1) & 1) << 2);
        return slot_px * 64 + g * 8 + (c4 & 4);
    };
    int d_fo[KSG][2], x_fo[KSG][2];     // [this group's k step][read]; group g takes k steps g, g + NG, ...
#pragma unroll
    for (int ks = 0; ks < KSG; ++ks)
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int px = (grp + NG * ks) * 16 + kh * 8 + r * 4 + (t >> 2);
            d_fo[ks][r] = row_off(px, wm * 32 + ch16 * 16 + (t & 3) * 4);
            x_fo[ks][r] = (px >> p.cw_log2) * pw + (px & (p.cw - 1));          // patch slot of tap (0,0); the channel part is added per tap
        }
    const int x_c4 = wn * 32 + ch16 * 16 + (t & 3) * 4;

    f32x16 acc[9];
#pragma unroll
    for (int tp = 0; tp < 9; ++tp)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[tp][q] = 0.f;
    if (cbeg < cend) dma_chunk(cbeg, 0);
    constexpr int TA[6] = {1, 0, 2, 0, 1, 0};            // plane pairs (dy, x) of the six products, smallest terms first
    constexpr int TB[6] = {1, 2, 0, 1, 0, 0};
    for (int c = cbeg; c < cend; ++c) {
        const int stage = (c - cbeg) & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // my copies of chunk c have landed ...
        __builtin_amdgcn_s_barrier();                         // ... everybody's have, and the other stage is no longer being read
        asm volatile("" ::: "memory");
        if (c + 1 < cend && !(ABL & 1)) dma_chunk(c + 1, stage ^ 1);
        const u16* D = smem + ((ABL & 1) ? 0 : stage) * STAGE;
        const u16* X = D + 3 * DPX * 64;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < KSG; ++ks) {
            bf16x8w a[3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) a[pl] = tr_frag(D + pl * DPX * 64 + d_fo[ks][0], D + pl * DPX * 64 + d_fo[ks][1]);
            // taps in pairs: the six products of a tap accumulate into one tile, so two taps' chains are interleaved (a dependent
            // MFMA issued back to back waits for its predecessor's last pass)
#pragma unroll
            for (int tp0 = 0; tp0 < 9; tp0 += W3X_TG) {
                bf16x8w b[W3X_TG][3];
#pragma unroll
                for (int u = 0; u < W3X_TG; ++u) {
                    const int tp = tp0 + u;
                    if (tp >= 9) break;
                    const int sh = (ABL & 4) ? 0 : (tp / 3) * pw + (tp % 3);      // (ablation 4: every tap reads tap 0's fragments -- the compiler merges the reads)
                    const int o0 = row_off(x_fo[ks][0] + sh, x_c4), o1 = row_off(x_fo[ks][1] + sh, x_c4);
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) b[u][pl] = tr_frag(X + pl * XPX * 64 + o0, X + pl * XPX * 64 + o1);
                }
                if (ABL & 2) {
#pragma unroll
                    for (int u = 0; u < W3X_TG; ++u)
                        if (tp0 + u < 9)
#pragma unroll
                            for (int pl = 0; pl < 3; ++pl) acc[0][pl] += (float)b[u][pl][0] + (float)a[pl][1];
                    continue;
                }
#pragma unroll
                for (int e = 0; e < 6; ++e)
#pragma unroll
                    for (int u = 0; u < W3X_TG; ++u)
                        if (tp0 + u < 9) acc[tp0 + u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[TA[e]], b[u][TB[e]], acc[tp0 + u], 0, 0, 0);
            }
        }
        __builtin_amdgcn_s_setprio(0);
    }
    if constexpr (NG > 1) {
        // group 1 -> LDS -> group 0, three taps per round (48 KB), fixed order: acc(group 0) + acc(group 1)
        float* R = smem_f + ((wave * 3) * 16) * 64 + lane;
#pragma unroll
        for (int rd = 0; rd < 3; ++rd) {
            __syncthreads();                                  // the stages (round 0) / the previous round's values are no longer needed
            if (grp == 1) {
#pragma unroll
                for (int tp = 0; tp < 3; ++tp)
#pragma unroll
                    for (int q = 0; q < 16; ++q) R[(tp * 16 + q) * 64] = acc[rd * 3 + tp][q];
            }
            __syncthreads();
            if (grp == 0) {
#pragma unroll
                for (int tp = 0; tp < 3; ++tp)
#pragma unroll
                    for (int q = 0; q < 16; ++q) acc[rd * 3 + tp][q] += R[(tp * 16 + q) * 64];
            }
        }
        if (grp != 0) return;
    }
    const int i = lane & 31;
    float* o = p.part + (long long)blockIdx.y * p.Cout * 9 * p.Cin;
#pragma unroll
    for (int tp = 0; tp < 9; ++tp)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int co = co0 + wm * 32 + mfma_row(q, lane);
            o[((long long)co * 9 + tp) * p.Cin + ci0 + wn * 32 + i] = acc[tp][q];
        }
}

// dW_oihw[co][ci][r][s] = sum_split part[split][co][tap][ci]   (fixed order)
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, int splits,
                                                           int Cout, int Cin, int RS, int accumulate) {
    // 256 consecutive KRSC outputs per workgroup (float4 per lane: 1 KiB per wave-load); the 4 waves take interleaved
    // splits, 4 independent loads in flight per lane; fixed summation order; combined through LDS.
    // n is a multiple of 256 (Cin % 64 == 0, Cout % 64 == 0).
    __shared__ f32x4 red[4][64];
    const long long n = (long long)Cout * RS * Cin;
    const int lane = threadIdx.x & 63, sg = threadIdx.x >> 6;
    const long long idx = ((long long)blockIdx.x * 64 + lane) * 4;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    int k = sg;
    for (; k + 12 < splits; k += 16) {
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(part + (long long)k * n + idx);
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(part + (long long)(k + 4) * n + idx);
        const f32x4 v2 = *reinterpret_cast<const f32x4*>(part + (long long)(k + 8) * n + idx);
        const f32x4 v3 = *reinterpret_cast<const f32x4*>(part + (long long)(k + 12) * n + idx);
        s += (v0 + v1) + (v2 + v3);
    }
    for (; k < splits; k += 4) s += *reinterpret_cast<const f32x4*>(part + (long long)k * n + idx);
    red[sg][lane] = s;
    __syncthreads();
    if (sg != 0) return;
    s = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
    const int ci = (int)(idx % Cin);           // 4 consecutive ci of one (co, tap)
    long long t = idx / Cin;
    const int tap = (int)(t % RS);
    const int co = (int)(t / RS);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const long long o = ((long long)co * Cin + ci + e) * RS + tap;
        dw[o] = accumulate ? dw[o] + s[e] : s[e];
    }
}

// =====================================================================================================
// stem weight gradient (no data gradient is needed: the network input does not require grad)
// =====================================================================================================
// The network input is the proxy representation: a silhouette + 17 truncated joint heat-maps, ~98 % exact zeros.  The
// kernel is organised around that: a workgroup owns ONE GROUP OF 4 INPUT CHANNELS (grid.y) and walks 2-row x 32-column
// output tiles; a pre-pass over the non-zero bit map of the input (straps_stem_nzmask -> stem_tileact_kernel) tells it --
// before touching LDS, x or dy -- which of its channels hold anything inside the tile's 9 x 72 input patch: most (tile,
// group) pairs are skipped outright (a workgroup fetches the activity of 64 tiles with one load), and inside an active
// pair only the active channels are loaded.
// A skipped contribution is 0 * dy = 0, so the result equals the dense one bit for bit; a dense input skips nothing.
// Per group the accumulator is dW[64][4*49 = 196 taps] = 14 MFMA tiles of 32x32 over the 4 waves (64 accumulator
// registers), so 4-5 workgroups are resident per CU and cover each other's load latency (the all-channel variant needed
// 472 registers: one wave per SIMD).
constexpr int TY = 2, TX = 32, PH = 2 * TY + 5, PW = 72, TPIX = TY * TX;
constexpr int CG = 4, GK = CG * 49, GNB = (GK + 31) / 32, GUNITS = 2 * GNB, UPW = (GUNITS + 3) / 4;   // 196 taps, 7 column blocks, 14 units, 4 per wave

// per (channel group, tile): which of the group's channels have a non-zero inside the tile's 9 x 72 input patch (bit cl)
__global__ __launch_bounds__(256) void stem_tileact_kernel(const unsigned* __restrict__ nzmask, uint8_t* __restrict__ tact, int C, int H, int W,
                                                           int tiles_x, int tiles_y, int ntiles, int groups) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= ntiles * groups) return;
    const int grp = idx / ntiles, tile = idx - grp * ntiles;
    int bid = tile;
    const int b = bid / (tiles_x * tiles_y);
    bid -= b * tiles_x * tiles_y;
    const int ty = bid / tiles_x, tx = bid - ty * tiles_x;
    const int hi0 = 2 * (ty * TY) - 3, wi0 = 2 * (tx * TX) - 3;
    const int HC = (H + 3) >> 2, WW = (W + 255) >> 8;
    const int rc0 = max(hi0, 0) >> 2, rc1 = min(hi0 + PH - 1, H - 1) >> 2;
    const int wa = max(wi0 - 1, 0), wb = min(wi0 - 2 + PW, W - 1);
    const int c0 = grp * CG, nc = min(CG, C - c0);
    unsigned chact = 0;
    for (int cl = 0; cl < nc; ++cl) {
        unsigned any = 0;
        for (int rcell = rc0; rcell <= rc1; ++rcell)
            for (int word = wa >> 8; word <= (wb >> 8); ++word) {
                const int lo = max(wa, word << 8), hi = min(wb, (word << 8) + 255);
                const int b0 = (lo >> 3) & 31, b1 = (hi >> 3) & 31;
                const unsigned bits = (b1 == 31 ? 0xffffffffu : ((1u << (b1 + 1)) - 1u)) & ~((1u << b0) - 1u);
                any |= nzmask[(((long long)b * C + c0 + cl) * HC + rcell) * WW + word] & bits;
            }
        if (any) chact |= 1u << cl;
    }
    tact[idx] = (uint8_t)chact;
}

// per tile: does ANY input channel hold a non-zero inside the tile's input patch (= is the tile of dy read by stem_wgrad_kernel at all).
// 32 lanes per tile, one channel each (a few independent word loads per lane), combined with a ballot.
__global__ __launch_bounds__(256) void stem_tileany_kernel(const unsigned* __restrict__ nzmask, uint8_t* __restrict__ tany, int C, int H, int W,
                                                           int tiles_x, int tiles_y, int ntiles) {
    const int lane = threadIdx.x & 63, sub = lane & 31;
    const int tile = (blockIdx.x * 256 + threadIdx.x) >> 5;
    unsigned any = 0;
    if (tile < ntiles) {
        int bid = tile;
        const int b = bid / (tiles_x * tiles_y);
        bid -= b * tiles_x * tiles_y;
        const int ty = bid / tiles_x, tx = bid - ty * tiles_x;
        const int hi0 = 2 * (ty * TY) - 3, wi0 = 2 * (tx * TX) - 3;
        const int HC = (H + 3) >> 2, WW = (W + 255) >> 8;
        const int rc0 = max(hi0, 0) >> 2, rc1 = min(hi0 + PH - 1, H - 1) >> 2;
        const int wa = max(wi0 - 1, 0), wb = min(wi0 - 2 + PW, W - 1);
        for (int c = sub; c < C; c += 32)
            for (int rcell = rc0; rcell <= rc1; ++rcell)
                for (int word = wa >> 8; word <= (wb >> 8); ++word) {
                    const int lo = max(wa, word << 8), hi = min(wb, (word << 8) + 255);
                    const int b0 = (lo >> 3) & 31, b1 = (hi >> 3) & 31;
                    const unsigned bits = (b1 == 31 ? 0xffffffffu : ((1u << (b1 + 1)) - 1u)) & ~((1u << b0) - 1u);
                    any |= nzmask[(((long long)b * C + c) * HC + rcell) * WW + word] & bits;
                }
    }
    const unsigned long long bal = __ballot(any != 0u);
    const unsigned mine = (unsigned)(bal >> (lane & 32));            // the 32 lanes of this tile
    if (sub == 0 && tile < ntiles) tany[tile] = mine ? 1 : 0;
}

__global__ __launch_bounds__(256) void stem_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                         float* __restrict__ part, const uint8_t* __restrict__ tact, int B, int C, int H,
                                                         int W, int Ho, int Wo,
                                                         int tiles_x, int tiles_y, int Kp, int ntiles, int rot) {
    __shared__ __attribute__((aligned(16))) float patch[2 * CG * PH * PW]; // [buffer][cl][PH][PW]
    __shared__ __attribute__((aligned(16))) float dys[2 * TPIX * 68];      // [buffer][pixel][64 + 4]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, h = lane >> 5;
    const int c0 = blockIdx.y * CG;
    const int nc = min(CG, C - c0);
    const int gk = nc * 49;                                                // valid taps of this group

    // this wave's units u = wave + 4*j = 2*nb + a: (tap column block nb, co half a); patch offset of this lane's tap; channel span
    int u_a[UPW], u_nb[UPW], u_ko[UPW];
    unsigned u_span[UPW];
#pragma unroll
    for (int j = 0; j < UPW; ++j) {
        const int u = wave + 4 * j;
        u_a[j] = u & 1;                     // consecutive column blocks x both co halves land on 4 different waves: the
        u_nb[j] = u >> 1;                   // 4 units of one active channel (49 taps ~ 1.5 column blocks) run in parallel
        const int k = u_nb[j] * 32 + i;
        int o = 0;
        if (u < GUNITS && k < gk) {
            const int cl = k / 49, rs = k - cl * 49;
            const int r = rs / 7, s_ = rs - r * 7;
            o = (cl * PH + r) * PW + s_ + 1;                               // +1: patch origin one column left of the receptive field
        }
        u_ko[j] = o;
        const int k0 = u_nb[j] * 32;
        u_span[j] = 0;
        if (u < GUNITS && k0 < gk) {
            const int cl0 = k0 / 49, cl1 = min(k0 + 31, gk - 1) / 49;
            u_span[j] = (1u << cl0) | (1u << cl1);
        }
    }
    f32x16 acc[UPW];
#pragma unroll
    for (int j = 0; j < UPW; ++j)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[j][q] = 0.f;

    // Rotated strided tile assignment (evens out body vs background tiles, see the walk below).  The activity bytes of 64 tiles are fetched with one
    // load and only the active tiles are visited -- in ascending order, so the accumulation order is fixed.  The visit is
    // software-pipelined over a double-buffered LDS: while the MFMAs of tile t run, the global loads of the next active
    // tile are already in flight (registers), and one barrier per tile is enough (a buffer is rewritten two tiles later,
    // after every wave has passed the barrier in between).
    constexpr int PIT = (CG * PH * (PW / 4) + 255) / 256;                  // 3 float4 of the patch per thread
    constexpr int DIT = TPIX * 16 / 256;                                   // 4 float4 of the dy tile per thread
    const bool vec = (W & 3) == 0;
    f32x4 pv[PIT], dv[DIT];
    auto tile_origin = [&](int tile, int& b, int& y0, int& x0) {
        b = tile / (tiles_x * tiles_y);
        const int bid = tile - b * tiles_x * tiles_y;
        const int ty = bid / tiles_x;
        y0 = ty * TY;
        x0 = (bid - ty * tiles_x) * TX;
    };
    auto issue_loads = [&](int tile, unsigned chact) {                     // global -> registers, nothing waits here
        int b, y0, x0;
        tile_origin(tile, b, y0, x0);
        const int hi0 = 2 * y0 - 3, wi0 = 2 * x0 - 3;
#pragma unroll
        for (int it = 0; it < PIT; ++it) {
            const int idx = tid + it * 256;
            const int q = idx % (PW / 4);
            const int rc = idx / (PW / 4);
            const int row = rc % PH, cl = rc / PH;
            const int hi = hi0 + row, wi = wi0 - 1 + 4 * q;                // patch column p holds input column wi0 - 1 + p
            pv[it] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (vec && cl < nc && ((chact >> cl) & 1u) && (unsigned)hi < (unsigned)H && wi >= 0 && wi < W)
                pv[it] = *reinterpret_cast<const f32x4*>(x + (((long long)b * C + c0 + cl) * H + hi) * W + wi);
        }
#pragma unroll
        for (int it = 0; it < DIT; ++it) {
            const int idx = tid + it * 256;
            const int pix = idx >> 4, c4 = idx & 15;
            const int yo = y0 + (pix >> 5), xo = x0 + (pix & 31);
            dv[it] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (yo < Ho && xo < Wo) dv[it] = *reinterpret_cast<const f32x4*>(dy + (((long long)b * Ho + yo) * Wo + xo) * 64 + c4 * 4);
        }
    };
    auto store_tile = [&](int tile, unsigned chact, float* pb, float* db) {        // registers -> LDS buffer
        if (vec) {
#pragma unroll
            for (int it = 0; it < PIT; ++it) {
                const int idx = tid + it * 256;
                if (idx < nc * PH * (PW / 4)) *reinterpret_cast<f32x4*>(pb + (idx / (PW / 4)) * PW + 4 * (idx % (PW / 4))) = pv[it];
            }
        } else {                                                           // odd widths: scalar fill, not pipelined
            int b, y0, x0;
            tile_origin(tile, b, y0, x0);
            const int hi0 = 2 * y0 - 3, wi0 = 2 * x0 - 3;
            for (int idx = tid; idx < nc * PH * PW; idx += 256) {
                const int col = idx % PW;
                const int rc = idx / PW;
                const int row = rc % PH, cl = rc / PH;
                const int hi = hi0 + row, wi = wi0 - 1 + col;
                float v = 0.f;
                if (((chact >> cl) & 1u) && (unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W) v = x[(((long long)b * C + c0 + cl) * H + hi) * W + wi];
                pb[idx] = v;
            }
        }
#pragma unroll
        for (int it = 0; it < DIT; ++it) {
            const int idx = tid + it * 256;
            *reinterpret_cast<f32x4*>(db + (idx >> 4) * 68 + (idx & 15) * 4) = dv[it];
        }
    };

    const uint8_t* myact = tact + (long long)blockIdx.y * ntiles;
    int buf = 0;
    // tile walk: "row" m of the tile list holds tiles m*G .. m*G + G - 1 (G = gridDim.x); this workgroup takes element (bx + 97 m) mod G of
    // every row.  With G = tiles per image (the B = 64, 256 x 256 case) a plain bx + m*G would pin a workgroup to ONE image position in all
    // images -- the body sits in the middle of every proxy image, so a quarter of the workgroups would do all the work; the rotation walks
    // each workgroup through the whole image instead.  Any G: a bijection per row, ascending tile order (fixed accumulation order).
    const int G = gridDim.x;
    for (int m0 = 0; (long long)m0 * G < ntiles; m0 += 64) {
        const int m = m0 + lane;
        const long long mine_l = (long long)m * G + (blockIdx.x + (long long)rot * m) % G;
        const int mine = mine_l < ntiles ? (int)mine_l : -1;
        const unsigned act = mine >= 0 ? myact[mine] : 0u;
        unsigned long long todo = __ballot(act != 0u);
        if (todo) {
            const int l0 = __ffsll((long long)todo) - 1;
            issue_loads(__shfl(mine, l0, 64), (unsigned)__shfl((int)act, l0, 64));
        }
        while (todo) {
            const int l = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            const int tile = __shfl(mine, l, 64);
            const unsigned chact = (unsigned)__shfl((int)act, l, 64);
            float* pb = patch + buf * (CG * PH * PW);
            float* db = dys + buf * (TPIX * 68);
            store_tile(tile, chact, pb, db);
            __syncthreads();
            if (todo) {                                                    // next active tile of this chunk: loads fly during the MFMAs
                const int ln = __ffsll((long long)todo) - 1;
                issue_loads(__shfl(mine, ln, 64), (unsigned)__shfl((int)act, ln, 64));
            }
            // contraction over the tile's 64 pixels: A[co][pix] = dys[pix][co], B[pix][k] = patch[koff[k] + 2*py*PW + 2*px].
            // Unit-outer / pixel-inner and fully unrolled: every LDS operand is base + immediate offset (a unit is skipped
            // when its taps read only all-zero channels of this tile -- wave-uniform).
#pragma unroll
            for (int j = 0; j < UPW; ++j) {
                if (!(u_span[j] & chact)) continue;
                const float* ap = db + (h * 4) * 68 + u_a[j] * 32 + i;
                const float* bp = pb + u_ko[j] + 8 * h;
                // all 64 operands of the unit are fetched first (the scheduling barrier keeps the compiler from re-serialising
                // read -> wait -> MFMA pairs through two registers), then the 32-MFMA chain runs without LDS waits
                float av[32], bv[32];
#pragma unroll
                for (int g = 0; g < TPIX / 8; ++g)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        av[g * 4 + e] = ap[(g * 8 + e) * 68];
                        bv[g * 4 + e] = bp[(g >> 2) * 2 * PW + (g & 3) * 16 + 2 * e];
                    }
                __builtin_amdgcn_sched_barrier(0);
                f32x16 c = acc[j];
#pragma unroll
                for (int t = 0; t < 32; ++t) c = mfma32(av[t], bv[t], c);
                acc[j] = c;
                __builtin_amdgcn_sched_barrier(0);
            }
            buf ^= 1;
        }
        __syncthreads();          // chunk boundary: the next chunk's first store may target the buffer still being read
    }
    // partial[block][co][k], k = (c0 + cl)*49 + rs: the group's taps are a contiguous column range
    float* o = part + (long long)blockIdx.x * 64 * Kp + c0 * 49;
#pragma unroll
    for (int j = 0; j < UPW; ++j) {
        const int kcol = u_nb[j] * 32 + i;
        if (wave + 4 * j < GUNITS && kcol < gk) {
#pragma unroll
            for (int q = 0; q < 16; ++q) o[(long long)(u_a[j] * 32 + mfma_row(q, lane)) * Kp + kcol] = acc[j][q];
        }
    }
}

// fixed-order sum of the per-block partials: 64 consecutive outputs per workgroup, the four waves take the blocks b = wave, wave + 4, ...
// (independent loads, eight in flight per lane), their four sums are added in wave order -- the same order on every launch.
__global__ __launch_bounds__(256) void stem_wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, int nblocks,
                                                                int K, int Kp, int accumulate) {
    __shared__ float red[4][64];
    const int o = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int idx = blockIdx.x * 64 + o;              // over [64][K]
    const bool live = idx < 64 * K;
    const int co = live ? idx / K : 0, k = live ? idx - co * K : 0;
    const float* src = part + (long long)co * Kp + k;
    const long long bstride = (long long)64 * Kp;
    float s = 0.f;
    if (live) {
#pragma unroll 8
        for (int b = g; b < nblocks; b += 4) s += src[b * bstride];
    }
    red[g][o] = s;
    __syncthreads();
    if (g == 0 && live) {
        const float t = ((red[0][o] + red[1][o]) + red[2][o]) + red[3][o];
        dw[idx] = accumulate ? dw[idx] + t : t;       // OIHW row co is (c,r,s)-ordered == k
    }
}

// =====================================================================================================
// BatchNorm (training) backward with fused ReLU mask
// =====================================================================================================
// pass 1: per-block partial sums of dz and dz*xhat, dz = dy * (y > 0 if masked).  Block (bx, by) owns rows
// [bx*rows_per_block, ...) x the 64 channels [64*by, 64*by+64): 16 float4 column lanes x 16 row lanes, so a deep layer
// (few rows, many channels) still fills the chip, and every wave load is 4 full 256-byte row segments.
// ReLU mask: (yact > 0), or -- when the activation is exactly relu(raw*msc + msh), no residual -- recomputed from raw
// with the forward's own fmaf (bit-identical to reading yact, one tensor read less).
//
// POOL variants (the stem's BatchNorm, whose output only feeds the 3x3/s2 max-pool): the incoming gradient is never materialised --
// each element gathers it from the <= 4 pooling windows that cover it, in the order straps_maxpool_bwd uses (bit-identical).
struct PoolSrc {
    const float* dyp;       // [B][Ho][Wo][C] gradient of the pooled tensor
    const uint8_t* idx;     // [B][Ho][Wo][C] arg-max tap of every window
    int H, W, Ho, Wo;
    // optional (the stem): tany[(b * tiles_y + h / 2) * tiles_x + w / 32] == 0 marks a 2-row x 32-column tile of the un-pooled grid whose
    // gradient nobody reads (the stem weight gradient skips it: no non-zero input under it) -- the apply pass leaves it unwritten
    const uint8_t* tany;
    int tiles_x, tiles_y;
};

__device__ __forceinline__ f32x4 pool_grad(const PoolSrc& ps, long long row, int c4, int C) {
    const int wi = (int)(row % ps.W);
    const long long t = row / ps.W;
    const int hi = (int)(t % ps.H);
    const long long b = t / ps.H;
    f32x4 g = {0.f, 0.f, 0.f, 0.f};
    const int ho_lo = hi >> 1, ho_hi = (hi + 1) >> 1;        // windows (ho,wo) with 2*ho-1 <= hi <= 2*ho+1
    const int wo_lo = wi >> 1, wo_hi = (wi + 1) >> 1;
    for (int ho = ho_lo; ho <= ho_hi; ++ho) {
        if (ho >= ps.Ho) continue;
        const int r = hi - (2 * ho - 1);
        for (int wo = wo_lo; wo <= wo_hi; ++wo) {
            if (wo >= ps.Wo) continue;
            const int want = r * 3 + wi - (2 * wo - 1);
            const long long o = ((b * ps.Ho + ho) * ps.Wo + wo) * C + c4 * 4;
            const uchar4 k = *reinterpret_cast<const uchar4*>(ps.idx + o);
            const f32x4 d = *reinterpret_cast<const f32x4*>(ps.dyp + o);
            if (k.x == want) g[0] += d[0];
            if (k.y == want) g[1] += d[1];
            if (k.z == want) g[2] += d[2];
            if (k.w == want) g[3] += d[3];
        }
    }
    return g;
}

// The three BatchNorm-backward kernels accumulate and subtract in DOUBLE: in the last stage the gradient that reaches a BatchNorm
// is almost constant per channel (global average pooling broadcasts one value to the 64 positions), so dz - mean(dz) cancels to
// rounding noise of fp32 sums -- measured 4e-4..2e-3 relative on the layer4 gradients against 2e-6 for the reference's CPU path,
// whose BatchNorm backward accumulates in double (ATen acc_type<float, cpu>).  The passes are HBM-bound; the fp64 VALU work is free.
template <bool POOL>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float* __restrict__ dy, const float* __restrict__ yact,
                                                            const float* __restrict__ raw, const float* __restrict__ mean,
                                                            const float* __restrict__ invstd, const float* __restrict__ msc,
                                                            const float* __restrict__ msh, double* __restrict__ part, long long rows,
                                                            int C, int rows_per_block, PoolSrc ps, const unsigned* __restrict__ bits) {
    // (bits: the ReLU decisions of yact as one word per (row, 32 channels) -- straps_bn_apply_bits_x3 -- read instead of yact itself)
    __shared__ double red[256][8];
    constexpr int TR = 16;
    const int tc = threadIdx.x & 15, tr = threadIdx.x >> 4;
    const int c4 = blockIdx.y * 16 + tc;
    const bool active = c4 * 4 < C;
    const long long r0 = (long long)blockIdx.x * rows_per_block;
    const long long r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
    double s1[4] = {0.0, 0.0, 0.0, 0.0}, s2[4] = {0.0, 0.0, 0.0, 0.0};       // s2 = sum g * (x - mean); invstd is applied once at the end
    f32x4 is = {0.f, 0.f, 0.f, 0.f};
    if (active) {
        const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + c4 * 4);
        is = *reinterpret_cast<const f32x4*>(invstd + c4 * 4);
        f32x4 ksc = {0.f, 0.f, 0.f, 0.f}, ksh = ksc;
        if (msc) { ksc = *reinterpret_cast<const f32x4*>(msc + c4 * 4); ksh = *reinterpret_cast<const f32x4*>(msh + c4 * 4); }
        // 4 rows per trip: 8-12 independent float4 loads in flight per lane before the first use
        long long r = r0 + tr;
        for (; r + 3 * TR < r1; r += 4 * TR) {
            f32x4 g[4], ya[4], xr[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long long o = (r + (long long)u * TR) * C + c4 * 4;
                if constexpr (POOL) g[u] = pool_grad(ps, r + (long long)u * TR, c4, C);
                else g[u] = *reinterpret_cast<const f32x4*>(dy + o);
                xr[u] = *reinterpret_cast<const f32x4*>(raw + o);
                if (bits) ya[u][0] = __uint_as_float(bits[(r + (long long)u * TR) * (C >> 5) + (c4 >> 3)]);
                else if (yact) ya[u] = *reinterpret_cast<const f32x4*>(yact + o);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (bits) {
                    const unsigned nib = __float_as_uint(ya[u][0]) >> ((c4 & 7) * 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) g[u][e] = ((nib >> e) & 1u) ? g[u][e] : 0.f;
                } else if (yact) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) g[u][e] = ya[u][e] > 0.f ? g[u][e] : 0.f;
                } else if (msc) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) g[u][e] = fmaf(xr[u][e], ksc[e], ksh[e]) > 0.f ? g[u][e] : 0.f;
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    s1[e] += (double)g[u][e];
                    s2[e] += (double)g[u][e] * ((double)xr[u][e] - (double)mu[e]);
                }
            }
        }
        for (; r < r1; r += TR) {
            const long long o = r * C + c4 * 4;
            f32x4 g;
            if constexpr (POOL) g = pool_grad(ps, r, c4, C);
            else g = *reinterpret_cast<const f32x4*>(dy + o);
            const f32x4 xr = *reinterpret_cast<const f32x4*>(raw + o);
            if (bits) {
                const unsigned nib = bits[r * (C >> 5) + (c4 >> 3)] >> ((c4 & 7) * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) g[e] = ((nib >> e) & 1u) ? g[e] : 0.f;
            } else if (yact) {
                const f32x4 ya = *reinterpret_cast<const f32x4*>(yact + o);
#pragma unroll
                for (int e = 0; e < 4; ++e) g[e] = ya[e] > 0.f ? g[e] : 0.f;
            } else if (msc) {
#pragma unroll
                for (int e = 0; e < 4; ++e) g[e] = fmaf(xr[e], ksc[e], ksh[e]) > 0.f ? g[e] : 0.f;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                s1[e] += (double)g[e];
                s2[e] += (double)g[e] * ((double)xr[e] - (double)mu[e]);
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) { red[threadIdx.x][e] = s1[e]; red[threadIdx.x][4 + e] = s2[e] * (double)is[e]; }
    __syncthreads();
    if (tr == 0 && active) {
        double t[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) t[e] = red[tc][e];
        for (int q = 1; q < TR; ++q)
#pragma unroll
            for (int e = 0; e < 8; ++e) t[e] += red[q * 16 + tc][e];
        double* o = part + ((long long)blockIdx.x * C + c4 * 4) * 2;
#pragma unroll
        for (int e = 0; e < 4; ++e) { o[e * 2 + 0] = t[e]; o[e * 2 + 1] = t[4 + e]; }
    }
}

// The same sums for the fused stem tail (bn1 + ReLU + max-pool), taken over the POOLED grid: every pooled element routes its gradient to
// exactly one un-pooled position (its arg-max tap), so  S1 = sum_q mask * dy_q,  S2 = sum_q mask * dy_q * (x[argmax(q)] - mean) --
// a quarter of the elements of the un-pooled sweep, one gathered raw value each, no four-window search per element.
// (mask = the ReLU derivative at the arg-max position, re-derived from raw like everywhere else.)
__global__ __launch_bounds__(256) void bn_bwd_reduce_pooled_kernel(const float* __restrict__ raw, const float* __restrict__ mean,
                                                                   const float* __restrict__ invstd, const float* __restrict__ msc,
                                                                   const float* __restrict__ msh, double* __restrict__ part, long long prows,
                                                                   int C, int rows_per_block, PoolSrc ps) {
    __shared__ double red[256][8];
    constexpr int TR = 16;
    const int tc = threadIdx.x & 15, tr = threadIdx.x >> 4;
    const int c4 = blockIdx.y * 16 + tc;
    const bool active = c4 * 4 < C;
    const long long r0 = (long long)blockIdx.x * rows_per_block;
    const long long r1 = r0 + rows_per_block < prows ? r0 + rows_per_block : prows;
    double s1[4] = {0.0, 0.0, 0.0, 0.0}, s2[4] = {0.0, 0.0, 0.0, 0.0};
    f32x4 is = {0.f, 0.f, 0.f, 0.f};
    if (active) {
        const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + c4 * 4);
        is = *reinterpret_cast<const f32x4*>(invstd + c4 * 4);
        const f32x4 ksc = *reinterpret_cast<const f32x4*>(msc + c4 * 4), ksh = *reinterpret_cast<const f32x4*>(msh + c4 * 4);
        for (long long r = r0 + tr; r < r1; r += TR) {
            const int wo = (int)(r % ps.Wo);
            const long long t = r / ps.Wo;
            const int ho = (int)(t % ps.Ho);
            const long long b = t / ps.Ho;
            const long long o = r * C + c4 * 4;
            const f32x4 d = *reinterpret_cast<const f32x4*>(ps.dyp + o);
            const uchar4 k4 = *reinterpret_cast<const uchar4*>(ps.idx + o);
            const int k[4] = {k4.x, k4.y, k4.z, k4.w};
            float xr[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int hi = 2 * ho - 1 + k[e] / 3, wi = 2 * wo - 1 + k[e] % 3;          // (the recorded tap lies inside the image)
                xr[e] = raw[((b * ps.H + hi) * ps.W + wi) * C + c4 * 4 + e];
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float g = fmaf(xr[e], ksc[e], ksh[e]) > 0.f ? d[e] : 0.f;
                s1[e] += (double)g;
                s2[e] += (double)g * ((double)xr[e] - (double)mu[e]);
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) { red[threadIdx.x][e] = s1[e]; red[threadIdx.x][4 + e] = s2[e] * (double)is[e]; }
    __syncthreads();
    if (tr == 0 && active) {
        double t[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) t[e] = red[tc][e];
        for (int q = 1; q < TR; ++q)
#pragma unroll
            for (int e = 0; e < 8; ++e) t[e] += red[q * 16 + tc][e];
        double* o = part + ((long long)blockIdx.x * C + c4 * 4) * 2;
#pragma unroll
        for (int e = 0; e < 4; ++e) { o[e * 2 + 0] = t[e]; o[e * 2 + 1] = t[4 + e]; }
    }
}

// per channel: dbeta = S1, dgamma = S2; coefficients for the apply pass: k1 = gamma*invstd (float), m1 = S1/N, m2 = S2/N (double)
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const double* __restrict__ part, int nblocks, int C, double count,
                                                             const float* __restrict__ gamma, const float* __restrict__ invstd,
                                                             float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                             double* __restrict__ coefd, float* __restrict__ k1, int accumulate) {
    int c;
    double s1, s2;
    if (bn_partials_sum4(part, nblocks, C, s1, s2, c)) {
        dbeta[c] = accumulate ? dbeta[c] + (float)s1 : (float)s1;
        dgamma[c] = accumulate ? dgamma[c] + (float)s2 : (float)s2;
        k1[c] = gamma[c] * invstd[c];
        coefd[c] = s1 / count;
        coefd[C + c] = s2 / count;
    }
}

// pass 2: draw = k1 * (dz - m1 - xhat*m2), evaluated in double and rounded once;  optionally also writes dz (the gradient the
// skip connection receives)
constexpr int POOL_CHUNK = 1024;        // float4 elements per workgroup of the pooled apply pass (64 pixels of a 64-channel tensor)
template <bool POOL>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ yact,
                                                           const float* __restrict__ raw, const float* __restrict__ mean,
                                                           const float* __restrict__ invstd, const double* __restrict__ coefd,
                                                           const float* __restrict__ k1p, const float* __restrict__ msc,
                                                           const float* __restrict__ msh, float* __restrict__ draw, float* dz_out,
                                                           u16* __restrict__ planes, long long pstride, long long n4, int C, PoolSrc ps,
                                                           const unsigned* __restrict__ bits, int tiled) {
    const int C4 = C >> 2;
    // POOL: a workgroup owns POOL_CHUNK consecutive elements (with the tile map most workgroups of an empty region exit at once and the
    // dispatcher hands out the rest: a grid-stride loop would pin every workgroup to one image position -- all work or none)
    const long long first = POOL ? (long long)blockIdx.x * POOL_CHUNK + threadIdx.x : (long long)blockIdx.x * 256 + threadIdx.x;
    const long long last = POOL ? (n4 < ((long long)blockIdx.x + 1) * POOL_CHUNK ? n4 : ((long long)blockIdx.x + 1) * POOL_CHUNK) : n4;
    const long long step = POOL ? 256 : (long long)gridDim.x * 256;
    const long long rows = n4 / C4;
    // per-channel constants of four channels: 5 float4 + 8 doubles.  Round 4: when the loop stride is a multiple of the row length (the host
    // sizes the grid so; always for the pooled form's 256-element stride on 64 channels) a thread keeps its four channels for the whole loop
    // and loads them ONCE -- they were 144 bytes of L1 traffic per 48-64 bytes of payload -- and the row index advances by a constant
    // instead of a 64-bit division per element.  Same arithmetic either way.
    struct Consts { f32x4 ksc, ksh, mu, is, k1; double m1[4], m2[4]; };
    auto load_consts = [&](int c4) {
        Consts k;
        if (!yact && !bits && msc) { k.ksc = *reinterpret_cast<const f32x4*>(msc + c4 * 4); k.ksh = *reinterpret_cast<const f32x4*>(msh + c4 * 4); }
        k.mu = *reinterpret_cast<const f32x4*>(mean + c4 * 4);
        k.is = *reinterpret_cast<const f32x4*>(invstd + c4 * 4);
        k.k1 = *reinterpret_cast<const f32x4*>(k1p + c4 * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { k.m1[e] = coefd[c4 * 4 + e]; k.m2[e] = coefd[C + c4 * 4 + e]; }
        return k;
    };
    auto body = [&](long long idx, long long row, int c4, const Consts& k) {
        f32x4 g;
        if constexpr (POOL) {
            if (ps.tany) {
                const int wi = (int)(row % ps.W);
                const long long t = row / ps.W;
                const int hi = (int)(t % ps.H);
                const long long b = t / ps.H;
                if (!ps.tany[(b * ps.tiles_y + (hi >> 1)) * ps.tiles_x + (wi >> 5)]) return;
            }
            g = pool_grad(ps, row, c4, C);
        } else {
            g = *reinterpret_cast<const f32x4*>(dy + idx * 4);
        }
        const f32x4 xr = *reinterpret_cast<const f32x4*>(raw + idx * 4);
        if (bits) {      // (ReLU decisions of yact as bits: word [row][C / 32], bit c & 31 -- straps_bn_apply_bits_x3)
            const unsigned nib = bits[row * (C >> 5) + (c4 >> 3)] >> ((c4 & 7) * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) g[e] = ((nib >> e) & 1u) ? g[e] : 0.f;
        } else if (yact) {
            const f32x4 ya = *reinterpret_cast<const f32x4*>(yact + idx * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) g[e] = ya[e] > 0.f ? g[e] : 0.f;
        } else if (msc) {
#pragma unroll
            for (int e = 0; e < 4; ++e) g[e] = fmaf(xr[e], k.ksc[e], k.ksh[e]) > 0.f ? g[e] : 0.f;
        }
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const double xh = ((double)xr[e] - (double)k.mu[e]) * (double)k.is[e];
            o[e] = (float)((double)k.k1[e] * (((double)g[e] - k.m1[e]) - xh * k.m2[e]));
        }
        if (dz_out) *reinterpret_cast<f32x4*>(dz_out + idx * 4) = g;
        if (draw) *reinterpret_cast<f32x4*>(draw + idx * 4) = o;      // (NULL: only the planes are consumed, see straps_bn_bwd_x3)
        if (planes) store_planes4_cm(planes, pstride, row, c4 * 4, rows, o);      // bf16x3 route: the data-gradient kernel's operand (chunk-major planes)
    };
    if (!POOL && tiled) {
        // a wave = 4 rows x 2 chunks of 32 channels: whole 256-byte runs of every chunk-major plane per wave store; `tiled` = the column
        // groups of 64 channels the workgroup's waves sit on side by side (bn_apply_kernel's tiled form, csrc/elementwise.hip; the host
        // passes it only for row / channel counts that tile exactly, with a grid that is a multiple of the column blocks)
        const int l = threadIdx.x & 63, w = threadIdx.x >> 6, wcg = tiled, ncb = (C4 >> 4) / wcg, trows = 16 / wcg;
        const int c4 = ((int)(blockIdx.x % ncb) * wcg + w % wcg) * 16 + (l >> 5) * 8 + (l & 7);
        const Consts k = load_consts(c4);
        const long long rstep = (long long)(gridDim.x / ncb) * trows;
        for (long long row = (long long)(blockIdx.x / ncb) * trows + (w / wcg) * 4 + ((l >> 3) & 3); row < rows; row += rstep) body(row * C4 + c4, row, c4, k);
    } else if (step % C4 == 0) {
        if (first >= last) return;
        const int c4 = (int)(first % C4);
        const Consts k = load_consts(c4);
        const long long rstep = step / C4;
        long long row = first / C4;
        for (long long idx = first; idx < last; idx += step, row += rstep) body(idx, row, c4, k);
    } else {
        for (long long idx = first; idx < last; idx += step) {
            const int c4 = (int)(idx % C4);
            body(idx, idx / C4, c4, load_consts(c4));
        }
    }
}

// =====================================================================================================
// pooling backward
// =====================================================================================================
// training-mode max-pool: also records the arg-max tap (first maximum in row-major scan order, like ATen)
__global__ __launch_bounds__(256) void maxpool_idx_kernel(const float* __restrict__ x, float* __restrict__ y, uint8_t* __restrict__ idx,
                                                          int B, int H, int W, int C, int Ho, int Wo) {
    const int C4 = C >> 2;
    const long long n = (long long)B * Ho * Wo * C4;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int c4 = (int)(i % C4);
        long long t = i / C4;
        const int wo = (int)(t % Wo); t /= Wo;
        const int ho = (int)(t % Ho);
        const int b = (int)(t / Ho);
        f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        int am[4] = {0, 0, 0, 0};
        bool first = true;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int hi = 2 * ho - 1 + r;
            if ((unsigned)hi >= (unsigned)H) continue;
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const int wi = 2 * wo - 1 + s;
                if ((unsigned)wi >= (unsigned)W) continue;
                const f32x4 v = *reinterpret_cast<const f32x4*>(x + (((long long)b * H + hi) * W + wi) * C + c4 * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (first || v[e] > m[e] || v[e] != v[e]) { m[e] = v[e]; am[e] = r * 3 + s; }
                first = false;
            }
        }
        *reinterpret_cast<f32x4*>(y + i * 4) = m;
        *reinterpret_cast<uchar4*>(idx + i * 4) = make_uchar4((unsigned char)am[0], (unsigned char)am[1], (unsigned char)am[2], (unsigned char)am[3]);
    }
}

// training-mode stem tail: max-pool 3x3/s2/p1 of y = relu(raw*scale + shift) straight from raw -- y itself is never written (the
// backward re-derives the ReLU mask from raw and gathers the pooled gradient, straps_bn_bwd_pooled).  Same fmaf / fmaxf / compare
// sequence as straps_bn_apply followed by straps_maxpool_fwd_idx: bit-identical outputs and arg-max taps.
__global__ __launch_bounds__(256) void bn_relu_maxpool_kernel(const float* __restrict__ raw, const float* __restrict__ scale,
                                                              const float* __restrict__ shift, float* __restrict__ y, uint8_t* __restrict__ idx,
                                                              u16* __restrict__ planes, long long pstride, int B, int H, int W, int C, int Ho, int Wo) {
    const int C4 = C >> 2;
    const long long n = (long long)B * Ho * Wo * C4;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int c4 = (int)(i % C4);
        long long t = i / C4;
        const int wo = (int)(t % Wo); t /= Wo;
        const int ho = (int)(t % Ho);
        const int b = (int)(t / Ho);
        const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + c4 * 4);
        const f32x4 sh = *reinterpret_cast<const f32x4*>(shift + c4 * 4);
        f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        int am[4] = {0, 0, 0, 0};
        bool first = true;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int hi = 2 * ho - 1 + r;
            if ((unsigned)hi >= (unsigned)H) continue;
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const int wi = 2 * wo - 1 + s;
                if ((unsigned)wi >= (unsigned)W) continue;
                f32x4 v = *reinterpret_cast<const f32x4*>(raw + (((long long)b * H + hi) * W + wi) * C + c4 * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = fmaxf(fmaf(v[e], sc[e], sh[e]), 0.f);
                    if (first || v[e] > m[e] || v[e] != v[e]) { m[e] = v[e]; am[e] = r * 3 + s; }
                }
                first = false;
            }
        }
        *reinterpret_cast<f32x4*>(y + i * 4) = m;
        *reinterpret_cast<uchar4*>(idx + i * 4) = make_uchar4((unsigned char)am[0], (unsigned char)am[1], (unsigned char)am[2], (unsigned char)am[3]);
        if (planes) store_planes4_cm(planes, pstride, i / C4, c4 * 4, (long long)B * Ho * Wo, m);
    }
}

__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* __restrict__ dy, const uint8_t* __restrict__ idx,
                                                          float* __restrict__ dx, int B, int H, int W, int C, int Ho, int Wo) {
    const int C4 = C >> 2;
    const long long n = (long long)B * H * W * C4;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int c4 = (int)(i % C4);
        long long t = i / C4;
        const int wi = (int)(t % W); t /= W;
        const int hi = (int)(t % H);
        const int b = (int)(t / H);
        f32x4 g = {0.f, 0.f, 0.f, 0.f};
        // windows (ho,wo) with 2*ho-1 <= hi <= 2*ho+1
        const int ho_lo = hi >> 1, ho_hi = (hi + 1) >> 1;
        const int wo_lo = wi >> 1, wo_hi = (wi + 1) >> 1;
        for (int ho = ho_lo; ho <= ho_hi; ++ho) {
            if (ho >= Ho) continue;
            const int r = hi - (2 * ho - 1);
            for (int wo = wo_lo; wo <= wo_hi; ++wo) {
                if (wo >= Wo) continue;
                const int s = wi - (2 * wo - 1);
                const long long o = (((long long)b * Ho + ho) * Wo + wo) * C + c4 * 4;
                const uchar4 k = *reinterpret_cast<const uchar4*>(idx + o);
                const f32x4 d = *reinterpret_cast<const f32x4*>(dy + o);
                const int want = r * 3 + s;
                if (k.x == want) g[0] += d[0];
                if (k.y == want) g[1] += d[1];
                if (k.z == want) g[2] += d[2];
                if (k.w == want) g[3] += d[3];
            }
        }
        *reinterpret_cast<f32x4*>(dx + i * 4) = g;
    }
}

__global__ __launch_bounds__(256) void gap_bwd_kernel(const float* __restrict__ dfeat, float* __restrict__ dx, int B, int HW, int C) {
    const long long n = (long long)B * HW * C;
    const float inv = 1.0f / (float)HW;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % C);
        const long long b = i / ((long long)HW * C);
        dx[i] = dfeat[b * C + c] * inv;
    }
}

// elementwise: y = x * (mask > 0) (+ y)
__global__ __launch_bounds__(256) void masked_copy_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ mask, int ldm,
                                                          float* y, int ldy, int M, int N, int accumulate) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)M * N) return;
    const int m = (int)(i / N), n = (int)(i - (long long)m * N);
    float v = x[(long long)m * ldx + n];
    if (mask) v = mask[(long long)m * ldm + n] > 0.f ? v : 0.f;
    float* o = y + (long long)m * ldy + n;
    *o = accumulate ? *o + v : v;
}

// =====================================================================================================
// rot6d backward (utils/rigid_transform_utils.py:27-41 differentiated by hand)
// =====================================================================================================
__global__ __launch_bounds__(256) void rot6d_bwd_kernel(const float* __restrict__ x6, long long ld, int per_row,
                                                        const float* __restrict__ dR, float* __restrict__ dx6, long long ldd,
                                                        long long n) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n) return;
    const long long row = idx / per_row;
    const int j = (int)(idx - row * per_row);
    const float* x = x6 + row * ld + j * 6;
    const float a1[3] = {x[0], x[2], x[4]}, a2[3] = {x[1], x[3], x[5]};
    const float* g = dR + idx * 9;                 // dL/dR, R columns = (b1,b2,b3)
    float gb1[3] = {g[0], g[3], g[6]}, gb2[3] = {g[1], g[4], g[7]};
    const float gb3[3] = {g[2], g[5], g[8]};
    const float n1r = sqrtf(a1[0] * a1[0] + a1[1] * a1[1] + a1[2] * a1[2]);
    const float n1 = fmaxf(n1r, 1e-12f);
    const float b1[3] = {a1[0] / n1, a1[1] / n1, a1[2] / n1};
    const float d = b1[0] * a2[0] + b1[1] * a2[1] + b1[2] * a2[2];
    const float u[3] = {a2[0] - d * b1[0], a2[1] - d * b1[1], a2[2] - d * b1[2]};
    const float n2r = sqrtf(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
    const float n2 = fmaxf(n2r, 1e-12f);
    const float b2[3] = {u[0] / n2, u[1] / n2, u[2] / n2};
    // b3 = b1 x b2:  gb1 += b2 x gb3, gb2 += gb3 x b1
    gb1[0] += b2[1] * gb3[2] - b2[2] * gb3[1];
    gb1[1] += b2[2] * gb3[0] - b2[0] * gb3[2];
    gb1[2] += b2[0] * gb3[1] - b2[1] * gb3[0];
    gb2[0] += gb3[1] * b1[2] - gb3[2] * b1[1];
    gb2[1] += gb3[2] * b1[0] - gb3[0] * b1[2];
    gb2[2] += gb3[0] * b1[1] - gb3[1] * b1[0];
    // b2 = u / max(|u|, eps)
    float gu[3];
    if (n2r > 1e-12f) {
        const float t = b2[0] * gb2[0] + b2[1] * gb2[1] + b2[2] * gb2[2];
#pragma unroll
        for (int e = 0; e < 3; ++e) gu[e] = (gb2[e] - b2[e] * t) / n2;
    } else {
#pragma unroll
        for (int e = 0; e < 3; ++e) gu[e] = gb2[e] / n2;
    }
    // u = a2 - d*b1, d = b1.a2
    const float gd = -(gu[0] * b1[0] + gu[1] * b1[1] + gu[2] * b1[2]);
    float ga2[3], ga1[3];
#pragma unroll
    for (int e = 0; e < 3; ++e) {
        ga2[e] = gu[e] + gd * b1[e];
        gb1[e] += -d * gu[e] + gd * a2[e];
    }
    // b1 = a1 / max(|a1|, eps)
    if (n1r > 1e-12f) {
        const float t = b1[0] * gb1[0] + b1[1] * gb1[1] + b1[2] * gb1[2];
#pragma unroll
        for (int e = 0; e < 3; ++e) ga1[e] = (gb1[e] - b1[e] * t) / n1;
    } else {
#pragma unroll
        for (int e = 0; e < 3; ++e) ga1[e] = gb1[e] / n1;
    }
    float* o = dx6 + row * ldd + j * 6;
    o[0] = ga1[0]; o[1] = ga2[0]; o[2] = ga1[1]; o[3] = ga2[1]; o[4] = ga1[2]; o[5] = ga2[2];
}

inline unsigned capped_grid(long long n) {
    long long g = (n + 255) / 256;
    return (unsigned)(g > 256 * 16 ? 256 * 16 : (g < 1 ? 1 : g));
}

// 128x128 tiles (32 flop per operand byte instead of 16) for the big 1x1 layers (resnet50): +10 % there; the 3x3 / strided layers
// and small pixel counts do better with 64x64 (more workgroups per tap)
inline bool wgrad_big_tile(long long M, int cin, int cout, int taps) {
    return cin % 128 == 0 && cout % 128 == 0 && taps == 1 && M >= 8192;
}

// x3: the per-tap kernel on the planes (its 128x128 tile takes 96 KB of LDS: one workgroup per CU is resident, so 256 workgroups = one
// round; 512 cost 10-18 % on resnet50's 1x1 layers -- tools/sweep_wgrad_x3.py, round 3).  The fp32 kernel (64 KB, two resident) keeps 512;
// the shared workspace size is the fp32 plan's, which is the larger.
inline int wgrad_splits(long long M, int tiles, bool big = false, bool x3 = false) {
    static const int t_big = STRAPS_TOOL_ENV_INT("STRAPS_WGRAD_WGS_BIG", 0);          // (A/B switches for tools)
    static const int t_small = STRAPS_TOOL_ENV_INT("STRAPS_WGRAD_WGS_SMALL", 1536);
    int s = ((big ? (t_big > 0 && x3 ? t_big : x3 ? 256 : 512) : t_small) + tiles - 1) / tiles;
    const long long max_s = (M + 127) / 128;      // at least 4 K-steps of 32 pixels per split
    if (s > max_s) s = (int)max_s;
    return s < 1 ? 1 : s;
}

}  // namespace

// ------------------------------------------------------------------------------------------ C ABI
// (library-internal, not part of the C ABI: conv_wgrad_x3f.hip reduces its split-K partials with the same fixed-order kernel)
int straps_internal_wgrad_reduce(const float* part, float* dw_oihw, int splits, int cout, int cin, int taps, int accumulate, hipStream_t st) {
    const long long n = (long long)cout * taps * cin;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)(n / 256)), dim3(256), 0, st, part, dw_oihw, splits, cout, cin, taps, accumulate);
    STRAPS_CHECK_LAUNCH("wgrad_reduce_kernel");
    return STRAPS_OK;
}

// plan of the halo-patch kernel; returns false when the layer must use the per-tap kernel
static bool wgrad3_plan(int batch, int h, int w, int cin, int cout, int kh, int kw, int stride, int pad, Wgrad3P* p, int* splits, int cp = 32) {
    if (!(kh == 3 && kw == 3 && stride == 1 && pad == 1)) return false;
    if (w < 8 || (w & (w - 1)) != 0) return false;              // chunk columns must tile the row exactly
    const int cw = w < 32 ? w : 32, rpc = cp / cw;              // cp pixels per chunk (32; the bf16x3 kernel also 64)
    if (h % rpc != 0) return false;
    int lg = 0;
    while ((1 << lg) < cw) ++lg;
    p->H = h; p->W = w; p->Cin = cin; p->Cout = cout; p->cw = cw; p->cw_log2 = lg; p->rpc = rpc;
    p->chunks_per_row = w / cw;
    p->chunk_rows_per_img = h / rpc;
    p->nchunks = batch * p->chunk_rows_per_img * p->chunks_per_row;
    p->it = cin / 64;
    const int tiles = (cout / 64) * (cin / 64);
    constexpr int W3_TARGET = 256;          // one 8-wave workgroup per CU (two 4-wave groups that share one partial)
    int s = (W3_TARGET + tiles - 1) / tiles;
    const int max_s = (p->nchunks + 3) / 4;                     // at least 4 chunks (576 MFMAs per wave) per workgroup
    if (s > max_s) s = max_s;
    if (s < 1) s = 1;
    p->chunks_per_split = (p->nchunks + s - 1) / s;
    *splits = (p->nchunks + p->chunks_per_split - 1) / p->chunks_per_split;
    return true;
}

extern "C" size_t straps_conv_wgrad_workspace_bytes(int batch, int h, int w, int cin, int cout, int kh, int kw, int stride, int pad) {
    Wgrad3P p3;
    int splits3;
    if (wgrad3_plan(batch, h, w, cin, cout, kh, kw, stride, pad, &p3, &splits3)) return (size_t)splits3 * cout * 9 * cin * sizeof(float);
    const int ho = (h + 2 * pad - kh) / stride + 1, wo = (w + 2 * pad - kw) / stride + 1;
    const long long M = (long long)batch * ho * wo;
    const bool big = wgrad_big_tile(M, cin, cout, kh * kw);
    const int tile = big ? 128 : 64;
    const int tiles = kh * kw * (cout / tile) * (cin / tile);
    return (size_t)wgrad_splits(M, tiles, big) * cout * kh * kw * cin * sizeof(float);
}

extern "C" int straps_conv_wgrad(const float* x, const float* dy, float* dw_oihw, void* workspace, int batch, int h, int w, int cin,
                                 int cout, int kh, int kw, int stride, int pad, int accumulate, void* stream) {
    STRAPS_REQUIRE(x && dy && dw_oihw && workspace, "straps_conv_wgrad: null pointer");
    STRAPS_REQUIRE(cin % 64 == 0 && cout % 64 == 0, "straps_conv_wgrad: need cin%%64==0 and cout%%64==0 (cin=%d cout=%d)", cin, cout);
    {
        Wgrad3P p3;
        int splits3;
        if (wgrad3_plan(batch, h, w, cin, cout, kh, kw, stride, pad, &p3, &splits3)) {
            p3.x = x; p3.dy = dy; p3.part = (float*)workspace;
            constexpr int NG3 = 2;
            const size_t lds = (size_t)NG3 * 2 * W3PX * 64 * sizeof(float);
            STRAPS_RAISE_LDS((conv_wgrad3x3_kernel<NG3>), lds, "conv_wgrad3x3_kernel");
            hipStream_t st3 = (hipStream_t)stream;
            hipLaunchKernelGGL(conv_wgrad3x3_kernel<NG3>, dim3((cout / 64) * (cin / 64), splits3), dim3(256 * NG3), lds, st3, p3);
            STRAPS_CHECK_LAUNCH("conv_wgrad3x3_kernel");
            const long long n3 = (long long)cout * 9 * cin;
            hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)(n3 / 256)), dim3(256), 0, st3, p3.part, dw_oihw, splits3, cout, cin, 9, accumulate);
            STRAPS_CHECK_LAUNCH("wgrad_reduce_kernel");
            return STRAPS_OK;
        }
    }
    WgradP p;
    p.x = x; p.dy = dy; p.part = (float*)workspace;
    p.B = batch; p.H = h; p.W = w; p.Cin = cin; p.Cout = cout; p.R = kh; p.S = kw; p.stride = stride; p.pad = pad;
    p.Ho = (h + 2 * pad - kh) / stride + 1;
    p.Wo = (w + 2 * pad - kw) / stride + 1;
    const long long M = (long long)batch * p.Ho * p.Wo;
    STRAPS_REQUIRE(M < (1LL << 31), "straps_conv_wgrad: problem too large");
    p.M = (int)M;
    const bool big = wgrad_big_tile(M, cin, cout, kh * kw);
    const int tile = big ? 128 : 64;
    p.ct = cout / tile; p.it = cin / tile;
    const int tiles = kh * kw * p.ct * p.it;
    const int splits = wgrad_splits(M, tiles, big);
    p.rows_per_split = (int)(((M + splits - 1) / splits + 31) / 32 * 32);
    hipStream_t st = (hipStream_t)stream;
    if (big) {
        STRAPS_RAISE_LDS((conv_wgrad_kernel<128, 128>), 64 * 1024, "conv_wgrad_kernel");
        hipLaunchKernelGGL((conv_wgrad_kernel<128, 128>), dim3(tiles, splits), dim3(256), 64 * 1024, st, p);
    } else {
        hipLaunchKernelGGL((conv_wgrad_kernel<64, 64>), dim3(tiles, splits), dim3(256), 32 * 1024, st, p);
    }
    STRAPS_CHECK_LAUNCH("conv_wgrad_kernel");
    const long long n = (long long)cout * kh * kw * cin;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)(n / 256)), dim3(256), 0, st, p.part, dw_oihw, splits, cout, cin, kh * kw, accumulate);
    STRAPS_CHECK_LAUNCH("wgrad_reduce_kernel");
    return STRAPS_OK;
}

// row blocks of tiles (grid.x): enough for 4-5 resident workgroups per CU across the channel groups, few enough that the
// per-block partials stay small
static int stem_wgrad_blocks(int ntiles) {
    static const int cap = STRAPS_TOOL_ENV_INT("STRAPS_STEM_WGRAD_BLOCKS", 256);      // (A/B switch for tools)
    return ntiles < cap ? ntiles : cap;
}

// weight gradient on the bf16x3 route: the 3x3 / stride 1 layers that fit the halo-patch plan run on the planes (x3, dy3: [3][plane
// stride] bf16, see straps_split3_bf16); every other shape falls through to the fp32 kernels of straps_conv_wgrad on (x, dy).
// which kernel straps_conv_wgrad_x3 runs when planes are given: 1 = halo-patch kernel on the planes, 2 = per-tap kernel on the planes,
// 0 = the fp32 kernels on (x, dy)
static int wgrad_x3_route(int batch, int h, int w, int cin, int cout, int kh, int kw, int stride, int pad) {
    Wgrad3P p3;
    int splits3;
    if (wgrad3_plan(batch, h, w, cin, cout, kh, kw, stride, pad, &p3, &splits3)) return 1;
    static const bool tap_x3 = !STRAPS_TOOL_ENV_INT("STRAPS_WGRAD_TAP_FP32", 0);      // (A/B switch for tools)
    // per-tap kernel on the planes where it beats the fp32 one (tools/sweep_wgrad_x3.py: 3x3 / stride 2: 87-93 vs 115-125 us; 1x1 with at
    // least 128 channels on both sides: +0..29 %; with a 64-channel side the fp32 kernel's 4-byte rows win: 52 vs 60 us).  Both stream
    // their operands from memory once per tile of the other channel dimension: bytes, not the matrix pipe, set their rate.
    return tap_x3 && (kh * kw > 1 || (cin >= 128 && cout >= 128)) ? 2 : 0;
}

// channel block (BCO x BCI) of the per-tap kernel on the planes.  Every operand is re-read once per block of the OTHER channel dimension
// (bytes per pixel = 6 (Cout Cin / BCI + Cin Cout / BCO)), every block costs split-K partials, and a block's LDS decides how many workgroups a
// CU holds: measured cold (operands from HBM, as inside a step) over the resnet50 / resnet18 layers by tools/sweep_wgrad_tiles.py
// (profiles/r04_wgrad_tile_sweep.txt), the rectangular blocks beat the square ones of round 3 by 8-20 % where the rule below picks them:
//   1x1, >= 32 768 pixels, Cout % 256 == 0           256 x 128   (l2 shortcut 256>512: 140 -> 112 us, l3.0 512>256: 114 -> 96)
//   1x1, <= 8 192 pixels                             64 x 128 when Cout > Cin at stride 1, else 128 x 64   (l4 2048>512: 52 -> 43, l4 shortcut: 94 -> 77)
//   3x3 / stride 2                                   256 x 128 where the channels allow, else 128 x 64   (l4: 113 -> 104, r18 l4.0: 115 -> 106)
// The tools build takes STRAPS_WGRAD_TILE = 1000 * BCO + BCI per call for the sweep.
static void wgrad_x3_block(long long M, int cin, int cout, int taps, bool big, int stride, int* bco, int* bci) {
    const int v = STRAPS_TOOL_ENV_INT("STRAPS_WGRAD_TILE", 0);      // (tools build: read at every call; the product build compiles this to 0)
    if (v) {
        const int o = v / 1000, i = v % 1000;
        const bool known = (o == 256 && i == 64) || (o == 64 && i == 256) || (o == 256 && i == 128) || (o == 128 && i == 256) || (o == 128 && i == 64) ||
                           (o == 64 && i == 128) || (o == 128 && i == 128) || (o == 64 && i == 64);
        if (known && cout % o == 0 && cin % i == 0 && !(o == 128 && i == 128 && !big)) { *bco = o; *bci = i; }
        return;
    }
    if (taps == 1) {
        if (M >= 32768 && cout % 256 == 0 && cin % 128 == 0) { *bco = 256; *bci = 128; }
        else if (M <= 8192 && cout % 128 == 0 && cin % 128 == 0) {
            if (stride == 1 && cout > cin) { *bco = 64; *bci = 128; }
            else { *bco = 128; *bci = 64; }
        }
    } else if (cout % 256 == 0 && cin % 128 == 0) { *bco = 256; *bci = 128; }
    else if (cout % 128 == 0 && cin % 64 == 0) { *bco = 128; *bci = 64; }
}

extern "C" int straps_conv_wgrad_x3_on_planes(int batch, int h, int w, int cin, int cout, int kh, int kw, int stride, int pad) {
    return wgrad_x3_route(batch, h, w, cin, cout, kh, kw, stride, pad) != 0;
}

extern "C" int straps_conv_wgrad_x3(const float* x, const float* dy, const unsigned short* x3, long long x_plane_stride, const unsigned short* dy3,
                                    long long dy_plane_stride, float* dw_oihw, void* workspace, int batch, int h, int w, int cin, int cout, int kh,
                                    int kw, int stride, int pad, int accumulate, void* stream) {
    STRAPS_REQUIRE(dw_oihw && workspace, "straps_conv_wgrad_x3: null pointer");
    const int route = (x3 && dy3) ? wgrad_x3_route(batch, h, w, cin, cout, kh, kw, stride, pad) : 0;
    STRAPS_REQUIRE(route != 0 || (x && dy), "straps_conv_wgrad_x3: this shape runs on the fp32 tensors, which are NULL (see straps_conv_wgrad_x3_on_planes)");
    STRAPS_REQUIRE(cin % 64 == 0 && cout % 64 == 0, "straps_conv_wgrad_x3: need cin%%64==0 and cout%%64==0 (cin=%d cout=%d)", cin, cout);
    Wgrad3P p3;
    int splits3;
    if (route == 1 && wgrad3_plan(batch, h, w, cin, cout, kh, kw, stride, pad, &p3, &splits3)) {
        STRAPS_REQUIRE(x_plane_stride % 8 == 0 && dy_plane_stride % 8 == 0, "straps_conv_wgrad_x3: plane strides must be multiples of 8 elements");
        Wgrad3XP q;
        q.x3 = x3; q.dy3 = dy3; q.xps = x_plane_stride; q.dps = dy_plane_stride; q.part = (float*)workspace;
        q.H = p3.H; q.W = p3.W; q.Cin = p3.Cin; q.Cout = p3.Cout; q.cw = p3.cw; q.cw_log2 = p3.cw_log2; q.rpc = p3.rpc;
        q.chunks_per_row = p3.chunks_per_row; q.chunk_rows_per_img = p3.chunk_rows_per_img; q.nchunks = p3.nchunks;
        q.chunks_per_split = p3.chunks_per_split; q.it = p3.it;
        q.rows = (long long)batch * h * w;
        constexpr int W3X_NG = 2;
        hipStream_t st3 = (hipStream_t)stream;
        // 64-pixel chunks where the rows allow it (fewer splits than the 32-pixel plan at most: the shared workspace size covers both)
        Wgrad3P p64;
        int splits64;
        const bool c64 = wgrad3_plan(batch, h, w, cin, cout, kh, kw, stride, pad, &p64, &splits64, 64) && splits64 <= splits3;
        if (c64) {
            q.rpc = p64.rpc; q.chunks_per_row = p64.chunks_per_row; q.chunk_rows_per_img = p64.chunk_rows_per_img; q.nchunks = p64.nchunks;
            q.chunks_per_split = p64.chunks_per_split;
            splits3 = splits64;
            const size_t lds = (size_t)2 * 3 * (64 + 136) * 64 * sizeof(u16);
            const dim3 grid3((cout / 64) * (cin / 64), splits3);
#ifdef STRAPS_TOOLS
            static const int abl = STRAPS_TOOL_ENV_INT("STRAPS_WGRAD3_ABL", 0);      // (tools/wgrad3_ablate.py; ablations compute wrong results)
            if (abl) {
                auto kern = abl == 1 ? conv_wgrad3x3_x3_kernel<W3X_NG, 64, 1> : abl == 2 ? conv_wgrad3x3_x3_kernel<W3X_NG, 64, 2>
                          : abl == 3 ? conv_wgrad3x3_x3_kernel<W3X_NG, 64, 3> : abl == 4 ? conv_wgrad3x3_x3_kernel<W3X_NG, 64, 4> : conv_wgrad3x3_x3_kernel<W3X_NG, 64, 5>;
                hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                hipLaunchKernelGGL(kern, grid3, dim3(256 * W3X_NG), lds, st3, q);
                STRAPS_CHECK_LAUNCH("conv_wgrad3x3_x3_kernel (ablation)");
                return STRAPS_OK;
            }
#endif
            STRAPS_RAISE_LDS((conv_wgrad3x3_x3_kernel<W3X_NG, 64>), lds, "conv_wgrad3x3_x3_kernel");
            hipLaunchKernelGGL((conv_wgrad3x3_x3_kernel<W3X_NG, 64>), grid3, dim3(256 * W3X_NG), lds, st3, q);
        } else {
            const size_t lds = (size_t)2 * 3 * (32 + 128) * 64 * sizeof(u16);
            STRAPS_RAISE_LDS((conv_wgrad3x3_x3_kernel<W3X_NG, 32>), lds, "conv_wgrad3x3_x3_kernel");
            hipLaunchKernelGGL((conv_wgrad3x3_x3_kernel<W3X_NG, 32>), dim3((cout / 64) * (cin / 64), splits3), dim3(256 * W3X_NG), lds, st3, q);
        }
        STRAPS_CHECK_LAUNCH("conv_wgrad3x3_x3_kernel");
        const long long n3 = (long long)cout * 9 * cin;
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)(n3 / 256)), dim3(256), 0, st3, q.part, dw_oihw, splits3, cout, cin, 9, accumulate);
        STRAPS_CHECK_LAUNCH("wgrad_reduce_kernel");
        return STRAPS_OK;
    }
    if (route == 2) {
        // same tiles, splits and partial layout as the fp32 kernel (the workspace size is shared)
        STRAPS_REQUIRE(x_plane_stride % 8 == 0 && dy_plane_stride % 8 == 0, "straps_conv_wgrad_x3: plane strides must be multiples of 8 elements");
        WgradXP q;
        q.x3 = x3; q.dy3 = dy3; q.xps = x_plane_stride; q.dps = dy_plane_stride; q.part = (float*)workspace;
        q.B = batch; q.H = h; q.W = w; q.Cin = cin; q.Cout = cout; q.R = kh; q.S = kw; q.stride = stride; q.pad = pad;
        q.Ho = (h + 2 * pad - kh) / stride + 1;
        q.Wo = (w + 2 * pad - kw) / stride + 1;
        const long long M = (long long)batch * q.Ho * q.Wo;
        STRAPS_REQUIRE(M < (1LL << 31), "straps_conv_wgrad_x3: problem too large");
        q.M = (int)M;
        const bool big = wgrad_big_tile(M, cin, cout, kh * kw);
        // channel block of a workgroup: square by default (128 x 128 where both sides allow it, else 64 x 64); RECTANGULAR blocks (round 4) where they
        // cut the operand stream -- every operand is re-read once per block of the OTHER channel dimension, bytes per pixel = 6 (Cout Cin / BCI +
        // Cin Cout / BCO) -- see wgrad_x3_block
        int bco = big ? 128 : 64, bci = bco;
        wgrad_x3_block(M, cin, cout, kh * kw, big, stride, &bco, &bci);
        q.ct = cout / bco; q.it = cin / bci;
        const int tiles = kh * kw * q.ct * q.it;
        const size_t lds = (size_t)2 * 3 * 32 * (bco + bci) * sizeof(u16);      // two stages of [3 planes][32 pixels][BCO + BCI]
        int splits;
        if (bco == bci) splits = wgrad_splits(M, tiles, bco == 128, true);
        else {
            const int per_cu = (int)((160 * 1024) / lds) < 1 ? 1 : (int)((160 * 1024) / lds);
            splits = (256 * per_cu + tiles - 1) / tiles;
            const long long max_s = (M + 127) / 128;
            if (splits > max_s) splits = (int)max_s;
            const int cap = wgrad_splits(M, kh * kw * (cout / (big ? 128 : 64)) * (cin / (big ? 128 : 64)), big);      // the shared workspace is sized by the fp32 plan
            if (splits > cap) splits = cap;
            if (splits < 1) splits = 1;
        }
        q.rows_per_split = (int)(((M + splits - 1) / splits + 31) / 32 * 32);
        hipStream_t st = (hipStream_t)stream;
        static const int nst_env = STRAPS_TOOL_ENV_INT("STRAPS_WGRAD_TAP_NST", 2);      // (A/B switch for tools: 3, 4 -- measured slower)
        constexpr int SB = 3 * 32 * 256 * 2, SS = 3 * 32 * 128 * 2;      // bytes per stage: 128x128 / 64x64 channel block
#define STRAPS_WGRAD_X3_LAUNCH(BCO_, BCI_)                                                                              \
        do {                                                                                                            \
            STRAPS_RAISE_LDS((conv_wgrad_x3_kernel<BCO_, BCI_, 2>), lds, "conv_wgrad_x3_kernel");                       \
            hipLaunchKernelGGL((conv_wgrad_x3_kernel<BCO_, BCI_, 2>), dim3(tiles, splits), dim3(256), lds, st, q);      \
        } while (0)
        if (bco == 256 && bci == 64) STRAPS_WGRAD_X3_LAUNCH(256, 64);
        else if (bco == 64 && bci == 256) STRAPS_WGRAD_X3_LAUNCH(64, 256);
        else if (bco == 256 && bci == 128) STRAPS_WGRAD_X3_LAUNCH(256, 128);
        else if (bco == 128 && bci == 256) STRAPS_WGRAD_X3_LAUNCH(128, 256);
        else if (bco == 128 && bci == 64) STRAPS_WGRAD_X3_LAUNCH(128, 64);
        else if (bco == 64 && bci == 128) STRAPS_WGRAD_X3_LAUNCH(64, 128);
        else if (bco == 128) {
            if (nst_env == 3) {
                STRAPS_RAISE_LDS((conv_wgrad_x3_kernel<128, 128, 3>), 3 * SB, "conv_wgrad_x3_kernel");
                hipLaunchKernelGGL((conv_wgrad_x3_kernel<128, 128, 3>), dim3(tiles, splits), dim3(256), 3 * SB, st, q);
            } else {
                STRAPS_RAISE_LDS((conv_wgrad_x3_kernel<128, 128, 2>), 2 * SB, "conv_wgrad_x3_kernel");
                hipLaunchKernelGGL((conv_wgrad_x3_kernel<128, 128, 2>), dim3(tiles, splits), dim3(256), 2 * SB, st, q);
            }
        } else if (nst_env == 3) {
            STRAPS_RAISE_LDS((conv_wgrad_x3_kernel<64, 64, 3>), 3 * SS, "conv_wgrad_x3_kernel");
            hipLaunchKernelGGL((conv_wgrad_x3_kernel<64, 64, 3>), dim3(tiles, splits), dim3(256), 3 * SS, st, q);
        } else if (nst_env == 4) {
            STRAPS_RAISE_LDS((conv_wgrad_x3_kernel<64, 64, 4>), 4 * SS, "conv_wgrad_x3_kernel");
            hipLaunchKernelGGL((conv_wgrad_x3_kernel<64, 64, 4>), dim3(tiles, splits), dim3(256), 4 * SS, st, q);
        } else {
            hipLaunchKernelGGL((conv_wgrad_x3_kernel<64, 64, 2>), dim3(tiles, splits), dim3(256), 2 * SS, st, q);
        }
#undef STRAPS_WGRAD_X3_LAUNCH
        STRAPS_CHECK_LAUNCH("conv_wgrad_x3_kernel");
        const long long n = (long long)cout * kh * kw * cin;
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)(n / 256)), dim3(256), 0, st, q.part, dw_oihw, splits, cout, cin, kh * kw, accumulate);
        STRAPS_CHECK_LAUNCH("wgrad_reduce_kernel");
        return STRAPS_OK;
    }
    return straps_conv_wgrad(x, dy, dw_oihw, workspace, batch, h, w, cin, cout, kh, kw, stride, pad, accumulate, stream);
}

extern "C" size_t straps_stem_wgrad_workspace_bytes(int batch, int cin, int h, int w) {
    const int Ho = (h - 1) / 2 + 1, Wo = (w - 1) / 2 + 1;
    const int ntiles = batch * ((Ho + TY - 1) / TY) * ((Wo + TX - 1) / TX);
    const int Kp = (cin * 49 + 31) / 32 * 32;
    // partials + room for an internally computed non-zero map (used when the caller passes none)
    const size_t groups = (cin + CG - 1) / CG;
    return (size_t)stem_wgrad_blocks(ntiles) * 64 * Kp * sizeof(float) + straps_stem_nzmask_words(batch, cin, h, w) * sizeof(uint32_t) +
           ((groups * ntiles + 3) & ~(size_t)3);
}

extern "C" int straps_stem_wgrad(const float* x_nchw, const float* dy_nhwc, float* dw_oihw, void* workspace, const uint32_t* nzmask,
                                 int batch, int cin, int h, int w, int accumulate, void* stream) {
    STRAPS_REQUIRE(x_nchw && dy_nhwc && dw_oihw && workspace, "straps_stem_wgrad: null pointer");
    STRAPS_REQUIRE(batch > 0 && cin > 0 && h >= 7 && w >= 7, "straps_stem_wgrad: bad shape B=%d C=%d H=%d W=%d", batch, cin, h, w);
    const int K = cin * 49, Kp = (K + 31) / 32 * 32;
    const int Ho = (h - 1) / 2 + 1, Wo = (w - 1) / 2 + 1;
    const int tiles_x = (Wo + TX - 1) / TX, tiles_y = (Ho + TY - 1) / TY;
    const int ntiles = batch * tiles_x * tiles_y;
    const int nblk = stem_wgrad_blocks(ntiles);
    const int groups = (cin + CG - 1) / CG;
    STRAPS_REQUIRE(groups <= 65535, "straps_stem_wgrad: too many input channels (%d)", cin);
    hipStream_t st = (hipStream_t)stream;
    float* part = (float*)workspace;
    uint32_t* own = (uint32_t*)(part + (size_t)nblk * 64 * Kp);
    uint8_t* tact = (uint8_t*)(own + straps_stem_nzmask_words(batch, cin, h, w));
    if (!nzmask) {
        const int rc = straps_stem_nzmask(x_nchw, own, batch, cin, h, w, stream);
        if (rc != STRAPS_OK) return rc;
        nzmask = own;
    }
    // (every (block, group) writes its whole column slice at the end, skipped tiles or not; padded columns are never read)
    hipLaunchKernelGGL(stem_tileact_kernel, dim3((ntiles * groups + 255) / 256), dim3(256), 0, st, nzmask, tact, cin, h, w, tiles_x, tiles_y, ntiles,
                       groups);
    STRAPS_CHECK_LAUNCH("stem_tileact_kernel");
    static const int rot = STRAPS_TOOL_ENV_INT("STRAPS_STEM_WGRAD_ROT", 97);      // (A/B switch for tools: 0 = plain strided walk)
    hipLaunchKernelGGL(stem_wgrad_kernel, dim3(nblk, groups), dim3(256), 0, st, x_nchw, dy_nhwc, part, tact, batch, cin, h, w, Ho, Wo,
                       tiles_x, tiles_y, Kp, ntiles, rot);
    STRAPS_CHECK_LAUNCH("stem_wgrad_kernel");
    hipLaunchKernelGGL(stem_wgrad_reduce_kernel, dim3((64 * K + 63) / 64), dim3(256), 0, st, (const float*)workspace, dw_oihw, nblk, K, Kp, accumulate);
    STRAPS_CHECK_LAUNCH("stem_wgrad_reduce_kernel");
    return STRAPS_OK;
}

// `accumulate` of the BatchNorm-backward entry points is a flag word: bit 0 = add to dgamma / dbeta instead of overwriting them, bit 1 =
// FROZEN statistics (eval-mode BatchNorm: mean / invstd are the running statistics, constants of the graph -- the two mean terms of
// the training-mode formula vanish: draw = gamma * invstd * dz).  Implemented as an infinite element count: m1 = S1 / N = 0, m2 = S2 / N = 0,
// dgamma = S2 and dbeta = S1 unchanged.
static double bn_bwd_count(long long rows, int flags) { return (flags & 2) ? (double)INFINITY : (double)rows; }

// row blocks of the reduction pass: ~2048 blocks in total over (row blocks x 64-channel column blocks), >= 64 rows each
extern "C" int straps_bn_bwd_blocks(long long rows, int c) {
    const int colblocks = (c + 63) / 64;
    long long b = 2048 / colblocks;
    const long long cap = (rows + 63) / 64;
    if (b > cap) b = cap;
    return (int)(b < 1 ? 1 : b);
}

static int bn_bwd_x3_impl(const float* dy, const float* yact, const unsigned* relu_bits, const float* raw, const float* save_mean, const float* save_invstd,
                          const float* gamma, const float* mask_scale, const float* mask_shift, float* dgamma, float* dbeta, float* draw,
                          float* dz_out, unsigned short* draw_planes, long long plane_stride, void* workspace, long long rows, int c,
                          int accumulate, void* stream) {
    STRAPS_REQUIRE(!relu_bits || (!yact && (c & 31) == 0), "straps_bn_bwd_bits_x3: the bits replace yact (and need c %% 32 == 0; c=%d)", c);
    STRAPS_REQUIRE(!draw_planes || (plane_stride >= rows * c && plane_stride % 8 == 0), "straps_bn_bwd_x3: plane_stride must be >= rows*c and a multiple of 8");
    STRAPS_REQUIRE(!draw_planes || (c & 31) == 0, "straps_bn_bwd_x3: chunk-major planes need c %% 32 == 0 (c=%d)", c);
    STRAPS_REQUIRE(dy && raw && save_mean && save_invstd && gamma && dgamma && dbeta && (draw || draw_planes) && workspace, "straps_bn_bwd: null pointer");
    STRAPS_REQUIRE(rows > 0 && c > 0 && (c & 3) == 0, "straps_bn_bwd: bad shape rows=%lld c=%d", rows, c);
    const int C4 = c >> 2;
    STRAPS_REQUIRE(C4 <= 256 ? (256 % C4 == 0) : (C4 % 256 == 0), "straps_bn_bwd: channel count %d not supported", c);
    hipStream_t st = (hipStream_t)stream;
    const int nblk = straps_bn_bwd_blocks(rows, c);
    const int rpb = (int)((rows + nblk - 1) / nblk);
    double* part = (double*)workspace;               // [nblk][c][2]
    double* coefd = part + (size_t)nblk * c * 2;     // [2][c]  m1, m2
    float* k1 = (float*)(coefd + 2 * (size_t)c);     // [c]
    hipLaunchKernelGGL(bn_bwd_reduce_kernel<false>, dim3(nblk, (c + 63) / 64), dim3(256), 0, st, dy, yact, raw, save_mean, save_invstd, mask_scale, mask_shift, part, rows, c, rpb, PoolSrc{}, relu_bits);
    STRAPS_CHECK_LAUNCH("bn_bwd_reduce_kernel");
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((c + 3) / 4), dim3(256), 0, st, part, nblk, c, bn_bwd_count(rows, accumulate), gamma, save_invstd, dgamma, dbeta, coefd, k1, accumulate & 1);
    STRAPS_CHECK_LAUNCH("bn_bwd_finalize_kernel");
    const long long n4 = rows * C4;
    const int tiled = straps_bn_tiled(rows, c >> 2);
    hipLaunchKernelGGL(bn_bwd_apply_kernel<false>, dim3(tiled ? straps_bn_tiled_grid(rows, c >> 2, tiled) : straps_grid256_rows(n4, c >> 2)), dim3(256), 0, st, dy, yact, raw, save_mean, save_invstd, coefd, k1, mask_scale, mask_shift, draw, dz_out, draw_planes, plane_stride, n4, c, PoolSrc{}, relu_bits, tiled);
    STRAPS_CHECK_LAUNCH("bn_bwd_apply_kernel");
    return STRAPS_OK;
}

// straps_bn_bwd_x3 when the two sums already exist as per-tile partials [nblk][c][2] (S1, invstd * S2) -- written by
// straps_conv_dgrad_x3_bn, the data-gradient launch that produced dy: finalize + apply only, no pass over (dy, raw) for the sums.
// workspace: (2 c) doubles + c floats.
extern "C" int straps_bn_bwd_x3(const float* dy, const float* yact, const float* raw, const float* save_mean, const float* save_invstd,
                                const float* gamma, const float* mask_scale, const float* mask_shift, float* dgamma, float* dbeta, float* draw,
                                float* dz_out, unsigned short* draw_planes, long long plane_stride, void* workspace, long long rows, int c,
                                int accumulate, void* stream) {
    return bn_bwd_x3_impl(dy, yact, nullptr, raw, save_mean, save_invstd, gamma, mask_scale, mask_shift, dgamma, dbeta, draw, dz_out, draw_planes,
                          plane_stride, workspace, rows, c, accumulate, stream);
}

// straps_bn_bwd_x3 with the ReLU mask as bits: relu_bits [rows][c / 32] (bit ch & 31 of word [row][ch / 32] = activation > 0, written by
// straps_bn_apply_bits_x3) takes the place of the fp32 activation `yact` -- 4 bytes per element less in each of the two passes.  There is no
// dz_out: the consumers of the masked gradient apply the same bits to dy themselves (straps_conv_dgrad_x3_bits / _bn_bits: addend_bits).
extern "C" int straps_bn_bwd_bits_x3(const float* dy, const unsigned* relu_bits, const float* raw, const float* save_mean, const float* save_invstd,
                                     const float* gamma, float* dgamma, float* dbeta, float* draw, unsigned short* draw_planes, long long plane_stride,
                                     void* workspace, long long rows, int c, int accumulate, void* stream) {
    STRAPS_REQUIRE(relu_bits, "straps_bn_bwd_bits_x3: null bit mask");
    return bn_bwd_x3_impl(dy, nullptr, relu_bits, raw, save_mean, save_invstd, gamma, nullptr, nullptr, dgamma, dbeta, draw, nullptr, draw_planes,
                          plane_stride, workspace, rows, c, accumulate, stream);
}

static int bn_bwd_finish_x3_impl(const float* dy, const float* yact, const unsigned* relu_bits, const float* raw, const float* save_mean,
                                 const float* save_invstd, const float* gamma, const float* mask_scale, const float* mask_shift, float* dgamma,
                                 float* dbeta, float* draw, float* dz_out, unsigned short* draw_planes, long long plane_stride, const double* partials,
                                 int nblk, void* workspace, long long rows, int c, int accumulate, void* stream) {
    STRAPS_REQUIRE(!relu_bits || (!yact && (c & 31) == 0), "straps_bn_bwd_finish_bits_x3: the bits replace yact (and need c %% 32 == 0; c=%d)", c);
    STRAPS_REQUIRE(!draw_planes || (plane_stride >= rows * c && plane_stride % 8 == 0), "straps_bn_bwd_finish_x3: plane_stride must be >= rows*c and a multiple of 8");
    STRAPS_REQUIRE(!draw_planes || (c & 31) == 0, "straps_bn_bwd_finish_x3: chunk-major planes need c %% 32 == 0 (c=%d)", c);
    STRAPS_REQUIRE(dy && raw && save_mean && save_invstd && gamma && dgamma && dbeta && (draw || draw_planes) && workspace && partials && nblk > 0,
                   "straps_bn_bwd_finish_x3: null pointer");
    STRAPS_REQUIRE(rows > 0 && c > 0 && (c & 3) == 0, "straps_bn_bwd_finish_x3: bad shape rows=%lld c=%d", rows, c);
    const int C4 = c >> 2;
    STRAPS_REQUIRE(C4 <= 256 ? (256 % C4 == 0) : (C4 % 256 == 0), "straps_bn_bwd_finish_x3: channel count %d not supported", c);
    hipStream_t st = (hipStream_t)stream;
    double* coefd = (double*)workspace;              // [2][c]  m1, m2
    float* k1 = (float*)(coefd + 2 * (size_t)c);     // [c]
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((c + 3) / 4), dim3(256), 0, st, partials, nblk, c, bn_bwd_count(rows, accumulate), gamma, save_invstd, dgamma, dbeta, coefd, k1, accumulate & 1);
    STRAPS_CHECK_LAUNCH("bn_bwd_finalize_kernel");
    const long long n4 = rows * C4;
    const int tiled = straps_bn_tiled(rows, c >> 2);
    hipLaunchKernelGGL(bn_bwd_apply_kernel<false>, dim3(tiled ? straps_bn_tiled_grid(rows, c >> 2, tiled) : straps_grid256_rows(n4, c >> 2)), dim3(256), 0, st, dy, yact, raw, save_mean, save_invstd, coefd, k1, mask_scale, mask_shift, draw, dz_out, draw_planes, plane_stride, n4, c, PoolSrc{}, relu_bits, tiled);
    STRAPS_CHECK_LAUNCH("bn_bwd_apply_kernel");
    return STRAPS_OK;
}

extern "C" int straps_bn_bwd_finish_x3(const float* dy, const float* yact, const float* raw, const float* save_mean, const float* save_invstd,
                                       const float* gamma, const float* mask_scale, const float* mask_shift, float* dgamma, float* dbeta,
                                       float* draw, float* dz_out, unsigned short* draw_planes, long long plane_stride, const double* partials,
                                       int nblk, void* workspace, long long rows, int c, int accumulate, void* stream) {
    return bn_bwd_finish_x3_impl(dy, yact, nullptr, raw, save_mean, save_invstd, gamma, mask_scale, mask_shift, dgamma, dbeta, draw, dz_out, draw_planes,
                                 plane_stride, partials, nblk, workspace, rows, c, accumulate, stream);
}

// straps_bn_bwd_finish_x3 with the ReLU mask as bits (see straps_bn_bwd_bits_x3; the partials come from straps_conv_dgrad_x3_bn_bits, which read
// the same bits)
extern "C" int straps_bn_bwd_finish_bits_x3(const float* dy, const unsigned* relu_bits, const float* raw, const float* save_mean,
                                            const float* save_invstd, const float* gamma, float* dgamma, float* dbeta, float* draw,
                                            unsigned short* draw_planes, long long plane_stride, const double* partials, int nblk, void* workspace,
                                            long long rows, int c, int accumulate, void* stream) {
    STRAPS_REQUIRE(relu_bits, "straps_bn_bwd_finish_bits_x3: null bit mask");
    return bn_bwd_finish_x3_impl(dy, nullptr, relu_bits, raw, save_mean, save_invstd, gamma, nullptr, nullptr, dgamma, dbeta, draw, nullptr, draw_planes,
                                 plane_stride, partials, nblk, workspace, rows, c, accumulate, stream);
}

extern "C" int straps_bn_bwd(const float* dy, const float* yact, const float* raw, const float* save_mean, const float* save_invstd,
                             const float* gamma, const float* mask_scale, const float* mask_shift, float* dgamma, float* dbeta, float* draw,
                             float* dz_out, void* workspace, long long rows, int c, int accumulate, void* stream) {
    return straps_bn_bwd_x3(dy, yact, raw, save_mean, save_invstd, gamma, mask_scale, mask_shift, dgamma, dbeta, draw, dz_out, nullptr, 0, workspace,
                            rows, c, accumulate, stream);
}

extern "C" size_t straps_stem_tiles(int batch, int h, int w) {
    const int Ho = (h - 1) / 2 + 1, Wo = (w - 1) / 2 + 1;
    return (size_t)batch * ((Wo + TX - 1) / TX) * ((Ho + TY - 1) / TY);
}

extern "C" int straps_stem_tile_activity(const uint32_t* nzmask, uint8_t* tile_active, int batch, int cin, int h, int w, void* stream) {
    STRAPS_REQUIRE(nzmask && tile_active, "straps_stem_tile_activity: null pointer");
    STRAPS_REQUIRE(batch > 0 && cin > 0 && h >= 7 && w >= 7, "straps_stem_tile_activity: bad shape B=%d C=%d H=%d W=%d", batch, cin, h, w);
    const int Ho = (h - 1) / 2 + 1, Wo = (w - 1) / 2 + 1;
    const int tiles_x = (Wo + TX - 1) / TX, tiles_y = (Ho + TY - 1) / TY;
    const int ntiles = batch * tiles_x * tiles_y;
    hipLaunchKernelGGL(stem_tileany_kernel, dim3((unsigned)(((long long)ntiles * 32 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, nzmask, tile_active, cin, h, w, tiles_x, tiles_y,
                       ntiles);
    STRAPS_CHECK_LAUNCH("stem_tileany_kernel");
    return STRAPS_OK;
}

extern "C" int straps_bn_bwd_pooled(const float* dy_pool, const uint8_t* idx, const float* raw, const float* save_mean, const float* save_invstd,
                                    const float* gamma, const float* mask_scale, const float* mask_shift, float* dgamma, float* dbeta,
                                    float* draw, void* workspace, int batch, int h, int w, int c, int accumulate, void* stream) {
    return straps_bn_bwd_pooled_sparse(dy_pool, idx, raw, save_mean, save_invstd, gamma, mask_scale, mask_shift, dgamma, dbeta, draw, workspace, batch, h, w,
                                       c, accumulate, nullptr, stream);
}

extern "C" int straps_bn_bwd_pooled_sparse(const float* dy_pool, const uint8_t* idx, const float* raw, const float* save_mean, const float* save_invstd,
                                           const float* gamma, const float* mask_scale, const float* mask_shift, float* dgamma, float* dbeta,
                                           float* draw, void* workspace, int batch, int h, int w, int c, int accumulate,
                                           const uint8_t* tile_active, void* stream) {
    STRAPS_REQUIRE(dy_pool && idx && raw && save_mean && save_invstd && gamma && mask_scale && mask_shift && dgamma && dbeta && draw && workspace,
                   "straps_bn_bwd_pooled: null pointer");
    STRAPS_REQUIRE(batch > 0 && h > 0 && w > 0 && c > 0 && (c & 3) == 0, "straps_bn_bwd_pooled: bad shape %dx%dx%dx%d", batch, h, w, c);
    const int C4 = c >> 2;
    STRAPS_REQUIRE(C4 <= 256 ? (256 % C4 == 0) : (C4 % 256 == 0), "straps_bn_bwd_pooled: channel count %d not supported", c);
    hipStream_t st = (hipStream_t)stream;
    const long long rows = (long long)batch * h * w;
    const int nblk = straps_bn_bwd_blocks(rows, c);
    const int rpb = (int)((rows + nblk - 1) / nblk);
    double* part = (double*)workspace;               // [nblk][c][2]  (straps_bn_bwd_workspace_bytes(rows, c))
    double* coefd = part + (size_t)nblk * c * 2;     // [2][c]  m1, m2
    float* k1 = (float*)(coefd + 2 * (size_t)c);     // [c]
    const PoolSrc ps{dy_pool, idx, h, w, (h - 1) / 2 + 1, (w - 1) / 2 + 1, tile_active, (w + TX - 1) / TX, (h + TY - 1) / TY};
    const long long prows = (long long)batch * ps.Ho * ps.Wo;
    const int prpb = (int)((prows + nblk - 1) / nblk);
    (void)rpb;
    hipLaunchKernelGGL(bn_bwd_reduce_pooled_kernel, dim3(nblk, (c + 63) / 64), dim3(256), 0, st, raw, save_mean, save_invstd, mask_scale, mask_shift, part, prows, c, prpb, ps);
    STRAPS_CHECK_LAUNCH("bn_bwd_reduce_pooled_kernel");
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((c + 3) / 4), dim3(256), 0, st, part, nblk, c, bn_bwd_count(rows, accumulate), gamma, save_invstd, dgamma, dbeta, coefd, k1, accumulate & 1);
    STRAPS_CHECK_LAUNCH("bn_bwd_finalize_kernel");
    const long long n4 = rows * C4;
    hipLaunchKernelGGL(bn_bwd_apply_kernel<true>, dim3((unsigned)((n4 + POOL_CHUNK - 1) / POOL_CHUNK)), dim3(256), 0, st, nullptr, nullptr, raw, save_mean, save_invstd, coefd, k1, mask_scale, mask_shift, draw, nullptr, (u16*)nullptr, 0LL, n4, c, ps, (const unsigned*)nullptr, 0);
    STRAPS_CHECK_LAUNCH("bn_bwd_apply_kernel<pool>");
    return STRAPS_OK;
}

extern "C" int straps_bn_relu_maxpool_fwd_x3(const float* raw, const float* scale, const float* shift, float* y_pool, uint8_t* idx,
                                             unsigned short* y_planes, long long plane_stride, int batch, int h, int w, int c, void* stream) {
    STRAPS_REQUIRE(raw && scale && shift && y_pool && idx && batch > 0 && h > 0 && w > 0 && c > 0 && (c & 3) == 0, "straps_bn_relu_maxpool_fwd: bad arguments");
    const int Ho = (h - 1) / 2 + 1, Wo = (w - 1) / 2 + 1;
    const long long n = (long long)batch * Ho * Wo * (c >> 2);
    STRAPS_REQUIRE(!y_planes || (plane_stride >= n * 4 && plane_stride % 8 == 0), "straps_bn_relu_maxpool_fwd_x3: plane_stride must be >= the output size and a multiple of 8");
    STRAPS_REQUIRE(!y_planes || (c & 31) == 0, "straps_bn_relu_maxpool_fwd_x3: chunk-major planes need c %% 32 == 0 (c=%d)", c);
    hipLaunchKernelGGL(bn_relu_maxpool_kernel, dim3(capped_grid(n)), dim3(256), 0, (hipStream_t)stream, raw, scale, shift, y_pool, idx, y_planes, plane_stride, batch, h, w, c, Ho, Wo);
    STRAPS_CHECK_LAUNCH("bn_relu_maxpool_kernel");
    return STRAPS_OK;
}

extern "C" int straps_bn_relu_maxpool_fwd(const float* raw, const float* scale, const float* shift, float* y_pool, uint8_t* idx, int batch, int h,
                                          int w, int c, void* stream) {
    return straps_bn_relu_maxpool_fwd_x3(raw, scale, shift, y_pool, idx, nullptr, 0, batch, h, w, c, stream);
}

extern "C" size_t straps_bn_bwd_workspace_bytes(long long rows, int c) {
    return ((size_t)straps_bn_bwd_blocks(rows, c) * c * 2 + 2 * (size_t)c) * sizeof(double) + (size_t)c * sizeof(float);
}

extern "C" int straps_maxpool_fwd_idx(const float* x, float* y, uint8_t* idx, int batch, int h, int w, int c, void* stream) {
    STRAPS_REQUIRE(x && y && idx && batch > 0 && (c & 3) == 0, "straps_maxpool_fwd_idx: bad arguments");
    const int Ho = (h - 1) / 2 + 1, Wo = (w - 1) / 2 + 1;
    const long long n = (long long)batch * Ho * Wo * (c >> 2);
    hipLaunchKernelGGL(maxpool_idx_kernel, dim3(capped_grid(n)), dim3(256), 0, (hipStream_t)stream, x, y, idx, batch, h, w, c, Ho, Wo);
    STRAPS_CHECK_LAUNCH("maxpool_idx_kernel");
    return STRAPS_OK;
}

extern "C" int straps_maxpool_bwd(const float* dy, const uint8_t* idx, float* dx, int batch, int h, int w, int c, void* stream) {
    STRAPS_REQUIRE(dy && idx && dx && batch > 0 && (c & 3) == 0, "straps_maxpool_bwd: bad arguments");
    const int Ho = (h - 1) / 2 + 1, Wo = (w - 1) / 2 + 1;
    const long long n = (long long)batch * h * w * (c >> 2);
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(capped_grid(n)), dim3(256), 0, (hipStream_t)stream, dy, idx, dx, batch, h, w, c, Ho, Wo);
    STRAPS_CHECK_LAUNCH("maxpool_bwd_kernel");
    return STRAPS_OK;
}

extern "C" int straps_gap_bwd(const float* dfeat, float* dx, int batch, int hw, int c, void* stream) {
    STRAPS_REQUIRE(dfeat && dx && batch > 0 && hw > 0 && c > 0, "straps_gap_bwd: bad arguments");
    const long long n = (long long)batch * hw * c;
    hipLaunchKernelGGL(gap_bwd_kernel, dim3(capped_grid(n)), dim3(256), 0, (hipStream_t)stream, dfeat, dx, batch, hw, c);
    STRAPS_CHECK_LAUNCH("gap_bwd_kernel");
    return STRAPS_OK;
}

extern "C" int straps_masked_copy(const float* x, int ldx, const float* mask, int ldmask, float* y, int ldy, int m, int n, int accumulate,
                                  void* stream) {
    STRAPS_REQUIRE(x && y && m > 0 && n > 0, "straps_masked_copy: bad arguments");
    const long long t = (long long)m * n;
    hipLaunchKernelGGL(masked_copy_kernel, dim3((unsigned)((t + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, ldx, mask, ldmask, y, ldy, m, n, accumulate);
    STRAPS_CHECK_LAUNCH("masked_copy_kernel");
    return STRAPS_OK;
}

extern "C" int straps_rot6d_bwd(const float* x6, long long ld, int per_row, const float* drot, float* dx6, long long ldd, long long rows,
                                void* stream) {
    STRAPS_REQUIRE(x6 && drot && dx6 && rows > 0 && per_row > 0, "straps_rot6d_bwd: bad arguments");
    const long long n = rows * per_row;
    hipLaunchKernelGGL(rot6d_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x6, ld, per_row, drot, dx6, ldd, n);
    STRAPS_CHECK_LAUNCH("rot6d_bwd_kernel");
    return STRAPS_OK;
}
