// elementwise.hip -- HBM-bound helpers of the encoder: weight repacking, BatchNorm fold / statistics /
// apply, max-pool, global average pool.  All NHWC fp32, float4 per lane, grid-stride where large.
#include "common.h"

namespace {

// OIHW -> KRSC: dst[((o*R + r)*S + s)*C + c] = src[((o*C + c)*R + r)*S + s]
__global__ __launch_bounds__(256) void pack_krsc_kernel(const float* __restrict__ src, float* __restrict__ dst, int O, int C, int R, int S) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long n = (long long)O * C * R * S;
    if (i >= n) return;
    const int c = (int)(i % C);
    long long t = i / C;
    const int s = (int)(t % S); t /= S;
    const int r = (int)(t % R);
    const int o = (int)(t / R);
    dst[i] = src[(((long long)o * C + c) * R + r) * S + s];
}

// OIHW -> [C][R][S][O] with flipped taps: dst[((c*R + r)*S + s)*O + o] = src[o][c][R-1-r][S-1-s]
__global__ __launch_bounds__(256) void pack_crsk_flip_kernel(const float* __restrict__ src, float* __restrict__ dst, int O, int C, int R, int S) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long n = (long long)O * C * R * S;
    if (i >= n) return;
    const int o = (int)(i % O);
    long long t = i / O;
    const int s = (int)(t % S); t /= S;
    const int r = (int)(t % R);
    const int c = (int)(t / R);
    dst[i] = src[(((long long)o * C + c) * R + (R - 1 - r)) * S + (S - 1 - s)];
}

// both packings of every layer in one launch.  Work unit = (layer, 32 output channels, 32 input channels, <= 9 taps): the source block
// (32 runs of 32*RS contiguous floats) is read coalesced into an LDS tile, then written once as KRSC (128-byte runs over ci) and once
// as flipped CRSK (128-byte runs over co) -- both sides of the transposes coalesced.  Blocks stride over the units; the layer of a
// unit is found by walking the (few dozen) descriptors, wave-uniformly.
// k3 / c3 (bf16x3 route): the three bf16 planes [3][ps] of the two packed layouts, element `first + offset inside the layer`, written
// in the same pass (the fp32 destinations of a descriptor may then be NULL: nothing on that route reads them).
constexpr int PK_T = 32, PK_RS = 9, PK_LD = PK_T * PK_RS + 1;
// chunk-major weight planes (bf16x3 route): the GEMM's B rows are `row` (output channels for the forward layout, input channels for the
// data-gradient one), its K index is (tap, k); the 32-wide K chunks are outermost, element (row, tap, k) at
//   ((tap * (K / 32) + (k >> 5)) * rows + row) * 32 + (k & 31)
// -- the 64 bytes of one (tap, chunk) of one row sit next to the neighbouring rows' (whole 128-byte lines per LDS-DMA fetch, common.h).
__device__ __forceinline__ long long wk_index(int row, int tap, int k, int rows, int K) {
    return (((long long)tap * (K >> 5) + (k >> 5)) * rows + row) * 32 + (k & 31);
}
__global__ __launch_bounds__(256) void pack_batched_kernel(const straps_pack_desc_t* __restrict__ descs, int n, u16* __restrict__ k3_all,
                                                           u16* __restrict__ c3_all, long long ps) {
    __shared__ float tile[PK_T * PK_LD];
    const int tid = threadIdx.x;
    int d = 0;
    long long base = 0;                                    // first unit of descriptor d
    for (long long u = blockIdx.x;; u += gridDim.x) {
        long long units = 0;
        int ot = 0, ct = 0, rt = 0;
        for (; d < n; ++d) {
            ot = (descs[d].o + PK_T - 1) / PK_T;
            ct = (descs[d].c + PK_T - 1) / PK_T;
            rt = (descs[d].r * descs[d].s + PK_RS - 1) / PK_RS;
            units = (long long)ot * ct * rt;
            if (u < base + units) break;
            base += units;
        }
        if (d >= n) return;
        const straps_pack_desc_t D = descs[d];
        const int RS = D.r * D.s;
        // planes exist only where the layout's K extent is a whole number of 32-wide chunks (the bf16x3 kernels need that anyway)
        u16* const k3 = (D.c & 31) ? nullptr : k3_all;
        u16* const c3 = (D.o & 31) ? nullptr : c3_all;
        int v = (int)(u - base);
        const int ri = v % rt; v /= rt;
        const int ci = v % ct;
        const int oi = v / ct;
        const int o0 = oi * PK_T, c0 = ci * PK_T, rs0 = ri * PK_RS;
        const int no = min(PK_T, D.o - o0), nc = min(PK_T, D.c - c0), nr = min(PK_RS, RS - rs0);
        // load: for each o a run of nc*RS floats (contiguous when the layer has <= 9 taps)
        const int run = nc * nr;
        if (nr == PK_RS && RS == PK_RS && nc == PK_T && (D.c & 3) == 0 && (reinterpret_cast<uintptr_t>(D.src) & 15) == 0) {
            // a full 3x3 unit: the run of an output channel IS its tile row (288 consecutive floats, 16-byte aligned): float4 loads
            constexpr int RUN4 = PK_T * PK_RS / 4;
            for (int idx = tid; idx < no * RUN4; idx += 256) {
                const int o = idx / RUN4, q = idx - o * RUN4;
                const f32x4 v = *reinterpret_cast<const f32x4*>(D.src + ((long long)(o0 + o) * D.c + c0) * PK_RS + q * 4);
                float* t = tile + o * PK_LD + q * 4;
                t[0] = v[0]; t[1] = v[1]; t[2] = v[2]; t[3] = v[3];
            }
        } else {
            for (int idx = tid; idx < no * run; idx += 256) {
                const int o = idx / run, k = idx - o * run;
                const int c = k / nr, j = k - c * nr;
                tile[o * PK_LD + c * PK_RS + j] = D.src[((long long)(o0 + o) * D.c + c0 + c) * RS + rs0 + j];
            }
        }
        __syncthreads();
        // planes of a layout go out four consecutive elements per lane (8-byte stores) when the layer allows it
        const bool v4k = k3 && ((D.c | D.first) & 3) == 0, v4c = c3 && ((D.o | D.first) & 3) == 0;
        // KRSC: dst[o][rs][c]
        if (D.dst_krsc || (k3 && !v4k)) {
            for (int idx = tid; idx < no * nr * PK_T; idx += 256) {
                const int c = idx & (PK_T - 1);
                const int k = idx >> 5;
                const int j = k % nr, o = k / nr;
                if (c < nc) {
                    const long long at = ((long long)(o0 + o) * RS + rs0 + j) * D.c + c0 + c;
                    const float v = tile[o * PK_LD + c * PK_RS + j];
                    if (D.dst_krsc) D.dst_krsc[at] = v;
                    if (k3 && !v4k) {
                        u16 b1, b2, b3;
                        split3(v, b1, b2, b3);
                        const long long w = D.first + wk_index(o0 + o, rs0 + j, c0 + c, D.o, D.c);
                        k3[w] = b1; k3[ps + w] = b2; k3[2 * ps + w] = b3;
                    }
                }
            }
        }
        if (v4k) {
            for (int idx = tid; idx < no * nr * (PK_T / 4); idx += 256) {
                const int c = (idx & (PK_T / 4 - 1)) * 4;
                const int k = idx >> 3;
                const int o = nr == PK_RS ? k / PK_RS : k / nr, j = k - o * nr;        // (constant divisor for the 3x3 layers)
                if (c < nc) {
                    const float* t = tile + o * PK_LD + c * PK_RS + j;
                    const f32x4 v = {t[0], t[PK_RS], t[2 * PK_RS], t[3 * PK_RS]};
                    store_planes4(k3, ps, D.first + wk_index(o0 + o, rs0 + j, c0 + c, D.o, D.c), v);
                }
            }
        }
        // flipped CRSK: dst[c][R-1-r][S-1-s][o] -- flipping (r, s) jointly is reversing the tap index rs
        if (D.dst_crsk || (c3 && !v4c)) {
            for (int idx = tid; idx < nc * nr * PK_T; idx += 256) {
                const int o = idx & (PK_T - 1);
                const int k = idx >> 5;
                const int j = k % nr, c = k / nr;
                if (o < no) {
                    const long long at = ((long long)(c0 + c) * RS + (RS - 1 - (rs0 + j))) * D.o + o0 + o;
                    const float v = tile[o * PK_LD + c * PK_RS + j];
                    if (D.dst_crsk) D.dst_crsk[at] = v;
                    if (c3 && !v4c) {
                        u16 b1, b2, b3;
                        split3(v, b1, b2, b3);
                        const long long w = D.first + wk_index(c0 + c, RS - 1 - (rs0 + j), o0 + o, D.c, D.o);
                        c3[w] = b1; c3[ps + w] = b2; c3[2 * ps + w] = b3;
                    }
                }
            }
        }
        if (v4c) {
            for (int idx = tid; idx < nc * nr * (PK_T / 4); idx += 256) {
                const int o = (idx & (PK_T / 4 - 1)) * 4;
                const int k = idx >> 3;
                const int c = nr == PK_RS ? k / PK_RS : k / nr, j = k - c * nr;
                if (o < no) {
                    const float* t = tile + o * PK_LD + c * PK_RS + j;
                    const f32x4 v = {t[0], t[PK_LD], t[2 * PK_LD], t[3 * PK_LD]};
                    store_planes4(c3, ps, D.first + wk_index(c0 + c, RS - 1 - (rs0 + j), o0 + o, D.c, D.o), v);
                }
            }
        }
        __syncthreads();
    }
}

__global__ void bn_fold_kernel(const float* __restrict__ g, const float* __restrict__ b, const float* __restrict__ mean,
                               const float* __restrict__ var, float eps, float* __restrict__ scale, float* __restrict__ shift, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float sc = g[c] / sqrtf(var[c] + eps);
    scale[c] = sc;
    shift[c] = b[c] - mean[c] * sc;
}

// MaxPool 3x3 / s2 / p1 over NHWC, one float4 of channels per thread
__global__ __launch_bounds__(256) void maxpool_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int H, int W, int C,
                                                      int Ho, int Wo) {
    const int C4 = C >> 2;
    const long long n = (long long)B * Ho * Wo * C4;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int c4 = (int)(i % C4);
        long long t = i / C4;
        const int wo = (int)(t % Wo); t /= Wo;
        const int ho = (int)(t % Ho);
        const int b = (int)(t / Ho);
        f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int hi = 2 * ho - 1 + r;
            if ((unsigned)hi >= (unsigned)H) continue;
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const int wi = 2 * wo - 1 + s;
                if ((unsigned)wi >= (unsigned)W) continue;
                const f32x4 v = *reinterpret_cast<const f32x4*>(x + (((long long)b * H + hi) * W + wi) * C + c4 * 4);
                m[0] = fmaxf(m[0], v[0]); m[1] = fmaxf(m[1], v[1]); m[2] = fmaxf(m[2], v[2]); m[3] = fmaxf(m[3], v[3]);
            }
        }
        *reinterpret_cast<f32x4*>(y + i * 4) = m;
    }
}

// global average pool: one thread per (b, c), coalesced across c
__global__ __launch_bounds__(256) void gap_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int HW, int C) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)B * C) return;
    const int c = (int)(i % C);
    const long long b = i / C;
    const float* p = x + b * HW * C + c;
    float s = 0.f;
    for (int k = 0; k < HW; ++k) s += p[(long long)k * C];
    y[i] = s / (float)HW;
}

// one wave per channel: fixed-order fp64 sum of the per-block (sum, sumsq) partials
__global__ __launch_bounds__(256) void bn_stats_finalize_kernel(const float* __restrict__ part, int nblocks, int C, double count,
                                                               const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                               float momentum, float* __restrict__ rmean, float* __restrict__ rvar,
                                                               float* __restrict__ scale, float* __restrict__ shift,
                                                               float* __restrict__ smean, float* __restrict__ sinv) {
    int c;
    double s1, s2;
    if (bn_partials_sum4(part, nblocks, C, s1, s2, c)) {
        const double mean = s1 / count;
        double var = s2 / count - mean * mean;
        if (var < 0.0) var = 0.0;
        const float inv = (float)(1.0 / sqrt(var + (double)eps));
        const float sc = gamma[c] * inv;
        scale[c] = sc;
        shift[c] = beta[c] - (float)mean * sc;
        if (smean) smean[c] = (float)mean;
        if (sinv) sinv[c] = inv;
        if (rmean) {
            const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
            rmean[c] = (1.f - momentum) * rmean[c] + momentum * (float)mean;
            rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unbiased;
        }
    }
}

__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                       const float* __restrict__ shift, const float* __restrict__ res, int relu,
                                                       float* __restrict__ y, u16* __restrict__ planes, long long ps, long long n4, int C4,
                                                       unsigned* __restrict__ bits, int tiled) {
    const long long first = (long long)blockIdx.x * 256 + threadIdx.x, step = (long long)gridDim.x * 256;
    const long long rows = n4 / C4;
    auto body = [&](long long i, long long r, int c4, const f32x4& sc, const f32x4& sh) {
        f32x4 v = *reinterpret_cast<const f32x4*>(x + i * 4);
        // explicit fmaf: the backward pass re-derives the ReLU mask from raw with the same operation (bn_bwd, mask_scale)
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = fmaf(v[q], sc[q], sh[q]);
        if (res) v += *reinterpret_cast<const f32x4*>(res + i * 4);
        if (relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
        if (y) *reinterpret_cast<f32x4*>(y + i * 4) = v;        // (NULL: only the planes are consumed -- bf16x3 route, see straps_bn_apply_x3)
        if (planes) store_planes4_cm(planes, ps, r, c4 * 4, rows, v);      // bf16x3 route: the next convolution's operand (chunk-major planes), written here instead of by a split pass
        if (bits) {
            // the ReLU decisions of this output as one word per (row, 32 channels), bit c & 31: what the backward pass needs of a residual unit's
            // activation (straps_bn_bwd_bits_x3, straps_conv_dgrad_x3_bn_bits) in 1/32 of its bytes.  Eight neighbouring lanes hold the eight
            // nibbles of a word (C4 % 8 == 0 and n4 % 8 == 0: they are in the same row and leave the loop together)
            unsigned wd = ((v[0] > 0.f ? 1u : 0u) | (v[1] > 0.f ? 2u : 0u) | (v[2] > 0.f ? 4u : 0u) | (v[3] > 0.f ? 8u : 0u)) << ((c4 & 7) * 4);
            wd |= __shfl_xor(wd, 1, 64);
            wd |= __shfl_xor(wd, 2, 64);
            wd |= __shfl_xor(wd, 4, 64);
            if ((c4 & 7) == 0) bits[r * (C4 >> 3) + (c4 >> 3)] = wd;
        }
    };
    if (tiled) {
        // round 4: a wave takes 4 rows x 2 chunks of 32 channels, 16 lanes 2 rows of one chunk -- every wave store covers 2 x 256 contiguous
        // bytes of each chunk-major plane (8 lanes = the 64 bytes of one row's chunk, the next row's 64 follow) instead of one row's 64-byte
        // pieces in up to 8 chunks, and every 8 lanes read one whole 128-byte line of the fp32 row.  The workgroup's 4 waves sit side by
        // side on `tiled` (1, 2 or 4) column groups of 64 channels and stack 4 / tiled deep in rows; gridDim.x is a multiple of the column
        // blocks (straps_bn_tiled_grid, common.h): a thread's channels stay fixed.
        const int l = threadIdx.x & 63, w = threadIdx.x >> 6, wcg = tiled, ncb = (C4 >> 4) / wcg, trows = 16 / wcg;
        const int c4 = ((int)(blockIdx.x % ncb) * wcg + w % wcg) * 16 + (l >> 5) * 8 + (l & 7);
        const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + c4 * 4);
        const f32x4 sh = *reinterpret_cast<const f32x4*>(shift + c4 * 4);
        const long long rstep = (long long)(gridDim.x / ncb) * trows;
        for (long long r = (long long)(blockIdx.x / ncb) * trows + (w / wcg) * 4 + ((l >> 3) & 3); r < rows; r += rstep) body(r * C4 + c4, r, c4, sc, sh);
    } else if (step % C4 == 0) {
        // round 4: the grid stride is a multiple of the row length (the host sizes the grid so), so a thread keeps ITS four channels for the whole
        // loop: scale / shift are loaded once, the row index advances by a constant -- no per-element 64-bit division, two 16-byte constant loads
        // less per 16 bytes of payload (the per-channel loads were 40-60 % of what went through the vector L1; same arithmetic, same results)
        const int c4 = (int)(first % C4);
        const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + c4 * 4);
        const f32x4 sh = *reinterpret_cast<const f32x4*>(shift + c4 * 4);
        const long long rstep = step / C4;
        long long r = first / C4;
        for (long long i = first; i < n4; i += step, r += rstep) body(i, r, c4, sc, sh);
    } else {
        for (long long i = first; i < n4; i += step) {
            const int c4 = (int)(i % C4);
            body(i, i / C4, c4, *reinterpret_cast<const f32x4*>(scale + c4 * 4), *reinterpret_cast<const f32x4*>(shift + c4 * 4));
        }
    }
}

// eval-mode BatchNorm as the four vectors the training-mode kernels take: scale, shift (the fold) and the statistics themselves
__global__ void bn_fold_stats_kernel(const float* __restrict__ g, const float* __restrict__ b, const float* __restrict__ mean,
                                     const float* __restrict__ var, float eps, float* __restrict__ scale, float* __restrict__ shift,
                                     float* __restrict__ smean, float* __restrict__ sinv, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float inv = 1.0f / sqrtf(var[c] + eps);
    const float sc = g[c] / sqrtf(var[c] + eps);          // (the same expression as bn_fold_kernel: identical scale / shift)
    scale[c] = sc;
    shift[c] = b[c] - mean[c] * sc;
    smean[c] = mean[c];
    sinv[c] = inv;
}

inline unsigned capped_grid(long long n) {
    long long g = (n + 255) / 256;
    return (unsigned)(g > 256 * 16 ? 256 * 16 : (g < 1 ? 1 : g));
}

}  // namespace

extern "C" int straps_pack_conv_weight(const float* w_oihw, float* w_krsc, int cout, int cin, int kh, int kw, void* stream) {
    STRAPS_REQUIRE(w_oihw && w_krsc && cout > 0 && cin > 0 && kh > 0 && kw > 0, "straps_pack_conv_weight: bad arguments");
    const long long n = (long long)cout * cin * kh * kw;
    hipLaunchKernelGGL(pack_krsc_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w_oihw, w_krsc, cout, cin, kh, kw);
    STRAPS_CHECK_LAUNCH("pack_krsc_kernel");
    return STRAPS_OK;
}

extern "C" int straps_pack_conv_weight_dgrad(const float* w_oihw, float* w_crsk, int cout, int cin, int kh, int kw, void* stream) {
    STRAPS_REQUIRE(w_oihw && w_crsk && cout > 0 && cin > 0 && kh > 0 && kw > 0, "straps_pack_conv_weight_dgrad: bad arguments");
    const long long n = (long long)cout * cin * kh * kw;
    hipLaunchKernelGGL(pack_crsk_flip_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w_oihw, w_crsk, cout, cin, kh, kw);
    STRAPS_CHECK_LAUNCH("pack_crsk_flip_kernel");
    return STRAPS_OK;
}

extern "C" int straps_pack_conv_weights_batched(const straps_pack_desc_t* descs, int n, long long total, void* stream) {
    STRAPS_REQUIRE(descs && n > 0 && total > 0, "straps_pack_conv_weights_batched: bad arguments");
    long long blocks = (total + PK_T * PK_T * PK_RS - 1) / (PK_T * PK_T * PK_RS);       // ~ one unit per block for 3x3 layers, grid-stride beyond
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(pack_batched_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, descs, n, (u16*)nullptr, (u16*)nullptr, 0LL);
    STRAPS_CHECK_LAUNCH("pack_batched_kernel");
    return STRAPS_OK;
}

extern "C" int straps_pack_conv_weights_batched_x3(const straps_pack_desc_t* descs, int n, long long total, unsigned short* krsc_planes,
                                                   unsigned short* crsk_planes, long long plane_stride, void* stream) {
    STRAPS_REQUIRE(descs && n > 0 && total > 0, "straps_pack_conv_weights_batched_x3: bad arguments");
    STRAPS_REQUIRE(krsc_planes || crsk_planes, "straps_pack_conv_weights_batched_x3: no planes given (use straps_pack_conv_weights_batched)");
    STRAPS_REQUIRE(plane_stride >= total && plane_stride % 8 == 0, "straps_pack_conv_weights_batched_x3: plane_stride must be >= total and a multiple of 8");
    STRAPS_REQUIRE((reinterpret_cast<uintptr_t>(krsc_planes) & 15) == 0 && (reinterpret_cast<uintptr_t>(crsk_planes) & 15) == 0,
                   "straps_pack_conv_weights_batched_x3: plane buffers must be 16-byte aligned");
    long long blocks = (total + PK_T * PK_T * PK_RS - 1) / (PK_T * PK_T * PK_RS);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(pack_batched_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, descs, n, krsc_planes, crsk_planes, plane_stride);
    STRAPS_CHECK_LAUNCH("pack_batched_kernel");
    return STRAPS_OK;
}

extern "C" int straps_bn_fold(const float* gamma, const float* beta, const float* mean, const float* var, float eps, float* scale,
                              float* shift, int c, void* stream) {
    STRAPS_REQUIRE(gamma && beta && mean && var && scale && shift && c > 0, "straps_bn_fold: bad arguments");
    hipLaunchKernelGGL(bn_fold_kernel, dim3((c + 255) / 256), dim3(256), 0, (hipStream_t)stream, gamma, beta, mean, var, eps, scale, shift, c);
    STRAPS_CHECK_LAUNCH("bn_fold_kernel");
    return STRAPS_OK;
}

extern "C" int straps_bn_fold_stats(const float* gamma, const float* beta, const float* mean, const float* var, float eps, float* scale,
                                    float* shift, float* save_mean, float* save_invstd, int c, void* stream) {
    STRAPS_REQUIRE(gamma && beta && mean && var && scale && shift && save_mean && save_invstd && c > 0, "straps_bn_fold_stats: bad arguments");
    hipLaunchKernelGGL(bn_fold_stats_kernel, dim3((c + 255) / 256), dim3(256), 0, (hipStream_t)stream, gamma, beta, mean, var, eps, scale, shift, save_mean,
                       save_invstd, c);
    STRAPS_CHECK_LAUNCH("bn_fold_stats_kernel");
    return STRAPS_OK;
}

extern "C" int straps_maxpool_fwd(const float* x, float* y, int batch, int h, int w, int c, void* stream) {
    STRAPS_REQUIRE(x && y && batch > 0 && h > 0 && w > 0 && c > 0 && (c & 3) == 0, "straps_maxpool_fwd: bad arguments (c%%4 must be 0)");
    const int Ho = (h + 2 - 3) / 2 + 1, Wo = (w + 2 - 3) / 2 + 1;
    const long long n = (long long)batch * Ho * Wo * (c >> 2);
    hipLaunchKernelGGL(maxpool_kernel, dim3(capped_grid(n)), dim3(256), 0, (hipStream_t)stream, x, y, batch, h, w, c, Ho, Wo);
    STRAPS_CHECK_LAUNCH("maxpool_kernel");
    return STRAPS_OK;
}

extern "C" int straps_gap_fwd(const float* x, float* y, int batch, int hw, int c, void* stream) {
    STRAPS_REQUIRE(x && y && batch > 0 && hw > 0 && c > 0, "straps_gap_fwd: bad arguments");
    const long long n = (long long)batch * c;
    hipLaunchKernelGGL(gap_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, y, batch, hw, c);
    STRAPS_CHECK_LAUNCH("gap_kernel");
    return STRAPS_OK;
}

extern "C" int straps_bn_stats_finalize(const float* stats_partial, int nblocks, int c, long long count, const float* gamma,
                                        const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                                        float* scale, float* shift, float* save_mean, float* save_invstd, void* stream) {
    STRAPS_REQUIRE(stats_partial && gamma && beta && scale && shift && nblocks > 0 && c > 0 && count > 0, "straps_bn_stats_finalize: bad arguments");
    STRAPS_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "straps_bn_stats_finalize: running stats must be given together");
    hipLaunchKernelGGL(bn_stats_finalize_kernel, dim3((c + 3) / 4), dim3(256), 0, (hipStream_t)stream, stats_partial, nblocks, c, (double)count, gamma, beta,
                       eps, momentum, running_mean, running_var, scale, shift, save_mean, save_invstd);
    STRAPS_CHECK_LAUNCH("bn_stats_finalize_kernel");
    return STRAPS_OK;
}

extern "C" int straps_bn_apply(const float* x, const float* scale, const float* shift, const float* residual, int relu, float* y,
                               long long rows, int c, void* stream) {
    STRAPS_REQUIRE(x && scale && shift && y && rows > 0 && c > 0 && (c & 3) == 0, "straps_bn_apply: bad arguments (c%%4 must be 0)");
    const long long n4 = rows * (c >> 2);
    hipLaunchKernelGGL(bn_apply_kernel, dim3(capped_grid(n4)), dim3(256), 0, (hipStream_t)stream, x, scale, shift, residual, relu, y, (u16*)nullptr, 0LL, n4, c >> 2, (unsigned*)nullptr, 0);
    STRAPS_CHECK_LAUNCH("bn_apply_kernel");
    return STRAPS_OK;
}

static int bn_apply_x3_impl(const char* who, const float* x, const float* scale, const float* shift, const float* residual, int relu, float* y,
                            unsigned short* y_planes, long long plane_stride, unsigned* relu_bits, long long rows, int c, void* stream) {
    // (the planes are chunk-major, common.h cm_index: 32-channel chunks outermost -- c % 32 != 0 would index past rows*c)
    // (round 6: y_planes may be NULL when every consumer reads the fp32 tensor -- the fp32-operand 1x1 route, conv_x3f.hip; then y is required)
    STRAPS_REQUIRE(x && scale && shift && (y_planes || y) && rows > 0 && c > 0 && (c & 31) == 0, "%s: bad arguments (c%%32 must be 0: chunk-major planes; c=%d)", who, c);
    STRAPS_REQUIRE(!y_planes || (plane_stride >= rows * c && plane_stride % 8 == 0), "%s: plane_stride must be >= rows*c and a multiple of 8", who);
    const long long n4 = rows * (c >> 2);
    const int tiled = straps_bn_tiled(rows, c >> 2);
    hipLaunchKernelGGL(bn_apply_kernel, dim3(tiled ? straps_bn_tiled_grid(rows, c >> 2, tiled) : straps_grid256_rows(n4, c >> 2)), dim3(256), 0, (hipStream_t)stream, x, scale, shift,
                       residual, relu, y, y_planes, plane_stride, n4, c >> 2, relu_bits, tiled);
    STRAPS_CHECK_LAUNCH("bn_apply_kernel");
    return STRAPS_OK;
}

extern "C" int straps_bn_apply_x3(const float* x, const float* scale, const float* shift, const float* residual, int relu, float* y,
                                  unsigned short* y_planes, long long plane_stride, long long rows, int c, void* stream) {
    return bn_apply_x3_impl("straps_bn_apply_x3", x, scale, shift, residual, relu, y, y_planes, plane_stride, nullptr, rows, c, stream);
}

// straps_bn_apply_x3 that also records its ReLU decisions: relu_bits [rows][c / 32] words, bit (ch & 31) of word [row][ch / 32] = (y[row][ch] > 0).
// The backward pass of a residual unit reads these instead of the fp32 activation (straps_bn_bwd_bits_x3, straps_bn_bwd_finish_bits_x3,
// straps_conv_dgrad_x3_bn_bits, straps_conv_dgrad_x3_bits).
extern "C" int straps_bn_apply_bits_x3(const float* x, const float* scale, const float* shift, const float* residual, float* y,
                                       unsigned short* y_planes, long long plane_stride, unsigned* relu_bits, long long rows, int c, void* stream) {
    STRAPS_REQUIRE(relu_bits, "straps_bn_apply_bits_x3: null bit buffer");
    return bn_apply_x3_impl("straps_bn_apply_bits_x3", x, scale, shift, residual, 1, y, y_planes, plane_stride, relu_bits, rows, c, stream);
}
