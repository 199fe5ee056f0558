// stem.hip -- conv 7x7 / stride 2 / pad 3 straight from the NCHW network input (models/resnet.py:145)
//
// The 18-channel 256x256 proxy representation is the only tensor that crosses the boundary in
// NCHW, and the stem is 29 % of resnet18's MACs.  Each workgroup produces a 4-row x 32-column tile
// of output pixels for all 64 output channels:
//   * the (2*4+5) x (2*32+5) input halo patch of every channel is loaded ONCE with row-contiguous
//     (coalesced) reads into LDS (C*13*72 floats = 67 KB at C = 18),
//   * the GEMM K index runs over (c, r, s) = C*49 taps in natural order -- a small LDS table maps
//     k -> patch offset, so no K padding per filter row is wasted on the slow fp32 MFMA,
//   * A fragments (pixels) are gathered from the patch with stride-2 ds_read_b32, B fragments
//     (weights, prepacked in fragment order by straps_pack_stem_weight) stream from L2 as
//     coalesced float4,
//   * ZERO SKIPPING: the proxy representation is a binary silhouette + 17 joint heat-maps that are non-zero only inside a
//     16x16 window around their joint (utils/label_conversions.py:58-87), so most (channel, patch row) strips are all zero.
//     The patch load flags the non-zero strips; each wave (one output row) compacts the 8-tap K groups that
//     touch a flagged strip of ITS 7 input rows into an LDS list and contracts only those.  A skipped group
//     would have added 0*w = 0 to every accumulator, so the result is the dense one bit for bit (finite weights);
//     a dense input simply skips nothing.
//   * epilogue writes NHWC (lane = channel) with BN scale/shift + ReLU fused (eval) or raw output
//     plus per-channel (sum, sumsq) partials (training).
#include "common.h"
#include <type_traits>

namespace {

constexpr int PH = 13, PW = 72;
constexpr int TY = 4, TX = 32;

// Non-zero map of the input: one bit per 4-row x 8-column cell, [B][C][ceil(H/4)][ceil(W/256)] 32-bit words
// (bit = (w >> 3) & 31).  With it the stem kernels do not even LOAD the all-zero cells of the proxy representation.
// One lane per cell (8 float4 loads, row-contiguous across lanes), the 32 bits of a word are gathered with a ballot:
// no atomics, no clear pass, same cost for dense and sparse inputs.
__global__ __launch_bounds__(256) void stem_nzmask_kernel(const float* __restrict__ x, unsigned* __restrict__ mask, long long nwords,
                                                          int H, int W, int HC, int WW) {
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long word = gid >> 5;
    const int bit = (int)(gid & 31);
    bool nz = false;
    if (word < nwords) {
        const int ww = (int)(word % WW);
        const long long t = word / WW;
        const int hc = (int)(t % HC);
        const long long bc = t / HC;
        const int w0 = ww * 256 + bit * 8;
        if (w0 < W) {
            const float* base = x + (bc * H + hc * 4) * (long long)W + w0;
            const int rows = min(4, H - hc * 4);
            if ((W & 3) == 0) {
                const bool two = w0 + 4 < W;
                for (int r = 0; r < rows; ++r) {
                    const f32x4 a = *reinterpret_cast<const f32x4*>(base + (long long)r * W);
                    nz = nz || a[0] != 0.f || a[1] != 0.f || a[2] != 0.f || a[3] != 0.f;
                    if (two) {
                        const f32x4 c = *reinterpret_cast<const f32x4*>(base + (long long)r * W + 4);
                        nz = nz || c[0] != 0.f || c[1] != 0.f || c[2] != 0.f || c[3] != 0.f;
                    }
                }
            } else {
                const int cols = min(8, W - w0);
                for (int r = 0; r < rows; ++r)
                    for (int c = 0; c < cols; ++c) nz = nz || base[(long long)r * W + c] != 0.f;
            }
        }
    }
    const unsigned long long m = __ballot(nz);
    if (bit == 0 && word < nwords) mask[word] = (unsigned)(m >> (32 * ((threadIdx.x & 63) >> 5)));
}

constexpr int NSLOT = 4;   // input channels resident in LDS per pass (slot 0 of the patch is the all-zero channel)

__global__ __launch_bounds__(256, 5) void stem_kernel(const float* __restrict__ x, const float* __restrict__ wfrag,
                                                   const float* __restrict__ scale, const float* __restrict__ shift, int relu,
                                                   float* __restrict__ y, float* __restrict__ stats,
                                                   const unsigned* __restrict__ nzmask, int B, int C, int H, int W,
                                                   int Ho, int Wo, int tiles_x, int tiles_y, int K, int Kp) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* patch = smem;                                                    // [1 + NSLOT][PH][PW]
    int* koff = reinterpret_cast<int*>(smem + (1 + NSLOT) * PH * PW);       // [Kp]   patch offset of tap k in the current pass
    int* krow = koff + Kp;                                                  // [Kp]   strip (c * PH + r) of tap k, -1 for the K padding
    int* rowflag = krow + Kp;                                               // [C*PH] strip (channel, patch row) holds a non-zero
    int* rank = rowflag + C * PH;                                           // [C]    position of channel c among the tile's active ones / -1
    int* chan = rank + C;                                                   // [C]    the active channels, ascending
    int* glist = chan + C;                                                  // [4 waves][Kp/8] active K groups of each wave
    __shared__ int n_active_s;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int bid = blockIdx.x;
    const int b = bid / (tiles_x * tiles_y);
    bid -= b * tiles_x * tiles_y;
    const int ty = bid / tiles_x, tx = bid - ty * tiles_x;
    const int y0 = ty * TY, x0 = tx * TX;
    const int hi0 = 2 * y0 - 3, wi0 = 2 * x0 - 3;

    // ---- strip flags: does (channel c, patch row) hold a non-zero?  One word of the caller's non-zero map per strip, BEFORE
    // anything is loaded; without a map every strip counts as non-zero (the plain dense convolution) ----
    const int HC = (H + 3) >> 2, WW = (W + 255) >> 8;
    int tile_any = 1;
    if (nzmask) {
        int mine = 0;
        const int wa = max(wi0 - 1, 0), wb = min(wi0 - 2 + PW, W - 1);
        for (int rc = tid; rc < C * PH; rc += 256) {
            const int row = rc % PH, c = rc / PH;
            const int hi = hi0 + row;
            unsigned any = 0;
            if ((unsigned)hi < (unsigned)H) {
                for (int word = wa >> 8; word <= (wb >> 8); ++word) {
                    const int lo = max(wa, word << 8), hi_c = min(wb, (word << 8) + 255);
                    const int b0 = (lo >> 3) & 31, b1 = (hi_c >> 3) & 31;
                    const unsigned bits = (b1 == 31 ? 0xffffffffu : ((1u << (b1 + 1)) - 1u)) & ~((1u << b0) - 1u);
                    any |= nzmask[(((long long)b * C + c) * HC + (hi >> 2)) * WW + word] & bits;
                }
            }
            rowflag[rc] = any != 0u;
            mine |= any != 0u;
        }
        tile_any = __syncthreads_or(mine);
    } else {
        for (int idx = tid; idx < C * PH; idx += 256) rowflag[idx] = 1;
        __syncthreads();
    }

    const int i = lane & 31, h = lane >> 5;
    const float* pbase = patch + (2 * wave) * PW + 2 * i;
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    if (tile_any) {      // (block-uniform) an all-zero tile goes straight to the epilogue with acc = 0
        // ---- the tile's active channels (ascending), the pass-independent tap -> strip table, the zero slot ----
        for (int k = tid; k < Kp; k += 256) {
            int v = -1;
            if (k < K) {
                const int c = k / 49, rs = k - c * 49;
                v = c * PH + rs / 7;
            }
            krow[k] = v;
        }
        for (int idx = tid; idx < PH * PW; idx += 256) patch[idx] = 0.f;
        if (wave == 0) {
            int n = 0;
            for (int cb = 0; cb < C; cb += 64) {
                const int c = cb + lane;
                bool act = false;
                if (c < C)
                    for (int row = 0; row < PH; ++row) act = act || rowflag[c * PH + row] != 0;
                const unsigned long long m = __ballot(act);
                const int pos = n + (int)__popcll(m & ((1ULL << lane) - 1ULL));
                if (c < C) rank[c] = act ? pos : -1;
                if (act) chan[pos] = c;
                n += (int)__popcll(m);
            }
            if (lane == 0) n_active_s = n;
        }
        __syncthreads();
        const int n_active = n_active_s;
        const f32x4* __restrict__ wp = reinterpret_cast<const f32x4*>(wfrag) + lane;
        const int G = Kp >> 3;
        int* gl = glist + wave * G;
        // ---- passes over the active channels: only NSLOT halo patches are resident in LDS (29 KB per workgroup instead of 67:
        // five workgroups per CU hide each other's load / table / list latencies); taps of every other channel point into the
        // zero slot.  Consecutive passes OVERLAP by one channel, so that a K group straddling two active channels always finds
        // both resident in exactly one pass; every group is contracted once, in ascending order over the whole tile, with all
        // of its non-zero taps -- the accumulators see exactly the sequence of a single dense sweep (bit-identical) ----
        for (int c0 = 0;; c0 += NSLOT - 1) {
            const int nch = min(NSLOT, n_active - c0);
            // patch column p holds input column wi0 - 1 + p: the patch origin is shifted one column left so that every row is
            // 18 ALIGNED float4 loads (wi0 - 1 = 2*x0 - 4 is a multiple of 4) instead of 69 scalar ones
            if ((W & 3) == 0) {
                const int nvec = nch * PH * (PW / 4);                 // <= 936: all loads of a thread in flight before the first LDS store
                f32x4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int idx = u * 256 + tid;
                    v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (idx < nvec) {
                        const int q = idx % (PW / 4);
                        const int rc = idx / (PW / 4);
                        const int row = rc % PH, c = chan[c0 + rc / PH];
                        const int hi = hi0 + row, wi = wi0 - 1 + 4 * q;
                        if (rowflag[c * PH + row] && (unsigned)hi < (unsigned)H && wi >= 0 && wi < W)
                            v[u] = *reinterpret_cast<const f32x4*>(x + (((long long)b * C + c) * H + hi) * W + wi);
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int idx = u * 256 + tid;
                    if (idx < nvec) *reinterpret_cast<f32x4*>(patch + PH * PW + 4 * idx) = v[u];     // (PW = 18 float4: slot/row/q linear in idx)
                }
            } else {
                const int npatch = nch * PH * PW;
                for (int idx = tid; idx < npatch; idx += 256) {
                    const int col = idx % PW;
                    const int rc = idx / PW;
                    const int row = rc % PH, c = chan[c0 + rc / PH];
                    const int hi = hi0 + row, wi = wi0 - 1 + col;
                    float v = 0.f;
                    if (rowflag[c * PH + row] && (unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W) v = x[(((long long)b * C + c) * H + hi) * W + wi];
                    patch[PH * PW + idx] = v;
                }
            }
            for (int k = tid; k < Kp; k += 256) {
                int o = 0;
                if (k < K) {
                    const int c = k / 49, rs = k - c * 49;
                    const int r = rs / 7, s = rs - r * 7;
                    const int rk = rank[c] - c0;
                    const int slot = (rk >= 0 && rk < nch) ? rk + 1 : 0;
                    o = (slot * PH + r) * PW + s + 1;      // +1: the patch origin sits one column left of the receptive field
                }
                koff[k] = o;
            }
            __syncthreads();
            // ---- this wave's K groups of this pass.  A group is needed iff one of its 8 taps reads a flagged strip of rows
            // 2*wave + r; it belongs to THIS pass iff the channels of those taps (one, or two neighbours) are all resident --
            // except that a group of the first resident channel alone was already done as the previous pass's last channel ----
            int ng = 0;
            for (int gb = 0; gb < G; gb += 64) {
                const int g = gb + lane;
                int lo = 1 << 30, hi = -1;
                if (g < G) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int rc = krow[8 * g + e];
                        if (rc >= 0 && rowflag[rc + 2 * wave] != 0) {
                            const int rk = rank[rc / PH];
                            lo = min(lo, rk);
                            hi = max(hi, rk);
                        }
                    }
                }
                const bool act = hi >= 0 && lo >= c0 && hi < c0 + nch && !(c0 > 0 && hi == c0);
                const unsigned long long m = __ballot(act);
                if (act) gl[ng + __popcll(m & ((1ULL << lane) - 1ULL))] = g;
                ng += (int)__popcll(m);
            }
            // (the list is written and read by this wave only; LDS operations of one wave retire in order)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // software pipeline over the active groups: the offset table is read two groups ahead and the (dependent) patch
            // gather one group ahead, so no MFMA of a group waits on an LDS round trip issued in that group
            if (ng > 0) {
                int gcur = gl[0];
                int gnext = gl[ng > 1 ? 1 : 0];
                f32x4 b0 = wp[gcur * 128], b1 = wp[gcur * 128 + 64];
                int4 ko_n = *reinterpret_cast<const int4*>(koff + 8 * gcur + 4 * h);
                float a0 = pbase[ko_n.x], a1 = pbase[ko_n.y], a2 = pbase[ko_n.z], a3 = pbase[ko_n.w];
                ko_n = *reinterpret_cast<const int4*>(koff + 8 * gnext + 4 * h);
                for (int j = 0; j < ng; ++j) {
                    f32x4 nb0 = b0, nb1 = b1;
                    float n0 = a0, n1 = a1, n2 = a2, n3 = a3;
                    if (j + 1 < ng) {
                        nb0 = wp[gnext * 128]; nb1 = wp[gnext * 128 + 64];
                        n0 = pbase[ko_n.x]; n1 = pbase[ko_n.y]; n2 = pbase[ko_n.z]; n3 = pbase[ko_n.w];
                    }
                    if (j + 2 < ng) {
                        gnext = gl[j + 2];
                        ko_n = *reinterpret_cast<const int4*>(koff + 8 * gnext + 4 * h);
                    }
                    acc0 = mfma32(a0, b0[0], acc0); acc1 = mfma32(a0, b1[0], acc1);
                    acc0 = mfma32(a1, b0[1], acc0); acc1 = mfma32(a1, b1[1], acc1);
                    acc0 = mfma32(a2, b0[2], acc0); acc1 = mfma32(a2, b1[2], acc1);
                    acc0 = mfma32(a3, b0[3], acc0); acc1 = mfma32(a3, b1[3], acc1);
                    b0 = nb0; b1 = nb1; a0 = n0; a1 = n1; a2 = n2; a3 = n3;
                }
            }
            if (c0 + nch >= n_active) break;
            __syncthreads();      // the next pass overwrites the patch and the tap table
        }
    }

    const int yo = y0 + wave;
    float s1[2], s2[2];
    // (a tile inside the image skips the per-element range test; the row base is one 64-bit address, the rest 32-bit offsets)
    const bool full = x0 + TX <= Wo && y0 + TY <= Ho;
    float* yrow = y + (((long long)b * Ho + min(yo, Ho - 1)) * Wo + x0) * 64 + i;
    auto rows = [&](auto full_c) {
        constexpr bool FULL = decltype(full_c)::value;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int n = nt * 32 + i;
            const float sc = scale ? scale[n] : 1.f;
            const float sh = shift ? shift[n] : 0.f;
            float t1 = 0.f, t2 = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int xr = mfma_row(r, lane);
                if (FULL || (x0 + xr < Wo && yo < Ho)) {
                    float v = nt == 0 ? acc0[r] : acc1[r];
                    t1 += v;
                    t2 = fmaf(v, v, t2);
                    if (scale) v = fmaf(v, sc, sh);
                    if (relu) v = fmaxf(v, 0.f);
                    yrow[xr * 64 + nt * 32] = v;
                }
            }
            s1[nt] = t1;
            s2[nt] = t2;
        }
    };
    if (full) rows(std::true_type{}); else rows(std::false_type{});
    if (stats) {
        __syncthreads();
        float* red = smem;   // [4 waves][64][2]
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const float u1 = s1[nt] + __shfl_xor(s1[nt], 32, 64);
            const float u2 = s2[nt] + __shfl_xor(s2[nt], 32, 64);
            if (lane < 32) {
                red[(wave * 64 + nt * 32 + lane) * 2 + 0] = u1;
                red[(wave * 64 + nt * 32 + lane) * 2 + 1] = u2;
            }
        }
        __syncthreads();
        if (tid < 128) {
            const int c = tid >> 1, q = tid & 1;
            stats[((long long)blockIdx.x * 64 + c) * 2 + q] =
                (red[(0 * 64 + c) * 2 + q] + red[(1 * 64 + c) * 2 + q]) + (red[(2 * 64 + c) * 2 + q] + red[(3 * 64 + c) * 2 + q]);
        }
    }
}

__global__ __launch_bounds__(256) void pack_stem_kernel(const float* __restrict__ w, float* __restrict__ wf, int C, int K, int Kp) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;   // over [G][2][64][4]
    if (idx >= (long long)(Kp >> 3) * 512) return;
    const int e = (int)(idx & 3), lane = (int)((idx >> 2) & 63), nt = (int)((idx >> 8) & 1), g = (int)(idx >> 9);
    const int k = 8 * g + 4 * (lane >> 5) + e;
    const int n = nt * 32 + (lane & 31);
    wf[idx] = (k < K) ? w[(long long)n * K + k] : 0.f;   // OIHW row n is already (c,r,s)-ordered
}

}  // namespace

extern "C" size_t straps_stem_weight_floats(int cin) { return (size_t)((cin * 49 + 7) / 8) * 512; }

extern "C" int straps_pack_stem_weight(const float* w_oihw, float* w_frag, int cin, void* stream) {
    STRAPS_REQUIRE(w_oihw && w_frag && cin > 0, "straps_pack_stem_weight: bad arguments");
    const int K = cin * 49, Kp = (K + 7) / 8 * 8;
    const long long n = (long long)(Kp >> 3) * 512;
    hipLaunchKernelGGL(pack_stem_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w_oihw, w_frag, cin, K, Kp);
    STRAPS_CHECK_LAUNCH("pack_stem_kernel");
    return STRAPS_OK;
}

extern "C" int straps_stem_stat_blocks(int batch, int h, int w) {
    const int Ho = (h + 6 - 7) / 2 + 1, Wo = (w + 6 - 7) / 2 + 1;
    return batch * ((Ho + TY - 1) / TY) * ((Wo + TX - 1) / TX);
}

extern "C" size_t straps_stem_nzmask_words(int batch, int cin, int h, int w) {
    if (batch <= 0 || cin <= 0 || h <= 0 || w <= 0) return 0;
    return (size_t)batch * cin * ((h + 3) / 4) * ((w + 255) / 256);
}

extern "C" int straps_stem_nzmask(const float* x, uint32_t* mask, int batch, int cin, int h, int w, void* stream) {
    STRAPS_REQUIRE(x && mask && batch > 0 && cin > 0 && h > 0 && w > 0, "straps_stem_nzmask: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const long long nwords = (long long)straps_stem_nzmask_words(batch, cin, h, w);
    STRAPS_REQUIRE((nwords * 32 + 255) / 256 < (1LL << 31), "straps_stem_nzmask: input too large for one launch");
    hipLaunchKernelGGL(stem_nzmask_kernel, dim3((unsigned)((nwords * 32 + 255) / 256)), dim3(256), 0, st, x, mask, nwords, h, w, (h + 3) / 4, (w + 255) / 256);
    STRAPS_CHECK_LAUNCH("stem_nzmask_kernel");
    return STRAPS_OK;
}

extern "C" int straps_stem_fwd(const float* x, const float* w_frag, const float* scale, const float* shift, int relu, float* y,
                               float* stats_partial, const uint32_t* nzmask, int batch, int cin, int h, int w, void* stream) {
    STRAPS_REQUIRE(x && w_frag && y, "straps_stem_fwd: null pointer");
    STRAPS_REQUIRE(batch > 0 && cin > 0 && h >= 7 && w >= 7, "straps_stem_fwd: bad shape B=%d C=%d H=%d W=%d", batch, cin, h, w);
    STRAPS_REQUIRE((scale == nullptr) == (shift == nullptr), "straps_stem_fwd: scale and shift must be given together");
    const int K = cin * 49, Kp = (K + 7) / 8 * 8;
    const size_t lds = (size_t)(1 + NSLOT) * PH * PW * sizeof(float) + (2 * (size_t)Kp + (size_t)cin * PH + 2 * (size_t)cin + 4 * (size_t)(Kp >> 3)) * sizeof(int);
    STRAPS_REQUIRE(lds <= 160 * 1024, "straps_stem_fwd: %d input channels need %zu B of LDS (max 160 KiB)", cin, lds);
    const int Ho = (h + 6 - 7) / 2 + 1, Wo = (w + 6 - 7) / 2 + 1;
    const int tiles_x = (Wo + TX - 1) / TX, tiles_y = (Ho + TY - 1) / TY;
    static size_t lds_set[64] = {};                 // per device: the limit grows with the channel count
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { straps_set_error("stem_kernel: hipGetDevice failed"); return STRAPS_EHIP; }
    if (lds > lds_set[dev & 63]) {
        hipError_t e = hipFuncSetAttribute((const void*)stem_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { straps_set_error("stem_kernel: cannot raise dynamic LDS to %zu: %s", lds, hipGetErrorString(e)); return STRAPS_EHIP; }
        lds_set[dev & 63] = lds;
    }
    const long long nblk = (long long)batch * tiles_x * tiles_y;
    STRAPS_REQUIRE(nblk < (1LL << 31), "straps_stem_fwd: grid too large");
    hipLaunchKernelGGL(stem_kernel, dim3((unsigned)nblk), dim3(256), lds, (hipStream_t)stream, x, w_frag, scale, shift, relu, y,
                       stats_partial, nzmask, batch, cin, h, w, Ho, Wo, tiles_x, tiles_y, K, Kp);
    STRAPS_CHECK_LAUNCH("stem_kernel");
    return STRAPS_OK;
}
