// smpl.hip -- SMPL forward for gfx950 (replaces models/smpl_official.py:27-41 -> smplx.lbs.lbs).
//
// Three kernels per call:
//   1. smpl_pose_kernel   : per (body, joint) lane -- joint regression from betas (linearity:
//                           J = J_template + J_shapedirs.beta), 24-joint kinematic chain by tree
//                           depth with wave shuffles, rest-pose removal, pose-feature rows.
//   2. smpl_verts_kernel  : the hot one.  Shape + pose-corrective blendshapes are ONE dense
//                           contraction  v_posed[b, (v,c)] = sum_k F[b,k] * D[k,(v,c)]  (K = 218)
//                           run on the exact-fp32 MFMA (32x32x2), 32 bodies x 32 vertices x 3
//                           coords per wave tile; linear blend skinning is then done per vertex
//                           on the VALU straight out of the accumulators (lane = body), staged
//                           through a per-wave LDS slice so the 12-byte vertices leave as
//                           coalesced rows.  8 waves per block, no barrier in the tile loop: the
//                           second wave of each SIMD skins/stores while the first contracts.
//   3. smpl_joints_kernel : picked vertices + the 45 sparse-regressed joints, gathered from the
//                           just-written vertices in a fixed order.
// Everything is deterministic (no atomics).
#include <stdlib.h>

#include "common.h"

namespace {

constexpr int KP = STRAPS_SMPL_KP;        // 224
constexpr int KG = KP / 8;                // 28 k-groups of 8
constexpr int NT = STRAPS_SMPL_TILES;     // 216 mesh-vertex tiles (virtual-vertex tiles follow, see straps_hip.h)
constexpr int NV = STRAPS_SMPL_V;         // 6890
constexpr int NW = 8;                     // waves per block of the vertex kernel (2 per SIMD)
constexpr int BT = 32;                    // bodies per block
constexpr int FS = 228;                   // LDS row strides (floats): 4*odd -> conflict-free b128
constexpr int AS = 292;
constexpr int HS = 50;                    // half-tile stage row stride: 48 floats + 2 (even: b64 row reads)

// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void smpl_pose_kernel(straps_smpl_model_t m, const float* __restrict__ betas,
                                                        const float* __restrict__ rotmats, float* __restrict__ F,
                                                        float* __restrict__ Amat, float* __restrict__ joints,
                                                        long long B) {
    const int lane = threadIdx.x & 63;
    const int j = lane & 31;
    const long long body = ((long long)blockIdx.x * 256 + threadIdx.x) >> 5;
    const bool vb = body < B;
    const bool vj = j < 24;
    const long long bb = vb ? body : 0;
    const int jj = vj ? j : 0;

    float beta[10];
#pragma unroll
    for (int l = 0; l < 10; ++l) beta[l] = betas[bb * 10 + l];
    float R[9];
#pragma unroll
    for (int e = 0; e < 9; ++e) R[e] = rotmats[(bb * 24 + jj) * 9 + e];
    float J[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float s = m.j_template[jj * 3 + c];
#pragma unroll
        for (int l = 0; l < 10; ++l) s = fmaf(m.j_shapedirs[(jj * 3 + c) * 10 + l], beta[l], s);
        J[c] = s;
    }
    const int par = m.parents[jj];
    const int dep = m.depth[jj];
    const int src = (lane & 32) + (par < 0 ? 0 : par);
    float rel[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float jp = __shfl(J[c], src, 64);
        rel[c] = (jj > 0) ? J[c] - jp : J[c];
    }
    float G[12];   // row-major 3x4 [R|t] of the global transform
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        G[r * 4 + 0] = R[r * 3 + 0];
        G[r * 4 + 1] = R[r * 3 + 1];
        G[r * 4 + 2] = R[r * 3 + 2];
        G[r * 4 + 3] = rel[r];
    }
    for (int d = 1; d <= m.max_depth; ++d) {
        float P[12];
#pragma unroll
        for (int e = 0; e < 12; ++e) P[e] = __shfl(G[e], src, 64);
        if (dep == d) {
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const float p0 = P[r * 4 + 0], p1 = P[r * 4 + 1], p2 = P[r * 4 + 2];
                G[r * 4 + 0] = p0 * R[0] + p1 * R[3] + p2 * R[6];
                G[r * 4 + 1] = p0 * R[1] + p1 * R[4] + p2 * R[7];
                G[r * 4 + 2] = p0 * R[2] + p1 * R[5] + p2 * R[8];
                G[r * 4 + 3] = p0 * rel[0] + p1 * rel[1] + p2 * rel[2] + P[r * 4 + 3];
            }
        }
    }
    if (!vb) return;
    float* Frow = F + body * KP;
    if (vj) {
        if (joints) {
            float* o = joints + (body * STRAPS_SMPL_NJOINTS_OUT + j) * 3;
            o[0] = G[3]; o[1] = G[7]; o[2] = G[11];
        }
        float* A = Amat + (body * 24 + j) * 12;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            A[r * 4 + 0] = G[r * 4 + 0];
            A[r * 4 + 1] = G[r * 4 + 1];
            A[r * 4 + 2] = G[r * 4 + 2];
            A[r * 4 + 3] = G[r * 4 + 3] - (G[r * 4 + 0] * J[0] + G[r * 4 + 1] * J[1] + G[r * 4 + 2] * J[2]);
        }
        if (j >= 1) {
#pragma unroll
            for (int e = 0; e < 9; ++e) Frow[11 + (j - 1) * 9 + e] = R[e] - ((e == 0 || e == 4 || e == 8) ? 1.0f : 0.0f);
        }
    } else if (j == 24) {
        Frow[0] = 1.0f;
#pragma unroll
        for (int l = 0; l < 10; ++l) Frow[1 + l] = beta[l];
    } else if (j == 25) {
#pragma unroll
        for (int e = 218; e < KP; ++e) Frow[e] = 0.0f;
    }
}

// ------------------------------------------------------------------------------------------
// 8 waves per block = 2 per SIMD: while one wave of a SIMD runs its 336-MFMA contraction the other
// does its skinning / stores on the VALU + LDS, so the two phases overlap in hardware (measured:
// run back to back by one wave per SIMD they cost ~6 ms + ~6 ms at B = 65536).  No block barrier
// inside the tile loop: each wave owns its stage slice.
__global__ __launch_bounds__(NW * 64) STRAPS_NO_PACKED_FP32 void smpl_verts_kernel(straps_smpl_model_t m, const float* __restrict__ F,
                                                             const float* __restrict__ Amat, float* __restrict__ verts,
                                                             float* __restrict__ vout, long long B, int btiles,
                                                             int rounds, int rounds_per_chunk) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Fs = smem;                      // [32][FS]
    float* As_ = Fs + BT * FS;             // [32][AS]
    float* stage = As_ + BT * AS;          // [NW][32][HS]  half a tile (16 vertices) per wave
    float* skin_lds = stage + NW * BT * HS; // [NW][256]    per-tile skinning weights + joint offsets

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int h = lane >> 5;
    const int bl = lane & 31;              // body within the block tile
    // chunk-major logical order, contiguous per XCD: an XCD streams one ~2 MB slice of the blend
    // fragments at a time (L2-resident) while it walks the body tiles
    const int logical = xcd_remap(blockIdx.x, gridDim.x);
    const int chunk = logical / btiles;
    const long long b0 = (long long)(logical - chunk * btiles) * BT;
    const int nb = (int)((B - b0) < BT ? (B - b0) : BT);

    // ---- stage the block's feature rows and joint transforms ----
    for (int i = tid; i < BT * (KP / 4); i += NW * 64) {
        const int b = i / (KP / 4), q = i % (KP / 4);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (b < nb) v = *reinterpret_cast<const f32x4*>(F + (b0 + b) * KP + q * 4);
        *reinterpret_cast<f32x4*>(Fs + b * FS + q * 4) = v;
    }
    for (int i = tid; i < BT * 72; i += NW * 64) {
        const int b = i / 72, q = i % 72;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (b < nb) v = *reinterpret_cast<const f32x4*>(Amat + (b0 + b) * 288 + q * 4);
        *reinterpret_cast<f32x4*>(As_ + b * AS + q * 4) = v;
    }
    __syncthreads();

    const int round0 = chunk * rounds_per_chunk;
    const int round1 = min(round0 + rounds_per_chunk, rounds);
    const int vrow_floats = (m.n_tiles - NT) * 96;      // virtual-vertex row of one body
    const f32x4* __restrict__ blend = reinterpret_cast<const f32x4*>(m.blend_frag);
    float* mystage = stage + wave * BT * HS;
    float* skw = skin_lds + wave * 256;    // [32][4] weights, then [32][4] joint offsets (int bits)
    const int KW = m.skin_k;
    const float* Ab = As_ + bl * AS;
    const float* frow = Fs + bl * FS + 4 * h;

    for (int rd = round0; rd < round1; ++rd) {
        const int tile = rd * NW + wave;
        // skinning weights / joint offsets of this tile's 32 vertices -> LDS now, so their L2 latency hides under the
        // MFMA loop instead of stalling the per-vertex loop (fast path: 4 weights per vertex, the real model's layout)
        if (KW == 4) {
            const int e2 = lane * 2;
            const f32x2 w2 = *reinterpret_cast<const f32x2*>(m.skin_w + tile * 128 + e2);
            const int2 j2 = *reinterpret_cast<const int2*>(m.skin_j + tile * 128 + e2);
            skw[e2] = w2[0]; skw[e2 + 1] = w2[1];
            reinterpret_cast<int*>(skw)[128 + e2] = j2.x * 12;
            reinterpret_cast<int*>(skw)[128 + e2 + 1] = j2.y * 12;
        }
        // ---------------- blendshape contraction on the fp32 MFMA ----------------
        f32x16 ax, ay, az;
#pragma unroll
        for (int r = 0; r < 16; ++r) { ax[r] = 0.f; ay[r] = 0.f; az[r] = 0.f; }
        {
            const f32x4* px = blend + ((long long)(tile * 3 + 0) * KG) * 64 + lane;
            const f32x4* py = px + KG * 64;
            const f32x4* pz = py + KG * 64;
            f32x4 cx0 = px[0], cy0 = py[0], cz0 = pz[0];
            f32x4 cx1 = px[64], cy1 = py[64], cz1 = pz[64];
#pragma unroll 2
            for (int g = 0; g < KG; g += 2) {
                f32x4 nx0 = cx0, ny0 = cy0, nz0 = cz0, nx1 = cx1, ny1 = cy1, nz1 = cz1;
                if (g + 2 < KG) {
                    nx0 = px[(g + 2) * 64]; ny0 = py[(g + 2) * 64]; nz0 = pz[(g + 2) * 64];
                    nx1 = px[(g + 3) * 64]; ny1 = py[(g + 3) * 64]; nz1 = pz[(g + 3) * 64];
                }
                const f32x4 f0 = *reinterpret_cast<const f32x4*>(frow + 8 * g);
                const f32x4 f1 = *reinterpret_cast<const f32x4*>(frow + 8 * g + 8);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    ax = mfma32(cx0[e], f0[e], ax);
                    ay = mfma32(cy0[e], f0[e], ay);
                    az = mfma32(cz0[e], f0[e], az);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    ax = mfma32(cx1[e], f1[e], ax);
                    ay = mfma32(cy1[e], f1[e], ay);
                    az = mfma32(cz1[e], f1[e], az);
                }
                cx0 = nx0; cy0 = ny0; cz0 = nz0; cx1 = nx1; cy1 = ny1; cz1 = nz1;
            }
        }
        // ---------------- linear blend skinning, lane = body, reg = vertex; 16 vertices at a time ----------------
#pragma unroll
        for (int half = 0; half < 2; ++half) {
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) {
                const int r = half * 8 + rr;
                const int vrow = (r & 3) + 8 * (r >> 2) + 4 * h;          // 0..31 within the tile
                const int vloc = vrow - 16 * half;                         // 0..15 within the half
                f32x4 t0 = {0.f, 0.f, 0.f, 0.f}, t1 = t0, t2 = t0;
                if (KW == 4) {
                    const f32x4 w4 = *reinterpret_cast<const f32x4*>(skw + vrow * 4);
                    const int4 j4 = *reinterpret_cast<const int4*>(reinterpret_cast<const int*>(skw) + 128 + vrow * 4);
                    const int jo[4] = {j4.x, j4.y, j4.z, j4.w};
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const f32x4 a0 = *reinterpret_cast<const f32x4*>(Ab + jo[k]);
                        const f32x4 a1 = *reinterpret_cast<const f32x4*>(Ab + jo[k] + 4);
                        const f32x4 a2 = *reinterpret_cast<const f32x4*>(Ab + jo[k] + 8);
                        t0 += w4[k] * a0; t1 += w4[k] * a1; t2 += w4[k] * a2;
                    }
                } else {
                    const int v = tile * 32 + vrow;
                    for (int k = 0; k < KW; ++k) {
                        const float w = m.skin_w[v * KW + k];
                        const int jo = m.skin_j[v * KW + k] * 12;
                        const f32x4 a0 = *reinterpret_cast<const f32x4*>(Ab + jo);
                        const f32x4 a1 = *reinterpret_cast<const f32x4*>(Ab + jo + 4);
                        const f32x4 a2 = *reinterpret_cast<const f32x4*>(Ab + jo + 8);
                        t0 += w * a0; t1 += w * a1; t2 += w * a2;
                    }
                }
                const float x = ax[r], y = ay[r], z = az[r];
                float* so = mystage + bl * HS + vloc * 3;
                so[0] = t0[0] * x + t0[1] * y + t0[2] * z + t0[3];
                so[1] = t1[0] * x + t1[1] * y + t1[2] * z + t1[3];
                so[2] = t2[0] * x + t2[1] * y + t2[2] * z + t2[3];
            }
            // the stage slice is private to this wave: wave-level ordering is enough (LDS ops of one wave retire in order)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // ---------------- coalesced row store: 32 bodies x 192 contiguous bytes ----------------
            {
                // mesh tiles go to verts [B][6890][3]; virtual tiles to the joint scratch [B][32*(n_tiles-216)][3]
                const bool mesh = tile < NT;
                const int v0 = (mesh ? tile : tile - NT) * 32 + half * 16;
                const int npair = mesh ? min(24, max(0, (NV - v0) * 3 / 2)) : 24;   // float2 columns that exist (NV*3 is even)
                const long long rstride = mesh ? (long long)(NV * 3) : (long long)vrow_floats;
                float* vbase = (mesh ? verts : vout) + b0 * rstride + (long long)v0 * 3;
                if (mesh || vout) {
#pragma unroll 4
                    for (int i = lane; i < BT * 24; i += 64) {
                        const int b = i / 24, c = i - b * 24;
                        if (b < nb && c < npair)
                            *reinterpret_cast<f32x2*>(vbase + b * rstride + c * 2) =
                                *reinterpret_cast<const f32x2*>(mystage + b * HS + c * 2);
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
}


// ------------------------------------------------------------------------------------------
// Split-precision variant of the vertex kernel (mode STRAPS_SMPL_SPLIT_F16): the K = 218 blend contraction -- 91 % of the
// forward's flops -- runs on the fp16 matrix pipe (v_mfma_f32_32x32x16_f16, 16x the fp32 MFMA rate) as THREE products of
// two-term fp16 splits with fp32 accumulation:
//     F . D  ~=  Fh.Dh + Fh.Dl + Fl.Dh ,   x = xh + xl,  xh = fp16(x),  xl = fp16(x - xh)
// Each factor keeps 22 mantissa bits, the dropped Fl.Dl term and the split residuals are ~3 * 2^-22 relative per product
// (7e-7), i.e. at the level of fp32 rounding of the result itself; products of fp16 values are exact in the fp32
// accumulator.  D is split on the host (pre-scaled by a power of two so the low halves stay normal numbers), F in the
// kernel while it is staged into LDS (scaled by 2^6: |feature| up to 1023 before fp16 overflows; betas live in +-10).
// Skinning, the joint chain and every other stage stay exact fp32.  Measured against the fp64 oracle the result is as
// close as the exact-fp32 kernel (tests/test_gpu_forward.py::test_smpl_split_precision_vs_oracle reports both).
// Fragment layout of blend_frag_h: [tile][kstep 14][coord 3][hi|lo][lane 64][8 halves]; lane l holds
// D[k = 16*kstep + 8*(l>>5) + j][vertex = 32*tile + (l&31)][coord], j = 0..7 (the A operand of the 32x32x16 MFMA; the
// B operand -- features, n = body -- uses the same k map, so the contraction is independent of the hardware's k order).
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
// out-of-range operands of the split kernels SATURATE (largest finite fp16) instead of overflowing to inf, whose low half x - inf would be
// NaN and poison the whole body: |feature| >= 1023 (e.g. a diverging regressor's betas) or |joint transform| >= 63 m then give finite,
// clipped meshes; inside the range nothing changes.  (NaN inputs stay NaN.)
__device__ __forceinline__ float sat_h(float x) { return fminf(fmaxf(x, -65504.f), 65504.f); }
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
constexpr int KS = KP / 16;               // 14 k-steps of 16
constexpr int FSH = 232;                  // halves per LDS feature row: 464 bytes = 4 * 29 dwords -> conflict-free b128 reads
constexpr float F_SCALE = 64.0f;          // 2^6

__device__ __forceinline__ f32x16 mfma16h(half8 a, half8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }

template <int PF>
__global__ __launch_bounds__(NW * 64) STRAPS_NO_PACKED_FP32 void smpl_verts_h_kernel(straps_smpl_model_t m, const float* __restrict__ F,
                                                               const float* __restrict__ Amat, float* __restrict__ verts,
                                                               float* __restrict__ vout, long long B, int btiles,
                                                               int rounds, int rounds_per_chunk) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    _Float16* Fh = reinterpret_cast<_Float16*>(smem);            // [32][FSH] high halves of the scaled features
    _Float16* Fl = Fh + BT * FSH;                                  // [32][FSH] low halves
    float* As_ = smem + BT * FSH;                                  // [32][AS]   (2 * 32 * FSH halves = 32 * FSH floats)
    float* stage = As_ + BT * AS;                                  // [NW][32][HS]
    float* skin_lds = stage + NW * BT * HS;                        // [NW][256]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int h = lane >> 5;
    const int bl = lane & 31;
    const int logical = xcd_remap(blockIdx.x, gridDim.x);
    const int chunk = logical / btiles;
    const long long b0 = (long long)(logical - chunk * btiles) * BT;
    const int nb = (int)((B - b0) < BT ? (B - b0) : BT);

    // ---- stage the block's feature rows (split into fp16 hi / lo) and joint transforms ----
    for (int i = tid; i < BT * (KP / 4); i += NW * 64) {
        const int b = i / (KP / 4), q = i % (KP / 4);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (b < nb) v = *reinterpret_cast<const f32x4*>(F + (b0 + b) * KP + q * 4);
        half4 hi, lo;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float x = sat_h(v[e] * F_SCALE);
            hi[e] = (_Float16)x;
            lo[e] = (_Float16)(x - (float)hi[e]);
        }
        *reinterpret_cast<half4*>(Fh + b * FSH + q * 4) = hi;
        *reinterpret_cast<half4*>(Fl + b * FSH + q * 4) = lo;
    }
    for (int i = tid; i < BT * 72; i += NW * 64) {
        const int b = i / 72, q = i % 72;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (b < nb) v = *reinterpret_cast<const f32x4*>(Amat + (b0 + b) * 288 + q * 4);
        *reinterpret_cast<f32x4*>(As_ + b * AS + q * 4) = v;
    }
    __syncthreads();

    const int round0 = chunk * rounds_per_chunk;
    const int round1 = min(round0 + rounds_per_chunk, rounds);
    const int vrow_floats = (m.n_tiles - NT) * 96;
    const half8* __restrict__ blend = reinterpret_cast<const half8*>(m.blend_frag_h);
    const float unscale = m.blend_h_unscale;
    float* mystage = stage + wave * BT * HS;
    float* skw = skin_lds + wave * 256;
    const int KW = m.skin_k;
    const float* Ab = As_ + bl * AS;
    const _Float16* fh_row = Fh + bl * FSH + 8 * h;
    const _Float16* fl_row = Fl + bl * FSH + 8 * h;

    half8 ring[PF + 1][6];
    if (round0 < round1) {
        const half8* q = blend + (long long)(round0 * NW + wave) * (KS * 6 * 64) + lane;
#pragma unroll
        for (int s = 0; s < PF; ++s)
#pragma unroll
            for (int f = 0; f < 6; ++f) ring[s][f] = q[s * 384 + f * 64];
    }

    for (int rd = round0; rd < round1; ++rd) {
        const int tile = rd * NW + wave;
        if (KW == 4) {
            const int e2 = lane * 2;
            const f32x2 w2 = *reinterpret_cast<const f32x2*>(m.skin_w + tile * 128 + e2);
            const int2 j2 = *reinterpret_cast<const int2*>(m.skin_j + tile * 128 + e2);
            skw[e2] = w2[0]; skw[e2 + 1] = w2[1];
            reinterpret_cast<int*>(skw)[128 + e2] = j2.x * 12;
            reinterpret_cast<int*>(skw)[128 + e2 + 1] = j2.y * 12;
        }
        // ---------------- blendshape contraction: 14 k-steps x 3 coordinates x 3 split products ----------------
        // The fragments come from L2 (each wave streams its own tile: 84 KB): one k-step is only 9 x 32 = 288 MFMA cycles, far
        // less than a loaded L2 round trip, so the loads run PF k-steps ahead in a register ring (fully unrolled: static
        // indices), and the first PF steps of the NEXT tile are issued before this tile's skinning phase starts.
        f32x16 ax, ay, az;
#pragma unroll
        for (int r = 0; r < 16; ++r) { ax[r] = 0.f; ay[r] = 0.f; az[r] = 0.f; }
        {
            const half8* p = blend + (long long)tile * (KS * 6 * 64) + lane;      // [kstep][coord][hi|lo][lane]
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                if (s + PF < KS) {
                    const half8* q = p + (s + PF) * 384;
#pragma unroll
                    for (int f = 0; f < 6; ++f) ring[(s + PF) % (PF + 1)][f] = q[f * 64];
                }
                // (the machine scheduler would otherwise sink every load to just before its first use -- register pressure --
                //  and the ring would prefetch nothing: pin the issue point)
                __builtin_amdgcn_sched_barrier(0);
                const half8 fh = *reinterpret_cast<const half8*>(fh_row + 16 * s);
                const half8 fl = *reinterpret_cast<const half8*>(fl_row + 16 * s);
                const half8* c = ring[s % (PF + 1)];
                ax = mfma16h(c[0], fh, ax); ay = mfma16h(c[2], fh, ay); az = mfma16h(c[4], fh, az);       // Dh . Fh
                ax = mfma16h(c[1], fh, ax); ay = mfma16h(c[3], fh, ay); az = mfma16h(c[5], fh, az);       // Dl . Fh
                ax = mfma16h(c[0], fl, ax); ay = mfma16h(c[2], fl, ay); az = mfma16h(c[4], fl, az);       // Dh . Fl
            }
            if (rd + 1 < round1) {                                                  // next tile's head: in flight during the skinning below
                const half8* q = p + (long long)NW * (KS * 6 * 64);
#pragma unroll
                for (int s = 0; s < PF; ++s)
#pragma unroll
                    for (int f = 0; f < 6; ++f) ring[s][f] = q[s * 384 + f * 64];
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---------------- linear blend skinning (exact fp32), lane = body, reg = vertex; 16 vertices at a time ----------------
#pragma unroll
        for (int half = 0; half < 2; ++half) {
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) {
                const int r = half * 8 + rr;
                const int vrow = (r & 3) + 8 * (r >> 2) + 4 * h;
                const int vloc = vrow - 16 * half;
                f32x4 t0 = {0.f, 0.f, 0.f, 0.f}, t1 = t0, t2 = t0;
                if (KW == 4) {
                    const f32x4 w4 = *reinterpret_cast<const f32x4*>(skw + vrow * 4);
                    const int4 j4 = *reinterpret_cast<const int4*>(reinterpret_cast<const int*>(skw) + 128 + vrow * 4);
                    const int jo[4] = {j4.x, j4.y, j4.z, j4.w};
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const f32x4 a0 = *reinterpret_cast<const f32x4*>(Ab + jo[k]);
                        const f32x4 a1 = *reinterpret_cast<const f32x4*>(Ab + jo[k] + 4);
                        const f32x4 a2 = *reinterpret_cast<const f32x4*>(Ab + jo[k] + 8);
                        t0 += w4[k] * a0; t1 += w4[k] * a1; t2 += w4[k] * a2;
                    }
                } else {
                    const int v = tile * 32 + vrow;
                    for (int k = 0; k < KW; ++k) {
                        const float w = m.skin_w[v * KW + k];
                        const int jo = m.skin_j[v * KW + k] * 12;
                        const f32x4 a0 = *reinterpret_cast<const f32x4*>(Ab + jo);
                        const f32x4 a1 = *reinterpret_cast<const f32x4*>(Ab + jo + 4);
                        const f32x4 a2 = *reinterpret_cast<const f32x4*>(Ab + jo + 8);
                        t0 += w * a0; t1 += w * a1; t2 += w * a2;
                    }
                }
                const float x = ax[r] * unscale, y = ay[r] * unscale, z = az[r] * unscale;      // power of two: exact
                float* so = mystage + bl * HS + vloc * 3;
                so[0] = t0[0] * x + t0[1] * y + t0[2] * z + t0[3];
                so[1] = t1[0] * x + t1[1] * y + t1[2] * z + t1[3];
                so[2] = t2[0] * x + t2[1] * y + t2[2] * z + t2[3];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            {
                const bool mesh = tile < NT;
                const int v0 = (mesh ? tile : tile - NT) * 32 + half * 16;
                const int npair = mesh ? min(24, max(0, (NV - v0) * 3 / 2)) : 24;
                const long long rstride = mesh ? (long long)(NV * 3) : (long long)vrow_floats;
                float* vbase = (mesh ? verts : vout) + b0 * rstride + (long long)v0 * 3;
                if (mesh || vout) {
#pragma unroll 4
                    for (int i = lane; i < BT * 24; i += 64) {
                        const int b = i / 24, c = i - b * 24;
                        if (b < nb && c < npair)
                            *reinterpret_cast<f32x2*>(vbase + b * rstride + c * 2) =
                                *reinterpret_cast<const f32x2*>(mystage + b * HS + c * 2);
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
}

// ------------------------------------------------------------------------------------------
// mode STRAPS_SMPL_SPLIT_F16_LBS: the skinning stage on the matrix pipe as well.  Measured on the kernel above (rocprofv3 PMC,
// profiles/r02_smpl_*): with the blend contraction down to 18 % MFMA utilisation the wave time is the per-vertex VALU skinning
// (~1200 VALU instructions and ~220 dependent LDS reads per 32-vertex tile) -- latency-bound with two waves per SIMD.
// The per-vertex transform  T[v][b] = sum_j W[v][j] A[b][j]  (3x4 entries) is itself a small GEMM: M = 32 vertices,
// N = 32 bodies (one 32x32 tile per entry e = 0..11), K = 24 joints padded to 32.  It runs as the same three-product fp16
// split (weights scaled 2^14 and split on the host, joint transforms scaled 2^10 and split while they are staged into LDS),
// its accumulators come out in the blend accumulators' layout (lane = body, register = vertex), so
//     out_c = (T[4c]*x + T[4c+1]*y + T[4c+2]*z) + T[4c+3]
// is 4 VALU operations per value: 72 MFMAs + ~250 VALU per tile instead of ~1200 VALU + the LDS gathers, any number of
// non-zero weights per vertex, no joint-index table.  Error of the split products: 3 * 2^-22 relative per product.
// Round 4: the three products of the skinning split are packed along K -- T = Ah.Wh + Al.Wh + Ah.Wl = [Ah | Al | Ah | -] . [Wh | Wh | Wl | 0] over
// 24 + 24 + 24 + 8 = 80 columns, 5 k-steps of 16 instead of 3 products x 2 k-steps (skin_frag_p, straps_hip.h): 5 MFMAs per entry instead of 6 and
// 3 LDS operand reads instead of 4 -- k-step 3 re-uses k-step 0's registers and k-step 4 those of k-step 1, whose last 8 columns meet zero
// weights.  Joint transforms are staged as packed [Ah 24 | Al 24 | 8 unused] rows of 112 bytes (= 4 * 7 dwords: conflict-free b128 reads).
// Both kernels of this mode (32 and 64 bodies per workgroup) use the same chain, so their results are bit-identical.
constexpr int ASP = 56;                   // halves per packed (entry, body) row
constexpr int KSP = 5;                    // k-steps of the K-packed skinning product
constexpr float A_SCALE = 1024.0f;        // 2^10: |A| < 63 (metres) before fp16 overflows
constexpr float W_UNSCALE = 1.0f / (16384.0f * 1024.0f);   // weights are scaled 2^14 on the host

// PD16 (mode STRAPS_SMPL_SPLIT_F16_LBS_PD16): from the second k step on -- columns 16.. of the contraction, all of them pose-corrective
// directions, whose summed contribution to a vertex is centimetres -- the directions are taken as plain fp16 (their low halves are
// neither fetched nor multiplied): two products (Fh.Dh + Fl.Dh) instead of three for 13 of the 14 k steps, 45 % fewer fragment
// bytes from L2 (the fragment stream, 19 MB per 32 bodies, is what bounds this kernel).  Template + shape (+ the first five pose
// features, k step 0) keep the three-product split.  Error: ~2^-12 of the pose-corrective sum -- measured in the tests.
// PD16 = 2 (mode STRAPS_SMPL_SPLIT_F16_LBS_P16): additionally the pose FEATURES of those columns as plain fp16 -- one product.
template <int NWV, int PF, int ABL, int PD16 = 0>
__global__ __launch_bounds__(NWV * 64) void smpl_verts_hh_kernel(straps_smpl_model_t m, const float* __restrict__ F,
                                                                 const float* __restrict__ Amat, float* __restrict__ verts,
                                                                 float* __restrict__ vout, long long B, int btiles,
                                                                 int ntiles, int rounds, int rounds_per_chunk, unsigned long long* clk) {
    // (clk: measurement aid, NULL in library use -- straps_set_clock_accumulator: workgroup 0 adds the shader-clock and wall-clock ticks of
    //  its lifetime, bench.py's sclk_mhz of the SMPL-only workload)
    unsigned long long clk_c0 = 0, clk_w0 = 0;
    if (clk) { clk_c0 = __builtin_amdgcn_s_memtime(); clk_w0 = __builtin_amdgcn_s_memrealtime(); }
    // (ABL: compile-time measurement switches, 0 in production -- 1 = no output stores, 2 = no fragment loads after the first
    //  k-step, 4 = no skinning MFMAs, 8 = no blend MFMAs; tools/smpl_ablate.sh)
    constexpr int ablate = ABL;
    // OPERAND ROLES ARE SWAPPED with respect to the two kernels above: the per-body operand (features, joint transforms) is the
    // MFMA's A side (i = body) and the per-vertex operand (blend directions, skinning weights) its B side (n = vertex), so the
    // accumulators hold lane = VERTEX, register = body row.  A lane's x,y,z of one (body, vertex) are then 12 contiguous bytes
    // of the output and the 32 lanes of a half-wave cover 384 contiguous bytes of one body's row: the tile leaves as sixteen
    // global_store_dwordx3 straight from the registers -- no LDS transpose, no staging buffer, no wave barrier.
    // (The fragment packing is symmetric in the two sides: blend_frag_h / skin_frag_p are read the same way from either side.)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    _Float16* Fh = reinterpret_cast<_Float16*>(smem);            // [32][FSH]
    _Float16* Fl = Fh + BT * FSH;                                  // [32][FSH]
    _Float16* Ap = Fl + BT * FSH;                                  // [12][32][ASP]  Ap[(e*32 + body)*ASP + (0 | 24) + joint]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int h = lane >> 5;
    const int bl = lane & 31;              // as A-side index: body; as output column: vertex within the tile
    const int logical = xcd_remap(blockIdx.x, gridDim.x);
    const int chunk = logical / btiles;
    const long long b0 = (long long)(logical - chunk * btiles) * BT;
    const int nb = (int)((B - b0) < BT ? (B - b0) : BT);

    // ---- stage: feature rows and joint transforms, both split into fp16 hi / lo ----
    // Every global load of the prologue is issued before the first value is converted (a load -> wait -> write loop had cost
    // ~19 us per workgroup, a fifth of the kernel; loads are unconditional from a clamped row and zeroed afterwards: a load inside
    // a divergent branch is followed by a full vmcnt(0) at the join).  Joint-transform items are (body, joint 0..31): the K
    // padding columns 24..31 are written as zeros by their own items, so there is no separate clearing pass / barrier; lanes map
    // to joints within a body row, which spreads the 2-byte LDS writes over the banks.
    {
        constexpr int FITEMS = BT * (KP / 4), FPER = (FITEMS + NWV * 64 - 1) / (NWV * 64);
        constexpr int AITEMS = BT * 32, APER = (AITEMS + NWV * 64 - 1) / (NWV * 64);
        f32x4 fv[FPER], av[APER][3];
#pragma unroll
        for (int t = 0; t < FPER; ++t) {
            const int i = tid + t * NWV * 64;
            const int b = i / (KP / 4), q = i % (KP / 4);
            const bool ok = i < FITEMS && b < nb;
            fv[t] = *reinterpret_cast<const f32x4*>(F + (b0 + (ok ? b : 0)) * KP + q * 4);
            if (!ok) fv[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int t = 0; t < APER; ++t) {
            const int i = tid + t * NWV * 64;
            const int b = i >> 5, j = i & 31;
            const bool ok = i < AITEMS && b < nb && j < 24;
            const long long row = ok ? ((b0 + b) * 24 + j) : (b0 * 24);
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                av[t][q] = *reinterpret_cast<const f32x4*>(Amat + row * 12 + q * 4);
                if (!ok) av[t][q] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
#pragma unroll
        for (int t = 0; t < FPER; ++t) {
            const int i = tid + t * NWV * 64;
            const int b = i / (KP / 4), q = i % (KP / 4);
            if (i < FITEMS) {
                half4 hi, lo;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float x = sat_h(fv[t][e] * F_SCALE);
                    hi[e] = (_Float16)x;
                    lo[e] = (_Float16)(x - (float)hi[e]);
                }
                *reinterpret_cast<half4*>(Fh + b * FSH + q * 4) = hi;
                *reinterpret_cast<half4*>(Fl + b * FSH + q * 4) = lo;
            }
        }
#pragma unroll
        for (int t = 0; t < APER; ++t) {
            const int i = tid + t * NWV * 64;
            const int b = i >> 5, j = i & 31;
            if (i < AITEMS) {
#pragma unroll
                for (int e = 0; e < 12; ++e) {
                    const float x = sat_h(av[t][e >> 2][e & 3] * A_SCALE);
                    const _Float16 hi = (_Float16)x;
                    if (j < 24) {
                        Ap[(e * BT + b) * ASP + j] = hi;
                        Ap[(e * BT + b) * ASP + 24 + j] = (_Float16)(x - (float)hi);
                    }
                }
            }
        }
    }
    __syncthreads();

    const int round0 = chunk * rounds_per_chunk;
    const int round1 = min(round0 + rounds_per_chunk, rounds);
    const int vrow_floats = (m.n_tiles - NT) * 96;
    const half8* __restrict__ blend = reinterpret_cast<const half8*>(m.blend_frag_h);
    const half8* __restrict__ skin = reinterpret_cast<const half8*>(m.skin_frag_p);
    const float us_blend = m.blend_h_unscale;
    const _Float16* fh_row = Fh + bl * FSH + 8 * h;
    const _Float16* fl_row = Fl + bl * FSH + 8 * h;
    const _Float16* a_row = Ap + bl * ASP + 8 * h;                  // entry e adds 32 * ASP, packed k-step s adds 16 s

    half8 ring[PF + 1][6];
    {
        const int t0 = round0 * NWV + wave;
        if (round0 < round1 && t0 < ntiles) {
            const half8* q = blend + (long long)t0 * (KS * 6 * 64) + lane;
#pragma unroll
            for (int s = 0; s < PF; ++s)
#pragma unroll
                for (int f = 0; f < 6; ++f)
                    if (!(PD16 && s >= 1 && (f & 1))) ring[s][f] = q[s * 384 + f * 64];
        }
    }
    // The stores of tile t-1 are issued AFTER the blend phase of tile t and BEFORE its skinning phase: on gfx9 one in-order
    // counter tracks loads and stores, so the first wait for a load issued behind stores drains the stores too (~1-2 us of
    // write acknowledgements).  Placed here the drain has the whole LDS/MFMA-only skinning phase to complete in.
    f32x16 outp[3];
    int ptile = -1;
    auto store_tile = [&](int t, const f32x16* o3) {
        const bool mesh = t < NT;
        const int v = (mesh ? t : t - NT) * 32 + bl;
        const long long rstride = mesh ? (long long)(NV * 3) : (long long)vrow_floats;
        float* vbase = (mesh ? verts : vout) + b0 * rstride + (long long)v * 3;
        const bool vok = mesh ? v < NV : (vout != nullptr);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int body = (r & 3) + 8 * (r >> 2) + 4 * h;
            if ((ablate & 1) && o3[0][r] != 12345.678f) continue;          // (never true: keeps the arithmetic alive)
            if (vok && body < nb) {
                float* o = vbase + body * rstride;
                o[0] = o3[0][r]; o[1] = o3[1][r]; o[2] = o3[2][r];
            }
        }
    };
    for (int rd = round0; rd < round1; ++rd) {
        const int tile = rd * NWV + wave;
        if (tile >= ntiles) break;                           // (wave-uniform: the last round may be ragged)
        // skinning-weight fragments of this tile [kstep 2][hi|lo][lane]: issued first, needed after the blend contraction
        const half8* sp = skin + (long long)tile * (KSP * 64) + lane;
        const half8 sw0 = sp[0], sw1 = sp[64], sw2 = sp[128], sw3 = sp[192], sw4 = sp[256];
        f32x16 ax, ay, az;
#pragma unroll
        for (int r = 0; r < 16; ++r) { ax[r] = 0.f; ay[r] = 0.f; az[r] = 0.f; }
        // LDS operands run one step ahead of the MFMAs that consume them (the wave would otherwise sit through an LDS round
        // trip per k-step / per skinning tile: 26 exposed round trips per tile were ~40 % of the wave's time)
        half8 fh = *reinterpret_cast<const half8*>(fh_row);
        half8 fl = *reinterpret_cast<const half8*>(fl_row);
        {
            const half8* p = blend + (long long)tile * (KS * 6 * 64) + lane;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                if (s + PF < KS && !(ablate & 2)) {
                    const half8* q = p + (s + PF) * 384;
#pragma unroll
                    for (int f = 0; f < 6; ++f)
                        if (!(PD16 && (f & 1))) ring[(s + PF) % (PF + 1)][f] = q[f * 64];          // (s + PF >= 1)
                }
                half8 fhn = fh, fln = fl;
                if (s + 1 < KS) {
                    fhn = *reinterpret_cast<const half8*>(fh_row + 16 * (s + 1));
                    fln = *reinterpret_cast<const half8*>(fl_row + 16 * (s + 1));
                }
                __builtin_amdgcn_sched_barrier(0);
                const half8* c = ring[s % (PF + 1)];
                if (!(ablate & 8)) {
                ax = mfma16h(fh, c[0], ax); ay = mfma16h(fh, c[2], ay); az = mfma16h(fh, c[4], az);       // Fh . Dh
                if (!(PD16 && s >= 1)) { ax = mfma16h(fh, c[1], ax); ay = mfma16h(fh, c[3], ay); az = mfma16h(fh, c[5], az); }       // Fh . Dl
                if (!(PD16 == 2 && s >= 1)) { ax = mfma16h(fl, c[0], ax); ay = mfma16h(fl, c[2], ay); az = mfma16h(fl, c[4], az); }       // Fl . Dh
                } else { ax[0] += (float)fh[0] + (float)c[0][0]; ay[0] += (float)fl[1] + (float)c[3][1]; az[0] += (float)c[5][2]; }
                fh = fhn; fl = fln;
            }
            // the head of the NEXT tile's fragments is issued before this tile's stores: a load issued behind stores can only be
            // waited for by draining the stores as well (one in-order counter for both on gfx9)
            const int tnext = tile + NWV;
            if (rd + 1 < round1 && tnext < ntiles) {
                const half8* q = p + (long long)NWV * (KS * 6 * 64);
#pragma unroll
                for (int s = 0; s < PF; ++s)
#pragma unroll
                    for (int f = 0; f < 6; ++f)
                        if (!(PD16 && s >= 1 && (f & 1))) ring[s][f] = q[s * 384 + f * 64];
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (ptile >= 0) store_tile(ptile, outp);
        __builtin_amdgcn_sched_barrier(0);
        // ---------------- skinning on the matrix pipe: T_e[body][vertex] = sum_j A[body][j][e] W[vertex][j] ----------------
        // Software pipeline over the twelve entries e: [LDS operands of e+1] -> [6-MFMA chain of e] -> [fold of e-1 on the VALU
        // while that chain runs].  out_c = (T[4c] x + T[4c+1] y + T[4c+2] z) us_rot + T[4c+3] us_w.
        f32x16 out[3];
        const float us_rot = us_blend * W_UNSCALE;       // (T * blend) carries both scales, the translation column only the skin scale
        const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        half8 a0 = *reinterpret_cast<const half8*>(a_row), a1 = *reinterpret_cast<const half8*>(a_row + 16), a2 = *reinterpret_cast<const half8*>(a_row + 32);
        f32x16 Tprev = zero16, acc = zero16;
#pragma unroll
        for (int e = 0; e <= 12; ++e) {
            half8 n0 = a0, n1 = a1, n2 = a2;
            if (e + 1 < 12) {
                n0 = *reinterpret_cast<const half8*>(a_row + (e + 1) * BT * ASP);
                n1 = *reinterpret_cast<const half8*>(a_row + (e + 1) * BT * ASP + 16);
                n2 = *reinterpret_cast<const half8*>(a_row + (e + 1) * BT * ASP + 32);
            }
            __builtin_amdgcn_sched_barrier(0);
            f32x16 T = zero16;
            if (e < 12 && !(ablate & 4)) {
                T = mfma16h(a0, sw0, zero16);
                T = mfma16h(a1, sw1, T);
                T = mfma16h(a2, sw2, T);
                T = mfma16h(a0, sw3, T);                      // (packed k-steps 3, 4: the operands of 0, 1 again)
                T = mfma16h(a1, sw4, T);
            } else if (e < 12) {
                T[0] = (float)a0[0] + (float)a2[1];
            }
            if (e > 0) {                                  // fold entry e-1 (Tprev) while the chain above is in the pipe
                const int pe = e - 1, c = pe >> 2, q = pe & 3;
                if (q == 0) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[r] = Tprev[r] * ax[r];
                } else if (q == 1) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[r] += Tprev[r] * ay[r];
                } else if (q == 2) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[r] += Tprev[r] * az[r];
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) out[c][r] = acc[r] * us_rot + Tprev[r] * W_UNSCALE;
                    asm volatile("" : "+v"(out[c]));
                }
                asm volatile("" : "+v"(acc));
            }
            __builtin_amdgcn_sched_barrier(0);
            Tprev = T;
            a0 = n0; a1 = n1; a2 = n2;
        }
        // (lane = vertex, register r = body (r&3) + 8 (r>>2) + 4 h: stored after the NEXT tile's blend phase, see above)
#pragma unroll
        for (int c = 0; c < 3; ++c) outp[c] = out[c];
        ptile = tile;
    }
    if (ptile >= 0) store_tile(ptile, outp);
    if (clk && blockIdx.x == 0 && threadIdx.x == 0) {
        atomicAdd(clk, (unsigned long long)__builtin_amdgcn_s_memtime() - clk_c0);
        atomicAdd(clk + 1, (unsigned long long)__builtin_amdgcn_s_memrealtime() - clk_w0);
    }
}

// ------------------------------------------------------------------------------------------
// The large-batch form of the kernel above (round 4): 64 BODIES PER WORKGROUP, ONE WAVE PER SIMD WITH THE WHOLE 512-ENTRY REGISTER FILE.
// What bounded smpl_verts_hh_kernel (profiles/r03_smpl_ablate.txt) was structural: every 32-body wave re-streams all direction tiles from
// L2 (38.9 GB per 65 536 bodies, ~57 B/clk/CU against the ~64 the L2 delivers), and with 256 registers per wave neither a deeper
// fragment ring nor a second body group fits, so stores, fragment loads and MFMAs added up instead of overlapping.  Here
//   * a wave holds TWO 32-body groups: every fetched direction fragment (the MFMA's B side) feeds both -- half the L2 -> register bytes,
//     twice the matrix work (18 MFMAs = 576 cycles) behind each k-step's loads, which are requested PF k-steps ahead;
//   * the three products of the skinning split are packed along K (skin_frag_p): T = [Ah | Al | Ah | -] . [Wh | Wh | Wl | 0] is 5 MFMAs per
//     entry instead of 6, with 3 LDS operand reads instead of 4 (k-step 3 re-uses k-step 0's registers, k-step 4 those of k-step 1:
//     its last 8 columns meet zero weights) -- 186 instead of 198 MFMAs per 32 x 32 tile;
//   * the skinning chains write VGPRs (TV = 1: the VGPR-destination encoding of the same instruction, through inline assembly): a kernel that
//     may use more than 256 registers gets every builtin MFMA in its AGPR-destination form, and the fold on the VALU would pay one
//     v_accvgpr_read per value (16 per entry, as many as the fold's own arithmetic).  The blend accumulators stay in AGPRs (read once);
//   * joint transforms are staged as packed [Ah 24 | Al 24 | pad 8] rows of 112 bytes (conflict-free b128 reads), 84 KB for 64 bodies next
//     to the 58 KB of split features: 142 KB of the 160 KB, one workgroup per CU.
// Same arithmetic class as the kernel above (every product split three ways, fp32 accumulate); the skinning products are added in one
// K-packed chain, so results may differ from it in the last bit.  Operand roles, output mapping and the store form are unchanged.
constexpr int WB = 64;                    // bodies per workgroup
constexpr int NWW = 4;                    // waves per workgroup = one per SIMD

// v_mfma_f32_32x32x16_f16 with VGPR C / D.  The hazard recogniser does not see inside inline assembly; the two rules this kernel relies on
// (MI355X guide, "inline assembly"; the compiler's own code for the builtin shows the same): an accumulate chain (D of one = C of the next,
// whole) needs no wait states; any other reader of D needs 11 (8-pass instruction) -- provided by construction, see the skinning loop.
__device__ __forceinline__ void mfma16h_v0(f32x16& d, const half8& a, const half8& b) {
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(d) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma16h_v(f32x16& d, const half8& a, const half8& b) {
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "v"(b));
}
// the first two links of a chain in ONE statement: nothing can be scheduled between them, so whatever follows is >= 2 MFMA issue periods
// (>= 64 cycles) behind the previous chain's last MFMA
__device__ __forceinline__ void mfma16h_v01(f32x16& d, const half8& a0, const half8& b0, const half8& a1, const half8& b1) {
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0\n\tv_mfma_f32_32x32x16_f16 %0, %3, %4, %0" : "=&v"(d) : "v"(a0), "v"(b0), "v"(a1), "v"(b1));
}

// ABL (tools build only; wrong results by design -- tools/smpl_w_ab.py --ablate): 1 = no output stores, 2 = no fragment loads after the
// prologue, 4 = no skinning MFMA chains, 8 = no blend MFMAs, 16 = no fold on the VALU, 32 = no LDS operand reads inside the tile loop
// SV: store / schedule variant bits -- 1 = non-temporal stores (the 5.5 GB of output stream through L2 without displacing the direction
// fragments the 32 CUs of an XCD share), 2 = the four waves of a workgroup start a quarter of a tile period apart (the chip's store bursts
// spread in time), 4 = the stores of tile t are issued inside the blend phase of tile t + 1, right after its last in-tile load request
template <int PF, int PD16, int TV, int ABL = 0, int SV = 0>
__global__ __launch_bounds__(NWW * 64) __attribute__((amdgpu_waves_per_eu(1, 1)))
void smpl_verts_w_kernel(straps_smpl_model_t m, const float* __restrict__ F, const float* __restrict__ Amat, float* __restrict__ verts,
                         float* __restrict__ vout, long long B, int bgroups, int ntiles, int rounds, int rounds_per_chunk,
                         unsigned long long* clk) {
    static_assert(PF >= 1 && PF < KS, "prefetch distance");
    constexpr int R = PF + 1;              // ring slots
    unsigned long long clk_c0 = 0, clk_w0 = 0;
    if (clk) { clk_c0 = __builtin_amdgcn_s_memtime(); clk_w0 = __builtin_amdgcn_s_memrealtime(); }
    extern __shared__ __attribute__((aligned(16))) float smem[];
    _Float16* Fh = reinterpret_cast<_Float16*>(smem);            // [64][FSH]
    _Float16* Fl = Fh + WB * FSH;                                  // [64][FSH]
    _Float16* Ap = Fl + WB * FSH;                                  // [12][64][ASP]   Ap[(e*64 + body)*ASP + (0|24) + joint]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5;
    const int il = lane & 31;              // as A-side index: body within its group; as output column: vertex within the tile
    const int logical = xcd_remap(blockIdx.x, gridDim.x);
    const int chunk = logical / bgroups;
    const long long b0 = (long long)(logical - chunk * bgroups) * WB;
    const int nb = (int)((B - b0) < WB ? (B - b0) : WB);

    // ---- stage: feature rows and joint transforms of the 64 bodies, split into fp16 hi / lo (all loads first, then the conversions) ----
    {
        constexpr int FPER = WB * (KP / 4) / (NWW * 64);          // 14 float4 per thread
        constexpr int APER = WB * 12 / (NWW * 64);                // 3 (body, joint pair) items per thread
        static_assert(FPER * NWW * 64 == WB * (KP / 4) && APER * NWW * 64 == WB * 12, "staging items must divide evenly");
        f32x4 fv[FPER], av[APER][6];
#pragma unroll
        for (int t = 0; t < FPER; ++t) {
            const int i = tid + t * NWW * 64;
            const int b = i / (KP / 4), q = i % (KP / 4);
            const bool ok = b < nb;
            fv[t] = *reinterpret_cast<const f32x4*>(F + (b0 + (ok ? b : 0)) * KP + q * 4);
            if (!ok) fv[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int t = 0; t < APER; ++t) {
            const int i = tid + t * NWW * 64;
            const int b = i / 12, jp = i % 12;
            const bool ok = b < nb;
            const float* src = Amat + ((b0 + (ok ? b : 0)) * 24 + 2 * jp) * 12;
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                av[t][q] = *reinterpret_cast<const f32x4*>(src + q * 4);
                if (!ok) av[t][q] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
#pragma unroll
        for (int t = 0; t < FPER; ++t) {
            const int i = tid + t * NWW * 64;
            const int b = i / (KP / 4), q = i % (KP / 4);
            half4 hi, lo;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float x = sat_h(fv[t][e] * F_SCALE);
                hi[e] = (_Float16)x;
                lo[e] = (_Float16)(x - (float)hi[e]);
            }
            *reinterpret_cast<half4*>(Fh + b * FSH + q * 4) = hi;
            *reinterpret_cast<half4*>(Fl + b * FSH + q * 4) = lo;
        }
        typedef _Float16 half2v __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int t = 0; t < APER; ++t) {
            const int i = tid + t * NWW * 64;
            const int b = i / 12, jp = i % 12;
#pragma unroll
            for (int e = 0; e < 12; ++e) {
                const float x0 = sat_h(av[t][e >> 2][e & 3] * A_SCALE), x1 = sat_h(av[t][3 + (e >> 2)][e & 3] * A_SCALE);
                half2v hi, lo;
                hi[0] = (_Float16)x0; hi[1] = (_Float16)x1;
                lo[0] = (_Float16)(x0 - (float)hi[0]); lo[1] = (_Float16)(x1 - (float)hi[1]);
                _Float16* row = Ap + (e * WB + b) * ASP + 2 * jp;
                *reinterpret_cast<half2v*>(row) = hi;
                *reinterpret_cast<half2v*>(row + 24) = lo;
            }
        }
    }
    __syncthreads();

    const int round0 = chunk * rounds_per_chunk;
    const int round1 = min(round0 + rounds_per_chunk, rounds);
    const int vrow_floats = (m.n_tiles - NT) * 96;
    const half8* __restrict__ blend = reinterpret_cast<const half8*>(m.blend_frag_h);
    const half8* __restrict__ skinp = reinterpret_cast<const half8*>(m.skin_frag_p);
    const float us_blend = m.blend_h_unscale;
    const float us_rot = us_blend * W_UNSCALE;       // (T * blend) carries both scales, the translation column only the skin scale
    const _Float16* fh_row = Fh + il * FSH + 8 * h;  // group g adds 32 * FSH, k-step s adds 16 s
    const _Float16* fl_row = Fl + il * FSH + 8 * h;
    const _Float16* a_row = Ap + il * ASP + 8 * h;   // group g adds 32 * ASP, entry e adds 64 * ASP, packed k-step s adds 16 s
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const bool full = nb == WB;

    auto load_step = [&](half8* slot, const half8* q, int ks) {      // the six fragments [coord][hi|lo] of one k-step
#pragma unroll
        for (int f = 0; f < 6; ++f)
            if (!(PD16 && ks >= 1 && (f & 1))) slot[f] = q[f * 64];
    };
    // one 12-byte-per-lane store: row r of group g's result tile t.  SV >> 8 selects the form: 0 = plain global store, 1 = non-temporal global
    // store, 100 + aux = buffer store with that cache policy (gfx940 aux bits: 1 = sc0, 2 = nt, 16 = sc1)
    constexpr int SP = SV >> 8;
    typedef float f32x3 __attribute__((ext_vector_type(3)));
    typedef unsigned u32x3 __attribute__((ext_vector_type(3)));
    __amdgpu_buffer_rsrc_t rs_mesh, rs_virt;
    if (SP >= 100) {
        rs_mesh = __builtin_amdgcn_make_buffer_rsrc(verts + b0 * (long long)(NV * 3), 0, 0x7fffffff, 0x00020000);
        rs_virt = __builtin_amdgcn_make_buffer_rsrc((vout ? vout : verts) + b0 * (long long)vrow_floats, 0, 0x7fffffff, 0x00020000);
    }
    auto store_one = [&](int t, int g, int r, const f32x16* o3, bool guard) {
        const bool mesh = t < NT;
        const int v = (mesh ? t : t - NT) * 32 + il;
        const long long rstride = mesh ? (long long)(NV * 3) : (long long)vrow_floats;
        const bool vok = t >= 0 && (mesh ? v < NV : (vout != nullptr));
        const int body = (r & 3) + 8 * (r >> 2) + 4 * h;
        if ((ABL & 1) && o3[0][r] != 12345.678f) return;          // (never true: keeps the arithmetic alive)
        if (vok && (!guard || g * 32 + body < nb)) {
            const int brow = (ABL & 64) ? (body & 7) : g * 32 + body;      // (ABL & 64: every store of the chip lands in the same 8 rows -- L2-resident)
            if (SP >= 100) {
                const unsigned off = (unsigned)((brow * (int)rstride + v * 3) * 4);
                u32x3 t3 = {__float_as_uint(o3[0][r]), __float_as_uint(o3[1][r]), __float_as_uint(o3[2][r])};
                if (mesh) __builtin_amdgcn_raw_buffer_store_b96(t3, rs_mesh, off, 0, SP - 100);
                else __builtin_amdgcn_raw_buffer_store_b96(t3, rs_virt, off, 0, SP - 100);
            } else {
                float* o = (mesh ? verts : vout) + (((ABL & 64) ? 0 : b0) + brow) * rstride + (long long)v * 3;
                if (SP == 1) {
                    f32x3 t3 = {o3[0][r], o3[1][r], o3[2][r]};
                    __builtin_nontemporal_store(t3, reinterpret_cast<f32x3*>(o));
                } else {
                    o[0] = o3[0][r]; o[1] = o3[1][r]; o[2] = o3[2][r];
                }
            }
        }
    };
    auto store_group = [&](int t, int g, const f32x16* o3, bool guard) {
#pragma unroll
        for (int r = 0; r < 16; ++r) store_one(t, g, r, o3, guard);
    };

    half8 ring[R][6];
    int tile = round0 * NWW + wave;
    if (SV & 2) {                                             // wave w starts w quarter periods (~3 700 cycles each; SV & 8: eighths) late
        for (int k = 0; k < wave; ++k) __builtin_amdgcn_s_sleep((SV & 8) ? 29 : 58);
    }
    if (round0 < round1 && tile < ntiles) {
        const half8* q = blend + (long long)tile * (KS * 6 * 64) + lane;
#pragma unroll
        for (int s = 0; s < PF; ++s) load_step(ring[s], q + s * 384, s);
    }
    f32x16 outp[2][3];                                        // (SV & 4: the previous tile's results, stored inside this tile's blend phase)
    int ptile = -1;
    unsigned long long ph_blend = 0, ph_skin0 = 0, ph_skin1 = 0;      // (ABL & 128: shader cycles per phase, summed over this wave's tiles)
    for (int rd = round0; rd < round1; ++rd, tile += NWW) {
        if (tile >= ntiles) break;                           // (wave-uniform: the last round may be ragged)
        unsigned long long ph_t0 = 0;
        if (ABL & 128) ph_t0 = __builtin_amdgcn_s_memtime();
        const bool has_next = rd + 1 < round1 && tile + NWW < ntiles;
        const half8* p = blend + (long long)tile * (KS * 6 * 64) + lane;
        const half8* pn = has_next ? p + (long long)NWW * (KS * 6 * 64) : p;      // (no next tile: the head loads re-read this one, unused)
        const half8* sp = skinp + (long long)tile * (KSP * 64) + lane;
        half8 sw[KSP];
        f32x16 acc[2][3];
        // ---------------- blend contraction: acc[g][c][body][vertex] = sum_k F[body][k] D[k][vertex][c], both body groups per fragment ----------------
        // Per k-step: 18 MFMAs (6 accumulators x 3 products), and ONE memory request after every third of them, fenced in place with
        // sched_barriers (left alone the scheduler issues the step's six loads in a block in front of its MFMAs: ~80 cycles of an idle
        // matrix pipe per k-step).  The request is k-step s + PF of this tile into the slot step s - 1 freed; in the last PF steps, where
        // the tile has nothing left to ask for, the slot takes the NEXT tile's head k-step of the same number when there is one (the rest of
        // the head follows the loop): those loads are then in front of this tile's stores (one in-order counter tracks loads and stores
        // on gfx9 -- a load issued behind stores can only be waited for by draining them) and land under the skinning phase.
        half8 fh[2], fl[2];
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            fh[g] = *reinterpret_cast<const half8*>(fh_row + g * 32 * FSH);
            fl[g] = *reinterpret_cast<const half8*>(fl_row + g * 32 * FSH);
        }
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            half8 fhn[2] = {fh[0], fh[1]}, fln[2] = {fl[0], fl[1]};
            if (s + 1 < KS && !(ABL & 32)) {
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    fhn[g] = *reinterpret_cast<const half8*>(fh_row + g * 32 * FSH + 16 * (s + 1));
                    fln[g] = *reinterpret_cast<const half8*>(fl_row + g * 32 * FSH + 16 * (s + 1));
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            const half8* c = ring[s % R];
            const bool p2 = !(PD16 && s >= 1), p3 = !(PD16 == 2 && s >= 1);
            const int nm = 6 * (1 + (p2 ? 1 : 0) + (p3 ? 1 : 0)), stride = nm / 6;
            const bool in_tile = s + PF < KS;
            const int hj = (s + R - 1) % R;                  // the slot step s - 1 freed
            const bool head = !in_tile && hj < PF;
            const int ks_l = in_tile ? s + PF : hj;          // the k-step requested at this step (of this tile / of the next one)
            half8* slot = ring[in_tile ? (s + PF) % R : hj];
            const half8* src = (in_tile ? p : pn) + ks_l * 384;
#pragma unroll
            for (int j = 0; j < nm; ++j) {
                int prod = j / 6;                            // 0: Fh.Dh, then Fh.Dl (unless skipped), then Fl.Dh
                if (prod == 1 && !p2) prod = 2;
                const int g = (j % 6) / 3, cc = j % 3;
                const half8& fa_ = prod == 2 ? fl[g] : fh[g];
                const half8& cb_ = c[2 * cc + (prod == 1 ? 1 : 0)];
                if (!(ABL & 8)) acc[g][cc] = mfma16h(fa_, cb_, (s == 0 && j < 6) ? zero16 : acc[g][cc]);
                else if (s == 0 && j < 6) { acc[g][cc] = zero16; acc[g][cc][0] = (float)fa_[0] + (float)cb_[1]; }
                if ((SV & 4) && j + 1 == stride && (s == KS - PF || s == KS - PF + 1) && ptile >= 0) {
                    // the previous tile's stores: behind this tile's last in-tile request, in front of the next tile's head requests, which are
                    // not waited for before the skinning phase is over
                    __builtin_amdgcn_sched_barrier(0);
                    const int gs = s - (KS - PF);
                    if (full) store_group(ptile, gs, outp[gs], false);
                    else store_group(ptile, gs, outp[gs], true);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if ((j + 1) % stride == 0) {
                    const int f = (j + 1) / stride - 1;      // 0..5
                    __builtin_amdgcn_sched_barrier(0);
                    if ((in_tile || head) && !(PD16 && ks_l >= 1 && (f & 1)) && !(ABL & 2)) slot[f] = src[f * 64];
                    if ((SV & 16) && (f & 1) && s * 3 + f / 2 < 32) {      // the previous tile's 32 stores, three per k-step: a steady write stream instead of a burst
                        const int idx = s * 3 + f / 2;
                        store_one(ptile, idx >> 4, idx & 15, outp[idx >> 4], !full);
                    }
                    if (s == ((SV & 4) ? KS - PF - 2 : KS - 3) && f < KSP) sw[f] = sp[f * 64];      // this tile's packed skinning weights: needed right after the contraction (SV & 4: requested in front of the delayed stores)
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#pragma unroll
            for (int g = 0; g < 2; ++g) { fh[g] = fhn[g]; fl[g] = fln[g]; }
        }
        // the head k-steps the loop had no free slot for
#pragma unroll
        for (int j = 0; j < PF; ++j) {
            bool in_loop = false;
#pragma unroll
            for (int s2 = KS - PF; s2 < KS; ++s2) in_loop = in_loop || ((s2 + R - 1) % R == j);
            if (!in_loop && !(ABL & 2)) load_step(ring[j], pn + j * 384, j);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (ABL & 128) { const unsigned long long t = __builtin_amdgcn_s_memtime(); ph_blend += t - ph_t0; ph_t0 = t; }
        // ---------------- skinning on the matrix pipe, K-packed: T_e[body][vertex] = [Ah | Al | Ah | -][body] . [Wh | Wh | Wl | 0][vertex] ----------------
        // Software pipeline over the twelve entries: [LDS operands of e + 1] -> [5-MFMA chain of e into one of two result buffers] -> [fold of
        // e - 1 from the other buffer on the VALU while that chain runs].  out_c = (T[4c] x + T[4c+1] y + T[4c+2] z) us_rot + T[4c+3] us_w.
        // TV = 1 (chains through inline assembly, VGPR results): the fold's reads of T(e - 1) sit behind the FIRST TWO links of chain e
        // (one asm statement, fenced by sched_barriers), i.e. >= 64 cycles after chain e - 1 issued its last MFMA: the 11 wait states an
        // 8-pass MFMA result needs before a VALU read; the last entry's fold, which has no chain in front of it, gets them as an s_nop.
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const _Float16* ar = a_row + g * 32 * ASP;
            half8 a[3], an[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) { a[q] = *reinterpret_cast<const half8*>(ar + 16 * q); an[q] = a[q]; }
            f32x16 T[2], fa = zero16;                         // (two result buffers, alternating)
            f32x16 out[3];
            if (!TV || (ABL & 4)) { T[0] = zero16; T[1] = zero16; }
#pragma unroll
            for (int e = 0; e <= 12; ++e) {
                if (e + 1 < 12 && !(ABL & 32)) {
#pragma unroll
                    for (int q = 0; q < 3; ++q) an[q] = *reinterpret_cast<const half8*>(ar + (e + 1) * WB * ASP + 16 * q);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (ABL & 4) {
                    if (e < 12) T[e & 1][0] += (float)a[0][0] + (float)a[1][1] + (float)a[2][2] + (float)sw[e % KSP][3];
                } else if (TV) {
                    if (e < 12) mfma16h_v01(T[e & 1], a[0], sw[0], a[1], sw[1]);
                    else asm volatile("s_nop 11");
                    __builtin_amdgcn_sched_barrier(0);
                    if (e < 12) {
                        mfma16h_v(T[e & 1], a[2], sw[2]);
                        mfma16h_v(T[e & 1], a[0], sw[3]);      // (k-steps 3, 4 re-use the operands of 0, 1)
                        mfma16h_v(T[e & 1], a[1], sw[4]);
                    }
                } else if (e < 12) {
#pragma unroll
                    for (int ks = 0; ks < KSP; ++ks) T[e & 1] = mfma16h(a[ks % 3], sw[ks], ks == 0 ? zero16 : T[e & 1]);
                }
                if (e > 0) {                                  // fold entry e - 1 while the chain above is in the pipe
                    const int pe = e - 1, c = pe >> 2, q = pe & 3;
                    const f32x16& Tp = T[pe & 1];
                    if (ABL & 16) {
                        if (q == 3) {
#pragma unroll
                            for (int r = 0; r < 16; ++r) out[c][r] = Tp[r];
                            out[c][0] += acc[g][c][0];
                        }
                    } else if (q == 0) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) fa[r] = Tp[r] * acc[g][0][r];
                    } else if (q == 1) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) fa[r] += Tp[r] * acc[g][1][r];
                    } else if (q == 2) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) fa[r] += Tp[r] * acc[g][2][r];
                    } else {
#pragma unroll
                        for (int r = 0; r < 16; ++r) out[c][r] = fa[r] * us_rot + Tp[r] * W_UNSCALE;
                    }
                    asm volatile("" : "+v"(fa));              // (pins the fold of THIS entry here: the vectoriser otherwise gathers all four entries of a coordinate at q == 3)
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < 3; ++q) a[q] = an[q];
            }
            // (lane = vertex, register r = body (r&3) + 8 (r>>2) + 4 h of group g: sixteen 12-byte stores per lane, straight from the registers)
            if (SV & (4 | 16)) {
#pragma unroll
                for (int c = 0; c < 3; ++c) outp[g][c] = out[c];
            } else {
                if (full) store_group(tile, g, out, false);
                else store_group(tile, g, out, true);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (ABL & 128) { const unsigned long long t = __builtin_amdgcn_s_memtime(); (g ? ph_skin1 : ph_skin0) += t - ph_t0; ph_t0 = t; }
        }
        ptile = tile;
    }
    if ((SV & (4 | 16)) && ptile >= 0) {
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            if (full) store_group(ptile, g, outp[g], false);
            else store_group(ptile, g, outp[g], true);
        }
    }
    if (clk && blockIdx.x == 0 && threadIdx.x == 0) {
        atomicAdd(clk, (unsigned long long)__builtin_amdgcn_s_memtime() - clk_c0);
        atomicAdd(clk + 1, (unsigned long long)__builtin_amdgcn_s_memrealtime() - clk_w0);
    }
    if ((ABL & 128) && clk && lane == 0 && (blockIdx.x % 61) == 0) {      // (tools: a sample of the workgroups; clk points at >= 6 counters then)
        atomicAdd(clk + 2, ph_blend); atomicAdd(clk + 3, ph_skin0); atomicAdd(clk + 4, ph_skin1); atomicAdd(clk + 5, (unsigned long long)(round1 - round0));
    }
}

// ------------------------------------------------------------------------------------------
// Output joints 24..89: 21 picked vertices, then the 45 regressed joints = fixed-order sums of their virtual vertices
// (contiguous in the scratch row).  One wave per body: the body's scratch row (<= 3 KB) is read once, coalesced, into LDS; each
// output scalar is then summed by one lane in ascending order -> deterministic, and bit-identical to the former one-thread-per-
// scalar kernel, which re-read the row with 12-byte accesses (220 us at 65 536 bodies; now the 200 MB row read at HBM speed).
constexpr int JW = 4;                      // bodies (waves) per workgroup of the joint kernel
__global__ __launch_bounds__(JW * 64) void smpl_joints_kernel(straps_smpl_model_t m, const float* __restrict__ verts,
                                                              const float* __restrict__ vout, float* __restrict__ joints,
                                                              long long B) {
    extern __shared__ __attribute__((aligned(16))) float jrow[];          // [JW][vrow]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long b = (long long)blockIdx.x * JW + wave;
    const int vrow = (m.n_tiles - NT) * 96;
    float* row = jrow + wave * vrow;
    if (b < B) {
        const float* vb = vout + b * (long long)vrow;
        for (int i = lane * 4; i < vrow; i += 256) *reinterpret_cast<f32x4*>(row + i) = *reinterpret_cast<const f32x4*>(vb + i);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (b >= B) return;
    constexpr int PER = (STRAPS_SMPL_NPICK + STRAPS_SMPL_NEXTRA) * 3;   // 198
    for (int r = lane; r < PER; r += 64) {
        float v;
        if (r < STRAPS_SMPL_NPICK * 3) {
            v = verts[b * (long long)(NV * 3) + m.pick_ids[r / 3] * 3 + (r % 3)];
        } else {
            const int rr = r - STRAPS_SMPL_NPICK * 3;
            const int j = rr / 3, c = rr - j * 3;
            const int e0 = m.vj_ptr[j], e1 = m.vj_ptr[j + 1];
            v = 0.f;
            for (int e = e0; e < e1; ++e) v += row[e * 3 + c];
        }
        joints[b * (STRAPS_SMPL_NJOINTS_OUT * 3) + 72 + r] = v;
    }
}

// rounds (of NW tiles) per block; chunks <= 0 -> auto: big batches take ~8 rounds per block so the F / A staging amortises -- split
// EVENLY (28 rounds -> 4 x 7, not 8 + 8 + 8 + 4: the uneven split cost 8 % at 65 536 bodies) -- small ones 1 round so a 64-body
// step still puts 2 x n_tiles/8 blocks on the chip
inline int resolve_rpc(int rounds, long long batch, int chunks) {
    if (chunks <= 0) {
        if (batch < 1024) return 1;
        chunks = (rounds + 7) / 8;
    }
    if (chunks > rounds) chunks = rounds;
    return (rounds + chunks - 1) / chunks;
}

}  // namespace

// shared with smpl_bwd.hip (which recomputes F and A before back-propagating)
int straps_smpl_launch_pose(const straps_smpl_model_t* model, const float* betas, const float* rotmats, float* F, float* Amat,
                            float* joints, long long batch, hipStream_t st) {
    const unsigned pose_blocks = (unsigned)((batch * 32 + 255) / 256);
    hipLaunchKernelGGL(smpl_pose_kernel, dim3(pose_blocks), dim3(256), 0, st, *model, betas, rotmats, F, Amat, joints, batch);
    STRAPS_CHECK_LAUNCH("smpl_pose_kernel");
    return STRAPS_OK;
}

extern "C" size_t straps_smpl_workspace_bytes(const straps_smpl_model_t* model, long long batch) {
    if (!model || batch <= 0 || model->n_tiles < NT) return 0;
    return (size_t)batch * (size_t)(KP + 288 + (model->n_tiles - NT) * 96) * sizeof(float);
}

extern "C" int straps_smpl_fwd(const straps_smpl_model_t* model, const float* betas, const float* rotmats,
                               float* verts, float* joints, void* workspace, long long batch, int chunks, int mode,
                               void* stream) {
    STRAPS_REQUIRE(model && betas && rotmats && verts && workspace, "straps_smpl_fwd: null pointer");
    STRAPS_REQUIRE(batch > 0, "straps_smpl_fwd: batch must be positive (got %lld)", batch);
    const bool wide_builtin = (mode & STRAPS_SMPL_KERNEL_WIDE_BUILTIN) != 0;
    mode &= ~STRAPS_SMPL_KERNEL_WIDE_BUILTIN;
    const int kflag = mode & (STRAPS_SMPL_KERNEL_WIDE | STRAPS_SMPL_KERNEL_NARROW);
    mode &= ~(STRAPS_SMPL_KERNEL_WIDE | STRAPS_SMPL_KERNEL_NARROW);
    STRAPS_REQUIRE(!wide_builtin || (kflag == STRAPS_SMPL_KERNEL_WIDE && mode == STRAPS_SMPL_SPLIT_F16_LBS),
                   "straps_smpl_fwd: STRAPS_SMPL_KERNEL_WIDE_BUILTIN goes with STRAPS_SMPL_KERNEL_WIDE and mode STRAPS_SMPL_SPLIT_F16_LBS");
    STRAPS_REQUIRE(kflag != (STRAPS_SMPL_KERNEL_WIDE | STRAPS_SMPL_KERNEL_NARROW), "straps_smpl_fwd: STRAPS_SMPL_KERNEL_WIDE and _NARROW exclude each other");
    STRAPS_REQUIRE(mode == STRAPS_SMPL_EXACT_F32 || mode == STRAPS_SMPL_SPLIT_F16 || mode == STRAPS_SMPL_SPLIT_F16_LBS || mode == STRAPS_SMPL_SPLIT_F16_LBS_PD16 ||
                   mode == STRAPS_SMPL_SPLIT_F16_LBS_P16,
                   "straps_smpl_fwd: unknown mode %d", mode);
    const int pd16 = mode == STRAPS_SMPL_SPLIT_F16_LBS_PD16 ? 1 : mode == STRAPS_SMPL_SPLIT_F16_LBS_P16 ? 2 : 0;
    if (pd16) mode = STRAPS_SMPL_SPLIT_F16_LBS;
    STRAPS_REQUIRE(mode != STRAPS_SMPL_SPLIT_F16_LBS || model->skin_frag_p, "straps_smpl_fwd: mode STRAPS_SMPL_SPLIT_F16_LBS needs skin_frag_p in the model");
    STRAPS_REQUIRE(mode == STRAPS_SMPL_EXACT_F32 || (model->blend_frag_h && model->blend_h_unscale > 0.f),
                   "straps_smpl_fwd: split-precision mode needs blend_frag_h / blend_h_unscale in the model");
    STRAPS_REQUIRE(model->skin_k >= 1 && model->skin_k <= 24, "straps_smpl_fwd: skin_k %d out of range", model->skin_k);
    STRAPS_REQUIRE(model->n_tiles >= NT && model->n_tiles % NW == 0 && model->vj_ptr,
                   "straps_smpl_fwd: n_tiles %d must be a multiple of %d >= %d with the virtual-vertex table set", model->n_tiles, NW, NT);
    hipStream_t st = (hipStream_t)stream;
    const int rounds = (joints ? model->n_tiles : NT) / NW;   // vertices only: the virtual (joint) tiles are skipped
    const int rpc = resolve_rpc(rounds, batch, chunks);
    const int nch = (rounds + rpc - 1) / rpc;
    float* F = (float*)workspace;
    float* Amat = F + batch * KP;
    float* vout = Amat + batch * 288;
    int rc = straps_smpl_launch_pose(model, betas, rotmats, F, Amat, joints, batch, st);
    if (rc != STRAPS_OK) return rc;
    const int split = mode == STRAPS_SMPL_SPLIT_F16 ? 1 : mode == STRAPS_SMPL_SPLIT_F16_LBS ? 2 : 0;
    // measurement knobs of the TOOLS build (tools/smpl_ablate.sh; the product library never reads the environment): STRAPS_SMPL_PF = depth of
    // the fragment ring (default 3 for the blend-split kernel, 2 for the matrix-pipe-skinning kernel), STRAPS_SMPL_ABLATE (compile-time
    // ablations of the latter), STRAPS_SMPL_RPC (rounds per workgroup)
    static const int pf_env = STRAPS_TOOL_ENV_INT("STRAPS_SMPL_PF", 0), rpc_env = STRAPS_TOOL_ENV_INT("STRAPS_SMPL_RPC", 0);
    const int pf = pf_env >= 1 && pf_env <= 4 ? pf_env : (split == 2 ? 2 : 3);
    constexpr int nwv = 8;              // (12 / 16 waves per workgroup need <= 168 / 128 VGPRs: the kernel spills and runs 3.5x slower)
    auto h_kernel = pf == 1 ? smpl_verts_h_kernel<1> : pf == 2 ? smpl_verts_h_kernel<2> : pf == 4 ? smpl_verts_h_kernel<4> : smpl_verts_h_kernel<3>;
    auto hh_kernel = pd16 == 1 ? (pf == 1 ? smpl_verts_hh_kernel<8, 1, 0, 1> : smpl_verts_hh_kernel<8, 2, 0, 1>)
                   : pd16 == 2 ? (pf == 1 ? smpl_verts_hh_kernel<8, 1, 0, 2> : smpl_verts_hh_kernel<8, 2, 0, 2>)
                   : pf == 1 ? smpl_verts_hh_kernel<8, 1, 0> : smpl_verts_hh_kernel<8, 2, 0>;      // (a deeper ring does not fit: PF = 2 uses 255 of the 256 registers two waves per SIMD leave a wave)
#ifdef STRAPS_TOOLS
    {   // ablation instantiations (wrong results by design; tools/smpl_ablate.sh) -- tools build only
        static const int ablate = STRAPS_TOOL_ENV_INT("STRAPS_SMPL_ABLATE", 0);
        if (!pd16 && ablate)
            hh_kernel = ablate == 1 ? smpl_verts_hh_kernel<8, 1, 1> : ablate == 2 ? smpl_verts_hh_kernel<8, 1, 2> : ablate == 3 ? smpl_verts_hh_kernel<8, 1, 3>
                      : ablate == 7 ? smpl_verts_hh_kernel<8, 1, 7> : ablate == 15 ? smpl_verts_hh_kernel<8, 1, 15> : hh_kernel;
    }
#endif
    const size_t lds = split == 2 ? (size_t)(2 * BT * FSH + 12 * BT * ASP) * sizeof(_Float16)
                                  : (size_t)(BT * (split ? FSH : FS) + BT * AS + NW * BT * HS + NW * 256) * sizeof(float);
    static unsigned long long lds_raised[5] = {0, 0, 0, 0, 0};          // per kernel variant: bit mask of the devices done
    {
        hipError_t e = straps_raise_dynamic_lds(split == 2 ? (const void*)hh_kernel : split ? (const void*)h_kernel : (const void*)smpl_verts_kernel,
                                                lds, lds_raised[split + pd16]);
        if (e != hipSuccess) { straps_set_error("smpl_verts_kernel: cannot raise dynamic LDS to %zu: %s", lds, hipGetErrorString(e)); return STRAPS_EHIP; }
    }
    const long long btiles = (batch + BT - 1) / BT;
    if (btiles * nch > 0x7fffffffLL || btiles > 0x3fffffLL) {
        straps_set_error("straps_smpl_fwd: batch %lld exceeds one launch; split it", batch);
        return STRAPS_EUNSUPPORTED;
    }
    // 64-body workgroups (smpl_verts_w_kernel) from 2048 bodies on: below that the grid would be a few dozen 256-thread workgroups
    const bool wide = split == 2 && model->skin_frag_p && (kflag == STRAPS_SMPL_KERNEL_WIDE || (kflag == 0 && batch >= 2048));
    STRAPS_REQUIRE(kflag != STRAPS_SMPL_KERNEL_WIDE || wide, "straps_smpl_fwd: STRAPS_SMPL_KERNEL_WIDE needs a STRAPS_SMPL_SPLIT_F16_LBS* mode and skin_frag_p in the model");
    if (wide) {
        const int ntiles = joints ? model->n_tiles : NT;
        const int rounds_w = (ntiles + NWW - 1) / NWW;
        const long long bgroups = (batch + WB - 1) / WB;
        // chunks of the tile rounds per body group: one (the staging of 64 bodies amortised over every tile) once the body groups alone fill
        // the chip, else enough to put ~one workgroup on every CU; split evenly
        int nchw = chunks > 0 ? chunks : (int)((256 + bgroups - 1) / bgroups);
        if (nchw > rounds_w) nchw = rounds_w;
        if (nchw < 1) nchw = 1;
        const int rpcw = rpc_env > 0 ? (rpc_env > rounds_w ? rounds_w : rpc_env) : (rounds_w + nchw - 1) / nchw;
        nchw = (rounds_w + rpcw - 1) / rpcw;
        if (bgroups * nchw > 0x7fffffffLL) {
            straps_set_error("straps_smpl_fwd: batch %lld exceeds one launch; split it", batch);
            return STRAPS_EUNSUPPORTED;
        }
        // product form (tools/smpl_w_ab.py, profiles/r04_smpl_w_ab.txt): prefetch distance 3, skinning chains with VGPR results, the previous tile's
        // stores spread over the blend phase as non-temporal buffer stores.  The tools build selects the A/B instantiations:
        // STRAPS_SMPL_WVAR = 10 * PF + TV (plain stores at the tile end), STRAPS_SMPL_WSV (store forms), STRAPS_SMPL_WABL (ablations, wrong results)
        constexpr int B_NT = (100 + 2) << 8, SV_PRODUCT = 16 | B_NT;
        auto w_kernel = pd16 == 1 ? smpl_verts_w_kernel<3, 1, 1, 0, SV_PRODUCT> : pd16 == 2 ? smpl_verts_w_kernel<3, 2, 1, 0, SV_PRODUCT>
                      : smpl_verts_w_kernel<3, 0, 1, 0, SV_PRODUCT>;
        int wslot = pd16;
        if (wide_builtin) {      // (the builtin-MFMA reference instantiation: TV = 0, otherwise the product form)
            w_kernel = smpl_verts_w_kernel<3, 0, 0, 0, SV_PRODUCT>;
            wslot = 6;
        }
#ifdef STRAPS_TOOLS
        static const int wvar = STRAPS_TOOL_ENV_INT("STRAPS_SMPL_WVAR", 0);
        if (!pd16 && wvar) {
            w_kernel = wvar == 31 ? smpl_verts_w_kernel<3, 0, 1> : wvar == 30 ? smpl_verts_w_kernel<3, 0, 0> : wvar == 21 ? smpl_verts_w_kernel<2, 0, 1>
                     : wvar == 41 ? smpl_verts_w_kernel<4, 0, 1> : smpl_verts_w_kernel<4, 0, 0>;
            wslot = 3;
        }
        static const int wsv = STRAPS_TOOL_ENV_INT("STRAPS_SMPL_WSV", 0);      // store forms: SV bits | form << 8
        if (!pd16 && wsv) {
            constexpr int NT1 = 1 << 8, B_SYS = (100 + 19) << 8, B_PLAIN = 100 << 8;
            w_kernel = wsv == 1 ? smpl_verts_w_kernel<3, 0, 1, 0, NT1> : wsv == 2 ? smpl_verts_w_kernel<3, 0, 1, 0, 16> : wsv == 3 ? smpl_verts_w_kernel<3, 0, 1, 0, 16 | NT1>
                     : wsv == 4 ? smpl_verts_w_kernel<3, 0, 1, 0, B_NT> : wsv == 5 ? smpl_verts_w_kernel<3, 0, 1, 0, B_SYS> : wsv == 7 ? smpl_verts_w_kernel<3, 0, 1, 0, B_PLAIN>
                     : wsv == 10 ? smpl_verts_w_kernel<3, 0, 1, 0, 4 | NT1> : wsv == 12 ? smpl_verts_w_kernel<4, 0, 1, 0, B_NT> : wsv == 13 ? smpl_verts_w_kernel<4, 0, 1, 0, 16 | B_NT>
                     : wsv == 16 ? smpl_verts_w_kernel<5, 0, 1, 0, B_NT> : smpl_verts_w_kernel<2, 0, 1, 0, 16 | B_NT>;
            wslot = 4;
        }
        static const int wabl = STRAPS_TOOL_ENV_INT("STRAPS_SMPL_WABL", 0);
        if (!pd16 && wabl) {
            w_kernel = wabl == 1 ? smpl_verts_w_kernel<3, 0, 1, 1> : wabl == 2 ? smpl_verts_w_kernel<3, 0, 1, 2> : wabl == 3 ? smpl_verts_w_kernel<3, 0, 1, 3>
                     : wabl == 4 ? smpl_verts_w_kernel<3, 0, 1, 4> : wabl == 8 ? smpl_verts_w_kernel<3, 0, 1, 8> : wabl == 12 ? smpl_verts_w_kernel<3, 0, 1, 12>
                     : wabl == 15 ? smpl_verts_w_kernel<3, 0, 1, 15> : wabl == 16 ? smpl_verts_w_kernel<3, 0, 1, 16> : wabl == 32 ? smpl_verts_w_kernel<3, 0, 1, 32>
                     : wabl == 64 ? smpl_verts_w_kernel<3, 0, 1, 64> : wabl == 66 ? smpl_verts_w_kernel<3, 0, 1, 66> : wabl == 128 ? smpl_verts_w_kernel<3, 0, 1, 128, SV_PRODUCT>
                     : wabl == 129 ? smpl_verts_w_kernel<3, 0, 1, 129, SV_PRODUCT> : wabl == 130 ? smpl_verts_w_kernel<3, 0, 1, 130, SV_PRODUCT>
                     : wabl == 131 ? smpl_verts_w_kernel<3, 0, 1, 131, SV_PRODUCT> : wabl == 151 ? smpl_verts_w_kernel<3, 0, 1, 151, SV_PRODUCT>
                     : wabl == 183 ? smpl_verts_w_kernel<3, 0, 1, 183, SV_PRODUCT> : wabl == 135 ? smpl_verts_w_kernel<3, 0, 1, 135, SV_PRODUCT>
                     : smpl_verts_w_kernel<3, 0, 1, 63>;
            wslot = 5;      // (one variant per process: the switches are read once)
        }
#endif
        const size_t ldsw = (size_t)(2 * WB * FSH + 12 * WB * ASP) * sizeof(_Float16);
        static unsigned long long lds_raised_w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        hipError_t e = straps_raise_dynamic_lds((const void*)w_kernel, ldsw, lds_raised_w[wslot]);
        if (e != hipSuccess) { straps_set_error("smpl_verts_w_kernel: cannot raise dynamic LDS to %zu: %s", ldsw, hipGetErrorString(e)); return STRAPS_EHIP; }
        hipLaunchKernelGGL(w_kernel, dim3((unsigned)(bgroups * nchw)), dim3(NWW * 64), ldsw, st, *model, F, Amat, verts, joints ? vout : nullptr,
                           batch, (int)bgroups, ntiles, rounds_w, rpcw, straps_clk_acc_current());
    } else if (split == 2) {
        // own round structure: nwv tiles per round, ragged last round
        const int ntiles = joints ? model->n_tiles : NT;
        const int rounds2 = (ntiles + nwv - 1) / nwv;
        const int rpc2 = rpc_env > 0 ? (rpc_env > rounds2 ? rounds2 : rpc_env) : resolve_rpc(rounds2, batch, chunks);
        const int nch2 = (rounds2 + rpc2 - 1) / rpc2;
        hipLaunchKernelGGL(hh_kernel, dim3((unsigned)(btiles * nch2)), dim3(nwv * 64), lds, st, *model, F, Amat, verts, joints ? vout : nullptr,
                           batch, (int)btiles, ntiles, rounds2, rpc2, straps_clk_acc_current());
    } else if (split)
        hipLaunchKernelGGL(h_kernel, dim3((unsigned)(btiles * nch)), dim3(NW * 64), lds, st, *model, F, Amat, verts,
                           joints ? vout : nullptr, batch, (int)btiles, rounds, rpc);
    else
        hipLaunchKernelGGL(smpl_verts_kernel, dim3((unsigned)(btiles * nch)), dim3(NW * 64), lds, st, *model, F, Amat, verts,
                           joints ? vout : nullptr, batch, (int)btiles, rounds, rpc);
    STRAPS_CHECK_LAUNCH("smpl_verts_kernel");
    if (joints) {
        const size_t jl = (size_t)JW * (model->n_tiles - NT) * 96 * sizeof(float);
        hipLaunchKernelGGL(smpl_joints_kernel, dim3((unsigned)((batch + JW - 1) / JW)), dim3(JW * 64), jl, st, *model, verts, vout, joints, batch);
        STRAPS_CHECK_LAUNCH("smpl_joints_kernel");
    }
    return STRAPS_OK;
}
