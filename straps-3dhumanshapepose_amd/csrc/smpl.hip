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
__global__ __launch_bounds__(NW * 64) void smpl_verts_kernel(straps_smpl_model_t m, const float* __restrict__ F,
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
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
constexpr int KS = KP / 16;               // 14 k-steps of 16
constexpr int FSH = 232;                  // halves per LDS feature row: 464 bytes = 4 * 29 dwords -> conflict-free b128 reads
constexpr float F_SCALE = 64.0f;          // 2^6

__device__ __forceinline__ f32x16 mfma16h(half8 a, half8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }

template <int PF>
__global__ __launch_bounds__(NW * 64) void smpl_verts_h_kernel(straps_smpl_model_t m, const float* __restrict__ F,
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
            const float x = v[e] * F_SCALE;
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
// Output joints 24..89: 21 picked vertices, then the 45 regressed joints = fixed-order sums of their
// virtual vertices (contiguous in the scratch row).  One thread per output scalar -> deterministic.
__global__ __launch_bounds__(256) void smpl_joints_kernel(straps_smpl_model_t m, const float* __restrict__ verts,
                                                          const float* __restrict__ vout, float* __restrict__ joints,
                                                          long long B) {
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    constexpr int PER = (STRAPS_SMPL_NPICK + STRAPS_SMPL_NEXTRA) * 3;   // 198
    if (gid >= B * PER) return;
    const long long b = gid / PER;
    const int r = (int)(gid - b * PER);
    float v;
    if (r < STRAPS_SMPL_NPICK * 3) {
        v = verts[b * (long long)(NV * 3) + m.pick_ids[r / 3] * 3 + (r % 3)];
    } else {
        const int rr = r - STRAPS_SMPL_NPICK * 3;
        const int j = rr / 3, c = rr - j * 3;
        const float* vb = vout + b * (long long)((m.n_tiles - NT) * 96);
        const int e0 = m.vj_ptr[j], e1 = m.vj_ptr[j + 1];
        v = 0.f;
        for (int e = e0; e < e1; ++e) v += vb[e * 3 + c];
    }
    joints[b * (STRAPS_SMPL_NJOINTS_OUT * 3) + 72 + r] = v;
}

// rounds (of NW tiles) per block; chunks <= 0 -> auto: big batches take 8 rounds per block so the F / A staging
// amortises, small ones 1 round so a 64-body step still puts 2 x n_tiles/8 blocks on the chip
inline int resolve_rpc(int rounds, long long batch, int chunks) {
    if (chunks <= 0) return (batch >= 1024) ? 8 : 1;
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
    STRAPS_REQUIRE(mode == STRAPS_SMPL_EXACT_F32 || mode == STRAPS_SMPL_SPLIT_F16, "straps_smpl_fwd: unknown mode %d", mode);
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
    const bool split = mode == STRAPS_SMPL_SPLIT_F16;
    // prefetch depth of the split kernel's fragment ring (tuning knob, STRAPS_SMPL_PF = 1..4; default 3)
    static int pf = 0;
    if (!pf) { const char* e = getenv("STRAPS_SMPL_PF"); pf = e ? atoi(e) : 3; if (pf < 1 || pf > 4) pf = 3; }
    auto h_kernel = pf == 1 ? smpl_verts_h_kernel<1> : pf == 2 ? smpl_verts_h_kernel<2> : pf == 4 ? smpl_verts_h_kernel<4> : smpl_verts_h_kernel<3>;
    const size_t lds = (size_t)(BT * (split ? FSH : FS) + BT * AS + NW * BT * HS + NW * 256) * sizeof(float);
    static bool attr_set[2] = {false, false};
    if (!attr_set[split]) {
        hipError_t e = hipFuncSetAttribute(split ? (const void*)h_kernel : (const void*)smpl_verts_kernel,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { straps_set_error("smpl_verts_kernel: cannot raise dynamic LDS to %zu: %s", lds, hipGetErrorString(e)); return STRAPS_EHIP; }
        attr_set[split] = true;
    }
    const long long btiles = (batch + BT - 1) / BT;
    if (btiles * nch > 0x7fffffffLL || btiles > 0x3fffffLL) {
        straps_set_error("straps_smpl_fwd: batch %lld exceeds one launch; split it", batch);
        return STRAPS_EUNSUPPORTED;
    }
    if (split)
        hipLaunchKernelGGL(h_kernel, dim3((unsigned)(btiles * nch)), dim3(NW * 64), lds, st, *model, F, Amat, verts,
                           joints ? vout : nullptr, batch, (int)btiles, rounds, rpc);
    else
        hipLaunchKernelGGL(smpl_verts_kernel, dim3((unsigned)(btiles * nch)), dim3(NW * 64), lds, st, *model, F, Amat, verts,
                           joints ? vout : nullptr, batch, (int)btiles, rounds, rpc);
    STRAPS_CHECK_LAUNCH("smpl_verts_kernel");
    if (joints) {
        const long long n = batch * (STRAPS_SMPL_NPICK + STRAPS_SMPL_NEXTRA) * 3;
        hipLaunchKernelGGL(smpl_joints_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, *model, verts, vout, joints, batch);
        STRAPS_CHECK_LAUNCH("smpl_joints_kernel");
    }
    return STRAPS_OK;
}
