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
//                           through LDS so the 12-byte vertices leave as coalesced rows; the
//                           45 regressed joints are accumulated from the staged tile (sparse).
//   3. smpl_joints_kernel : picked vertices + fixed-order sum of the per-chunk joint partials.
// Everything is deterministic (no atomics).
#include "common.h"

namespace {

constexpr int KP = STRAPS_SMPL_KP;        // 224
constexpr int KG = KP / 8;                // 28 k-groups of 8
constexpr int NT = STRAPS_SMPL_TILES;     // 216 vertex tiles
constexpr int NV = STRAPS_SMPL_V;         // 6890
constexpr int NROUNDS = NT / 4;           // 54 rounds of 4 tiles (one per wave)
constexpr int BT = 32;                    // bodies per block
constexpr int FS = 228;                   // LDS row strides (floats): 4*odd -> conflict-free b128
constexpr int AS = 292;
constexpr int SS = 97;                    // stage row stride (odd -> conflict-free b32)
constexpr int JS = 180;                   // jacc row stride: 45 joints x vec4

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
__global__ __launch_bounds__(256, 1) void smpl_verts_kernel(straps_smpl_model_t m, const float* __restrict__ F,
                                                            const float* __restrict__ Amat,
                                                            float* __restrict__ verts, float* __restrict__ partial,
                                                            long long B, int rounds_per_chunk) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Fs = smem;                      // [32][FS]
    float* As_ = Fs + BT * FS;             // [32][AS]
    float* jacc = As_ + BT * AS;           // [32][JS]
    float* stage = jacc + BT * JS;         // [4][32][SS]
    float* skin_lds = stage + 4 * BT * SS; // [4 waves][256]: per-tile skinning weights + joint offsets

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int h = lane >> 5;
    const int bl = lane & 31;              // body within the block tile
    const int chunk = blockIdx.x;
    const long long b0 = (long long)blockIdx.y * BT;
    const int nb = (int)((B - b0) < BT ? (B - b0) : BT);

    // ---- stage the block's feature rows and joint transforms, zero the joint accumulators ----
    for (int i = tid; i < BT * (KP / 4); i += 256) {
        const int b = i / (KP / 4), q = i % (KP / 4);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (b < nb) v = *reinterpret_cast<const f32x4*>(F + (b0 + b) * KP + q * 4);
        *reinterpret_cast<f32x4*>(Fs + b * FS + q * 4) = v;
    }
    for (int i = tid; i < BT * 72; i += 256) {
        const int b = i / 72, q = i % 72;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (b < nb) v = *reinterpret_cast<const f32x4*>(Amat + (b0 + b) * 288 + q * 4);
        *reinterpret_cast<f32x4*>(As_ + b * AS + q * 4) = v;
    }
    for (int i = tid; i < BT * JS; i += 256) jacc[i] = 0.f;
    __syncthreads();

    const int round0 = chunk * rounds_per_chunk;
    const int round1 = min(round0 + rounds_per_chunk, NROUNDS);
    const f32x4* __restrict__ blend = reinterpret_cast<const f32x4*>(m.blend_frag);
    float* mystage = stage + wave * BT * SS;
    const int KW = m.skin_k;

    for (int rd = round0; rd < round1; ++rd) {
        const int tile = rd * 4 + wave;
        // skinning weights / joint offsets of this tile's 32 vertices -> LDS now, so their L2 latency hides under the
        // MFMA loop instead of stalling the per-vertex loop (fast path: 4 weights per vertex, the real model's layout)
        float* skw = skin_lds + wave * 256;                    // [32][4] weights, then [32][4] joint offsets (int bits)
        if (KW == 4) {
            const int e2 = lane * 2;
            const f32x2 w2 = *reinterpret_cast<const f32x2*>(m.skin_w + tile * 128 + e2);
            const int2 j2 = *reinterpret_cast<const int2*>(m.skin_j + tile * 128 + e2);
            skw[e2] = w2[0]; skw[e2 + 1] = w2[1];
            reinterpret_cast<int*>(skw)[128 + e2] = j2.x * 12;
            reinterpret_cast<int*>(skw)[128 + e2 + 1] = j2.y * 12;
        }
        // ---------------- blendshape contraction on the fp32 MFMA ----------------
        {
            f32x16 ax, ay, az;
#pragma unroll
            for (int r = 0; r < 16; ++r) { ax[r] = 0.f; ay[r] = 0.f; az[r] = 0.f; }
            const f32x4* px = blend + ((long long)(tile * 3 + 0) * KG) * 64 + lane;
            const f32x4* py = px + KG * 64;
            const f32x4* pz = py + KG * 64;
            const float* frow = Fs + bl * FS + 4 * h;
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
            // ---------------- linear blend skinning, lane = body, reg = vertex ----------------
            const float* Ab = As_ + bl * AS;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int vrow = (r & 3) + 8 * (r >> 2) + 4 * h;
                const int v = tile * 32 + vrow;
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
                float* so = mystage + bl * SS + vrow * 3;
                so[0] = t0[0] * x + t0[1] * y + t0[2] * z + t0[3];
                so[1] = t1[0] * x + t1[1] * y + t1[2] * z + t1[3];
                so[2] = t2[0] * x + t2[1] * y + t2[2] * z + t2[3];
            }
        }
        __syncthreads();
        // ---------------- coalesced row store of this wave's staged tile ----------------
        {
            const int ncol = min(96, (NV - tile * 32) * 3);
            for (int i = lane; i < BT * 96; i += 64) {
                const int b = i / 96, c = i - b * 96;
                if (b < nb && c < ncol) verts[(b0 + b) * (long long)(NV * 3) + tile * 96 + c] = mystage[b * SS + c];
            }
        }
        // ---------------- sparse joint regression: this wave owns joints == wave (mod 4) ----------------
        if (h == 0) {
            const int q = rd * 4 + wave;
            const int e0 = m.jr_ptr[q], e1 = m.jr_ptr[q + 1];
            for (int e = e0; e < e1; ++e) {
                const int code = m.jr_code[e];
                const float w = m.jr_w[e];
                const float* sv = stage + ((code >> 16) * BT + bl) * SS + ((code >> 8) & 255) * 3;
                f32x4* acc = reinterpret_cast<f32x4*>(jacc + bl * JS + (code & 255) * 4);
                f32x4 a = *acc;
                a[0] = fmaf(w, sv[0], a[0]); a[1] = fmaf(w, sv[1], a[1]); a[2] = fmaf(w, sv[2], a[2]);
                *acc = a;
            }
        }
        __syncthreads();
    }
    if (partial) {
        for (int i = tid; i < BT * STRAPS_SMPL_NEXTRA * 3; i += 256) {
            const int b = i / (STRAPS_SMPL_NEXTRA * 3), r = i - b * (STRAPS_SMPL_NEXTRA * 3);
            if (b < nb)
                partial[((long long)chunk * B + b0 + b) * (STRAPS_SMPL_NEXTRA * 3) + r] = jacc[b * JS + (r / 3) * 4 + (r % 3)];
        }
    }
}

// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void smpl_joints_kernel(straps_smpl_model_t m, const float* __restrict__ verts,
                                                          const float* __restrict__ partial, float* __restrict__ joints,
                                                          long long B, int chunks) {
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
        v = 0.f;
        for (int c = 0; c < chunks; ++c) v += partial[((long long)c * B + b) * (STRAPS_SMPL_NEXTRA * 3) + rr];
    }
    joints[b * (STRAPS_SMPL_NJOINTS_OUT * 3) + 72 + r] = v;
}

inline int resolve_rpc(long long batch, int chunks) {
    if (chunks <= 0) chunks = (batch >= 1024) ? 8 : 54;
    if (chunks > NROUNDS) chunks = NROUNDS;
    return (NROUNDS + chunks - 1) / chunks;
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

extern "C" size_t straps_smpl_workspace_bytes(long long batch, int chunks) {
    const int rpc = resolve_rpc(batch, chunks);
    const int nch = (NROUNDS + rpc - 1) / rpc;
    return (size_t)batch * (size_t)(KP + 288 + nch * STRAPS_SMPL_NEXTRA * 3) * sizeof(float);
}

extern "C" int straps_smpl_fwd(const straps_smpl_model_t* model, const float* betas, const float* rotmats,
                               float* verts, float* joints, void* workspace, long long batch, int chunks,
                               void* stream) {
    STRAPS_REQUIRE(model && betas && rotmats && verts && workspace, "straps_smpl_fwd: null pointer");
    STRAPS_REQUIRE(batch > 0, "straps_smpl_fwd: batch must be positive (got %lld)", batch);
    STRAPS_REQUIRE(model->skin_k >= 1 && model->skin_k <= 24, "straps_smpl_fwd: skin_k %d out of range", model->skin_k);
    hipStream_t st = (hipStream_t)stream;
    const int rpc = resolve_rpc(batch, chunks);
    const int nch = (NROUNDS + rpc - 1) / rpc;
    float* F = (float*)workspace;
    float* Amat = F + batch * KP;
    float* partial = Amat + batch * 288;
    const unsigned pose_blocks = (unsigned)((batch * 32 + 255) / 256);
    hipLaunchKernelGGL(smpl_pose_kernel, dim3(pose_blocks), dim3(256), 0, st, *model, betas, rotmats, F, Amat, joints, batch);
    STRAPS_CHECK_LAUNCH("smpl_pose_kernel");
    const size_t lds = (size_t)(BT * FS + BT * AS + BT * JS + 4 * BT * SS + 4 * 256) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)smpl_verts_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { straps_set_error("smpl_verts_kernel: cannot raise dynamic LDS to %zu: %s", lds, hipGetErrorString(e)); return STRAPS_EHIP; }
        attr_set = true;
    }
    const long long btiles = (batch + BT - 1) / BT;
    if (btiles > 65535) {
        straps_set_error("straps_smpl_fwd: batch %lld exceeds one launch (max %d bodies); split it", batch, 65535 * BT);
        return STRAPS_EUNSUPPORTED;
    }
    hipLaunchKernelGGL(smpl_verts_kernel, dim3(nch, (unsigned)btiles), dim3(256), lds, st, *model, F, Amat, verts,
                       joints ? partial : nullptr, batch, rpc);
    STRAPS_CHECK_LAUNCH("smpl_verts_kernel");
    if (joints) {
        const long long n = batch * (STRAPS_SMPL_NPICK + STRAPS_SMPL_NEXTRA) * 3;
        hipLaunchKernelGGL(smpl_joints_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, *model, verts, partial, joints, batch, nch);
        STRAPS_CHECK_LAUNCH("smpl_joints_kernel");
    }
    return STRAPS_OK;
}
