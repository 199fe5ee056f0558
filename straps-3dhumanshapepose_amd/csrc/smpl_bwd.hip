// smpl_bwd.hip -- gradient of the SMPL forward w.r.t. (betas, rotmats) given dL/dvertices, dL/djoints
// (what autograd does through smplx.lbs for pred_smpl_output, train/train_synthetic_otf_rendering.py:196-232).
//
//   smpl_verts_bwd_kernel : same tiling as the forward (32 bodies x vertex chunk per workgroup, one 32-vertex tile
//       per wave per round).  Per tile: (1) recompute v_posed with the K=218 MFMA contraction, (2) gather the vertex
//       gradient (dverts + sparse J-regressor / picked-vertex contributions of djoints) through an LDS-staged tile,
//       (3) per-vertex skinning backward on the VALU: g_vposed = T_R^T g; v_posed is staged next to g and, after a
//       barrier, dA_j += w_j * g (x) [v_posed;1] is accumulated BY JOINT OWNER: lane (body, h) of wave w owns joints
//       w + 4*(2i + h), i = 0..2, walks the round's (tile, vertex, weight) entries of those joints in a fixed order and
//       keeps the 3 x 12 sums in registers -- no atomics, bit-reproducible, (4) second MFMA contraction dF[b][k] += sum_{v,c} D[k][v][c] * g_vposed[b][v][c] with the transposed
//       blend fragments; per-chunk partials of dF and dA are written out.
//   smpl_pose_bwd_kernel  : lane = (body, joint): sums the chunk partials, back-propagates through rest-pose removal
//       and the kinematic chain (children -> parents by depth with wave shuffles), the joint regression and the pose
//       feature, and emits dbetas [B,10] and drotmats [B,24,3,3].
#include "common.h"

namespace {

constexpr int KP = STRAPS_SMPL_KP, KG = KP / 8, NT = STRAPS_SMPL_TILES, NV = STRAPS_SMPL_V, NROUNDS = NT / 4;
constexpr int BT = 32, AS = 292, SS = 193;        // SS: stage row = 32 vertices x (g[3], v_posed[3]) + 1
constexpr int NJS = STRAPS_SMPL_NPICK + STRAPS_SMPL_NEXTRA;   // 66 joint-gradient sources (joints 24..89)

__global__ __launch_bounds__(256, 1) STRAPS_NO_PACKED_FP32 void smpl_verts_bwd_kernel(straps_smpl_model_t m, const float* __restrict__ F,
                                                                const float* __restrict__ Amat, const float* __restrict__ dverts,
                                                                const float* __restrict__ djoints, float* __restrict__ dFp,
                                                                float* __restrict__ dAp, long long B, int rounds_per_chunk) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As_ = smem;                     // [32][AS]
    float* stage = As_ + BT * AS;          // [4][32][SS]   (the F rows are read straight from global / L1: 16 B per lane per step)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, bl = lane & 31;
    const int chunk = blockIdx.x;
    const long long b0 = (long long)blockIdx.y * BT;
    const int nb = (int)((B - b0) < BT ? (B - b0) : BT);

    for (int i = tid; i < BT * 72; i += 256) {
        const int b = i / 72, q = i % 72;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (b < nb) v = *reinterpret_cast<const f32x4*>(Amat + (b0 + b) * 288 + q * 4);
        *reinterpret_cast<f32x4*>(As_ + b * AS + q * 4) = v;
    }
    __syncthreads();

    const int round0 = chunk * rounds_per_chunk;
    const int round1 = min(round0 + rounds_per_chunk, NROUNDS);
    const f32x4* __restrict__ blend = reinterpret_cast<const f32x4*>(m.blend_frag);
    const f32x4* __restrict__ blend_t = reinterpret_cast<const f32x4*>(m.blend_frag_t);
    float* mystage = stage + wave * BT * SS;
    const int KW = m.skin_k;
    const bool vbody = bl < nb;

    float dacc[3][12];                     // dA of this lane's three joints
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int e = 0; e < 12; ++e) dacc[i][e] = 0.f;
    f32x16 accF[7];
#pragma unroll
    for (int f = 0; f < 7; ++f)
#pragma unroll
        for (int q = 0; q < 16; ++q) accF[f][q] = 0.f;

    for (int rd = round0; rd < round1; ++rd) {
        const int tile = rd * 4 + wave;
        // ---- (1) recompute v_posed for this tile ----
        f32x16 ax, ay, az;
#pragma unroll
        for (int r = 0; r < 16; ++r) { ax[r] = 0.f; ay[r] = 0.f; az[r] = 0.f; }
        {
            const f32x4* px = blend + ((long long)(tile * 3 + 0) * KG) * 64 + lane;
            const f32x4* py = px + KG * 64;
            const f32x4* pz = py + KG * 64;
            const float* frow = F + (b0 + (vbody ? bl : 0)) * KP + 4 * h;
            f32x4 cx0 = px[0], cy0 = py[0], cz0 = pz[0];
            f32x4 fn = *reinterpret_cast<const f32x4*>(frow);
#pragma unroll 2
            for (int g = 0; g < KG; ++g) {
                f32x4 nx0 = cx0, ny0 = cy0, nz0 = cz0;
                const f32x4 f0 = vbody ? fn : f32x4{0.f, 0.f, 0.f, 0.f};
                if (g + 1 < KG) {
                    nx0 = px[(g + 1) * 64]; ny0 = py[(g + 1) * 64]; nz0 = pz[(g + 1) * 64];
                    fn = *reinterpret_cast<const f32x4*>(frow + 8 * (g + 1));
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    ax = mfma32(cx0[e], f0[e], ax);
                    ay = mfma32(cy0[e], f0[e], ay);
                    az = mfma32(cz0[e], f0[e], az);
                }
                cx0 = nx0; cy0 = ny0; cz0 = nz0;
            }
        }
        // ---- (2) stage dL/dverts of this tile (coalesced rows), then add the joint-gradient contributions ----
        {
            const int ncol = min(96, (NV - tile * 32) * 3);
            for (int i = lane; i < BT * 96; i += 64) {
                const int b = i / 96, c = i - b * 96;
                float v = 0.f;
                if (dverts && b < nb && c < ncol) v = dverts[(b0 + b) * (long long)(NV * 3) + tile * 96 + c];
                mystage[b * SS + (c / 3) * 6 + (c % 3)] = v;
            }
        }
        __syncthreads();
        if (djoints && h == 0 && vbody) {
            const float* dj = djoints + (b0 + bl) * (STRAPS_SMPL_NJOINTS_OUT * 3) + 72;   // joints 24..89
            const int e0 = m.jrt_ptr[tile], e1 = m.jrt_ptr[tile + 1];
            for (int e = e0; e < e1; ++e) {
                const int code = m.jrt_code[e];
                const float w = m.jrt_w[e];
                float* sv = mystage + bl * SS + (code >> 8) * 6;
                const float* g = dj + (code & 255) * 3;
                sv[0] = fmaf(w, g[0], sv[0]); sv[1] = fmaf(w, g[1], sv[1]); sv[2] = fmaf(w, g[2], sv[2]);
            }
        }
        __syncthreads();
        // ---- (3) skinning backward per (body, vertex) ----
        {
            const float* Ab = As_ + bl * AS;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int vrow = (r & 3) + 8 * (r >> 2) + 4 * h;
                const int v = tile * 32 + vrow;
                float* sv = mystage + bl * SS + vrow * 6;
                const float gx = sv[0], gy = sv[1], gz = sv[2];
                sv[3] = ax[r]; sv[4] = ay[r]; sv[5] = az[r];              // v_posed, for the joint owners below
                float t00 = 0.f, t01 = 0.f, t02 = 0.f, t10 = 0.f, t11 = 0.f, t12 = 0.f, t20 = 0.f, t21 = 0.f, t22 = 0.f;
                for (int k = 0; k < KW; ++k) {
                    const float w = m.skin_w[v * KW + k];
                    const int jo = m.skin_j[v * KW + k] * 12;
                    const f32x4 a0 = *reinterpret_cast<const f32x4*>(Ab + jo);
                    const f32x4 a1 = *reinterpret_cast<const f32x4*>(Ab + jo + 4);
                    const f32x4 a2 = *reinterpret_cast<const f32x4*>(Ab + jo + 8);
                    t00 += w * a0[0]; t01 += w * a0[1]; t02 += w * a0[2];
                    t10 += w * a1[0]; t11 += w * a1[1]; t12 += w * a1[2];
                    t20 += w * a2[0]; t21 += w * a2[1]; t22 += w * a2[2];
                }
                // g_vposed = T_R^T g
                ax[r] = t00 * gx + t10 * gy + t20 * gz;
                ay[r] = t01 * gx + t11 * gy + t21 * gz;
                az[r] = t02 * gx + t12 * gy + t22 * gz;
            }
        }
        __syncthreads();   // all four tiles of the round are staged (g and v_posed)
        // ---- (3b) dA_j += w * g (x) [v_posed; 1], by joint owner, entries in table order ----
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int j = wave + 4 * (2 * i + h);
            const int e0 = m.dj_ptr[rd * 24 + j], e1 = m.dj_ptr[rd * 24 + j + 1];
            for (int e = e0; e < e1; ++e) {
                const int code = m.dj_code[e];                            // tile_in_round << 5 | vertex row
                const float w = m.dj_w[e];
                const float* sv = stage + ((code >> 5) * BT + bl) * SS + (code & 31) * 6;
                const float wx = w * sv[0], wy = w * sv[1], wz = w * sv[2];
                const float x = sv[3], y = sv[4], z = sv[5];
                dacc[i][0] = fmaf(wx, x, dacc[i][0]); dacc[i][1] = fmaf(wx, y, dacc[i][1]); dacc[i][2] = fmaf(wx, z, dacc[i][2]); dacc[i][3] += wx;
                dacc[i][4] = fmaf(wy, x, dacc[i][4]); dacc[i][5] = fmaf(wy, y, dacc[i][5]); dacc[i][6] = fmaf(wy, z, dacc[i][6]); dacc[i][7] += wy;
                dacc[i][8] = fmaf(wz, x, dacc[i][8]); dacc[i][9] = fmaf(wz, y, dacc[i][9]); dacc[i][10] = fmaf(wz, z, dacc[i][10]); dacc[i][11] += wz;
            }
        }
        // ---- (4) dF^T[k][b] += sum_v D[k][v][c] * g_vposed[v][b][c] ----
        {
            const f32x4* pt = blend_t + ((long long)tile * 3 * 7 * 4) * 64 + lane;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
#pragma unroll
                for (int f = 0; f < 7; ++f) {
#pragma unroll
                    for (int rq = 0; rq < 4; ++rq) {
                        const f32x4 a = pt[((c * 7 + f) * 4 + rq) * 64];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float g = c == 0 ? ax[rq * 4 + e] : (c == 1 ? ay[rq * 4 + e] : az[rq * 4 + e]);
                            accF[f] = mfma32(a[e], g, accF[f]);
                        }
                    }
                }
            }
        }
        __syncthreads();
    }
    // ---- per-chunk partials ----
    // each wave holds dF sums over ITS tiles: fixed-order tree reduction over the 4 waves through the (now free)
    // F/A staging area: (0 += 2, 1 += 3) then (0 += 1)
    {
        __syncthreads();
        float* red = smem;                                   // 2 x 7168 floats fit in As_ + stage (34048 floats)
        if (wave >= 2) {
#pragma unroll
            for (int f = 0; f < 7; ++f)
#pragma unroll
                for (int q = 0; q < 16; ++q) red[(wave - 2) * 7168 + (f * 16 + q) * 64 + lane] = accF[f][q];
        }
        __syncthreads();
        if (wave < 2) {
#pragma unroll
            for (int f = 0; f < 7; ++f)
#pragma unroll
                for (int q = 0; q < 16; ++q) accF[f][q] += red[wave * 7168 + (f * 16 + q) * 64 + lane];
        }
        __syncthreads();
        if (wave == 1) {
#pragma unroll
            for (int f = 0; f < 7; ++f)
#pragma unroll
                for (int q = 0; q < 16; ++q) red[(f * 16 + q) * 64 + lane] = accF[f][q];
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int f = 0; f < 7; ++f)
#pragma unroll
                for (int q = 0; q < 16; ++q) accF[f][q] += red[(f * 16 + q) * 64 + lane];
        }
    }
    if (vbody && wave == 0) {
        float* o = dFp + ((long long)chunk * B + b0 + bl) * KP;
#pragma unroll
        for (int f = 0; f < 7; ++f)
#pragma unroll
            for (int q = 0; q < 16; ++q) o[f * 32 + mfma_row(q, lane)] = accF[f][q];
    }
    if (vbody) {
        float* o = dAp + ((long long)chunk * B + b0 + bl) * 288;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int j = wave + 4 * (2 * i + h);
#pragma unroll
            for (int e = 0; e < 12; ++e) o[j * 12 + e] = dacc[i][e];
        }
    }
}

// ------------------------------------------------------------------------------------------
// Lane exchanges of smpl_pose_bwd_kernel.  XCHG = 0 is the product: __shfl = ds_bpermute_b32, several in flight, waits placed by the compiler.
// The other forms exist in the tools build only (round 5, DESIGN section 1: this kernel is not bit-reproducible beside a bf16x3 convolution
// workgroup on its compute unit -- WHAT do the wrong values look like, and does the exchange have to go through the LDS unit for it?):
//   1  v_readlane_b32 + select, 64 of them per exchange: no LDS-unit instruction in the kernel at all
//   2  ds_bpermute_b32 TWICE, each followed by s_waitcnt lgkmcnt(0) (inline assembly: one in flight), both checked against form 1; mismatches logged
//   4  ds_bpermute_b32 + s_waitcnt lgkmcnt(0), unchecked: is "one in flight" alone enough to make it reproducible?
//   5  the product's __shfl (several in flight), checked against form 1 afterwards; mismatches logged
#ifdef STRAPS_TOOLS
__device__ unsigned g_xchg_log[4 + 8 * 4096];      // [0] mismatches seen, [1] exchanges checked (one count per wave-level call); records of 8 words
#endif
__device__ __forceinline__ int xchg_readlane(int v, int src) {
    int r = 0;
#pragma unroll
    for (int k = 0; k < 64; ++k) {
        const int t = __builtin_amdgcn_readlane(v, k);
        r = (src == k) ? t : r;
    }
    return r;
}
__device__ __forceinline__ int xchg_bpermute_now(int v, int src) {
    int r;
    asm volatile("ds_bpermute_b32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(src << 2), "v"(v) : "memory");
    return r;
}
template <int XCHG>
__device__ __forceinline__ float lane_get(float value, int src, int site, int trip, int& prev) {
    if constexpr (XCHG == 0) {
        return __shfl(value, src, 64);
    } else {
        const int v = __float_as_int(value);
        src &= 63;
        if constexpr (XCHG == 1) return __int_as_float(xchg_readlane(v, src));
        if constexpr (XCHG == 4) return __int_as_float(xchg_bpermute_now(v, src));
#ifdef STRAPS_TOOLS
        int a, b;
        if constexpr (XCHG == 2) { a = xchg_bpermute_now(v, src); b = xchg_bpermute_now(v, src); }
        else { a = __float_as_int(__shfl(value, src, 64)); b = a; }
        const int truth = xchg_readlane(v, src);
        if ((threadIdx.x & 63) == 0) atomicAdd(&g_xchg_log[1], 1u);
        if (a != truth || b != truth) {
            int from = -1;                            // does the wrong value belong to ANOTHER lane of the same register?
            for (int k = 63; k >= 0; --k) from = (__builtin_amdgcn_readlane(v, k) == a) ? k : from;
            const unsigned slot = atomicAdd(&g_xchg_log[0], 1u);
            if (slot < 4096) {
                unsigned* r = g_xchg_log + 4 + slot * 8;
                r[0] = (unsigned)site | ((unsigned)trip << 8) | ((unsigned)(from & 0xff) << 16) | ((unsigned)(a != truth) << 24) | ((unsigned)(b != truth) << 25);
                r[1] = threadIdx.x | (blockIdx.x << 16);
                r[2] = (unsigned)src; r[3] = (unsigned)truth; r[4] = (unsigned)a; r[5] = (unsigned)b; r[6] = (unsigned)v; r[7] = (unsigned)prev;
            }
        }
        prev = a;
        return __int_as_float(a);
#else
        // (the checked forms 2 / 5 exist in the tools build only -- instantiating one in the product must not compile to "no exchange")
        static_assert(XCHG == 0 || XCHG == 1 || XCHG == 4, "lane_get: the checked exchange forms (2, 5) need -DSTRAPS_TOOLS");
        return value;
#endif
    }
}

// SC = 1 (tools build): the chunk partials are read with system-scope loads (sc0 sc1: past the L1 and the L2) -- does a stale cache line explain it?
template <int SC>
__device__ __forceinline__ float partial_load(const float* p) {
    if constexpr (SC) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    else return *p;
}
#ifdef STRAPS_TOOLS
__global__ void pose_bwd_gap_kernel() {}
// DBG = 1: workgroup 0 dumps its intermediate values, field-major ([field][thread], 256 threads), for the probe to name the FIRST quantity that differs
constexpr int POSE_DBG_FIELDS = 112;
__device__ float g_pose_dbg[POSE_DBG_FIELDS * 256];
#define POSE_DBG(field, value) do { if constexpr (DBG) { if (blockIdx.x == 0) g_pose_dbg[(field) * 256 + threadIdx.x] = (value); } } while (0)
#else
#define POSE_DBG(field, value) do { } while (0)
#endif

// The kernel as it was (packed fp32 instructions allowed: the reproducer's victim, and the forms that write one out in assembly) exists in a tools build made with
// STRAPS_TOOLS_SMPL_BWD_FLAGS=-DSTRAPS_POSE_BWD_PACKED only; every other build compiles it without packed fp32 instructions (common.h, STRAPS_NO_PACKED_FP32).
#if defined(STRAPS_TOOLS) && defined(STRAPS_POSE_BWD_PACKED)
#define POSE_BWD_ATTR
#else
#define POSE_BWD_ATTR STRAPS_NO_PACKED_FP32
#endif
template <int XCHG, int SC = 0, int DBG = 0>
__global__ __launch_bounds__(256) POSE_BWD_ATTR void smpl_pose_bwd_kernel(straps_smpl_model_t m, const float* __restrict__ betas,
                                                            const float* __restrict__ rotmats, const float* __restrict__ dFp,
                                                            const float* __restrict__ dAp, const float* __restrict__ djoints,
                                                            float* __restrict__ dbetas, float* __restrict__ drot, long long B,
                                                            int chunks) {
    const int lane = threadIdx.x & 63;
    const int j = lane & 31;
    const int base = lane & 32;
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
    [[maybe_unused]] int par_early = 0, dep_early = 0;
    typedef float pose_f4 __attribute__((ext_vector_type(4)));
    [[maybe_unused]] pose_f4 sd_first = {0.f, 0.f, 0.f, 0.f};
    if constexpr (DBG == 2 || DBG == 3) {
        // (tools build; the first value that differs in an event is J[0], short of its l = 1 term in lanes 48..63.  Here the joint's first four shape
        //  coefficients are loaded by ONE dwordx4 load written out in assembly and awaited on the spot (vmcnt(0)); its four result registers are copied
        //  right away ("early") and again at the kernel's end ("late": the same registers read ~10^4 cycles later) -- was the value never written, or
        //  written after the counter said so?  DBG = 3 sleeps ~500 cycles between the wait and the early copy)
        const float* ap = m.j_shapedirs + jj * 30;
        if constexpr (DBG == 3) asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)\n\ts_sleep 8" : "=&v"(sd_first) : "v"(ap) : "memory");
        else asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=&v"(sd_first) : "v"(ap) : "memory");
        float e0, e1, e2, e3;
        asm volatile("v_mov_b32 %0, %4\n\tv_mov_b32 %1, %5\n\tv_mov_b32 %2, %6\n\tv_mov_b32 %3, %7" : "=&v"(e0), "=&v"(e1), "=&v"(e2), "=&v"(e3)
                     : "v"(sd_first.x), "v"(sd_first.y), "v"(sd_first.z), "v"(sd_first.w));
        POSE_DBG(104, e0); POSE_DBG(105, e1); POSE_DBG(106, e2); POSE_DBG(107, e3);
        float sd[30];
#pragma unroll
        for (int q = 4; q < 30; ++q) sd[q] = m.j_shapedirs[jj * 30 + q];
        sd[0] = e0; sd[1] = e1; sd[2] = e2; sd[3] = e3;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float s = m.j_template[jj * 3 + c];
#pragma unroll
            for (int l = 0; l < 10; ++l) s = fmaf(sd[c * 10 + l], beta[l], s);
            J[c] = s;
        }
    }
#if defined(STRAPS_TOOLS) && defined(STRAPS_POSE_BWD_PACKED)
    else if constexpr (DBG == 4 || DBG == 5 || DBG == 6) {
        // (tools build.  The wrong J[0] is the result of ONE instruction: the l = 1 step of the (J[0], J[1]) pair, v_pk_fma_f32 ... op_sel:[0,1,0] -- the only step
        //  whose LOW half takes the HIGH register of a source.  DBG = 4: that instruction written out, and the same product-sum computed again by a plain
        //  v_fma_f32 from the same registers right behind it; a difference is logged with the operands.  DBG = 5: the kernel's usual code, but every load
        //  of the prologue awaited (and ~250 cycles slept) before the first arithmetic instruction: no load result arrives while the packed instructions run)
        float sd[30], t[3];
#pragma unroll
        for (int q = 0; q < 30; ++q) sd[q] = m.j_shapedirs[jj * 30 + q];
#pragma unroll
        for (int c = 0; c < 3; ++c) t[c] = m.j_template[jj * 3 + c];
        par_early = m.parents[jj];
        dep_early = m.depth[jj];
        if constexpr (DBG == 5) {
#pragma unroll
            for (int q = 0; q < 30; ++q) asm volatile("" : "+v"(sd[q]));
#pragma unroll
            for (int c = 0; c < 3; ++c) asm volatile("" : "+v"(t[c]));
#pragma unroll
            for (int l = 0; l < 10; ++l) asm volatile("" : "+v"(beta[l]));
#pragma unroll
            for (int e = 0; e < 9; ++e) asm volatile("" : "+v"(R[e]));
            asm volatile("" : "+v"(par_early), "+v"(dep_early));
            asm volatile("s_waitcnt vmcnt(0)\n\ts_sleep 4" ::: "memory");
            // (pinned again BEHIND the sleep: volatile statements keep their order, and the arithmetic below depends on these)
#pragma unroll
            for (int q = 0; q < 30; ++q) asm volatile("" : "+v"(sd[q]));
#pragma unroll
            for (int c = 0; c < 3; ++c) asm volatile("" : "+v"(t[c]));
#pragma unroll
            for (int l = 0; l < 10; ++l) asm volatile("" : "+v"(beta[l]));
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float s = t[c];
#pragma unroll
                for (int l = 0; l < 10; ++l) s = fmaf(sd[c * 10 + l], beta[l], s);
                J[c] = s;
            }
        } else {
            typedef float pose_f2 __attribute__((ext_vector_type(2)));
            pose_f2 acc = {fmaf(sd[0], beta[0], t[0]), fmaf(sd[10], beta[0], t[1])};
            pose_f2 a = {sd[1], sd[11]}, b = {beta[0], beta[1]}, d;
            float plain;
            asm volatile("" : "+v"(a), "+v"(b), "+v"(acc));      // (the three register pairs exist as pairs from here on: the plain v_fma_f32 reads their halves)
            if constexpr (DBG == 6)      // (DBG = 6: the same pair of instructions with every load of the wave awaited and ~250 cycles slept in front of them)
                asm volatile("s_waitcnt vmcnt(0)\n\ts_sleep 4\n\tv_pk_fma_f32 %0, %2, %3, %4 op_sel:[0,1,0]\n\tv_fma_f32 %1, %5, %6, %7"
                             : "=&v"(d), "=&v"(plain) : "v"(a), "v"(b), "v"(acc), "v"(a.x), "v"(b.y), "v"(acc.x) : "memory");
            else
                asm volatile("v_pk_fma_f32 %0, %2, %3, %4 op_sel:[0,1,0]\n\tv_fma_f32 %1, %5, %6, %7"
                             : "=&v"(d), "=&v"(plain) : "v"(a), "v"(b), "v"(acc), "v"(a.x), "v"(b.y), "v"(acc.x));
            if ((threadIdx.x & 63) == 0) atomicAdd(&g_xchg_log[1], 1u);
            if (__float_as_uint(d.x) != __float_as_uint(plain)) {
                const unsigned slot = atomicAdd(&g_xchg_log[0], 1u);
                if (slot < 4096) {
                    unsigned* r = g_xchg_log + 4 + slot * 8;
                    r[0] = 200; r[1] = threadIdx.x | (blockIdx.x << 16); r[2] = __float_as_uint(a.x); r[3] = __float_as_uint(b.y); r[4] = __float_as_uint(acc.x);
                    r[5] = __float_as_uint(d.x); r[6] = __float_as_uint(plain); r[7] = __float_as_uint(d.y);
                }
            }
            float s0 = d.x, s1 = d.y, s2 = t[2];
#pragma unroll
            for (int l = 2; l < 10; ++l) { s0 = fmaf(sd[l], beta[l], s0); s1 = fmaf(sd[10 + l], beta[l], s1); }
#pragma unroll
            for (int l = 0; l < 10; ++l) s2 = fmaf(sd[20 + l], beta[l], s2);
            J[0] = s0; J[1] = s1; J[2] = s2;
        }
    }
#endif
    else {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float s = m.j_template[jj * 3 + c];
#pragma unroll
            for (int l = 0; l < 10; ++l) s = fmaf(m.j_shapedirs[(jj * 3 + c) * 10 + l], beta[l], s);
            J[c] = s;
        }
    }
#if !(defined(STRAPS_TOOLS) && defined(STRAPS_POSE_BWD_PACKED))
    static_assert(DBG < 4, "forms 4..6 need -DSTRAPS_POSE_BWD_PACKED");
#endif
    const int par = (DBG >= 4) ? par_early : m.parents[jj];
    const int dep = vj ? ((DBG >= 4) ? dep_early : m.depth[jj]) : -1;
    const int src = base + (par < 0 ? 0 : par);
    int xprev = 0;      // (tools forms: the value the previous exchange delivered to this lane)
    float rel[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float jp = lane_get<XCHG>(J[c], src, 0 + c, 0, xprev);
        rel[c] = (jj > 0) ? J[c] - jp : J[c];
    }
    for (int c = 0; c < 3; ++c) { POSE_DBG(0 + c, J[c]); POSE_DBG(3 + c, rel[c]); }
    for (int e = 0; e < 9; ++e) POSE_DBG(6 + e, R[e]);
    POSE_DBG(15, beta[0]); POSE_DBG(16, beta[9]); POSE_DBG(17, (float)par); POSE_DBG(18, (float)dep);
    // forward chain: G (own global transform) and P (parent's rotation), recomputed
    float G[12], PR[9];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        G[r * 4 + 0] = R[r * 3 + 0]; G[r * 4 + 1] = R[r * 3 + 1]; G[r * 4 + 2] = R[r * 3 + 2]; G[r * 4 + 3] = rel[r];
    }
#pragma unroll
    for (int e = 0; e < 9; ++e) PR[e] = (e == 0 || e == 4 || e == 8) ? 1.f : 0.f;
    for (int d = 1; d <= m.max_depth; ++d) {
        float P[12];
#pragma unroll
        for (int e = 0; e < 12; ++e) P[e] = lane_get<XCHG>(G[e], src, 8 + e, d, xprev);
        if (dep == d) {
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const float p0 = P[r * 4 + 0], p1 = P[r * 4 + 1], p2 = P[r * 4 + 2];
                PR[r * 3 + 0] = p0; PR[r * 3 + 1] = p1; PR[r * 3 + 2] = p2;
                G[r * 4 + 0] = p0 * R[0] + p1 * R[3] + p2 * R[6];
                G[r * 4 + 1] = p0 * R[1] + p1 * R[4] + p2 * R[7];
                G[r * 4 + 2] = p0 * R[2] + p1 * R[5] + p2 * R[8];
                G[r * 4 + 3] = p0 * rel[0] + p1 * rel[1] + p2 * rel[2] + P[r * 4 + 3];
            }
        }
    }
    for (int e = 0; e < 12; ++e) POSE_DBG(19 + e, G[e]);
    for (int e = 0; e < 9; ++e) POSE_DBG(31 + e, PR[e]);
    // gradient arriving at A_j (sum of chunk partials) and at the posed joint
    float gA[12];
#pragma unroll
    for (int e = 0; e < 12; ++e) gA[e] = 0.f;
    if (vb && vj)
        for (int c = 0; c < chunks; ++c) {
            const float* p = dAp + (((long long)c * B + body) * 24 + j) * 12;
#pragma unroll
            for (int e = 0; e < 12; ++e) gA[e] += partial_load<SC>(p + e);
        }
    float gGR[9], gGt[3], gJ[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const float gat = gA[r * 4 + 3];
        gGR[r * 3 + 0] = gA[r * 4 + 0] - gat * J[0];
        gGR[r * 3 + 1] = gA[r * 4 + 1] - gat * J[1];
        gGR[r * 3 + 2] = gA[r * 4 + 2] - gat * J[2];
        gGt[r] = gat + ((vb && vj && djoints) ? djoints[(body * STRAPS_SMPL_NJOINTS_OUT + j) * 3 + r] : 0.f);
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) gJ[c] = -(G[0 * 4 + c] * gA[3] + G[1 * 4 + c] * gA[7] + G[2 * 4 + c] * gA[11]);

    for (int e = 0; e < 12; ++e) POSE_DBG(40 + e, gA[e]);
    for (int e = 0; e < 9; ++e) POSE_DBG(52 + e, gGR[e]);
    for (int c = 0; c < 3; ++c) { POSE_DBG(61 + c, gGt[c]); POSE_DBG(64 + c, gJ[c]); }
    int child[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) child[c] = m.children[jj * 3 + c];
    for (int c = 0; c < 3; ++c) POSE_DBG(67 + c, (float)child[c]);
    float gR[9];
#pragma unroll
    for (int e = 0; e < 9; ++e) gR[e] = 0.f;
    for (int d = m.max_depth; d >= 1; --d) {
        // message of a depth-d joint to its parent: gP_R (9), gP_t (3), -grel (3)
        float M[15];
        const bool mine = (dep == d);
        float grel[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) grel[c] = PR[0 * 3 + c] * gGt[0] + PR[1 * 3 + c] * gGt[1] + PR[2 * 3 + c] * gGt[2];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c)
                M[r * 3 + c] = mine ? (gGR[r * 3 + 0] * R[c * 3 + 0] + gGR[r * 3 + 1] * R[c * 3 + 1] + gGR[r * 3 + 2] * R[c * 3 + 2] + gGt[r] * rel[c]) : 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) { M[9 + c] = mine ? gGt[c] : 0.f; M[12 + c] = mine ? -grel[c] : 0.f; }
        if (mine) {
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    gR[r * 3 + c] = PR[0 * 3 + r] * gGR[0 * 3 + c] + PR[1 * 3 + r] * gGR[1 * 3 + c] + PR[2 * 3 + r] * gGR[2 * 3 + c];
#pragma unroll
            for (int c = 0; c < 3; ++c) gJ[c] += grel[c];
        }
        // parents gather from their (up to 3) children
#pragma unroll
        for (int cc = 0; cc < 3; ++cc) {
            const int ch = child[cc];
            const int cs = base + (ch < 0 ? 0 : ch);
            float in[15];
#pragma unroll
            for (int e = 0; e < 15; ++e) in[e] = lane_get<XCHG>(M[e], cs, 32 + cc * 16 + e, d, xprev);
            if (vj && ch >= 0) {       // in[] is zero unless that child is at depth d
#pragma unroll
                for (int e = 0; e < 9; ++e) gGR[e] += in[e];
#pragma unroll
                for (int c = 0; c < 3; ++c) { gGt[c] += in[9 + c]; gJ[c] += in[12 + c]; }
            }
        }
    }
    for (int e = 0; e < 9; ++e) { POSE_DBG(70 + e, gGR[e]); POSE_DBG(79 + e, gR[e]); }
    for (int c = 0; c < 3; ++c) { POSE_DBG(88 + c, gGt[c]); POSE_DBG(91 + c, gJ[c]); }
    if (j == 0) {   // root: G = [R | J]
#pragma unroll
        for (int e = 0; e < 9; ++e) gR[e] = gGR[e];
#pragma unroll
        for (int c = 0; c < 3; ++c) gJ[c] += gGt[c];
    }
    // pose-feature and direct beta gradients from dF (sum of chunk partials)
    float gbeta_direct = 0.f;
    if (vb && vj) {
        for (int c = 0; c < chunks; ++c) {
            const float* p = dFp + ((long long)c * B + body) * KP;
            if (j >= 1) {
#pragma unroll
                for (int e = 0; e < 9; ++e) gR[e] += partial_load<SC>(p + 11 + (j - 1) * 9 + e);
            }
            if (j < 10) gbeta_direct += partial_load<SC>(p + 1 + j);
        }
#pragma unroll
        for (int e = 0; e < 9; ++e) drot[(body * 24 + j) * 9 + e] = gR[e];
    }
    for (int e = 0; e < 9; ++e) POSE_DBG(94 + e, gR[e]);
    POSE_DBG(103, gbeta_direct);
    if constexpr (DBG == 2 || DBG == 3) {
        float l0, l1, l2, l3;
        asm volatile("v_mov_b32 %0, %4\n\tv_mov_b32 %1, %5\n\tv_mov_b32 %2, %6\n\tv_mov_b32 %3, %7" : "=&v"(l0), "=&v"(l1), "=&v"(l2), "=&v"(l3)
                     : "v"(sd_first.x), "v"(sd_first.y), "v"(sd_first.z), "v"(sd_first.w));
        POSE_DBG(108, l0); POSE_DBG(109, l1); POSE_DBG(110, l2); POSE_DBG(111, l3);
    }
    // dbeta[l] = dF[1+l] + sum_j sum_c Js[j][c][l] * gJ_j[c]
#pragma unroll
    for (int l = 0; l < 10; ++l) {
        float s = vj ? (m.j_shapedirs[(jj * 3 + 0) * 10 + l] * gJ[0] + m.j_shapedirs[(jj * 3 + 1) * 10 + l] * gJ[1] +
                        m.j_shapedirs[(jj * 3 + 2) * 10 + l] * gJ[2]) : 0.f;
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) {
            if constexpr (XCHG == 0) s += __shfl_xor(s, o, 64);
            else s += lane_get<XCHG>(s, lane ^ o, 96 + l, o, xprev);
        }
        const float direct = lane_get<XCHG>(gbeta_direct, base + l, 112 + l, 0, xprev);
        if (vb && j == l) dbetas[body * 10 + l] = s + direct;
    }
}

inline int resolve_rpc(long long batch, int chunks) {
    if (chunks <= 0) chunks = (batch >= 1024) ? 8 : 54;
    if (chunks > NROUNDS) chunks = NROUNDS;
    return (NROUNDS + chunks - 1) / chunks;
}

}  // namespace

// forward pose kernel is shared with smpl.hip (recomputes F and A for the backward)
int straps_smpl_launch_pose(const straps_smpl_model_t* model, const float* betas, const float* rotmats, float* F, float* Amat,
                            float* joints, long long batch, hipStream_t st);

extern "C" size_t straps_smpl_bwd_workspace_bytes(long long batch, int chunks) {
    const int rpc = resolve_rpc(batch, chunks);
    const int nch = (NROUNDS + rpc - 1) / rpc;
    return (size_t)batch * (size_t)(KP + 288 + nch * (KP + 288)) * sizeof(float);
}

extern "C" int straps_smpl_bwd(const straps_smpl_model_t* model, const float* betas, const float* rotmats, const float* dverts,
                               const float* djoints, float* dbetas, float* drotmats, void* workspace, long long batch, int chunks,
                               void* stream) {
    STRAPS_REQUIRE(model && betas && rotmats && dbetas && drotmats && workspace, "straps_smpl_bwd: null pointer");
    STRAPS_REQUIRE(model->blend_frag_t && model->children && model->jrt_ptr && model->dj_ptr && model->dj_code && model->dj_w,
                   "straps_smpl_bwd: model lacks the backward tables");
    STRAPS_REQUIRE(batch > 0, "straps_smpl_bwd: batch must be positive");
    hipStream_t st = (hipStream_t)stream;
    const int rpc = resolve_rpc(batch, chunks);
    const int nch = (NROUNDS + rpc - 1) / rpc;
    float* F = (float*)workspace;
    float* Amat = F + batch * KP;
    float* dFp = Amat + batch * 288;
    float* dAp = dFp + (long long)nch * batch * KP;
    int rc = straps_smpl_launch_pose(model, betas, rotmats, F, Amat, nullptr, batch, st);
    if (rc != STRAPS_OK) return rc;
    const size_t lds = (size_t)(BT * AS + 4 * BT * SS) * sizeof(float);
    STRAPS_RAISE_LDS((smpl_verts_bwd_kernel), lds, "smpl_verts_bwd_kernel");
#ifdef STRAPS_TOOLS
    static const int poison = STRAPS_TOOL_ENV_INT("STRAPS_POSE_BWD_POISON", 0);    // (1: the partials are NaN before their producer runs -- a result that is NaN read what this call never wrote)
    if (poison && hipMemsetAsync(dFp, 0xff, (size_t)nch * batch * (KP + 288) * sizeof(float), st) != hipSuccess) { straps_set_error("straps_smpl_bwd: poison failed"); return STRAPS_EHIP; }
#endif
    const long long btiles = (batch + BT - 1) / BT;
    if (btiles > 65535) { straps_set_error("straps_smpl_bwd: batch %lld exceeds one launch; split it", batch); return STRAPS_EUNSUPPORTED; }
    hipLaunchKernelGGL(smpl_verts_bwd_kernel, dim3(nch, (unsigned)btiles), dim3(256), lds, st, *model, F, Amat, dverts, djoints, dFp, dAp, batch, rpc);
    STRAPS_CHECK_LAUNCH("smpl_verts_bwd_kernel");
    // (Round 5, DESIGN section 1: this kernel was the one whose results differed between two processes on one GPU.  Cause: a packed fp32 instruction with a
    //  low-half operand select the compiler had formed in it -- the kernel is compiled without packed fp32 instructions now, see POSE_BWD_ATTR.  The mitigation
    //  that came first, 96 KB of dynamic LDS the kernel never touches so that no bf16x3 convolution workgroup fits beside it on a compute unit, is kept as a
    //  switch of the tools build for A/B runs -- STRAPS_POSE_BWD_FENCE=1 -- and is off everywhere else: profiles/r05_packed_fp32_fix.txt ran without it.)
    size_t fence = 0;
    auto pose_bwd = smpl_pose_bwd_kernel<0, 0>;
#ifdef STRAPS_TOOLS
    static const int xchg = STRAPS_TOOL_ENV_INT("STRAPS_POSE_BWD_XCHG", 0);        // (exchange forms, see lane_get)
    static const int fenced = STRAPS_TOOL_ENV_INT("STRAPS_POSE_BWD_FENCE", 0);     // (1: 96 KB of unused LDS, the first mitigation)
    pose_bwd = xchg == 1 ? smpl_pose_bwd_kernel<1> : xchg == 2 ? smpl_pose_bwd_kernel<2> : xchg == 4 ? smpl_pose_bwd_kernel<4> : xchg == 5 ? smpl_pose_bwd_kernel<5> : pose_bwd;
    if (fenced) fence = 96 * 1024;
    static const int sc = STRAPS_TOOL_ENV_INT("STRAPS_POSE_BWD_SC", 0);            // (1: partials read past the caches)
    if (sc) pose_bwd = smpl_pose_bwd_kernel<0, 1>;
    static const int dbg = STRAPS_TOOL_ENV_INT("STRAPS_POSE_BWD_DBG", 0);          // (1: intermediate values dumped, straps_tool_pose_dbg fetches them)
#ifdef STRAPS_POSE_BWD_PACKED
    if (dbg >= 4) pose_bwd = dbg == 4 ? smpl_pose_bwd_kernel<0, 0, 4> : dbg == 5 ? smpl_pose_bwd_kernel<0, 0, 5> : smpl_pose_bwd_kernel<0, 0, 6>;
    else
#endif
    if (dbg) pose_bwd = dbg == 2 ? smpl_pose_bwd_kernel<0, 0, 2> : dbg == 3 ? smpl_pose_bwd_kernel<0, 0, 3> : xchg == 1 ? smpl_pose_bwd_kernel<1, 0, 1> : smpl_pose_bwd_kernel<0, 0, 1>;
    static const int gap = STRAPS_TOOL_ENV_INT("STRAPS_POSE_BWD_GAP", 0);          // (1: an empty kernel between the producer of the partials and this kernel)
    if (gap) hipLaunchKernelGGL(pose_bwd_gap_kernel, dim3(1), dim3(64), 0, st);
#endif
    if (fence) STRAPS_RAISE_LDS(pose_bwd, fence, "smpl_pose_bwd_kernel");
    hipLaunchKernelGGL(pose_bwd, dim3((unsigned)((batch * 32 + 255) / 256)), dim3(256), fence, st, *model, betas, rotmats, dFp, dAp,
                       djoints, dbetas, drotmats, batch, nch);
    STRAPS_CHECK_LAUNCH("smpl_pose_bwd_kernel");
    return STRAPS_OK;
}

#ifdef STRAPS_TOOLS
// tools build only: the intermediate values smpl_pose_bwd_kernel<.., .., 1> dumped, copied device -> device on `stream` ([112][256] floats)
extern "C" int straps_tool_pose_dbg(float* device_dst, void* stream) {
    STRAPS_REQUIRE(device_dst, "straps_tool_pose_dbg: null pointer");
    const hipError_t e = hipMemcpyFromSymbolAsync(device_dst, HIP_SYMBOL(g_pose_dbg), sizeof(float) * POSE_DBG_FIELDS * 256, 0, hipMemcpyDeviceToDevice, (hipStream_t)stream);
    if (e != hipSuccess) { straps_set_error("straps_tool_pose_dbg: %s", hipGetErrorString(e)); return STRAPS_EHIP; }
    return STRAPS_OK;
}
// tools build only: the exchange log of smpl_pose_bwd_kernel<2 | 5> copied to the host (words: 4 + 8 * 4096), then cleared when `reset`
extern "C" int straps_tool_xchg_log(unsigned* host_words, int reset) {
    STRAPS_REQUIRE(host_words, "straps_tool_xchg_log: null pointer");
    static unsigned zeros[4 + 8 * 4096];
    hipError_t e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpyFromSymbol(host_words, HIP_SYMBOL(g_xchg_log), sizeof(zeros));
    if (e == hipSuccess && reset) e = hipMemcpyToSymbol(HIP_SYMBOL(g_xchg_log), zeros, sizeof(zeros));
    if (e != hipSuccess) { straps_set_error("straps_tool_xchg_log: %s", hipGetErrorString(e)); return STRAPS_EHIP; }
    return STRAPS_OK;
}
#endif

#ifdef STRAPS_TOOLS
// tools build only (round 5): a victim of nothing but packed fp32 instructions.  One workgroup-sized launch like smpl_pose_bwd_kernel's (no LDS,
// a few loads, then arithmetic); every trip runs seven forms of v_pk_{fma,mul,add}_f32, each followed by plain v_fma / v_mul / v_add of the SAME registers, and logs
// every lane whose packed result differs (g_xchg_log: form | trip << 16, thread, operands, both results).  Which operand-select forms lose their product beside a
// bf16x3 convolution workgroup -- only "low result from a HIGH register" (op_sel), or others too -- and only in a wave's first instructions, or any time?
namespace {
typedef float tool_f2 __attribute__((ext_vector_type(2)));
#define PK_CHECK(form, PK_ASM, LO_ASM, HI_ASM)                                                                                          \
    do {                                                                                                                                \
        tool_f2 d; float lo, hi;                                                                                                        \
        asm volatile(PK_ASM "\n\t" LO_ASM "\n\t" HI_ASM : "=&v"(d), "=&v"(lo), "=&v"(hi)                                               \
                     : "v"(a), "v"(b), "v"(c), "v"(a.x), "v"(a.y), "v"(b.x), "v"(b.y), "v"(c.x), "v"(c.y));                             \
        if (__float_as_uint(d.x) != __float_as_uint(lo) || __float_as_uint(d.y) != __float_as_uint(hi)) {                               \
            const unsigned slot = atomicAdd(&g_xchg_log[0], 1u);                                                                        \
            if (slot < 4096) {                                                                                                          \
                unsigned* r = g_xchg_log + 4 + slot * 8;                                                                                \
                r[0] = 300u + (form) + ((unsigned)trip << 16); r[1] = threadIdx.x | (blockIdx.x << 16);                                 \
                r[2] = __float_as_uint(d.x); r[3] = __float_as_uint(lo); r[4] = __float_as_uint(d.y); r[5] = __float_as_uint(hi);       \
                r[6] = __float_as_uint(c.x); r[7] = __float_as_uint(c.y);                                                               \
            }                                                                                                                           \
        }                                                                                                                               \
        sum += d.x + d.y;                                                                                                               \
    } while (0)
// operands: %3 a, %4 b, %5 c (pairs); %6 a.lo %7 a.hi %8 b.lo %9 b.hi %10 c.lo %11 c.hi
__global__ __launch_bounds__(256) void pk_victim_kernel(const float* __restrict__ in, float* __restrict__ out, int trips) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    tool_f2 a = {in[t * 6 + 0], in[t * 6 + 1]}, b = {in[t * 6 + 2], in[t * 6 + 3]}, c = {in[t * 6 + 4], in[t * 6 + 5]};
    float sum = 0.f;
    for (int trip = 0; trip < trips; ++trip) {
        asm volatile("" : "+v"(a), "+v"(b), "+v"(c));
        if ((threadIdx.x & 63) == 0) atomicAdd(&g_xchg_log[1], 1u);
        PK_CHECK(0, "v_pk_fma_f32 %0, %3, %4, %5 op_sel:[0,1,0]", "v_fma_f32 %1, %6, %9, %10", "v_fma_f32 %2, %7, %9, %11");
        PK_CHECK(1, "v_pk_fma_f32 %0, %3, %4, %5 op_sel:[1,0,0]", "v_fma_f32 %1, %7, %8, %10", "v_fma_f32 %2, %7, %9, %11");
        PK_CHECK(2, "v_pk_fma_f32 %0, %3, %4, %5 op_sel:[0,0,1]", "v_fma_f32 %1, %6, %8, %11", "v_fma_f32 %2, %7, %9, %11");
        PK_CHECK(3, "v_pk_fma_f32 %0, %3, %4, %5 op_sel_hi:[1,0,1]", "v_fma_f32 %1, %6, %8, %10", "v_fma_f32 %2, %7, %8, %11");
        PK_CHECK(4, "v_pk_fma_f32 %0, %3, %4, %5", "v_fma_f32 %1, %6, %8, %10", "v_fma_f32 %2, %7, %9, %11");
        PK_CHECK(5, "v_pk_mul_f32 %0, %3, %4 op_sel:[0,1]", "v_mul_f32 %1, %6, %9", "v_mul_f32 %2, %7, %9");
        PK_CHECK(6, "v_pk_add_f32 %0, %3, %4 op_sel:[0,1]", "v_add_f32 %1, %6, %9", "v_add_f32 %2, %7, %9");
        a.x += 0.001f; b.y -= 0.002f; c.x += 0.003f;
    }
    out[t] = sum;
}
#undef PK_CHECK
}  // namespace
extern "C" int straps_tool_pk_victim(const float* in, float* out, int blocks, int trips, void* stream) {
    STRAPS_REQUIRE(in && out && blocks > 0 && trips > 0, "straps_tool_pk_victim: bad argument");
    hipLaunchKernelGGL(pk_victim_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, in, out, trips);
    STRAPS_CHECK_LAUNCH("pk_victim_kernel");
    return STRAPS_OK;
}
#endif
