/* straps_hip.h -- C ABI of the MI355X-native STRAPS hot path (libstraps_hip.so, gfx950 only).
 *
 * The reference has no FFI of its own: its operator boundary is the Python nn.Module surface
 * (SURVEY.md 8b).  These entry points are what a ctypes binding placed under that surface calls;
 * each one names the reference code it replaces (paths relative to the reference repo root).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the parameter name ends in _host;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); nothing synchronises;
 *   - no hidden allocation: outputs and workspaces are caller-owned, sizes given by *_bytes();
 *   - return value: 0 = STRAPS_OK, otherwise an error code; straps_last_error() gives the text;
 *   - all arithmetic is IEEE fp32 (fp32-input MFMA = exact fmaf chains; no reduced precision);
 *   - activations inside the encoder are NHWC fp32; the boundary tensor (network input) is NCHW
 *     exactly as the reference passes it (models/regressor.py:43).
 */
#ifndef STRAPS_HIP_H
#define STRAPS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define STRAPS_ABI_VERSION 9

#define STRAPS_OK 0
#define STRAPS_EINVAL 1       /* bad argument (shape, alignment, null pointer) */
#define STRAPS_EHIP 2         /* a HIP runtime call / kernel launch failed */
#define STRAPS_EUNSUPPORTED 3 /* valid request this build does not cover */

int straps_abi_version(void);
/* calibration: `blocks` x 4 waves each issue 4*iters register-resident fp32 MFMAs (32x32x2);
 * seed512 = 512 floats, out = blocks*256 floats.  Time it to get the board's sustained MFMA rate. */
int straps_selftest_mfma_peak(const float* seed512, float* out, int blocks, int iters, void* stream);
/* calibration of the bf16 matrix pipe: `blocks` x 4 waves each issue iters x 48 register-resident v_mfma_f32_32x32x16_bf16 on
 * operand-like data (out = blocks*256 floats; clk2 optional: shader / wall ticks of workgroup 0).  Timed, it gives the rate the pipe
 * SUSTAINS under the board's power budget -- the ceiling of the bf16x3 convolution kernels (bench.py: roofline.sustained_*). */
int straps_selftest_mfma_bf16(float* out, unsigned long long* clk2, int blocks, int iters, void* stream);
/* the same with a DENSE issue stream (round 6): eight independent accumulators per wave, two waves per SIMD at blocks = 512; `blocks` x 4 waves each
 * issue iters x 96 MFMAs; data = 0: all-zero operands (no power limit on the pipe's data path), 1: operand-like bit patterns as above.  bench.py
 * reports both (`roofline.sustained_mfma`), with MfmaUtil from its counter pass. */
int straps_selftest_mfma_bf16_dense(float* out, unsigned long long* clk2, int blocks, int iters, int data, void* stream);
const char* straps_last_error(void);
/* number of visible HIP devices (0 => the product path must refuse to run) */
int straps_device_count(void);
/* measurement aid (bench.py `sclk_mhz`; no reference counterpart): with acc2 != NULL (a zeroed device pair of 64-bit counters),
 * workgroup 0 of every implicit-GEMM convolution (and matrix-pipe SMPL vertex kernel) launched AFTERWARDS (incl. launches captured into a hipGraph afterwards) adds the
 * shader-clock ticks and the constant-rate wall ticks of its lifetime to acc2[0] / acc2[1]; sustained shader clock in MHz =
 * acc2[0] / acc2[1] * straps_wall_clock_khz() / 1000.  NULL switches it off (the default: library use pays nothing).  The setting is
 * PER DEVICE (the current one); clear it before freeing the buffer, and do not replay hipGraphs captured with it afterwards.           */
int straps_wall_clock_khz(void);
int straps_set_clock_accumulator(unsigned long long* acc2);

/* ------------------------------------------------------------------------------------------
 * Encoder -- replaces nn.Conv2d / nn.BatchNorm2d / nn.ReLU / nn.MaxPool2d / AdaptiveAvgPool2d as
 * used by models/resnet.py:145-157 (stem), :61-77 (BasicBlock), :101-121 (Bottleneck), :201-216.
 * ------------------------------------------------------------------------------------------ */

/* OIHW [cout][cin][kh][kw] (PyTorch layout, models/resnet.py:30,36) -> KRSC [cout][kh][kw][cin]. */
int straps_pack_conv_weight(const float* w_oihw, float* w_krsc, int cout, int cin, int kh, int kw,
                            void* stream);
/* transposed copy for the data-gradient pass: OIHW -> [cin][kh][kw][cout] with the taps flipped
 * (dgrad of a conv is a conv with rotated filters).                                           */
int straps_pack_conv_weight_dgrad(const float* w_oihw, float* w_crsk, int cout, int cin, int kh,
                                  int kw, void* stream);
/* both packings of MANY conv layers in one launch (a training step repacks every layer after the
 * optimiser update: 2 x 19 / 2 x 52 tiny launches otherwise).  descs: DEVICE array of n entries sorted by
 * `first` (prefix sum of cout*cin*kh*kw); dst_crsk may be NULL (forward only); total = sum of elements. */
typedef struct {
    const float* src;          /* OIHW */
    float* dst_krsc;
    float* dst_crsk;
    int32_t o, c, r, s;
    long long first;
} straps_pack_desc_t;
int straps_pack_conv_weights_batched(const straps_pack_desc_t* descs, int n, long long total,
                                     void* stream);
/* the same launch for the bf16x3 route: the three bf16 planes [3][plane_stride] of the two packed layouts are written in the same
 * pass -- no split pass over the packed weights.  Planes are CHUNK-MAJOR (what straps_conv_fwd_x3 / straps_conv_dgrad_x3 read):
 * inside a layer's slot (element `first` ..) the weight of GEMM row `row`, tap `tap`, reduction index k sits at
 *     ((tap * (K / 32) + k / 32) * rows + row) * 32 + k % 32
 * with (row, k, rows, K) = (cout, cin, Cout, Cin) for the forward planes and (cin, cout, Cin, Cout) with flipped taps for the
 * data-gradient planes: the 64 bytes of one 32-wide K chunk of one row lie next to the neighbouring rows' (whole 128-byte lines
 * per LDS-DMA fetch).  A layer's planes are written only if its K is a multiple of 32 (others cannot run on this route).
 * Either plane buffer may be NULL; dst_krsc / dst_crsk of a descriptor may be NULL when only the planes are consumed.       */
int straps_pack_conv_weights_batched_x3(const straps_pack_desc_t* descs, int n, long long total,
                                        unsigned short* krsc_planes, unsigned short* crsk_planes,
                                        long long plane_stride, void* stream);

/* stem weights OIHW [64][cin][7][7] (models/resnet.py:145) -> MFMA fragment order
 * [ceil(cin*49/8)][2][64][4]; straps_stem_weight_floats gives the element count.              */
size_t straps_stem_weight_floats(int cin);
int straps_pack_stem_weight(const float* w_oihw, float* w_frag, int cin, void* stream);

/* BatchNorm eval-mode fold (models/resnet.py:147 in .eval()): scale = gamma/sqrt(var+eps),
 * shift = beta - mean*scale.                                                                  */
int straps_bn_fold(const float* gamma, const float* beta, const float* mean, const float* var,
                   float eps, float* scale, float* shift, int c, void* stream);
/* the same fold plus the statistics as the training-mode kernels take them (save_mean = mean, save_invstd = 1/sqrt(var+eps)):
 * an eval-mode BatchNorm that must back-propagate (nn.BatchNorm2d in .eval() with requires_grad parameters -- fine-tuning with
 * frozen statistics) runs the training-mode kernels on these vectors, with bit 1 of the backward entry points' `accumulate`. */
int straps_bn_fold_stats(const float* gamma, const float* beta, const float* mean, const float* var,
                         float eps, float* scale, float* shift, float* save_mean, float* save_invstd,
                         int c, void* stream);

/* diagnostics: with tile_cfg bit 5 set, straps_conv_fwd / straps_conv_dgrad run a build of the implicit-GEMM kernel whose
 * wave 0 of every workgroup writes 8 shader-clock sums (copy wait, barrier, copy issue, MFMA burst, chunks, total, prologue,
 * epilogue) to trace[workgroup][8] (device int64; NULL switches the writes off).  tools/igemm_trace.py.                 */
int straps_conv_trace_buffer(long long* trace);

/* conv7x7/s2/p3 over the NCHW network input, fused y = relu?(conv*scale+shift) -> NHWC.
 * scale/shift may be NULL (raw conv output, used in training mode).  If stats_partial != NULL it
 * receives per-block per-channel (sum, sum of squares) of the RAW conv output:
 * [straps_stem_stat_blocks()][64][2] floats (training-mode BatchNorm statistics).
 * ZERO SKIPPING (exact): the proxy input is a silhouette + truncated joint heat-maps, ~98 % exact
 * zeros.  The kernel contracts only the K groups whose input strips hold a non-zero (a skipped
 * group adds 0*w = 0), so the result equals the dense one bit for bit for finite weights; a dense
 * input skips nothing.  nzmask (optional, from straps_stem_nzmask; NULL = probe by loading) lets it
 * skip even the loads of all-zero 4x8 cells.                                                    */
int straps_stem_stat_blocks(int batch, int h, int w);
size_t straps_stem_nzmask_words(int batch, int cin, int h, int w);
/* one bit per 4-row x 8-column cell of x: mask[B][cin][ceil(h/4)][ceil(w/256)] uint32            */
int straps_stem_nzmask(const float* x_nchw, uint32_t* mask, int batch, int cin, int h, int w,
                       void* stream);
int straps_stem_fwd(const float* x_nchw, const float* w_frag, const float* scale,
                    const float* shift, int relu, float* y_nhwc, float* stats_partial,
                    const uint32_t* nzmask, int batch, int cin, int h, int w, void* stream);

/* generic implicit-GEMM convolution over NHWC fp32 (3x3 pad 1 stride 1|2, 1x1 stride 1|2):
 *   y[m][co] = epilogue( sum_{r,s,ci} x[b][ho*stride+r-pad][wo*stride+s-pad][ci] * w[co][r][s][ci] )
 * epilogue: *scale[co] + shift[co] (if scale), + residual[m][co] (if residual), relu (if relu).
 * cin % 32 == 0 and cout % 64 == 0 required.  stats_partial as for the stem:
 * [straps_conv_stat_blocks(...)][cout][2].  tile_cfg: 0 = auto, 1 = 128x128, 2 = 128x64,
 * 3 = 64x64 block tile.                                                                       */
int straps_conv_stat_blocks(int batch, int ho, int wo, int cout, int kdim /* kh*kw*cin */, int tile_cfg);
int straps_conv_fwd(const float* x_nhwc, const float* w_krsc, const float* scale,
                    const float* shift, const float* residual, int relu, float* y_nhwc,
                    float* stats_partial, int batch, int h, int w, int cin, int cout, int kh,
                    int kw, int stride, int pad, int tile_cfg, void* stream);

/* MaxPool2d(3, stride 2, pad 1) over NHWC (models/resnet.py:149); c % 4 == 0.                  */
int straps_maxpool_fwd(const float* x_nhwc, float* y_nhwc, int batch, int h, int w, int c,
                       void* stream);
/* AdaptiveAvgPool2d(1) + flatten over NHWC (models/resnet.py:157,213-214): [B,hw,c] -> [B,c].  */
int straps_gap_fwd(const float* x_nhwc, float* y, int batch, int hw, int c, void* stream);

/* training-mode BatchNorm (models/resnet.py:47 in .train()): reduce the per-block partials to
 * batch mean / biased variance, emit scale/shift for the apply pass and mean/invstd for backward,
 * and update running_mean / running_var (unbiased, momentum) in place.                         */
int straps_bn_stats_finalize(const float* stats_partial, int nblocks, int c, long long count,
                             const float* gamma, const float* beta, float eps, float momentum,
                             float* running_mean, float* running_var, float* scale, float* shift,
                             float* save_mean, float* save_invstd, void* stream);
/* y = relu?(x*scale[c] + shift[c] (+ residual)) elementwise over NHWC; n = B*H*W rows.          */
int straps_bn_apply(const float* x, const float* scale, const float* shift, const float* residual,
                    int relu, float* y, long long rows, int c, void* stream);

/* ------------------------------------------------------------------------------------------
 * IEF regressor -- replaces nn.Linear x3 x iterations (models/ief_module.py:16-18,48-64).
 * ------------------------------------------------------------------------------------------ */
/* out[m][n] = act( addend[m][n]? + bias[n]? + sum_k x[m][k] * w[n][k] ), fp32 MFMA.
 * x: [m][ldx] (k < kdim valid, ldx % 4 == 0, kdim % 8 == 0, padding columns must be zero),
 * w: [n_pad][ldw] rows n >= n are zero-padded up to a multiple of 32, out/addend: [m][ldo].
 * out may alias addend (in-place residual update of the estimate, models/ief_module.py:57).     */
int straps_linear_fwd(const float* x, int ldx, const float* w, int ldw, const float* bias,
                      const float* addend, float* out, int ldo, int m, int n, int kdim, int relu,
                      void* stream);
/* broadcast the initial estimate (models/ief_module.py:50-52): est[m][0..157) = init, pad = 0  */
int straps_broadcast_rows(const float* row, int cols, float* dst, int ld_dst, int m, void* stream);
/* the three repacked weight views the IEF kernels read, one launch per parameter update:
 * fc1.weight [h1][f + p] -> w1f [h1][f] (feature columns) and w1e [h1][ld_e] (estimate columns, zero padded to ld_e >= p);
 * fc3.weight [p][h2] -> w3 [p rounded up to 32][h2] (zero rows).  models/ief_module.py:16-18,54.                             */
int straps_ief_pack(const float* fc1_w, const float* fc3_w, float* w1f, float* w1e, float* w3, int f, int p,
                    int h1, int h2, int ld_e, void* stream);
/* up to STRAPS_GEMM_MULTI_MAX small strided fp32-MFMA GEMMs in ONE launch (backward of the IEF's nn.Linear layers,
 * models/ief_module.py:55-58 differentiated): per problem
 *     v[m][n] = sum_k a[m*sam + k*sak] * b[k*sbk + n*sbn]  (+ addend[m*ldadd + n])  (then 0 where mask[m*ldmask + n] <= 0)
 *     c[m*ldc + n] (+)= v ;  c2[m*ldc2 + n] (+)= v   (c2 optional)
 * The descriptors are HOST memory (copied into the kernel arguments).  With the three iterations' activations stacked
 * row-wise, a weight gradient is ONE problem with k = 3 x batch, a bias gradient the same with a = a single 1.0f (sam = sak = 0). */
#define STRAPS_GEMM_MULTI_MAX 8
typedef struct {
    const float* a; long long sam, sak;
    const float* b; long long sbk, sbn;
    float* c; int32_t ldc, accumulate;
    const float* addend; int32_t ldadd, reserved0;
    const float* mask; int32_t ldmask, reserved1;
    float* c2; int32_t ldc2, accumulate2;
    int32_t m, n, k, reserved2;
} straps_gemm_desc_t;
int straps_gemm_multi(const straps_gemm_desc_t* descs_host, int n, void* stream);

/* ------------------------------------------------------------------------------------------
 * Pose representation -- utils/rigid_transform_utils.py:27-41 and smplx.lbs.batch_rodrigues.
 * ------------------------------------------------------------------------------------------ */
/* x6 [rows][ld]: each row holds `per_row` 6-D rotations (interleaved a1x,a2x,a1y,a2y,a1z,a2z)
 * starting at column 0 -> R [rows*per_row][3][3] row-major (columns b1,b2,b3).                  */
int straps_rot6d_fwd(const float* x6, long long ld, int per_row, float* rotmats, long long rows,
                     void* stream);
/* axis-angle [n][3] -> R [n][3][3] (angle = ||r + 1e-8||).                                      */
int straps_rodrigues_fwd(const float* aa, float* rotmats, long long n, void* stream);
/* utils/cam_utils.py:5-26 orthographic_project_torch: points [batch][n][3], cam rows [s, tx, ty] (row stride ld_cam)
 * -> out [batch][n][2] = s * (x + tx, y + ty); and its gradient (dpoints [batch][n][3] and / or dcam [batch][3]; per-body
 * sums in a fixed order).                                                                          */
int straps_orthographic_project(const float* points, const float* cam, int ld_cam, float* out,
                                long long batch, int n, void* stream);
int straps_orthographic_project_bwd(const float* points, const float* cam, int ld_cam, const float* dout,
                                    float* dpoints, float* dcam, long long batch, int n, void* stream);
/* utils/cam_utils.py:40-71 perspective_project_torch: p = R x + t, p /= p_z, out = (K p)[:2]; rotation [batch][3][3],
 * translation [batch][3], cam_k one [3][3] (k_per_body = 0) or [batch][3][3] (k_per_body = 1) -> out [batch][n][2].           */
int straps_perspective_project(const float* points, const float* rotation, const float* translation,
                               const float* cam_k, int k_per_body, float* out, long long batch, int n,
                               void* stream);

/* ------------------------------------------------------------------------------------------
 * SMPL forward -- models/smpl_official.py:27-41 -> smplx.SMPL.forward -> smplx.lbs.lbs.
 * The model constants are packed ON THE HOST by the Python side (layout documented at
 * straps_smpl_model_t) and uploaded once.
 * ------------------------------------------------------------------------------------------ */
#define STRAPS_SMPL_V 6890
#define STRAPS_SMPL_VPAD 6912   /* 216 tiles of 32 vertices */
#define STRAPS_SMPL_TILES 216
#define STRAPS_SMPL_KP 224      /* 1 (template) + 10 (betas) + 207 (pose feats) + 6 zero pad */
#define STRAPS_SMPL_NJ 24
#define STRAPS_SMPL_NEXTRA 45   /* 9 extra + 19 cocoplus + 17 h36m regressed joints */
#define STRAPS_SMPL_NPICK 21    /* vertices appended by the vertex-joint selector */
#define STRAPS_SMPL_NJOINTS_OUT 90

typedef struct {
    /* blend directions in MFMA A-fragment order: [tile n_tiles][coord 3][kgroup 28][lane 64][4],
     * element = D[k = 8*g + 4*(lane>>5) + e][vertex = 32*tile + (lane&31)][coord], where D row 0
     * is v_template, rows 1..10 shapedirs[..., l], rows 11..217 posedirs, rest zero.
     * Tiles 0..215 are the 6890 mesh vertices; tiles 216..n_tiles-1 hold the VIRTUAL vertices that
     * carry the 45 sparse-regressed joints (see vj_ptr), zero padded to a multiple of 8 tiles.    */
    const float* blend_frag;
    const float* j_template;   /* [24][3]  J_regressor @ v_template (host fp64)              */
    const float* j_shapedirs;  /* [24][3][10] J_regressor @ shapedirs                          */
    const int32_t* parents;    /* [24] */
    const int32_t* depth;      /* [24] depth of each joint in the kinematic tree               */
    int32_t max_depth;
    int32_t skin_k;            /* non-zeros kept per vertex (4 for the real model)              */
    const float* skin_w;       /* [32*n_tiles][skin_k] */
    const int32_t* skin_j;     /* [32*n_tiles][skin_k] joint index of each weight               */
    /* Extra joints (J_regressor_extra | cocoplus | h36m, 45 rows) without a gather: regrouping
     *   joint_j = sum_v R[j,v] sum_k w[v,k] A_k.[v_posed_v;1]  by bone k gives one rigidly skinned
     * virtual vertex per (joint, bone) pair: blend directions sum_v R[j,v] w[v,k] D[:,v] / s and
     * skinning weight s = sum_v R[j,v] w[v,k] on bone k.  They ride through the same contraction
     * as extra tiles; joint j is the sum of virtual vertices [vj_ptr[j], vj_ptr[j+1]).           */
    const int32_t* vj_ptr;     /* [45 + 1] */
    int32_t n_tiles;           /* 216 + virtual tiles, multiple of 8                            */
    int32_t reserved0;
    const int32_t* pick_ids;   /* [21] vertex ids appended as joints 24..44                    */
    /* ---- tables used only by straps_smpl_bwd (may be NULL for forward-only use) ---- */
    /* transposed blend fragments [tile 216][coord 3][kblock 7][rq 4][lane 64][4]:
     * element = D[k = 32*kblock + (lane&31)][vertex = 32*tile + row(4*rq+e, lane>>5)][coord],
     * row(r,h) = (r&3) + 8*(r>>2) + 4*h (the MFMA C-layout row of accumulator register r).    */
    const float* blend_frag_t;
    const int32_t* children;   /* [24][3] child joints, -1 padded                              */
    /* joint-gradient sources per vertex tile: entries of tile t in [jrt_ptr[t], jrt_ptr[t+1]);
     * jrt_code = (v_local << 8) | src, src 0..20 = picked-vertex joints 24..44 (weight 1),
     * src 21..65 = regressed joints 45..89.                                                   */
    const int32_t* jrt_ptr;    /* [216 + 1] */
    const int32_t* jrt_code;
    const float* jrt_w;
    /* skinning weights regrouped for the joint-transform gradient: the non-zero weights of round rd (= 4 tiles of 32
     * vertices) on joint j live in [dj_ptr[rd*24 + j], dj_ptr[rd*24 + j + 1]), ascending vertex;
     * dj_code = (tile_in_round << 5) | vertex_in_tile.                                                           */
    const int32_t* dj_ptr;     /* [54*24 + 1] */
    const int32_t* dj_code;
    const float* dj_w;
    /* ---- split-precision forward (mode STRAPS_SMPL_SPLIT_F16; may be NULL / 0 otherwise) ----
     * fp16 two-term split of the blend directions scaled by a power of two S_D: [tile][kstep 14][coord 3][hi|lo][lane 64][8],
     * element = split(S_D * D[k = 16*kstep + 8*(lane>>5) + j][vertex = 32*tile + (lane&31)][coord]), hi = fp16(x),
     * lo = fp16(x - hi); blend_h_unscale = 1 / (S_D * 64) (the kernel scales the features by 64).               */
    const void* blend_frag_h;
    float blend_h_unscale;
    int32_t reserved1;
    /* ---- skinning on the matrix pipe (modes STRAPS_SMPL_SPLIT_F16_LBS*; may be NULL otherwise) ----
     * fp16 two-term split (hi, lo) of 2^14 * W, W = dense skinning weights [32*n_tiles][24] (the virtual vertices carry their single
     * weight), with the three products of the split packed along K:
     *   T = Ah.Wh + Al.Wh + Ah.Wl = [Ah | Al | Ah | -] . [Wh | Wh | Wl | 0]  over 24 + 24 + 24 + 8 = 80 columns = 5 k-steps of 16
     * (instead of 3 products x 2 k-steps of the 24 joints padded to 32): [tile][kstep 5][lane 64][8], element =
     * P[vertex = 32*tile + (lane&31)][col = 16*kstep + 8*(lane>>5) + j],  P = [hi[0:24] | hi[0:24] | lo[0:24] | 0 x 8].               */
    const void* skin_frag_p;
} straps_smpl_model_t;

/* kernel choice of the STRAPS_SMPL_SPLIT_F16_LBS* modes, OR-ed into `mode` (default 0 = by batch size): _WIDE = 64 bodies per workgroup,
 * one wave per SIMD with the whole 512-entry register file (every fetched direction fragment feeds two body groups, skinning
 * products K-packed; the large-batch kernel), _NARROW = 32 bodies per workgroup, two waves per SIMD (small batches).               */
#define STRAPS_SMPL_KERNEL_WIDE 0x100
#define STRAPS_SMPL_KERNEL_NARROW 0x200
/* _WIDE_BUILTIN (with _WIDE; round 6): the 64-body kernel with its skinning chains issued through the compiler's MFMA builtin (AGPR results, every
 * hazard resolved by the compiler) instead of the hand-placed VGPR-result assembly statements of the product form -- the reference instantiation the
 * product form is compared with bit for bit (tests/test_gpu_forward.py): same arithmetic, ~5 % slower.  Plain-fp16x3_lbs mode only.            */
#define STRAPS_SMPL_KERNEL_WIDE_BUILTIN 0x400
/* arithmetic of the blend contraction (v_template + shapedirs + posedirs, K = 218) in straps_smpl_fwd */
#define STRAPS_SMPL_EXACT_F32 0 /* fp32-input MFMA: exact fmaf chains (the reference's fp32 arithmetic)                  */
/* three fp16-MFMA products of two-term splits, fp32 accumulate: ~7e-7 relative per product at 16x the matrix rate;
 * features must satisfy |f| < 1023 (betas and R - I always do)                                                  */
#define STRAPS_SMPL_SPLIT_F16 1
/* as above, and the skinning transforms T[v][b] = sum_j W[v][j] A[b][j] as the same kind of split product on the matrix
 * pipe (any number of weights per vertex); |A| < 63 (metres)                                                     */
#define STRAPS_SMPL_SPLIT_F16_LBS 2
/* as STRAPS_SMPL_SPLIT_F16_LBS with the pose-corrective directions from column 16 on as plain fp16 (two products per term,
 * their low halves are not fetched): ~2^-12 of the pose-corrective displacement (measured: tests/test_gpu_forward.py),
 * inside north_star's 1e-4 m; template and shape directions keep the three-product split                          */
#define STRAPS_SMPL_SPLIT_F16_LBS_PD16 3
/* and the pose features of those columns as plain fp16 too (one product per term)                                  */
#define STRAPS_SMPL_SPLIT_F16_LBS_P16 4

/* bytes of caller-owned scratch for `batch` bodies (depends on the model's virtual-tile count)  */
size_t straps_smpl_workspace_bytes(const straps_smpl_model_t* model, long long batch);
/* verts [B][6890][3], joints [B][90][3] (may be NULL: vertices only).  betas [B][10],
 * rotmats [B][24][3][3] row-major.  chunks: split of the n_tiles/8 tile rounds over blocks, 0 = auto.
 * mode: STRAPS_SMPL_EXACT_F32 or STRAPS_SMPL_SPLIT_F16 (skinning and the joint chain are exact fp32 in both).   */
int straps_smpl_fwd(const straps_smpl_model_t* model, const float* betas, const float* rotmats,
                    float* verts, float* joints, void* workspace, long long batch, int chunks, int mode,
                    void* stream);

/* gradient of straps_smpl_fwd w.r.t. betas [B,10] and rotmats [B,24,3,3] given dverts [B,6890,3]
 * and/or djoints [B,90,3] (either may be NULL = zero) -- what autograd does through smplx.lbs for
 * pred_smpl_output in loss.backward() (train loop :196,:232).                                    */
size_t straps_smpl_bwd_workspace_bytes(long long batch, int chunks);
int straps_smpl_bwd(const straps_smpl_model_t* model, const float* betas, const float* rotmats,
                    const float* dverts, const float* djoints, float* dbetas, float* drotmats,
                    void* workspace, long long batch, int chunks, void* stream);

/* ------------------------------------------------------------------------------------------
 * Backward of the encoder / IEF / rot6d -- the work of loss.backward() (train loop :232) that
 * cuDNN / cuBLAS / ATen kernels do for the reference.
 * ------------------------------------------------------------------------------------------ */
/* data gradient of straps_conv_fwd (same geometry arguments as the forward; h,w = forward INPUT
 * size): dx = conv_transpose(dy, w) (+ addend, e.g. the skip-connection gradient).
 * w_crsk from straps_pack_conv_weight_dgrad.  cout % 32 == 0, cin % 64 == 0, stride 1 or 2.      */
int straps_conv_dgrad(const float* dy_nhwc, const float* w_crsk, const float* addend,
                      float* dx_nhwc, int batch, int h, int w, int cin, int cout, int kh, int kw,
                      int stride, int pad, int tile_cfg, void* stream);

/* ------------------------------------------------------------------------------------------
 * The same implicit-GEMM convolution on the bf16 matrix pipe at fp32 accuracy (csrc/conv_x3.hip):
 * every fp32 operand is carried as three bf16 planes x = x1 + x2 + x3 (exact, round-to-nearest
 * splits) and a product is the six bf16 products of weight >= 2^-16, accumulated in fp32 --
 * relative error <= ~2^-23 per product, the accuracy class of the fp32 chain, at 2.67x its
 * matrix-pipe rate.  planes: [3][plane_stride] bf16 bit patterns (uint16), plane_stride >= n,
 * a multiple of 8.  Geometry / epilogue arguments exactly as straps_conv_fwd /
 * straps_conv_dgrad; x3 / dy3 are the CHUNK-MAJOR planes of the NHWC tensor [rows = B*H*W][C]:
 * element (row, c) at ((c / 32) * rows + row) * 32 + c % 32 -- 32-channel chunks outermost, so the 64 bytes
 * one K chunk of a pixel contributes lie next to the neighbouring pixels' and an LDS-DMA fetch of a chunk
 * for 16 rows reads 1 KiB of whole 128-byte lines (plain NHWC planes made the same fetch touch half of each
 * of 16 lines: 16-18 vs 31-36 TB/s of useful L2 -> LDS bytes, tools/l2_line_probe.hip); written by
 * straps_split3_bf16_cm and the fused producers below.  w3: the chunk-major weight planes of
 * straps_pack_conv_weights_batched_x3.
 * tile_cfg: 0 = auto (incl. the halo-patch kernel for 3x3 / stride-1 layers: the tile's input patch is
 * copied once per channel chunk and the nine taps are shifted LDS addresses), 1..12 = explicit tiles
 * of the im2col kernel (8..12: software-pipelined loop; csrc/conv_x3.hip), + 256 = auto without the
 * halo-patch kernel, + 512 = the two-buffer halo-patch kernel wherever it applies, + 1024 = the single-buffer one (the rule for
 * 64-channel outputs); bits 6 / 7 select
 * measurement builds with wrong results (no MFMAs / no operand copies; tools/pmc_x3.sh).
 * ------------------------------------------------------------------------------------------ */
/* plain split of n values, planes in the order of x (any fp32 array; the exactness tests)                  */
int straps_split3_bf16(const float* x, unsigned short* planes, long long n, long long plane_stride,
                       void* stream);
/* split of an NHWC tensor [rows][c] (c % 32 == 0) into the chunk-major planes the convolutions read       */
int straps_split3_bf16_cm(const float* x, unsigned short* planes, long long rows, int c,
                          long long plane_stride, void* stream);
/* statistics partials of straps_conv_fwd_x3 for this geometry: [blocks][cout][2]                          */
int straps_conv_x3_stat_blocks(int batch, int h, int w, int cin, int cout, int kh, int kw, int stride,
                               int pad, int tile_cfg);

/* ---- 1x1 convolutions with the A operand read from the fp32 tensor (round 6; csrc/conv_x3f.hip) -------------------------------------
 * models/resnet.py:34-36 (conv1x1) as used by the bottleneck units :80-121 and the shortcut convolutions :172-177.  Same arithmetic as
 * the plane entry points above (three bf16 parts per fp32 value, six products per term, fp32 accumulate -- the two routes give the same
 * bits on the same values), but the kernel splits the activation / gradient itself: the producer writes (and this reads) 4 bytes per
 * element instead of 6-10, and the BatchNorm + ReLU in front of the convolution can ride in the operand path (a_scale / a_shift).       */
/* 1 if straps_conv_fwd_x3f / straps_conv_dgrad_x3f / straps_conv_wgrad_x3f cover the geometry (1x1, pad 0, stride 1 | 2, channels % 64) */
int straps_conv_x3f_supported(int cin, int cout, int kh, int kw, int stride, int pad);
/* y = conv1x1(act(x)) [* scale + shift] [+ residual] [relu]; x: fp32 NHWC [batch][h][w][cin].  a_scale / a_shift ([cin], 16-byte aligned;
 * both or neither): act(x) = x * a_scale + a_shift per channel, then ReLU if a_relu -- x is then the RAW output of the previous
 * convolution and the normalised activation is never materialised; NULL: act(x) = x.  w3: the weights' planes as for
 * straps_conv_fwd_x3.  stats_partial (training): [straps_conv_x3f_stat_blocks][cout][2] partial (sum, sum of squares) of y.            */
int straps_conv_fwd_x3f(const float* x, const float* a_scale, const float* a_shift, int a_relu, const unsigned short* w3,
                        long long w_plane_stride, const float* scale, const float* shift, const float* residual, int relu,
                        float* y, float* stats_partial, int batch, int h, int w, int cin, int cout, int kh, int kw, int stride,
                        int pad, int tile_cfg, void* stream);
int straps_conv_x3f_stat_blocks(int batch, int h, int w, int cin, int cout, int kh, int kw, int stride, int pad, int tile_cfg);
/* dx = dgrad1x1(dy) [+ addend, masked by addend_bits]; dy: fp32 [batch][ho][wo][cout] (no planes of it need exist).  Optional, as in
 * straps_conv_dgrad_x3_bn_bits: the two BatchNorm-backward sums of the BatchNorm (+ ReLU) whose output the convolution read (bn_raw !=
 * NULL; mask = bn_out_bits, else relu'(bn_raw * bn_mask_scale + bn_mask_shift)) into bn_partials [straps_conv_dgrad_x3f_bn_blocks][cin][2]. */
int straps_conv_dgrad_x3f(const float* dy, const unsigned short* w3_crsk, long long w_plane_stride, const float* addend,
                          const unsigned* addend_bits, float* dx, int batch, int h, int w, int cin, int cout, int kh, int kw,
                          int stride, int pad, int tile_cfg, const float* bn_raw, const unsigned* bn_out_bits,
                          const float* bn_mask_scale, const float* bn_mask_shift, const float* bn_mean, const float* bn_invstd,
                          double* bn_partials, void* stream);
int straps_conv_dgrad_x3f_bn_blocks(int batch, int h, int w, int cin, int cout, int kh, int kw, int stride, int pad, int tile_cfg);
/* dW (OIHW [cout][cin][1][1]) of a 1x1 convolution with BOTH operands read from the fp32 tensors (csrc/conv_wgrad_x3f.hip): x [batch][h][w][cin]
 * -- with a_scale / a_shift / a_relu as in straps_conv_fwd_x3f: x is then the RAW output of the previous convolution, the producer's
 * BatchNorm (+ ReLU) is applied in the operand path -- and dy [batch][ho][wo][cout].  workspace: straps_conv_wgrad_x3f_workspace_bytes
 * (split-K partials); accumulate: dW += instead of =.  Same six-product bf16 arithmetic as straps_conv_wgrad_x3.                        */
size_t straps_conv_wgrad_x3f_workspace_bytes(int batch, int h, int w, int cin, int cout, int kh, int kw, int stride, int pad);
int straps_conv_wgrad_x3f(const float* x, const float* a_scale, const float* a_shift, int a_relu, const float* dy, float* dw_oihw,
                          void* workspace, int batch, int h, int w, int cin, int cout, int kh, int kw, int stride, int pad,
                          int accumulate, void* stream);
int straps_conv_fwd_x3(const unsigned short* x3, long long x_plane_stride,
                       const unsigned short* w3_krsc, long long w_plane_stride, const float* scale,
                       const float* shift, const float* residual, int relu, float* y_nhwc,
                       float* stats_partial, int batch, int h, int w, int cin, int cout, int kh,
                       int kw, int stride, int pad, int tile_cfg, void* stream);
/* straps_conv_fwd_x3 (no statistics) that also writes its result as planes -- y_nhwc may be NULL when only they are consumed   */
int straps_conv_fwd_x3p(const unsigned short* x3, long long x_plane_stride,
                        const unsigned short* w3_krsc, long long w_plane_stride, const float* scale,
                        const float* shift, const float* residual, int relu, float* y_nhwc,
                        unsigned short* y_planes, long long y_plane_stride, int batch, int h, int w,
                        int cin, int cout, int kh, int kw, int stride, int pad, int tile_cfg, void* stream);
int straps_conv_dgrad_x3(const unsigned short* dy3, long long dy_plane_stride,
                         const unsigned short* w3_crsk, long long w_plane_stride,
                         const float* addend, float* dx_nhwc, int batch, int h, int w, int cin,
                         int cout, int kh, int kw, int stride, int pad, int tile_cfg, void* stream);
/* producers of the bf16x3 route that write their fp32 output AND its three planes in one pass (instead of a
 * straps_split3_bf16 pass over the output): straps_bn_apply / straps_bn_relu_maxpool_fwd / straps_bn_bwd with
 * two more arguments (planes [3][plane_stride], plane_stride >= element count, a multiple of 8; draw_planes
 * may be NULL).  The fp32 output of straps_bn_apply_x3 (y) and of straps_bn_bwd_x3 (draw) may be NULL when only
 * bf16x3 kernels consume it: 4 of the 10 bytes per element are then not written.                            */
int straps_bn_apply_x3(const float* raw, const float* scale, const float* shift, const float* residual,
                       int relu, float* y, unsigned short* y_planes, long long plane_stride,
                       long long rows, int c, void* stream);
int straps_bn_relu_maxpool_fwd_x3(const float* raw, const float* scale, const float* shift,
                                  float* y_pool, uint8_t* idx, unsigned short* y_planes,
                                  long long plane_stride, int batch, int h, int w, int c, void* stream);
int straps_bn_bwd_x3(const float* dy, const float* yact, const float* raw, const float* save_mean,
                     const float* save_invstd, const float* gamma, const float* mask_scale,
                     const float* mask_shift, float* dgamma, float* dbeta, float* draw, float* dz_out,
                     unsigned short* draw_planes, long long plane_stride, void* workspace,
                     long long rows, int c, int accumulate, void* stream);
/* The data gradient with the sums of the NEXT BatchNorm backward fused into its epilogue: dx (= the gradient dy entering the
 * BatchNorm + ReLU that produced this convolution's input) is written as by straps_conv_dgrad_x3, and per M tile one partial
 * bn_partials[block][cin][2] = (S1, invstd * S2) of  S1 = sum mask*dy,  S2 = sum mask*dy*(raw - mean)  (double; mask = bn_out > 0
 * if bn_out, else fma(raw, bn_mask_scale, bn_mask_shift) > 0) -- the reduction pass of straps_bn_bwd_x3 over (dy, raw) is not
 * needed: straps_bn_bwd_finish_x3 takes the partials (blocks from straps_conv_dgrad_x3_bn_blocks; workspace: 2 c doubles + c floats). */
int straps_conv_dgrad_x3_bn_blocks(int batch, int h, int w, int cin, int cout, int kh, int kw, int stride,
                                   int pad, int tile_cfg);
int straps_conv_dgrad_x3_bn(const unsigned short* dy3, long long dy_plane_stride,
                            const unsigned short* w3_crsk, long long w_plane_stride, const float* addend,
                            float* dx_nhwc, int batch, int h, int w, int cin, int cout, int kh, int kw,
                            int stride, int pad, int tile_cfg, const float* bn_raw, const float* bn_out,
                            const float* bn_mask_scale, const float* bn_mask_shift, const float* bn_mean,
                            const float* bn_invstd, double* bn_partials, void* stream);
int straps_bn_bwd_finish_x3(const float* dy, const float* yact, const float* raw, const float* save_mean,
                            const float* save_invstd, const float* gamma, const float* mask_scale,
                            const float* mask_shift, float* dgamma, float* dbeta, float* draw, float* dz_out,
                            unsigned short* draw_planes, long long plane_stride, const double* partials,
                            int nblk, void* workspace, long long rows, int c, int accumulate, void* stream);
/* ReLU decisions as BITS (ABI 8).  The backward pass of a residual unit needs the unit's fp32 activation only for its sign:
 * as the ReLU mask of the last BatchNorm's two passes, of the sums fused into the data gradient that produces dy, and of the gradient
 * the skip connection receives (which the fp32 form materialises as dz_out).  straps_bn_apply_bits_x3 (= straps_bn_apply_x3 with
 * relu = 1) also writes  relu_bits[rows][c / 32]:  bit (ch & 31) of word [row][ch / 32] = (y[row][ch] > 0);  the *_bits forms below
 * read those words where the plain forms read an fp32 tensor:
 *   straps_bn_bwd_bits_x3 / straps_bn_bwd_finish_bits_x3   relu_bits instead of yact; no dz_out (consumers mask dy themselves);
 *   straps_conv_dgrad_x3_bits / _bn_bits                   addend_bits: dx = dgrad + (bit ? addend : 0), the addend being the UNMASKED
 *                                                          gradient of the later unit's output;  bn_out_bits instead of bn_out
 *                                                          (either may be NULL: that operand is then used as in the plain form).
 * Results are bit-identical to the fp32-mask forms (models/resnet.py:72-74,115-117: out += identity; out = relu(out)).
 * c (cin for the data gradients) must be a multiple of 32. */
int straps_bn_apply_bits_x3(const float* raw, const float* scale, const float* shift, const float* residual, float* y,
                            unsigned short* y_planes, long long plane_stride, unsigned* relu_bits, long long rows,
                            int c, void* stream);
int straps_bn_bwd_bits_x3(const float* dy, const unsigned* relu_bits, const float* raw, const float* save_mean,
                          const float* save_invstd, const float* gamma, float* dgamma, float* dbeta, float* draw,
                          unsigned short* draw_planes, long long plane_stride, void* workspace, long long rows, int c,
                          int accumulate, void* stream);
int straps_bn_bwd_finish_bits_x3(const float* dy, const unsigned* relu_bits, const float* raw, const float* save_mean,
                                 const float* save_invstd, const float* gamma, float* dgamma, float* dbeta, float* draw,
                                 unsigned short* draw_planes, long long plane_stride, const double* partials, int nblk,
                                 void* workspace, long long rows, int c, int accumulate, void* stream);
int straps_conv_dgrad_x3_bits(const unsigned short* dy3, long long dy_plane_stride, const unsigned short* w3_crsk,
                              long long w_plane_stride, const float* addend, float* dx_nhwc, int batch, int h, int w,
                              int cin, int cout, int kh, int kw, int stride, int pad, int tile_cfg,
                              const unsigned* addend_bits, void* stream);
int straps_conv_dgrad_x3_bn_bits(const unsigned short* dy3, long long dy_plane_stride,
                                 const unsigned short* w3_crsk, long long w_plane_stride, const float* addend,
                                 float* dx_nhwc, int batch, int h, int w, int cin, int cout, int kh, int kw,
                                 int stride, int pad, int tile_cfg, const float* bn_raw, const float* bn_out,
                                 const float* bn_mask_scale, const float* bn_mask_shift, const float* bn_mean,
                                 const float* bn_invstd, double* bn_partials, const unsigned* addend_bits,
                                 const unsigned* bn_out_bits, void* stream);
/* weight gradient on the bf16x3 route: straps_conv_wgrad's arguments plus the planes of x and dy.  3x3 / stride 1 / pad 1
 * layers (power-of-two width >= 8) run the halo-patch kernel on the planes (ds_read_b64_tr_b16 operand gathers); the other
 * 3x3 layers and the 1x1 layers whose channel counts are both >= 128 run the per-tap kernel on the planes; what is left
 * (1x1 with a 64-channel side) -- or NULL planes -- uses the fp32 kernels on (x, dy).  Workspace as for straps_conv_wgrad.   */
/* 1 if straps_conv_wgrad_x3 handles this geometry on the planes alone (x_nhwc / dy_nhwc may then be NULL, and the producers
 * -- straps_bn_apply_x3's y, straps_bn_bwd_x3's draw -- need not write their fp32 copies), 0 if it needs the fp32 tensors.      */
int straps_conv_wgrad_x3_on_planes(int batch, int h, int w, int cin, int cout, int kh, int kw,
                                   int stride, int pad);
int straps_conv_wgrad_x3(const float* x_nhwc, const float* dy_nhwc, const unsigned short* x3,
                         long long x_plane_stride, const unsigned short* dy3, long long dy_plane_stride,
                         float* dw_oihw, void* workspace, int batch, int h, int w, int cin, int cout,
                         int kh, int kw, int stride, int pad, int accumulate, void* stream);


/* weight gradient in the parameter's own OIHW layout: dw (+)= sum_pixels dy (x) x.                */
size_t straps_conv_wgrad_workspace_bytes(int batch, int h, int w, int cin, int cout, int kh,
                                         int kw, int stride, int pad);
int straps_conv_wgrad(const float* x_nhwc, const float* dy_nhwc, float* dw_oihw, void* workspace,
                      int batch, int h, int w, int cin, int cout, int kh, int kw, int stride,
                      int pad, int accumulate, void* stream);
/* stem weight gradient straight from the NCHW input (the input itself needs no gradient).        */
size_t straps_stem_wgrad_workspace_bytes(int batch, int cin, int h, int w);
int straps_stem_wgrad(const float* x_nchw, const float* dy_nhwc, float* dw_oihw, void* workspace,
                      const uint32_t* nzmask /* optional, see straps_stem_fwd */, int batch, int cin,
                      int h, int w, int accumulate, void* stream);
/* training-mode BatchNorm backward with the ReLU mask fused: dz = dy * (yact > 0) (yact NULL = no
 * ReLU), dgamma/dbeta, draw = gamma*invstd*(dz - mean(dz) - xhat*mean(dz*xhat)); dz_out (optional,
 * may alias dy) receives dz for the skip connection.  When the activation was exactly
 * relu(raw*scale + shift) (no residual), pass yact = NULL and the forward's scale/shift as
 * mask_scale/mask_shift: the mask is then recomputed from raw with the forward's fmaf (bit-identical,
 * one tensor read less); both NULL with yact NULL = no ReLU.
 * `accumulate` (here and in straps_bn_bwd_x3 / _finish_x3 / _pooled[_sparse]) is a flag word: bit 0 = add to dgamma / dbeta,
 * bit 1 = FROZEN statistics (eval-mode BatchNorm, models/resnet.py:47 under .eval(): save_mean / save_invstd are constants, the
 * two mean terms vanish: draw = gamma*invstd*dz; dgamma = sum dz*xhat and dbeta = sum dz as before).                          */
int straps_bn_bwd_blocks(long long rows, int c);
size_t straps_bn_bwd_workspace_bytes(long long rows, int c);
int straps_bn_bwd(const float* dy, const float* yact, const float* raw, const float* save_mean,
                  const float* save_invstd, const float* gamma, const float* mask_scale,
                  const float* mask_shift, float* dgamma, float* dbeta, float* draw, float* dz_out,
                  void* workspace, long long rows, int c, int accumulate, void* stream);
/* The stem's training-mode tail fused (models/resnet.py:146-149 bn1 -> relu -> maxpool, whose activation feeds nothing but
 * the pool): straps_bn_relu_maxpool_fwd = straps_bn_apply(relu, no residual) + straps_maxpool_fwd_idx without ever writing
 * the activation; straps_bn_bwd_pooled = straps_maxpool_bwd + straps_bn_bwd(mask from raw) without ever writing the
 * un-pooled gradient (each element gathers it from the <= 4 windows covering it).  Same arithmetic in the same order as the
 * unfused calls: bit-identical results, 1.3 GB less HBM traffic per B=64 step.  workspace: straps_bn_bwd_workspace_bytes(
 * batch*h*w, c).                                                                                   */
int straps_bn_relu_maxpool_fwd(const float* raw_nhwc, const float* scale, const float* shift,
                               float* y_pool_nhwc, uint8_t* idx, int batch, int h, int w, int c,
                               void* stream);
int straps_bn_bwd_pooled(const float* dy_pool_nhwc, const uint8_t* idx, const float* raw,
                         const float* save_mean, const float* save_invstd, const float* gamma,
                         const float* mask_scale, const float* mask_shift, float* dgamma, float* dbeta,
                         float* draw, void* workspace, int batch, int h, int w, int c, int accumulate,
                         void* stream);
/* straps_bn_bwd_pooled that leaves unwritten what nobody reads: the only consumer of `draw` is straps_stem_wgrad, which visits a
 * 2-row x 32-column tile of it only if some input channel has a non-zero under the tile (its 9 x 72 input patch).
 * straps_stem_tile_activity derives that map (straps_stem_tiles(batch, in_h, in_w) bytes, 1 = read) from the input's non-zero
 * bit map (straps_stem_nzmask); tile_active == NULL writes everything.  dgamma / dbeta are sums over the whole tensor either way;
 * on the proxy representation ~40 % of the tiles are never read (tools/stem_tile_activity.py).                                  */
size_t straps_stem_tiles(int batch, int in_h, int in_w);
int straps_stem_tile_activity(const uint32_t* nzmask, uint8_t* tile_active, int batch, int cin, int in_h,
                              int in_w, void* stream);
int straps_bn_bwd_pooled_sparse(const float* dy_pool_nhwc, const uint8_t* idx, const float* raw,
                                const float* save_mean, const float* save_invstd, const float* gamma,
                                const float* mask_scale, const float* mask_shift, float* dgamma,
                                float* dbeta, float* draw, void* workspace, int batch, int h, int w, int c,
                                int accumulate, const uint8_t* tile_active, void* stream);
/* max-pool forward that also records the arg-max tap (uint8 per element), and its backward.      */
int straps_maxpool_fwd_idx(const float* x_nhwc, float* y_nhwc, uint8_t* idx, int batch, int h,
                           int w, int c, void* stream);
int straps_maxpool_bwd(const float* dy_nhwc, const uint8_t* idx, float* dx_nhwc, int batch, int h,
                       int w, int c, void* stream);
int straps_gap_bwd(const float* dfeat, float* dx_nhwc, int batch, int hw, int c, void* stream);
/* y[m][n] (+)= x[m][n] * (mask[m][n] > 0)        (ReLU backward on small matrices)                */
int straps_masked_copy(const float* x, int ldx, const float* mask, int ldmask, float* y, int ldy,
                       int m, int n, int accumulate, void* stream);
/* gradient of straps_rot6d_fwd: drot [rows*per_row][3][3] -> dx6 [rows][ldd].                    */
int straps_rot6d_bwd(const float* x6, long long ld, int per_row, const float* drot, float* dx6,
                     long long ldd, long long rows, void* stream);

/* ------------------------------------------------------------------------------------------
 * Train-step glue: input construction, prediction heads + multi-task loss, Adam.
 * ------------------------------------------------------------------------------------------ */
/* utils/label_conversions.py:48-55 + :90-127 + train loop :178-182 in ONE pass: seg [B,H,W] part
 * ids -> channel 0 = (seg != 0), channels 1..nj = 16x16 truncated Gaussian heat-maps of the
 * (int-truncated) joints2d [B,nj,2]; writes the NCHW network input [B,1+nj,wh,wh].                */
int straps_build_proxy_input(const float* seg, const float* joints2d, float* out_nchw, int batch,
                             int nj, int wh, void* stream);
/* the same with the heat-maps' standard deviation as an argument (utils/label_conversions.py:90: `std`, an integer; the patch is
 * 4 std x 4 std samples of torch.linspace(-2 std, 2 std, 4 std)); straps_build_proxy_input is std = 4, the value of every call site. */
int straps_build_proxy_input_std(const float* seg, const float* joints2d, float* out_nchw, int batch,
                                 int nj, int wh, int std, void* stream);
/* the same, and in the same pass the non-zero bit map of the planes it writes: nzmask[straps_stem_nzmask_words(batch, nj + 1, wh, wh)] ==
 * what straps_stem_nzmask(out_nchw, ...) would read back from them (the 302 MB read of that pass at 64 bodies is not made; cells that cannot
 * intersect a joint's window are written as zeros without evaluating the Gaussian).  wh % 8 == 0. */
int straps_build_proxy_input_nz(const float* seg, const float* joints2d, float* out_nchw, uint32_t* nzmask,
                                int batch, int nj, int wh, int std, void* stream);
/* prediction heads + HomoscedasticUncertaintyWeightedMultiTaskLoss (losses/multi_task_loss.py:76-119,
 * reduction 'mean') fused with its own backward.  From pred joints [B,90,3], cam [B,3] (row
 * stride ld_est) it forms joints2D = orthographic projection of the 17 COCO joints
 * (utils/cam_utils.py:5-26, config.py:27) and joints3D = 14 H36M-LSP joints (config.py:28-32);
 * visibility of the target 2D joints per utils/joints2d_utils.py:23-32.
 * loss_out[0] = total, [1..5] = weighted task losses (verts, joints2D, joints3D, shape, pose),
 * [6..10] = raw MSEs, [11] = number of visible joints.  Gradients: dverts [B,6890,3],
 * djoints [B,90,3], dest [B,ld_est] (cam cols 0..2, shape cols 147..156, rest zero),
 * drot [B,24,3,3], dlogvar[5] (order verts, joints2D, joints3D, shape_params, pose_params).        */
size_t straps_loss_workspace_bytes(long long batch);
int straps_loss_fwd_bwd(const float* pred_verts, const float* pred_joints, const float* est,
                        int ld_est, const float* pred_rot, const float* tgt_verts,
                        const float* tgt_joints2d, const float* tgt_joints3d,
                        const float* tgt_shape, const float* tgt_rot, const float* log_vars,
                        float* loss_out, float* dverts, float* djoints, float* dest, float* drot,
                        float* dlogvar, void* workspace, long long batch, int img_wh, void* stream);
/* data parallel with the GLOBAL visibility-masked mean: the joints2D task's denominator becomes
 * 2 * j2d_count_global[0] * count_scale (device scalar = the job's visible-joint count, count_scale = 1 / world size) instead of
 * this rank's own count -- the average over ranks of loss_out and of every gradient (what the sum all-reduce + 1/world of the
 * step computes) is then the masked mean over the global batch, whatever the split of visible joints between ranks.
 * j2d_count_global = NULL: straps_loss_fwd_bwd.  straps_count_visible gives a rank's own count (as a float, to be sum all-reduced). */
int straps_loss_fwd_bwd_gm(const float* pred_verts, const float* pred_joints, const float* est, int ld_est,
                           const float* pred_rot, const float* tgt_verts, const float* tgt_joints2d,
                           const float* tgt_joints3d, const float* tgt_shape, const float* tgt_rot,
                           const float* log_vars, float* loss_out, float* dverts, float* djoints, float* dest,
                           float* drot, float* dlogvar, void* workspace, long long batch, int img_wh,
                           const float* j2d_count_global, float count_scale, void* stream);
int straps_count_visible(const float* tgt_joints2d, float* out_count, long long batch, int nj, int img_wh,
                         void* stream);
/* row-masked mean-squared error used by the drop-in criterion module (losses/multi_task_loss.py:78-112):
 * out3 = {sum of squares, kept element count, mean}; the target is read as tgt*tgt_scale + tgt_shift
 * (the 2x/256-1 normalisation of :92); row_mask (uint8 per row, may be NULL) is labels['vis'].
 * workspace: 512 floats.  straps_mse_bwd: grad = coef[0] * (pred - target) on kept rows.             */
int straps_mse_fwd(const float* pred, const float* tgt, const uint8_t* row_mask, long long rows,
                   int cols, float tgt_scale, float tgt_shift, float* out3, void* workspace,
                   void* stream);
int straps_mse_bwd(const float* pred, const float* tgt, const uint8_t* row_mask, long long rows,
                   int cols, float tgt_scale, float tgt_shift, const float* coef, float* grad,
                   void* stream);
/* random_remove_bodyparts + random_occlude (augmentation/proxy_rep_augmentation.py:52-101) in one
 * pass over the part-id segmentation [B,wh,wh]: uniforms [B][9] in [0,1) (6 part-removal draws,
 * 1 occlusion draw, 2 box-centre draws), remove_prob[6] device array.                              */
int straps_augment_seg(const float* seg, const float* uniforms, const float* remove_prob,
                       float occlude_prob, int box_dim, float* out, int batch, int wh, void* stream);
/* ------------------------------------------------------------------------------------------
 * Random draws + the augmentations that consume them (train loop :121-129,146-151,173-175).  The reference mixes
 * torch's device generator and numpy's host generator; here ONE counter-based generator (Philox4x32-10, 10 rounds,
 * key = seed, counter = {index/4, step, sub-stream}) fills draw buffers on the device.  `step_dev` (optional device
 * int64) overrides step_host so a replayed hipGraph advances the sequence with straps_counter_add.
 * Range / scale parameters of the augmentations are doubles -- the reference's Python scalars: (h - l) is formed in double
 * and rounded once to the fp32 scalar its tensor expression uses, so results are bit-identical given the draws.
 * kind 0: uniform [0,1) = (x >> 8) * 2^-24;  kind 1: standard normal (Box-Muller on consecutive pairs).
 * ------------------------------------------------------------------------------------------ */
int straps_philox_fill(unsigned long long seed, const long long* step_dev, long long step_host,
                       unsigned substream, float* out, long long n, int kind, void* stream);
/* counters[i] += delta for i < n (device int64: the generator's step, Adam's step, the BatchNorm num_batches_tracked row) */
int straps_counter_add(long long* counters, int n, long long delta, void* stream);
/* dst[i] = src[index[i]], i < n (device index array): the step keeps the five loss log-variances in the kernel's task
 * order while the parameters sit in the criterion's registration order (losses/multi_task_loss.py:46-55).            */
int straps_gather_f32(const float* src, const int* index, float* dst, int n, void* stream);
/* hipMemsetAsync(ptr, 0, bytes) on `stream` (the step zeroes its flat gradient buffer with it) */
int straps_memset_zero(void* ptr, size_t bytes, void* stream);
/* augment_smpl (augmentation/smpl_augmentation.py:27-61) + the dataset gather in front of it:
 * body b takes pose_rows[b] ([n_rows][72] axis-angle, global orientation first), or with u_index [B] in [0,1) the
 * row floor(u * n_rows) of a resident pose pool (stand-in for data/synthetic_training_dataset.py);
 * out_rotmats [B][24][3][3] = batch_rodrigues of it (joint 0 = glob_rotmats, 1..23 = pose_rotmats);
 * out_shape [B][10]: shape_mode 0 = orig_shape, 1 = shape_draws * std_vector + mean_shape (normal_sample_shape :18-25,
 * shape_draws ~ N(0,1)), 2 = (range_hi - range_lo) * shape_draws + range_lo + mean_shape (uniform_sample_shape :6-15,
 * shape_draws ~ U[0,1)).  out_pose (optional [B][72]) receives the gathered axis-angle rows.                   */
int straps_augment_smpl(const float* pose_rows, long long n_rows, const float* u_index,
                        const float* orig_shape, const float* mean_shape, const float* shape_draws,
                        int shape_mode, const float* std_vector, double range_lo, double range_hi,
                        float* out_shape, float* out_rotmats, float* out_pose, long long batch,
                        void* stream);
/* augment_cam_t (augmentation/cam_augmentation.py:4-14): out[:, :2] = mean[:, :2] + normals_xy [B][2] * xy_std,
 * out[:, 2] = mean[:, 2] + (z_hi - z_lo) * uniform_z [B] + z_lo.                                               */
int straps_augment_cam_t(const float* mean_cam_t, const float* normals_xy, const float* uniform_z,
                         double xy_std, double z_lo, double z_hi, float* out_cam_t, long long batch,
                         void* stream);
/* random_joints2D_deviation (augmentation/proxy_rep_augmentation.py:25-49) on [B][17][2] COCO joints:
 * out = joints + (hi - lo) * u + lo, the hip joints 11 and 12 with their own range; uniforms [B][17][2].        */
int straps_deviate_joints2d(const float* joints2d, const float* uniforms, double lo, double hi,
                            double hip_lo, double hip_hi, float* out, long long batch, void* stream);
/* random_verts2D_deviation (augmentation/proxy_rep_augmentation.py:5-22) as a materialised copy: out [n][3] = verts with
 * x,y += (hi - lo) * u + lo, uniforms [n][2].  (The training step does not call this: straps_rasterize_parts applies the
 * same noise inside its projection kernel.)                                                                     */
int straps_deviate_verts2d(const float* verts, const float* uniforms, double lo, double hi, float* out,
                           long long nverts_total, void* stream);
/* target-side heads of the train step (train loop :138-143): joints3d [B,14,3] = H36M-LSP subset of
 * joints [B,90,3]; joints2d [B,17,2] = perspective projection of the COCO subset with identity
 * rotation, translation cam_t [B,3] and intrinsics fx,fy,cx,cy (utils/cam_utils.py:40-71).         */
int straps_project_targets(const float* joints, const float* cam_t, float fx, float fy, float cx,
                           float cy, float* joints2d, float* joints3d, long long batch, void* stream);
/* renderers/nmr_renderer.py:84-100 (NMRRenderer.forward, rend_parts_seg=True; train loop :155): body-part id per pixel.
 * neural_renderer's 'projection' camera (x_cam = R v + t, pin-hole K, NDC, flipped v axis), pixel-centre sampling,
 * two-sided faces, nearest depth in (near, far), lower face id on ties, final vertical flip; the texture + cube_parts
 * decode of get_parts is the per-face table face_parts (0..255).  verts [B][nverts][3], faces [nfaces][3],
 * cam_K / cam_R [3][3] shared (cam_per_body = 0) or [B][3][3], cam_t [B][3].  parts [B][wh][wh] float part ids
 * (0 = background) and/or depth [B][wh][wh] (`far` where empty); either may be NULL.  Deterministic.             */
size_t straps_rasterize_workspace_bytes(long long batch, int nverts, int wh);
int straps_rasterize_parts(const float* verts, const int32_t* faces, const uint8_t* face_parts,
                           const float* cam_K, const float* cam_R, const float* cam_t, float* parts,
                           float* depth, void* workspace, long long batch, int nverts, int nfaces, int wh,
                           int cam_per_body, float near, float far, const float* vert_noise_u, double noise_lo,
                           double noise_hi, void* stream);
/* (vert_noise_u: NULL, or uniforms [B][nverts][2] in [0,1): the rendered copy of the mesh gets x,y += (hi-lo)*u + lo --
 *  random_verts2D_deviation, augmentation/proxy_rep_augmentation.py:5-22, train loop :146-151 -- inside the projection
 *  kernel; the caller's vertices, i.e. the loss targets, are not touched.)                                        */
/* On-device bounding-box crop + nearest-neighbour resize (utils/image_utils.py:44-105,
 * train loop :161-170): per sample, box of the non-zero pixels of seg [B,wh,wh] -> centre / max(h,w) * scale
 * (scale = orig_scale_factor + U(delta_scale), centre += U(delta_centre); uniforms [B][3] in [0,1), NULL = no
 * jitter) -> int16-truncated corners -> crop -> resize to out_wh with cv2.INTER_NEAREST index rule; joints are
 * shifted by the top-left corner and scaled by out_wh / crop size.  boxes [B][6] receives {r0,c0,r1,c1,shift_r,shift_c}.
 * The scale / range parameters are doubles: the box arithmetic is the reference's float64 arithmetic on the given draws.  */
int straps_crop_resize(const float* seg, const float* joints2d, const float* uniforms,
                       double orig_scale_factor, double delta_scale_lo, double delta_scale_hi,
                       double delta_centre_lo, double delta_centre_hi, float* out_seg,
                       float* out_joints2d, int* boxes, int batch, int wh, int out_wh, int nj,
                       void* stream);
/* On-device evaluation metrics (metrics/train_loss_and_metrics_tracker.py:127-197 +
 * utils/eval_utils.py:7-85): for each sample b, out3[b] = { sum_n |p-t|,
 * sum_n |scale_and_translation_transform(p) - t|, sum_n |procrustes(p) - t| } over npoints 3-D points
 * (6890 vertices -> PVE / PVE-SC / PVE-PA sums, 14 joints -> MPJPE / -SC / -PA sums).                */
int straps_point_metrics(const float* pred, const float* target, float* out3, long long batch,
                         int npoints, void* stream);
/* torch.optim.Adam defaults (run_train.py:200-201) over one flat fp32 buffer:
 * m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2; p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps);
 * grad_scale multiplies g first (1/world_size after a sum all-reduce).  step_dev (optional device
 * int64) overrides `step` so that the step count can live on the device (straps_counter_add).       */
int straps_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                     long long n, int step, float lr, float beta1, float beta2, float eps,
                     float grad_scale, const long long* step_dev, void* stream);

/* ---- gradient exchange of data-parallel training -----------------------------------------------
 * The reference trains on one GPU (run_train.py:23-26) and so has no counterpart; north_star asks for "a single RCCL all-reduce of
 * grads over xGMI per step".  These entry points give a host WITHOUT torch that exchange: the flat fp32 gradient buffer every
 * backward kernel of this library writes into (one buffer, straps_adam_step's `grads`) is summed over the ranks in place.
 * RCCL is resolved at run time: the copy already loaded in the process (a torch host's) is preferred so that communicator handles
 * created by the host are valid here; otherwise librccl.so.1 is loaded.  STRAPS_EUNSUPPORTED when no RCCL can be found.
 *   straps_comm_unique_id : rank 0 generates the 128-byte rendezvous id (ncclGetUniqueId); the host ships it to the other ranks.
 *   straps_comm_init_rank : every rank joins (ncclCommInitRank; collective, blocks until all `nranks` have called it).  One
 *                           communicator per process per GPU; the current HIP device is the one it binds to.
 *   straps_allreduce_grads: in-place sum all-reduce of n floats, enqueued on `stream` (asynchronous like every kernel launch here;
 *                           calls on one communicator must be issued in the same order on every rank).  `comm` may equally be
 *                           an ncclComm_t the host created itself with the same RCCL library.
 *   straps_comm_size      : number of ranks of the communicator (0 on error);  straps_comm_library: which librccl is in use.  */
#define STRAPS_COMM_ID_BYTES 128
int straps_comm_unique_id(void* id128);
int straps_comm_init_rank(const void* id128, int nranks, int rank, void** comm);
int straps_comm_destroy(void* comm);
int straps_comm_size(void* comm);
const char* straps_comm_library(void);
int straps_allreduce_grads(float* flat_g, long long n, void* comm, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* STRAPS_HIP_H */
