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

#define STRAPS_ABI_VERSION 1

#define STRAPS_OK 0
#define STRAPS_EINVAL 1       /* bad argument (shape, alignment, null pointer) */
#define STRAPS_EHIP 2         /* a HIP runtime call / kernel launch failed */
#define STRAPS_EUNSUPPORTED 3 /* valid request this build does not cover */

int straps_abi_version(void);
const char* straps_last_error(void);
/* number of visible HIP devices (0 => the product path must refuse to run) */
int straps_device_count(void);

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

/* stem weights OIHW [64][cin][7][7] (models/resnet.py:145) -> MFMA fragment order
 * [ceil(cin*49/8)][2][64][4]; straps_stem_weight_floats gives the element count.              */
size_t straps_stem_weight_floats(int cin);
int straps_pack_stem_weight(const float* w_oihw, float* w_frag, int cin, void* stream);

/* BatchNorm eval-mode fold (models/resnet.py:147 in .eval()): scale = gamma/sqrt(var+eps),
 * shift = beta - mean*scale.                                                                  */
int straps_bn_fold(const float* gamma, const float* beta, const float* mean, const float* var,
                   float eps, float* scale, float* shift, int c, void* stream);

/* conv7x7/s2/p3 over the NCHW network input, fused y = relu?(conv*scale+shift) -> NHWC.
 * scale/shift may be NULL (raw conv output, used in training mode).  If stats_partial != NULL it
 * receives per-block per-channel (sum, sum of squares) of the RAW conv output:
 * [straps_stem_stat_blocks()][64][2] floats (training-mode BatchNorm statistics).             */
int straps_stem_stat_blocks(int batch, int h, int w);
int straps_stem_fwd(const float* x_nchw, const float* w_frag, const float* scale,
                    const float* shift, int relu, float* y_nhwc, float* stats_partial, int batch,
                    int cin, int h, int w, void* stream);

/* generic implicit-GEMM convolution over NHWC fp32 (3x3 pad 1 stride 1|2, 1x1 stride 1|2):
 *   y[m][co] = epilogue( sum_{r,s,ci} x[b][ho*stride+r-pad][wo*stride+s-pad][ci] * w[co][r][s][ci] )
 * epilogue: *scale[co] + shift[co] (if scale), + residual[m][co] (if residual), relu (if relu).
 * cin % 32 == 0 and cout % 64 == 0 required.  stats_partial as for the stem:
 * [straps_conv_stat_blocks(...)][cout][2].  tile_cfg: 0 = auto, 1 = 128x128, 2 = 128x64,
 * 3 = 64x64 block tile.                                                                       */
int straps_conv_stat_blocks(int batch, int ho, int wo, int cout, int tile_cfg);
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
/* copy a [rows][cols] slice of a row-major matrix into a zero-padded [rows_pad][ld_dst] one
 * (splits fc1.weight [H][F+157] into its feature and estimate parts, pads 157 -> 160).          */
int straps_pad_copy(const float* src, int ld_src, int col0, int rows, int cols, float* dst,
                    int ld_dst, int rows_pad, void* stream);
/* broadcast the initial estimate (models/ief_module.py:50-52): est[m][0..157) = init, pad = 0  */
int straps_broadcast_rows(const float* row, int cols, float* dst, int ld_dst, int m, void* stream);

/* ------------------------------------------------------------------------------------------
 * Pose representation -- utils/rigid_transform_utils.py:27-41 and smplx.lbs.batch_rodrigues.
 * ------------------------------------------------------------------------------------------ */
/* x6 [rows][ld]: each row holds `per_row` 6-D rotations (interleaved a1x,a2x,a1y,a2y,a1z,a2z)
 * starting at column 0 -> R [rows*per_row][3][3] row-major (columns b1,b2,b3).                  */
int straps_rot6d_fwd(const float* x6, long long ld, int per_row, float* rotmats, long long rows,
                     void* stream);
/* axis-angle [n][3] -> R [n][3][3] (angle = ||r + 1e-8||).                                      */
int straps_rodrigues_fwd(const float* aa, float* rotmats, long long n, void* stream);

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
    /* blend directions in MFMA A-fragment order: [tile 216][coord 3][kgroup 28][lane 64][4],
     * element = D[k = 8*g + 4*(lane>>5) + e][vertex = 32*tile + (lane&31)][coord], where D row 0
     * is v_template, rows 1..10 shapedirs[..., l], rows 11..217 posedirs, rest zero.           */
    const float* blend_frag;
    const float* j_template;   /* [24][3]  J_regressor @ v_template (host fp64)              */
    const float* j_shapedirs;  /* [24][3][10] J_regressor @ shapedirs                          */
    const int32_t* parents;    /* [24] */
    const int32_t* depth;      /* [24] depth of each joint in the kinematic tree               */
    int32_t max_depth;
    int32_t skin_k;            /* non-zeros kept per vertex (4 for the real model)              */
    const float* skin_w;       /* [VPAD][skin_k] */
    const int32_t* skin_j;     /* [VPAD][skin_k] joint index of each weight                     */
    /* extra-joint regressors as sparse entries grouped by (round = tile/4, owner = joint%4):
     * entries of group q = round*4 + owner live in [jr_ptr[q], jr_ptr[q+1]).
     * jr_code = (tile_in_round << 16) | (v_local << 8) | joint(0..44).                        */
    const int32_t* jr_ptr;     /* [54*4 + 1] */
    const int32_t* jr_code;
    const float* jr_w;
    const int32_t* pick_ids;   /* [21] vertex ids appended as joints 24..44                    */
} straps_smpl_model_t;

/* bytes of caller-owned scratch for `batch` bodies split into `chunks` vertex chunks           */
size_t straps_smpl_workspace_bytes(long long batch, int chunks);
/* verts [B][6890][3], joints [B][90][3] (may be NULL: vertices only).  betas [B][10],
 * rotmats [B][24][3][3] row-major.  chunks: 0 = auto (8 for big batches = one per XCD).        */
int straps_smpl_fwd(const straps_smpl_model_t* model, const float* betas, const float* rotmats,
                    float* verts, float* joints, void* workspace, long long batch, int chunks,
                    void* stream);

#ifdef __cplusplus
}
#endif
#endif /* STRAPS_HIP_H */
