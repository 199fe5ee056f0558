"""SMPL body model module -- drop-in for reference models/smpl_official.py:10-41 (class SMPL).

Same call surface (`SMPL(model_dir, batch_size=B)`; `smpl(body_pose=, global_orient=, betas=,
pose2rot=)`; output object with .vertices [B,6890,3], .joints [B,90,3], .global_orient, .body_pose,
.betas, .full_pose) but no `smplx` dependency: the forward is three HIP kernels
(csrc/smpl.hip) behind `straps_smpl_fwd`.  The model constants are packed once on the host into
the layouts the kernels consume (see include/straps_hip.h, straps_smpl_model_t).
"""
import ctypes as C
from collections import namedtuple

import numpy as np
import torch
import torch.nn as nn

from . import hipabi
from .synthetic_smpl import load_smpl_model

ModelOutput = namedtuple('ModelOutput', ['vertices', 'joints', 'full_pose', 'betas', 'global_orient', 'body_pose'])
ModelOutput.__new__.__defaults__ = (None,) * len(ModelOutput._fields)

V, VPAD, TILES, KP = 6890, 6912, 216, 224
# STRAPS_SMPL_EXACT_F32 / STRAPS_SMPL_SPLIT_F16 (blend contraction split) / STRAPS_SMPL_SPLIT_F16_LBS (blend + skinning split)
PRECISIONS = {'fp32': 0, 'fp16x3': 1, 'fp16x3_lbs': 2, 'fp16x3_lbs_pd16': 3, 'fp16x3_lbs_p16': 4}
KERNELS = {'auto': 0, 'wide': 0x100, 'narrow': 0x200, 'wide_builtin': 0x100 | 0x400}      # STRAPS_SMPL_KERNEL_WIDE / _NARROW, OR-ed into the mode argument


def pack_smpl_model(model):
    """numpy model dict -> dict of numpy arrays in kernel layout (host side, fp64 accumulation)."""
    vt = np.asarray(model['v_template'], np.float64)
    sd = np.asarray(model['shapedirs'], np.float64)
    pd = np.asarray(model['posedirs'], np.float64)
    Jr = np.asarray(model['J_regressor'], np.float64)
    W = np.asarray(model['weights'], np.float32)
    assert vt.shape == (V, 3) and sd.shape == (V, 3, 10) and pd.shape == (207, V * 3) and W.shape == (V, 24)
    # blend directions D[k][v][c]: k=0 template, 1..10 shapedirs, 11..217 posedirs
    D = np.zeros((KP, VPAD, 3), np.float32)
    D[0, :V] = vt
    D[1:11, :V] = np.transpose(sd, (2, 0, 1))
    D[11:218, :V] = pd.reshape(207, V, 3)
    # ---- the 45 sparse-regressed joints as "virtual vertices" (no gather over the written mesh) ----
    #   joint_j = sum_v R[j,v] sum_k W[v,k] (R_k v_posed_v + t_k) = sum_k ( R_k (F . E_jk) + s_jk t_k ),
    #   E_jk = sum_v R[j,v] W[v,k] D[:,v],  s_jk = sum_v R[j,v] W[v,k]
    # i.e. one virtual vertex per (joint, bone) pair with blend directions E_jk / s_jk, skinned rigidly with weight
    # s_jk on bone k.  A pair whose coefficients cancel (|s| small against sum |c|) is split by sign so s never
    # vanishes.  Accumulated in fp64 on the host.
    R45 = np.concatenate([np.asarray(model[n], np.float64) for n in
                          ('J_regressor_extra', 'J_regressor_cocoplus', 'J_regressor_h36m')], axis=0)
    assert R45.shape == (45, V)
    D64 = np.zeros((V, 218, 3), np.float64)
    D64[:, 0] = vt
    D64[:, 1:11] = np.transpose(sd, (0, 2, 1))
    D64[:, 11:] = pd.reshape(207, V, 3).transpose(1, 0, 2)
    D64 = D64.reshape(V, 218 * 3)
    W64 = W.astype(np.float64)
    vdirs, vs_, vk_, vjn = [], [], [], np.zeros(45, np.int64)
    for j in range(45):
        cols = np.nonzero(R45[j])[0]
        if cols.size == 0:
            continue
        Cj = R45[j, cols, None] * W64[cols]                                  # [n][24]
        for kb in np.nonzero(np.any(Cj != 0, axis=0))[0]:
            c = Cj[:, kb]
            parts = [c] if abs(c.sum()) >= 0.25 * np.abs(c).sum() else [np.where(c > 0, c, 0.0), np.where(c < 0, c, 0.0)]
            for cp in parts:
                s_ = cp.sum()
                if s_ == 0:
                    continue
                vdirs.append((cp @ D64[cols]) / s_)
                vs_.append(s_)
                vk_.append(kb)
                vjn[j] += 1
    nvirt = len(vs_)
    vj_ptr = np.concatenate([[0], np.cumsum(vjn)]).astype(np.int32)
    n_tiles = ((TILES + (nvirt + 31) // 32 + 7) // 8) * 8
    Dall = np.zeros((KP, n_tiles * 32, 3), np.float32)
    Dall[:, :VPAD] = D
    if nvirt:
        Dall[:218, VPAD:VPAD + nvirt] = np.asarray(vdirs).reshape(nvirt, 218, 3).transpose(1, 0, 2)
    # fragment order [tile][coord][g][h][i][e]  <-  D[8g+4h+e][32t+i][c]
    frag = Dall.reshape(28, 2, 4, n_tiles, 32, 3).transpose(3, 5, 0, 1, 4, 2).copy()
    # split-precision operand (straps_hip.h: blend_frag_h): two-term fp16 split of S_D * D, S_D the largest power of two that
    # keeps |S_D * D| <= 32768 (at most 2^14): the low halves then stay normal fp16 numbers for every |D| > 2^-14 * 2^-3.
    dmax = float(np.abs(Dall).max())
    sd_exp = 14 if dmax == 0 else int(min(14, np.floor(np.log2(32768.0 / dmax))))
    Ds = Dall.astype(np.float32) * np.float32(2.0 ** sd_exp)                              # exact (power of two)
    Dh = Ds.astype(np.float16)
    Dl = (Ds - Dh.astype(np.float32)).astype(np.float16)                                  # the difference is exact in fp32
    assert np.isfinite(Dh.astype(np.float32)).all()
    # [tile][kstep][coord][hi|lo][hh][i][j]  <-  D[16s + 8hh + j][32t + i][c]
    frag_h = np.stack([Dh, Dl], axis=0).reshape(2, KP // 16, 2, 8, n_tiles, 32, 3).transpose(4, 1, 6, 0, 2, 5, 3).copy()
    parents = np.asarray(model['parents'], np.int32).copy()
    depth = np.zeros(24, np.int32)
    for j in range(1, 24):
        assert 0 <= parents[j] < j, 'parents must be topologically ordered'
        depth[j] = depth[parents[j]] + 1
    nnz = (W != 0).sum(1)
    k = int(max(1, nnz.max()))
    order = np.argsort(-(W != 0).astype(np.int8), axis=1, kind='stable')[:, :k]      # non-zeros first, by joint id
    sw = np.zeros((n_tiles * 32, k), np.float32)
    sj = np.zeros((n_tiles * 32, k), np.int32)
    sw[:V] = np.take_along_axis(W, order, axis=1)
    sj[:V] = np.where(sw[:V] != 0, order, 0)
    if nvirt:
        sw[VPAD:VPAD + nvirt, 0] = np.asarray(vs_, np.float32)
        sj[VPAD:VPAD + nvirt, 0] = np.asarray(vk_, np.int32)
    # dense skinning weights for the matrix-pipe skinning (straps_hip.h: skin_frag_p): two-term fp16 split of 2^14 * W (24 joints)
    Wd = np.zeros((n_tiles * 32, 32), np.float32)
    np.add.at(Wd, (np.repeat(np.arange(n_tiles * 32), k), sj.reshape(-1)), sw.reshape(-1))
    assert float(np.abs(Wd).max()) < 3.9, 'skinning weight too large for the fp16 split (|w| * 2^14 must stay below 65504)'
    Ws = Wd * np.float32(2.0 ** 14)
    Wh = Ws.astype(np.float16)
    Wl = (Ws - Wh.astype(np.float32)).astype(np.float16)
    # with the three products of the split packed along K:
    #   T = Ah.Wh + Al.Wh + Ah.Wl = [Ah | Al | Ah | .] . [Wh | Wh | Wl | 0]   over 24 + 24 + 24 + 8 = 80 columns = 5 k-steps (not 3 x 2 = 6)
    # [tile][kstep 5][hh][i][j]  <-  P[32t + i][16 ks + 8 hh + j]
    Wp = np.concatenate([Wh[:, :24], Wh[:, :24], Wl[:, :24], np.zeros((n_tiles * 32, 8), np.float16)], axis=1)
    skin_frag_p = Wp.reshape(n_tiles, 32, 5, 2, 8).transpose(0, 2, 3, 1, 4).copy()
    jj, vv = np.nonzero(R45)                                                 # (backward tables below)
    o = np.lexsort((vv, jj))
    jj, vv = jj[o], vv[o]
    # ---- backward tables ----
    # transposed fragments [t][c][f][rq][h][i][e] <- D[32f+i][32t + row(4rq+e, h)][c], row(r,h) = (r&3)+8(r>>2)+4h
    r_ = np.arange(16)
    rows = (r_[None, :] & 3) + 8 * (r_[None, :] >> 2) + 4 * np.arange(2)[:, None]            # [h][r]
    Dk = D.reshape(7, 32, TILES, 32, 3)                                                          # [f][i][t][vl][c]
    Dt = Dk[:, :, :, rows, :]                                                                    # [f][i][t][h][r][c]
    frag_t = Dt.reshape(7, 32, TILES, 2, 4, 4, 3).transpose(2, 6, 0, 4, 3, 1, 5).copy()          # [t][c][f][rq][h][i][e]
    children = -np.ones((24, 3), np.int32)
    for j in range(1, 24):
        slot = int((children[parents[j]] >= 0).sum())
        assert slot < 3, 'a joint with more than 3 children is not supported'
        children[parents[j], slot] = j
    # joint-gradient sources per tile: picked vertices (src 0..20, weight 1) + regressed joints (src 21..65)
    pj = np.arange(21)
    pv = np.asarray(model['extra_vertex_ids'], np.int64)
    src = np.concatenate([pj, 21 + jj])
    vs = np.concatenate([pv, vv])
    ws = np.concatenate([np.ones(21, np.float32), R45[jj, vv].astype(np.float32)])
    o2 = np.lexsort((src, vs))
    src, vs, ws = src[o2], vs[o2], ws[o2]
    jrt_ptr = np.zeros(TILES + 1, np.int32)
    np.add.at(jrt_ptr, vs // 32 + 1, 1)
    jrt_ptr = np.cumsum(jrt_ptr).astype(np.int32)
    # skinning weights by (round of 4 tiles, joint) for the joint-transform gradient (fixed summation order)
    dv, dk = np.nonzero(sw[:VPAD])
    dj, dw = sj[:VPAD][dv, dk], sw[:VPAD][dv, dk]
    o3 = np.lexsort((dv, dj, dv // 128))
    dv, dj, dw = dv[o3], dj[o3], dw[o3]
    dj_ptr = np.zeros(54 * 24 + 1, np.int32)
    np.add.at(dj_ptr, (dv // 128) * 24 + dj + 1, 1)
    dj_ptr = np.cumsum(dj_ptr).astype(np.int32)
    dj_code = ((((dv % 128) // 32) << 5) | (dv % 32)).astype(np.int32)
    return {
        'dj_ptr': dj_ptr, 'dj_code': dj_code, 'dj_w': dw.astype(np.float32),
        'blend_frag_t': frag_t.reshape(-1), 'children': children,
        'jrt_ptr': jrt_ptr, 'jrt_code': (((vs % 32) << 8) | src).astype(np.int32), 'jrt_w': ws,
        'blend_frag': frag.reshape(-1),
        'blend_frag_h': frag_h.reshape(-1), 'blend_h_unscale': float(2.0 ** -(sd_exp + 6)),         'skin_frag_p': skin_frag_p.reshape(-1),
        'j_template': (Jr @ vt).astype(np.float32),
        'j_shapedirs': np.einsum('jv,vcl->jcl', Jr, sd).astype(np.float32),
        'parents': parents, 'depth': depth, 'max_depth': int(depth.max()), 'skin_k': k,
        'skin_w': sw, 'skin_j': sj,
        'vj_ptr': vj_ptr, 'n_tiles': int(n_tiles),
        'pick_ids': np.asarray(model['extra_vertex_ids'], np.int32),
    }


class SMPL(nn.Module):
    """`model_path`: directory / file of a real SMPL model (reference run_train.py:109), or a model
    dict (e.g. `synthetic_smpl_model()`).  Default pose/shape parameters are registered exactly as
    smplx does (zeros, shaped by batch_size) so `smpl(betas=...)` alone works (train loop :144)."""

    NUM_BODY_JOINTS = 23

    def __init__(self, model_path, batch_size=1, gender='neutral', extra_regressor_paths=None, **kwargs):
        super().__init__()
        model = model_path if isinstance(model_path, dict) else load_smpl_model(model_path, gender, extra_regressor_paths)
        self.batch_size = batch_size
        # mesh topology like smplx's `.faces` (numpy); face_parts is this package's per-face body-part table (nmr_renderer.py)
        self.faces = None if model.get('faces') is None else np.asarray(model['faces'])
        self.face_parts = None if model.get('face_parts') is None else np.asarray(model['face_parts'])
        packed = pack_smpl_model(model)
        self.max_depth, self.skin_k, self.n_tiles = packed.pop('max_depth'), packed.pop('skin_k'), packed.pop('n_tiles')
        self.blend_h_unscale = packed.pop('blend_h_unscale')
        precision = kwargs.pop('precision', 'fp32')
        if precision not in PRECISIONS:
            raise ValueError("SMPL: precision must be one of %s" % (sorted(PRECISIONS),))
        self.precision = precision
        for name, arr in packed.items():
            self.register_buffer('_k_' + name, torch.from_numpy(np.ascontiguousarray(arr)), persistent=False)
        # buffers with the names smplx / the reference expose (state-dict visible, used by callers)
        self.register_buffer('v_template', torch.tensor(model['v_template'], dtype=torch.float32))
        self.register_buffer('shapedirs', torch.tensor(model['shapedirs'], dtype=torch.float32))
        self.register_buffer('posedirs', torch.tensor(model['posedirs'], dtype=torch.float32))
        self.register_buffer('J_regressor', torch.tensor(model['J_regressor'], dtype=torch.float32))
        self.register_buffer('lbs_weights', torch.tensor(model['weights'], dtype=torch.float32))
        self.register_buffer('parents', torch.tensor(model['parents'], dtype=torch.long))
        self.register_buffer('J_regressor_extra', torch.tensor(model['J_regressor_extra'], dtype=torch.float32))
        self.register_buffer('J_regressor_cocoplus', torch.tensor(model['J_regressor_cocoplus'], dtype=torch.float32))
        self.register_buffer('J_regressor_h36m', torch.tensor(model['J_regressor_h36m'], dtype=torch.float32))
        self.betas = nn.Parameter(torch.zeros(batch_size, 10))
        self.global_orient = nn.Parameter(torch.zeros(batch_size, 3))
        self.body_pose = nn.Parameter(torch.zeros(batch_size, self.NUM_BODY_JOINTS * 3))
        self.transl = nn.Parameter(torch.zeros(batch_size, 3))
        self._struct = None
        self._struct_key = None

    def _model_struct(self):
        key = self._k_blend_frag.data_ptr()
        if self._struct_key != key:
            s = hipabi.SmplModelStruct()
            for f in ('blend_frag', 'blend_frag_h', 'skin_frag_p', 'j_template', 'j_shapedirs', 'parents', 'depth', 'skin_w', 'skin_j', 'vj_ptr',
                      'pick_ids', 'blend_frag_t', 'children', 'jrt_ptr', 'jrt_code', 'jrt_w', 'dj_ptr', 'dj_code', 'dj_w'):
                setattr(s, f, getattr(self, '_k_' + f).data_ptr())
            s.max_depth, s.skin_k, s.n_tiles = self.max_depth, self.skin_k, self.n_tiles
            s.blend_h_unscale = self.blend_h_unscale
            self._struct, self._struct_key = s, key
        return self._struct

    @hipabi.on_tensor_device
    def forward_arrays(self, betas, rotmats, want_joints=True, chunks=0, out_verts=None, out_joints=None, precision=None, kernel='auto'):
        """raw entry: betas [B,10], rotmats [B,24,3,3] (contiguous fp32 GPU) -> (verts, joints|None).
        out_verts / out_joints: optional resident output buffers ([B,6890,3] / [B,90,3], contiguous fp32).
        precision: 'fp32' (exact fp32 everywhere), 'fp16x3' (blend contraction as a three-product fp16 split with fp32
        accumulate: same accuracy class, 16x the matrix rate) or 'fp16x3_lbs' (the skinning transforms on the matrix pipe as
        well); None = the module's setting.
        Range of the split modes: |beta| and the pose features below 1023, joint transforms (rotations and joint positions in metres)
        below 63, skinning weights below 3.9 -- far outside anything a body model produces; operands beyond it SATURATE at the
        largest fp16 value (finite, clipped meshes; csrc/smpl.hip sat_h) rather than turning the body into NaNs.  'fp32' has
        no such range.
        kernel (the 'fp16x3_lbs*' modes only): 'auto' = by batch size (64-body workgroups with one 512-register wave per SIMD from 2048
        bodies on, 32-body workgroups below), 'wide' / 'narrow' force one of the two (A/B and tests).  The two kernels are
        BIT-IDENTICAL -- the same accumulation chain per output value, so the automatic switch at 2048 bodies changes no result
        (tests/test_gpu_forward.py::test_smpl_wide_kernel_vs_oracle_and_narrow asserts torch.equal at every batch size)."""
        if kernel not in KERNELS:
            raise ValueError("SMPL.forward_arrays: kernel must be one of %s" % (sorted(KERNELS),))
        hipabi.require_gpu_tensor(betas, 'betas', torch.float32)
        hipabi.require_gpu_tensor(rotmats, 'rotmats', torch.float32)
        hipabi.require_gpu_tensor(self._k_blend_frag, 'SMPL model buffers (call .to(device))')
        B = betas.shape[0]
        if tuple(betas.shape) != (B, 10) or tuple(rotmats.shape) != (B, 24, 3, 3):
            raise RuntimeError('SMPL: expected betas [B,10] and rotmats [B,24,3,3], got %s and %s'
                               % (tuple(betas.shape), tuple(rotmats.shape)))
        betas, rotmats = betas.contiguous(), rotmats.contiguous()
        L = hipabi.lib()
        verts = torch.empty(B, V, 3, device=betas.device, dtype=torch.float32) if out_verts is None else out_verts
        joints = None
        if want_joints:
            joints = torch.empty(B, 90, 3, device=betas.device, dtype=torch.float32) if out_joints is None else out_joints
        for t, shp, nm in ((verts, (B, V, 3), 'out_verts'), (joints, (B, 90, 3), 'out_joints')):
            if t is not None and (tuple(t.shape) != shp or not t.is_contiguous() or t.dtype != torch.float32 or t.device != betas.device):
                raise RuntimeError('SMPL: %s must be a contiguous fp32 %s tensor on %s' % (nm, shp, betas.device))
        ws = torch.empty(L.straps_smpl_workspace_bytes(C.byref(self._model_struct()), B) // 4, device=betas.device, dtype=torch.float32)
        hipabi.check(L.straps_smpl_fwd(C.byref(self._model_struct()), hipabi.ptr(betas), hipabi.ptr(rotmats),
                                       hipabi.ptr(verts), hipabi.ptr(joints), hipabi.ptr(ws), B, chunks,
                                       PRECISIONS[self.precision if precision is None else precision] | KERNELS[kernel], hipabi.stream_ptr()), 'straps_smpl_fwd')
        return verts, joints

    @hipabi.on_tensor_device
    def forward(self, betas=None, body_pose=None, global_orient=None, transl=None, pose2rot=True, **kwargs):
        # kwargs swallows get_skin etc. exactly like the reference (models/smpl_official.py:28)
        from .rigid_transform_utils import batch_rodrigues
        betas = self.betas if betas is None else betas
        body_pose = self.body_pose if body_pose is None else body_pose
        global_orient = self.global_orient if global_orient is None else global_orient
        B = max(betas.shape[0], body_pose.shape[0], global_orient.shape[0])
        if betas.shape[0] != B:
            betas = betas.expand(B, -1)
        if pose2rot:
            full_pose = torch.cat([global_orient.reshape(-1, 3).expand(B, -1) if global_orient.shape[0] != B
                                   else global_orient.reshape(B, 3), body_pose.reshape(body_pose.shape[0], -1).expand(B, -1)],
                                  dim=1).detach()
            rotmats = batch_rodrigues(full_pose.reshape(-1, 3)).view(B, 24, 3, 3)
        else:
            full_pose = torch.cat([global_orient.reshape(-1, 1, 3, 3), body_pose.reshape(-1, 23, 3, 3)], dim=1)
            rotmats = full_pose
        if torch.is_grad_enabled() and (betas.requires_grad or rotmats.requires_grad):
            from .autograd_ops import smpl_forward_autograd
            verts, joints = smpl_forward_autograd(self, betas, rotmats)
        else:
            verts, joints = self.forward_arrays(betas.detach().float(), rotmats.detach().float())
        tr = self.transl if transl is None else transl
        if transl is not None:          # module default is zeros: skip the add
            verts = verts + tr[:, None]
            joints = joints + tr[:, None]
        return ModelOutput(vertices=verts, joints=joints, full_pose=full_pose if pose2rot else None, betas=betas,
                           global_orient=global_orient, body_pose=body_pose)
