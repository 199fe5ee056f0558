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
    # fragment order [tile][coord][g][h][i][e]  <-  D[8g+4h+e][32t+i][c]
    frag = D.reshape(28, 2, 4, TILES, 32, 3).transpose(3, 5, 0, 1, 4, 2).copy()
    parents = np.asarray(model['parents'], np.int32).copy()
    depth = np.zeros(24, np.int32)
    for j in range(1, 24):
        assert 0 <= parents[j] < j, 'parents must be topologically ordered'
        depth[j] = depth[parents[j]] + 1
    nnz = (W != 0).sum(1)
    k = int(max(1, nnz.max()))
    order = np.argsort(-(W != 0).astype(np.int8), axis=1, kind='stable')[:, :k]      # non-zeros first, by joint id
    sw = np.zeros((VPAD, k), np.float32)
    sj = np.zeros((VPAD, k), np.int32)
    sw[:V] = np.take_along_axis(W, order, axis=1)
    sj[:V] = np.where(sw[:V] != 0, order, 0)
    # sparse extra regressors grouped by (round = tile//4, owner = joint%4)
    R45 = np.concatenate([np.asarray(model[n], np.float32) for n in
                          ('J_regressor_extra', 'J_regressor_cocoplus', 'J_regressor_h36m')], axis=0)
    assert R45.shape == (45, V)
    jj, vv = np.nonzero(R45)
    tile = vv // 32
    q = (tile // 4) * 4 + (jj % 4)
    o = np.lexsort((vv, jj, q))
    jj, vv, q, tile = jj[o], vv[o], q[o], tile[o]
    code = ((tile % 4) << 16) | ((vv % 32) << 8) | jj
    jr_ptr = np.zeros(54 * 4 + 1, np.int32)
    np.add.at(jr_ptr, q + 1, 1)
    jr_ptr = np.cumsum(jr_ptr).astype(np.int32)
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
    return {
        'blend_frag_t': frag_t.reshape(-1), 'children': children,
        'jrt_ptr': jrt_ptr, 'jrt_code': (((vs % 32) << 8) | src).astype(np.int32), 'jrt_w': ws,
        'blend_frag': frag.reshape(-1),
        'j_template': (Jr @ vt).astype(np.float32),
        'j_shapedirs': np.einsum('jv,vcl->jcl', Jr, sd).astype(np.float32),
        'parents': parents, 'depth': depth, 'max_depth': int(depth.max()), 'skin_k': k,
        'skin_w': sw, 'skin_j': sj,
        'jr_ptr': jr_ptr, 'jr_code': np.ascontiguousarray(code.astype(np.int32) if code.size else np.zeros(1, np.int32)),
        'jr_w': np.ascontiguousarray(R45[jj, vv].astype(np.float32) if jj.size else np.zeros(1, np.float32)),
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
        packed = pack_smpl_model(model)
        self.max_depth, self.skin_k = packed.pop('max_depth'), packed.pop('skin_k')
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
            for f in ('blend_frag', 'j_template', 'j_shapedirs', 'parents', 'depth', 'skin_w', 'skin_j', 'jr_ptr',
                      'jr_code', 'jr_w', 'pick_ids', 'blend_frag_t', 'children', 'jrt_ptr', 'jrt_code', 'jrt_w'):
                setattr(s, f, getattr(self, '_k_' + f).data_ptr())
            s.max_depth, s.skin_k = self.max_depth, self.skin_k
            self._struct, self._struct_key = s, key
        return self._struct

    def forward_arrays(self, betas, rotmats, want_joints=True, chunks=0):
        """raw entry: betas [B,10], rotmats [B,24,3,3] (contiguous fp32 GPU) -> (verts, joints|None)."""
        hipabi.require_gpu_tensor(betas, 'betas', torch.float32)
        hipabi.require_gpu_tensor(rotmats, 'rotmats', torch.float32)
        hipabi.require_gpu_tensor(self._k_blend_frag, 'SMPL model buffers (call .to(device))')
        B = betas.shape[0]
        if tuple(betas.shape) != (B, 10) or tuple(rotmats.shape) != (B, 24, 3, 3):
            raise RuntimeError('SMPL: expected betas [B,10] and rotmats [B,24,3,3], got %s and %s'
                               % (tuple(betas.shape), tuple(rotmats.shape)))
        betas, rotmats = betas.contiguous(), rotmats.contiguous()
        L = hipabi.lib()
        verts = torch.empty(B, V, 3, device=betas.device, dtype=torch.float32)
        joints = torch.empty(B, 90, 3, device=betas.device, dtype=torch.float32) if want_joints else None
        ws = torch.empty(L.straps_smpl_workspace_bytes(B, chunks) // 4, device=betas.device, dtype=torch.float32)
        hipabi.check(L.straps_smpl_fwd(C.byref(self._model_struct()), hipabi.ptr(betas), hipabi.ptr(rotmats),
                                       hipabi.ptr(verts), hipabi.ptr(joints), hipabi.ptr(ws), B, chunks,
                                       hipabi.stream_ptr()), 'straps_smpl_fwd')
        return verts, joints

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
