"""Body-part segmentation renderer -- drop-in for reference renderers/nmr_renderer.py:9-100 (class NMRRenderer) in
its `rend_parts_seg=True` configuration, the one the training loop uses (train loop :155).

The reference wraps the third-party CUDA extension `neural_renderer`; here the z-buffer pass and the part look-up
are three HIP kernels behind `straps_rasterize_parts` (csrc/raster.hip).  Same constructor / call surface:

    renderer = NMRRenderer(batch_size, cam_K, cam_R, img_wh=256, rend_parts_seg=True)
    parts = renderer(vertices, cam_ts)            # [B, wh, wh] long, 0 background, 1..6 body parts

Mesh topology and part labels: the reference reads additional/{smpl_faces,vertex_texture,cube_parts}.npy
(config.py).  Those assets are not redistributable; pass `faces` / `face_parts` explicitly (e.g. from
`synthetic_smpl_model()`), or leave them None to load the reference's files from `config`.  The RGB mode
(rend_parts_seg=False, visualisation only) is out of scope and raises.
"""
import os

import numpy as np
import torch
import torch.nn as nn

from . import config, hipabi


def face_parts_from_texture(vertex_texture, cube_parts):
    """per-face part id from the reference's assets: vertex_texture [1,F,T,T,T,3] colours decoded through
    cube_parts[floor(100 r), floor(100 g), floor(100 b)] exactly like get_parts (renderers/nmr_renderer.py:93-100);
    a face whose texels disagree (part boundary) takes the majority part, lowest id on ties."""
    vertex_texture = np.asarray(vertex_texture, np.float32)
    tex = vertex_texture.reshape(-1, int(np.prod(vertex_texture.shape[-4:-1])), 3)           # [F][texels][3]
    idx = np.floor(100.0 * tex).astype(np.int64)
    labels = np.asarray(cube_parts)[idx[..., 0], idx[..., 1], idx[..., 2]].astype(np.int64)      # [F, texels]
    counts = np.stack([(labels == k).sum(1) for k in range(7)], axis=1)
    return counts.argmax(1).astype(np.uint8)


class NMRRenderer(nn.Module):
    def __init__(self, batch_size, cam_K, cam_R, img_wh=256, rend_parts_seg=False, faces=None, face_parts=None,
                 near=0.1, far=100.0):
        super().__init__()
        if not rend_parts_seg:
            raise RuntimeError('NMRRenderer: only rend_parts_seg=True (the training-loop configuration) is built; the RGB '
                               'visualisation mode of renderers/nmr_renderer.py is out of scope')
        if faces is None:
            for pth in (config.SMPL_FACES_PATH, config.VERTEX_TEXTURE_PATH, config.CUBE_PARTS_PATH):
                if not os.path.isfile(pth):
                    raise RuntimeError('NMRRenderer: %s not found; pass faces= and face_parts= (e.g. from synthetic_smpl_model())' % pth)
            faces = np.load(config.SMPL_FACES_PATH)
            face_parts = face_parts_from_texture(np.load(config.VERTEX_TEXTURE_PATH), np.load(config.CUBE_PARTS_PATH))
        faces = np.ascontiguousarray(np.asarray(faces).astype(np.int32))
        face_parts = np.ascontiguousarray(np.asarray(face_parts).astype(np.uint8))
        if faces.ndim != 2 or faces.shape[1] != 3 or face_parts.shape != (faces.shape[0],):
            raise RuntimeError('NMRRenderer: faces must be [F,3] and face_parts [F] (got %s, %s)' % (faces.shape, face_parts.shape))
        self.register_buffer('faces', torch.from_numpy(faces))
        self.register_buffer('face_parts', torch.from_numpy(face_parts))
        cam_K = torch.as_tensor(cam_K, dtype=torch.float32)
        cam_R = torch.as_tensor(cam_R, dtype=torch.float32)
        if (cam_K.ndim == 3) != (cam_R.ndim == 3):
            raise RuntimeError('NMRRenderer: cam_K and cam_R must both be [3,3] or both [B,3,3]')
        self.register_buffer('cam_K', cam_K.contiguous())
        self.register_buffer('cam_R', cam_R.contiguous())
        self.batch_size, self.img_wh, self.rend_parts_seg = batch_size, int(img_wh), True
        self.near, self.far = float(near), float(far)

    @hipabi.on_tensor_device
    def render_arrays(self, vertices, cam_ts, want_depth=False, vert_noise_u=None, noise_range=(-0.01, 0.01), out=None):
        """raw entry: vertices [B,N,3], cam_ts [B,3] (fp32 GPU) -> float part map [B,wh,wh] (+ depth).
        vert_noise_u: optional uniforms [B,N,2] in [0,1): the rendered copy of the mesh gets x,y += (h-l)*u + l
        (random_verts2D_deviation, proxy_rep_augmentation.py:5-22) inside the projection kernel."""
        hipabi.require_gpu_tensor(vertices, 'vertices', torch.float32)
        hipabi.require_gpu_tensor(cam_ts, 'cam_ts', torch.float32)
        hipabi.require_gpu_tensor(self.faces, 'NMRRenderer buffers (call .to(device))')
        B, N = vertices.shape[0], vertices.shape[1]
        if cam_ts.ndim == 3:
            cam_ts = cam_ts[:, 0]
        if tuple(vertices.shape) != (B, N, 3) or tuple(cam_ts.shape) != (B, 3):
            raise RuntimeError('NMRRenderer: expected vertices [B,N,3] and cam_ts [B,3] / [B,1,3], got %s and %s'
                               % (tuple(vertices.shape), tuple(cam_ts.shape)))
        per_body = self.cam_K.ndim == 3
        if per_body and self.cam_K.shape[0] != B:
            raise RuntimeError('NMRRenderer: %d cameras for a batch of %d' % (self.cam_K.shape[0], B))
        vertices, cam_ts = vertices.contiguous(), cam_ts.contiguous()
        L = hipabi.lib()
        wh = self.img_wh
        parts = torch.empty(B, wh, wh, device=vertices.device, dtype=torch.float32) if out is None else out
        if vert_noise_u is not None:
            hipabi.require_gpu_tensor(vert_noise_u, 'vert_noise_u', torch.float32)
            if vert_noise_u.numel() != B * N * 2 or not vert_noise_u.is_contiguous():
                raise RuntimeError('NMRRenderer: vert_noise_u must be contiguous [B,N,2] uniforms')
        depth = torch.empty(B, wh, wh, device=vertices.device, dtype=torch.float32) if want_depth else None
        ws = torch.empty(L.straps_rasterize_workspace_bytes(B, N, wh) // 4, device=vertices.device, dtype=torch.float32)
        hipabi.check(L.straps_rasterize_parts(hipabi.ptr(vertices), hipabi.ptr(self.faces), hipabi.ptr(self.face_parts), hipabi.ptr(self.cam_K),
                                              hipabi.ptr(self.cam_R), hipabi.ptr(cam_ts), hipabi.ptr(parts), hipabi.ptr(depth), hipabi.ptr(ws),
                                              B, N, self.faces.shape[0], wh, 1 if per_body else 0, self.near, self.far,
                                              hipabi.ptr(vert_noise_u), float(noise_range[0]), float(noise_range[1]),
                                              hipabi.stream_ptr()), 'straps_rasterize_parts')
        return (parts, depth) if want_depth else parts

    def forward(self, vertices, cam_ts):
        """vertices (B, N, 3), cam_ts (B, 1, 3) or (B, 3) -> (B, wh, wh) long part ids (renderers/nmr_renderer.py:76-91)."""
        return self.render_arrays(vertices, cam_ts).long()
