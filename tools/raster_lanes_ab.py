#!/usr/bin/env python3
"""tools/raster_lanes_ab.py -- lanes per face of the part-segmentation rasteriser (csrc/raster.hip, raster_face_kernel<G>): a face of the
13 776-face SMPL mesh rendered at 256 x 256 covers one or two samples and its bounding box four to nine, so most of a 16-lane group idles
after repeating the face's set-up sixteen times.  Times straps_rasterize_parts on the training step's geometry (64 posed bodies, cam_t =
(0, 0.2, 42), run_train.py:119-124) for G = 16, 64, 32, 8, 4 -- one process per value (tools build: STRAPS_RASTER_LANES is read once) -- and
checks every variant's image against G = 16 bit for bit (the z-buffer minimum does not depend on which lane visits which sample)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(out_path):
    sys.path.insert(0, ROOT)
    import torch
    import straps_amd
    from straps_amd import hipabi, config
    from straps_amd.nmr_renderer import NMRRenderer
    hipabi.use_library(hipabi.build(tools=True))
    dev = torch.device('cuda:0')
    B = int(os.environ.get('RASTER_B', '64'))
    model = straps_amd.synthetic_smpl_model(0)
    smpl = straps_amd.SMPL(model, batch_size=1).to(dev)
    g = torch.Generator().manual_seed(0)
    betas = (torch.randn(B, 10, generator=g) * 1.5).to(dev)
    R = straps_amd.batch_rodrigues((torch.randn(B, 72, generator=g) * 0.4).to(dev).view(-1, 3)).view(B, 24, 3, 3).contiguous()
    verts, _ = smpl.forward_arrays(betas, R)
    K = torch.tensor([[config.FOCAL_LENGTH, 0., config.REGRESSOR_IMG_WH / 2.], [0., config.FOCAL_LENGTH, config.REGRESSOR_IMG_WH / 2.], [0., 0., 1.]])
    rend = NMRRenderer(B, K, torch.eye(3), config.REGRESSOR_IMG_WH, rend_parts_seg=True, faces=smpl.faces, face_parts=smpl.face_parts).to(dev)
    cam_t = torch.tensor([0., 0.2, 42.], device=dev).expand(B, 3).contiguous()
    img = rend.render_arrays(verts, cam_t)
    img = img[0] if isinstance(img, (tuple, list)) else img
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    s.record()
    for _ in range(n):
        rend.render_arrays(verts, cam_t)
    e.record()
    torch.cuda.synchronize()
    cov = float((img > 0).float().mean())
    print('lanes per face %2s: %.1f us per straps_rasterize_parts call (project + clear + faces + resolve), B = %d, %.1f %% of the pixels covered'
          % (os.environ.get('STRAPS_RASTER_LANES', '16'), s.elapsed_time(e) / n * 1e3, B, 100 * cov), flush=True)
    torch.save(img.cpu(), out_path)


if __name__ == '__main__':
    if len(sys.argv) > 2 and sys.argv[1] == '--child':
        child(sys.argv[2])
        sys.exit(0)
    import torch
    ref = None
    for lanes in (16, 64, 32, 8, 4):
        out = '/tmp/raster_lanes_%d.pt' % lanes
        subprocess.run([sys.executable, os.path.abspath(__file__), '--child', out], env=dict(os.environ, STRAPS_RASTER_LANES=str(lanes)), check=True, timeout=600)
        img = torch.load(out)
        if ref is None:
            ref = img
        else:
            print('    identical to 16 lanes: %s' % bool(torch.equal(img, ref)), flush=True)
