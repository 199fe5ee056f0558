import os, sys, json
import numpy as np, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'oracle'))
import straps_amd, straps_oracle as O
from straps_amd import hipabi
from detgen import det_uniform, det_state_dict
GOLD = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')
MP = straps_amd.synthetic_mean_params(0)
dev = torch.device('cuda:0')
L = hipabi.load()
man = json.load(open(os.path.join(GOLD, 'state_dict_keys_r18.json')))['keys']
sd = {k: torch.from_numpy(v) for k, v in det_state_dict(man).items()}
B = 3
x = torch.from_numpy(det_uniform((B, 18, 256, 256), 4343, 0.0, 1.0))
x[:, 1:, ::2] = 0.0
coef = torch.from_numpy(det_uniform((B, 157), 556))
torch.set_num_threads(32)
sdo = {k: (v.clone().double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
w = sdo['image_encoder.conv1.weight'].requires_grad_(True)
# oracle with a tap on the stem conv output gradient
raw = F.conv2d(x.double(), w, None, 2, 3)
raw.retain_grad()
y = F.relu(F.batch_norm(raw, sdo['image_encoder.bn1.running_mean'], sdo['image_encoder.bn1.running_var'], sdo['image_encoder.bn1.weight'], sdo['image_encoder.bn1.bias'], False, 0.1, 1e-5))
y = F.max_pool2d(y, 3, 2, 1)
# rest of the encoder through the oracle: reuse resnet_forward pieces by calling it on a patched input is awkward; do layers manually
def rest(y):
    p = 'image_encoder.'
    for li, nblk in enumerate([2, 2, 2, 2]):
        for bi in range(nblk):
            q = '%slayer%d.%d.' % (p, li + 1, bi)
            stride = 2 if (li > 0 and bi == 0) else 1
            idt = y
            o = F.conv2d(y, sdo[q + 'conv1.weight'], None, stride, 1)
            o = F.relu(O._bn(o, sdo, q + 'bn1', False))
            o = F.conv2d(o, sdo[q + 'conv2.weight'], None, 1, 1)
            o = O._bn(o, sdo, q + 'bn2', False)
            if (q + 'downsample.0.weight') in sdo:
                idt = O._bn(F.conv2d(y, sdo[q + 'downsample.0.weight'], None, stride, 0), sdo, q + 'downsample.1', False)
            y = F.relu(o + idt)
    return y.mean(dim=(2, 3))
feat = rest(y)
_, _, _, est = O.ief_forward(feat, sdo, O.ief_init_estimate(MP['pose'], MP['shape']).double(), 3)
(est * coef.double()).sum().backward()
ref_dw, ref_draw = w.grad.clone(), raw.grad.clone()
print('oracle: |dw|max %.3e  |draw|max %.3e  draw mean %.3e' % (float(ref_dw.abs().max()), float(ref_draw.abs().max()), float(ref_draw.mean())))
# unit kernel on the oracle's draw
xd = x.to(dev)
gd = ref_draw.float().permute(0, 2, 3, 1).contiguous().to(dev)
ws = torch.empty(L.straps_stem_wgrad_workspace_bytes(B, 18, 256, 256) // 4, device=dev)
dw = torch.empty(64, 18, 7, 7, device=dev)
hipabi.check(L.straps_stem_wgrad(hipabi.ptr(xd), hipabi.ptr(gd), hipabi.ptr(dw), hipabi.ptr(ws), None, B, 18, 256, 256, 0, None), 'wgrad')
e = (dw.cpu().double() - ref_dw).abs() / ref_dw.abs().max()
print('unit kernel on the oracle draw: max rel err %.2e' % float(e.max()))
ref2 = torch.nn.grad.conv2d_weight(x.double(), (64, 18, 7, 7), ref_draw.float().double(), stride=2, padding=3)
print('fp64 wgrad of the fp32-rounded draw vs autograd: %.2e' % float((ref2 - ref_dw).abs().max() / ref_dw.abs().max()))
# the model path
reg = straps_amd.SingleInputRegressor(18, 18, 3, mean_params=MP)
reg.load_state_dict(sd, strict=True)
reg = reg.to(dev).eval()
cam, pose, shape = reg(xd)
(torch.cat([cam, pose, shape], 1) * coef.to(dev)).sum().backward()
g = reg.image_encoder.conv1.weight.grad.cpu().double()
e = (g - ref_dw).abs() / ref_dw.abs().max()
print('model path: max rel err %.2e; by input channel %s' % (float(e.max()), ['%.0e' % float(e[:, c].max()) for c in range(18)]))
print('by tap row', ['%.0e' % float(e[:, :, r].max()) for r in range(7)], 'by tap col', ['%.0e' % float(e[:, :, :, s].max()) for s in range(7)])
print('by output channel (first 16)', ['%.0e' % float(e[c].max()) for c in range(16)])
ec = e.reshape(64, -1).max(1).values
bad = [int(c) for c in torch.nonzero(ec > 1e-4).flatten()]
print('bad output channels', bad)
yb = F.batch_norm(raw.detach(), sdo['image_encoder.bn1.running_mean'], sdo['image_encoder.bn1.running_var'], sdo['image_encoder.bn1.weight'], sdo['image_encoder.bn1.bias'], False, 0.1, 1e-5)
for c in bad[:6] + [0, 1]:
    print('ch %d: gamma %.4f beta %.4f rmean %.4f rvar %.4f | active frac %.4f | min|y| %.3e | count(|y|<1e-5) %d | dw err %.2e | |dw|max %.3e' % (
        c, float(sdo['image_encoder.bn1.weight'][c]), float(sdo['image_encoder.bn1.bias'][c]), float(sdo['image_encoder.bn1.running_mean'][c]),
        float(sdo['image_encoder.bn1.running_var'][c]), float((yb[:, c] > 0).double().mean()), float(yb[:, c].abs().min()), int((yb[:, c].abs() < 1e-5).sum()),
        float(ec[c]), float(ref_dw[c].abs().max())))
# ---- capture draw of the stem BatchNorm on the unfused path
from straps_amd import autograd_ops as A
saved = {}
orig = A._bn_bwd
def spy(L_, rec, dy, masked, want_dz, grads, planes_sink=None, keep_fp32=True):
    draw, dz = orig(L_, rec, dy, masked, want_dz, grads, planes_sink, keep_fp32)
    if rec['bn'] is reg2.image_encoder.bn1:
        saved['draw'] = draw.clone(); saved['dy'] = dy.clone(); saved['raw'] = rec['raw'].clone(); saved['ss'] = rec['stats'].clone()
    return draw, dz
A._bn_bwd = spy
reg2 = straps_amd.SingleInputRegressor(18, 18, 3, mean_params=MP)
reg2.load_state_dict(sd, strict=True)
reg2 = reg2.to(dev).eval()
reg2.image_encoder.unfused_stem_tail = True
cam, pose, shape = reg2(xd)
(torch.cat([cam, pose, shape], 1) * coef.to(dev)).sum().backward()
gdraw = saved['draw'].cpu().double().permute(0, 3, 1, 2)
d = (gdraw - ref_draw).abs()
pc = d.amax(dim=(0, 2, 3)) / ref_draw.abs().amax(dim=(0, 2, 3))
print('draw rel err per channel: worst %.2e at ch %d; ch32 %.2e' % (float(pc.max()), int(pc.argmax()), float(pc[32])))
ss = saved['ss'].cpu()
print('ch32 scale %.6f shift %.6f mean %.6f invstd %.6f | oracle scale %.6f' % (float(ss[0, 32]), float(ss[1, 32]), float(ss[2, 32]), float(ss[3, 32]),
      float(sdo['image_encoder.bn1.weight'][32] / (sdo['image_encoder.bn1.running_var'][32] + 1e-5).sqrt())))
graw = saved['raw'].cpu().double().permute(0, 3, 1, 2)
print('raw err %.2e' % float((graw - raw.detach()).abs().max()))
m_gpu = (saved['raw'].cpu()[..., 32] * ss[0, 32] + ss[1, 32]) > 0
m_ref = yb[:, 32] > 0
print('mask mismatches ch32: %d of %d; ref inactive %d; gpu draw nonzero where ref inactive: %d' % (int((m_gpu != m_ref).sum()), m_ref.numel(), int((~m_ref).sum()),
      int(((gdraw[:, 32] != 0) & ~m_ref).sum())))
# ---- is it a max-pool near-tie?  arg-max of the GPU's own stem activation vs the float64 one, channel 32
yg = torch.relu(saved['raw'].cpu() * ss[0] + ss[1]).permute(0, 3, 1, 2).double()          # GPU activation (fp32 values)
_, ig = F.max_pool2d(yg[:, 32:33], 3, 2, 1, return_indices=True)
y64 = torch.relu(yb)
_, i64 = F.max_pool2d(y64[:, 32:33], 3, 2, 1, return_indices=True)
diff = (ig != i64)
print('channel 32: %d of %d pooling windows pick another position than float64' % (int(diff.sum()), diff.numel()))
for b, _, ho, wo in diff.nonzero().tolist()[:5]:
    win = y64[b, 32, max(0, 2 * ho - 1):2 * ho + 2, max(0, 2 * wo - 1):2 * wo + 2].reshape(-1).sort(descending=True).values
    print('  window (%d,%d,%d): float64 top two %.9f %.9f (gap %.2e, relative %.1e)' % (b, ho, wo, float(win[0]), float(win[1]), float(win[0] - win[1]), float((win[0] - win[1]) / win[0])))
