"""Per-layer rates of the MFMA kernels INSIDE the real training step (eager launches, HIP-event pairs): where the step's
convolution time goes, layer by layer (forward / data gradient / weight gradient), resnet18 at B=64."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import straps_amd
from straps_amd import hipabi
from straps_amd.train_step import TrainStep

dev = torch.device('cuda:0')
L = hipabi.load()
recs = []


def out(h, k, s, p):
    return (h + 2 * p - k) // s + 1


def wrap(kind, geo, fn):
    B, H, W, Cin, Cout, kh, kw, stride, pad = geo
    fl = 2.0 * B * out(H, kh, stride, pad) * out(W, kw, stride, pad) * Cout * Cin * kh * kw
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); r = fn(); e.record()
    recs.append(('%s %3dx%-3d %4d>%-4d k%d s%d' % (kind, H, W, Cin, Cout, kh, stride), fl, s, e))
    return r


class Proxy:
    def __getattr__(self, k):
        return getattr(L, k)

    def straps_conv_fwd(self, *a):
        return wrap('fwd  ', a[8:17], lambda: L.straps_conv_fwd(*a))

    def straps_conv_dgrad(self, *a):
        return wrap('dgrad', a[4:13], lambda: L.straps_conv_dgrad(*a))

    def straps_conv_wgrad(self, *a):
        return wrap('wgrad', a[4:13], lambda: L.straps_conv_wgrad(*a))




hipabi.lib = lambda: Proxy()
mp = straps_amd.synthetic_mean_params(0)
reg = straps_amd.SingleInputRegressor(18, 18, 3, mean_params=mp).to(dev).train()
smpl = straps_amd.SMPL(straps_amd.synthetic_smpl_model(0), batch_size=64).to(dev)
crit = straps_amd.HomoscedasticUncertaintyWeightedMultiTaskLoss(['verts', 'shape_params', 'pose_params', 'joints2D', 'joints3D']).to(dev)
ts = TrainStep(reg, smpl, crit, 64, mean_shape=mp['shape'], pipeline_data=False)
for _ in range(3):
    ts.step()
torch.cuda.synchronize()
recs.clear()
N = 5
for _ in range(N):
    ts.step()
torch.cuda.synchronize()
agg = {}
for name, fl, s, e in recs:
    a = agg.setdefault(name, [0, 0.0, 0.0])
    a[0] += 1; a[1] += fl; a[2] += s.elapsed_time(e) * 1e-3
tot = sum(v[2] for v in agg.values())
for name, (n, fl, t) in sorted(agg.items(), key=lambda kv: -kv[1][2]):
    print('%-34s x%2d/step %7.1f us %6.1f TFLOP/s %5.1f %%' % (name, n // N, t / n * 1e6, fl / t / 1e12, 100 * t / tot))
print('total %.3f ms/step' % (tot / N * 1e3))
