"""CPU test of the oracle's forced-ReLU-decision mode of the IEF head (oracle/straps_oracle.py::ief_forward, used by the GPU whole-step
tests to evaluate the gradient of the function the GPU differentiated when a pre-activation is a rounding-level tie)."""
import torch

import straps_oracle as O


def _sd(F_, H, P, g):
    sd = {}
    for name, (o, i) in (('fc1', (H, F_ + P)), ('fc2', (H, H)), ('fc3', (P, H))):
        sd['ief_module.%s.weight' % name] = (torch.randn(o, i, generator=g, dtype=torch.float64) / i ** 0.5).requires_grad_()
        sd['ief_module.%s.bias' % name] = (torch.randn(o, generator=g, dtype=torch.float64) * 0.1).requires_grad_()
    return sd


def test_forced_masks_reproduce_relu_and_flip_one_unit_by_its_rank_one_term():
    g = torch.Generator().manual_seed(0)
    F_, H, P, B = 24, 32, 157, 5
    sd = _sd(F_, H, P, g)
    feat = torch.randn(B, F_, generator=g, dtype=torch.float64)
    init = torch.randn(P, generator=g, dtype=torch.float64) * 0.1
    taps = []
    est = O.ief_forward(feat, sd, init, 3, taps=taps)[3]
    assert len(taps) == 3 and taps[0][0].shape == (B, H)
    masks = [(z1 > 0, z2 > 0) for z1, z2 in taps]
    est_m = O.ief_forward(feat, sd, init, 3, relu_masks=masks)[3]
    assert torch.equal(est, est_m)
    # gradients agree too
    ga = torch.autograd.grad(est.square().sum(), [sd['ief_module.fc1.weight']])[0]
    gb = torch.autograd.grad(est_m.square().sum(), [sd['ief_module.fc1.weight']])[0]
    assert torch.equal(ga, gb)
    # flipping the decision of the unit closest to zero changes the output by at most |z| * |W3 column| (a tie is a small perturbation
    # of the value but a whole rank-one term of the gradient)
    z2 = taps[2][1]
    r, u = divmod(int(z2.abs().argmin()), H)
    flipped = [(a.clone(), b.clone()) for a, b in masks]
    flipped[2][1][r, u] = ~flipped[2][1][r, u]
    est_f = O.ief_forward(feat, sd, init, 3, relu_masks=flipped)[3]
    bound = float(z2[r, u].abs() * sd['ief_module.fc3.weight'][:, u].abs().max())
    assert 0 < float((est_f - est).abs().max()) <= bound * (1 + 1e-12)
    assert torch.equal(est_f[torch.arange(B) != r], est[torch.arange(B) != r])


def _encoder_decisions_of(x, sd, layers, training=False):
    """the oracle's own ReLU / max-pool decisions, recorded by running its plain forward with hooks on the two functionals"""
    import torch.nn.functional as F
    rec = {'relu': [], 'pool': None}
    relu0, pool0 = F.relu, F.max_pool2d

    def relu(z, *a, **k):
        rec['relu'].append(z > 0)
        return relu0(z, *a, **k)

    def pool(y, *a, **k):
        out, idx = F.max_pool2d_with_indices(y, 3, 2, 1)
        W = y.shape[3]
        hi, wi = idx // W, idx % W
        ho = torch.arange(out.shape[2]).view(1, 1, -1, 1)
        wo = torch.arange(out.shape[3]).view(1, 1, 1, -1)
        rec['pool'] = (hi - (2 * ho - 1)) * 3 + (wi - (2 * wo - 1))
        return out
    F.relu, F.max_pool2d = relu, pool
    try:
        sdc = {k: v.clone() for k, v in sd.items()}
        with torch.no_grad():
            O.resnet_forward(x, sdc, layers, training)
    finally:
        F.relu, F.max_pool2d = relu0, pool0
    return rec


def test_forced_encoder_decisions_reproduce_the_plain_forward_and_its_gradients():
    import straps_amd
    torch.manual_seed(3)
    for layers in (18, 50):
        net = straps_amd.SingleInputRegressor(18, layers, 3, mean_params=straps_amd.synthetic_mean_params(0))
        sd = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in net.state_dict().items()}
        x = torch.rand(2, 18, 64, 64, dtype=torch.float64)
        for training in (False, True):
            dec = _encoder_decisions_of(x, sd, layers, training)
            n_relu = {18: 1 + 8 * 2, 50: 1 + 16 * 3}[layers]
            assert len(dec['relu']) == n_relu and dec['pool'].shape == (2, 64, 16, 16)
            assert int(dec['pool'].min()) >= 0 and int(dec['pool'].max()) <= 8
            w = 'image_encoder.layer1.0.conv1.weight'
            outs = []
            for d in (None, dec):
                sdc = {k: v.clone() for k, v in sd.items()}
                sdc[w].requires_grad_(True)
                sdc['image_encoder.conv1.weight'].requires_grad_(True)
                f = O.resnet_forward(x, sdc, layers, training, decisions=d)
                g = torch.autograd.grad(f.square().sum(), [sdc[w], sdc['image_encoder.conv1.weight']])
                outs.append((f.detach(), g))
            assert torch.equal(outs[0][0], outs[1][0])
            for a, b in zip(outs[0][1], outs[1][1]):
                assert float((a - b).abs().max()) <= 1e-12 * float(a.abs().max())
