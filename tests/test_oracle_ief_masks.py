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
