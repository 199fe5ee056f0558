"""Test helper: the piecewise-linear decisions (ReLU masks, max-pool arg-max taps) the GPU encoder took on an input, compared unit by
unit with the float64 oracle's, so that a gradient comparison can be made against the function the GPU actually differentiated.

A ReLU whose pre-activation sits within the evaluation error of zero -- or a pooling window whose two largest candidates agree to within
it -- may be decided differently in fp32 than in float64; both gradients are then right, for two functions that differ by that unit's
whole term.  The whole-network gradient tests do not widen their bars for this: they identify every differing decision, check that it IS a
tie (|z64| / the float64 gap within a few times the activation error observed on the same layer), and re-run the float64 oracle with the
GPU's decisions forced (oracle/straps_oracle.py resnet_forward(decisions=...), ief_forward(relu_masks=...))."""
import torch

from straps_amd.encoder_exec import encoder_forward


def gpu_encoder_decisions(net, x):
    """-> decisions_from_tape() of a taped forward on x (the kernels of loss.backward()'s forward: deterministic), features"""
    tape = {}
    with torch.no_grad():
        feat = encoder_forward(net, x, tape)
    return decisions_from_tape(net, tape), feat


def decisions_from_tape(net, tape):
    """tape: the dict encoder_forward filled.  -> {'relu': [bool NCHW masks in the oracle's evaluation order], 'pool': long
    [B,64,Hp,Wp], 'act': [post-ReLU activations, NCHW float64 on the CPU, same order]}"""

    def act(rec):
        if rec.get('out') is not None and rec['out'].numel():
            y = rec['out']
        else:
            # not materialised (fused stem tail / plane-only activations): the kernels' own expression, fmaf(raw, scale, shift) (+ residual);
            # the float64 product is exact and the sum correctly rounded, so the sign -- the decision -- is the fmaf's
            ss = rec['stats']
            y = (rec['raw'].double() * ss[0].double() + ss[1].double()).float()
            if rec.get('residual') is not None:
                y = y + rec['residual']
            y = y.clamp_min(0)
        return y.permute(0, 3, 1, 2).cpu().double()
    acts = [act(tape['stem'])]
    for li in range(1, 5):
        for unit in getattr(net, 'layer%d' % li):
            for conv, _ in unit.conv_bn_pairs():
                acts.append(act(tape[id(conv)]))
    pool = tape['maxpool']['idx'].permute(0, 3, 1, 2).cpu().long()
    return {'relu': [a > 0 for a in acts], 'pool': pool, 'act': acts}


ERR_CAP = 2e-5      # a layer's activation error may be at most this fraction of the layer's largest float64 pre-activation (resnet18: measured 4.6e-6)
ERR_CAP_R50 = 2e-4  # resnet50: 53 convolutions deep, training-mode BatchNorm over 8 .. 32 bodies: measured 4.2e-5 .. 5.1e-5 on its worst layer, resnet18 4.7e-6 .. 5.7e-6 (a kernel that is wrong sits at 1e-2)


def compare_encoder_decisions(gpu, rec64, err_cap=ERR_CAP):
    """gpu: gpu_encoder_decisions()[0]; rec64: the oracle's {'record': True} dict after its float64 forward.
    -> (number of differing ReLU decisions, number of differing pooling decisions, worst |z64| of a differing ReLU relative to the
    activation error observed on its layer, worst float64 gap of a differing window relative to the stem's activation error, worst
    activation error of a layer relative to its largest float64 pre-activation).
    A ratio <= ~4 is a tie: the other side of the decision lies within the evaluation error.
    The tie scale is the GPU's OWN observed error on the layer (max |a - z64| over the units both sides switch on), so it is capped here
    against the float64 values alone (a layer that was systematically off must not widen its own tie window): the
    error has to stay below err_cap x max |z64| of the layer -- fp32-class evaluation -- or the comparison fails outright."""
    assert len(gpu['relu']) == len(gpu['act']) == len(rec64['z']), 'the two evaluations list different numbers of ReLUs'
    n_relu, worst_relu, worst_rel, worst_li = 0, 0.0, 0.0, -1
    errs = []
    for li, (m, a, z) in enumerate(zip(gpu['relu'], gpu['act'], rec64['z'])):
        z = z.double()
        both = m & (z > 0)
        err = float((a - z)[both].abs().max()) if both.any() else 0.0
        zmax = float(z.abs().max())
        if err / max(zmax, 1e-30) > worst_rel:
            worst_rel, worst_li = err / max(zmax, 1e-30), li
        errs.append(err)
        diff = m != (z > 0)
        k = int(diff.sum())
        if k:
            n_relu += k
            worst_relu = max(worst_relu, float(z[diff].abs().max()) / max(err, 1e-30))
    win = rec64['pool_windows'].double()
    best = win.max(dim=4).values
    chosen = win.gather(4, gpu['pool'][..., None]).squeeze(4)
    diff = chosen < best
    n_pool = int(diff.sum())
    worst_pool = float((best - chosen)[diff].max()) / max(errs[0], 1e-30) if n_pool else 0.0
    assert worst_rel <= err_cap, ('ReLU layer %d: activation error %.2e of the layer\'s largest float64 pre-activation exceeds the cap %.1e -- the layer is off '
                                  'by more than fp32 evaluation error; its tie window would be meaningless' % (worst_li, worst_rel, err_cap))
    return n_relu, n_pool, worst_relu, worst_pool, worst_rel


def gpu_ief_masks(ief, feat):
    """ReLU decisions of the IEF head on the GPU for the features `feat`: [(mask fc1, mask fc2)] per iteration (CPU bool [B,H])"""
    tape = []
    with torch.no_grad():
        ief.forward_estimate(feat, tape)
    return [(rec['h1'].detach().cpu() > 0, rec['h2'].detach().cpu() > 0) for rec in tape]
