"""Executor of the encoder forward on the HIP kernels (replaces ResNet.forward, reference
models/resnet.py:201-216).  NHWC fp32 activations; one C-ABI call per fused stage.

eval  : conv -> (folded BN scale/shift, residual, ReLU) fused in the conv epilogue.
train : conv emits raw output + per-block (sum, sumsq) partials -> straps_bn_stats_finalize
        (batch mean / biased var, running-stat update with the unbiased var) -> straps_bn_apply.
"""
import torch

from . import hipabi


# The fp32-operand route of the 1x1 layers (csrc/conv_x3f.hip, round 6): a 1x1 convolution whose input has at least this many pixel rows reads the
# fp32 activation itself (and its data gradient the fp32 gradient), so the tensors around it are produced without bf16 planes.  0 switches the
# route off (A/B: bench.py --no-x3f; tests compare the two routes).  The byte-bound layers are the long ones: resnet50's layer1 / layer2 at 32 bodies.
X3F_MIN_ROWS = 16384      # (same-box A/B of the resnet50 step at 32 bodies, profiles/r06_x3f_step_ab.txt: 16384 and 4096 +9.7 %, 2048 +8.4 %, 65536 +6.3 % over the plane route)
# the BatchNorm + ReLU in front of a 1x1 convolution on that route applied in the convolution's operand path (the normalised activation is never
# materialised: straps_conv_fwd_x3f / straps_conv_wgrad_x3f with a_scale); False = an apply pass that writes the fp32 activation (A/B)
X3F_OPERAND_BN = True


def x3f_mode(ctx, conv, B, H, W):
    """does this convolution run on the fp32-operand route in a training step?  1x1 without padding on the bf16x3 route with a long enough pixel axis:
    forward (straps_conv_fwd_x3f), data gradient (straps_conv_dgrad_x3f) and weight gradient (straps_conv_wgrad_x3f) then all read fp32 tensors, and no
    planes of its input activation / output gradient are ever written"""
    if not ctx.x3 or not X3F_MIN_ROWS:
        return False
    Cout, Cin, k = conv.weight.shape[0], conv.weight.shape[1], conv.weight.shape[2]
    stride, pad = conv.stride[0], conv.padding[0]
    if not ctx.L.straps_conv_x3f_supported(Cin, Cout, k, conv.weight.shape[3], stride, pad):
        return False
    Ho, Wo = _conv_out(H, k, stride, pad), _conv_out(W, k, stride, pad)
    return B * Ho * Wo >= X3F_MIN_ROWS


# ReLU decisions of a residual unit as bits for the backward pass (straps_bn_apply_bits_x3 and the *_bits backward entry points); False = the
# fp32-mask forms of rounds 1-3 (tests compare the two: identical gradients)
_RELU_BITS = True

def _conv_out(h, k, s, p):
    return (h + 2 * p - k) // s + 1


class _Ctx:
    """per-forward scratch: library handle, stream, device, training flag, optional tape for backward."""

    def __init__(self, device, training, tape=None):
        self.L = hipabi.lib()
        self.device = device
        self.training = training
        self.tape = tape
        self.defer_nbt = False        # True: the caller bumps every BatchNorm's num_batches_tracked itself (one fused add)
        self.x3 = False               # bf16x3 convolution route (net.conv_precision)
        self.planes = {}              # bf16x3 route: id(tensor) -> (tensor, planes, plane stride) of the activations split so far
        self.net = None

    def empty(self, *shape):
        return torch.empty(*shape, device=self.device, dtype=torch.float32)


def _new_planes(t):
    ps = (t.numel() + 7) // 8 * 8
    return torch.empty(3, ps, device=t.device, dtype=torch.int16), ps


def _bn_train_finish(ctx, bn, raw, part, nblk, rows, residual, relu, rec, apply=True, keep_fp32=True, planes=True):
    C = bn.weight.shape[0]
    L = ctx.L
    if not ctx.training:
        # eval mode with a tape (gradients through frozen BatchNorm statistics, models/resnet.py:47 under .eval()): the running
        # statistics in the role of the batch statistics, nothing is updated; the backward runs with the FROZEN flag
        ss = ctx.net._frozen_bn(bn)
        if rec is not None:
            rec['frozen'] = True
    else:
        ss = ctx.empty(4, C)          # scale, shift, save_mean, save_invstd
        track = bn.track_running_stats and bn.running_mean is not None
        mom = 0.1 if bn.momentum is None else bn.momentum
        hipabi.check(L.straps_bn_stats_finalize(hipabi.ptr(part), nblk, C, rows, hipabi.ptr(bn.weight), hipabi.ptr(bn.bias), bn.eps,
                                                mom, hipabi.ptr(bn.running_mean if track else None),
                                                hipabi.ptr(bn.running_var if track else None), hipabi.ptr(ss[0]), hipabi.ptr(ss[1]),
                                                hipabi.ptr(ss[2]), hipabi.ptr(ss[3]), hipabi.stream_ptr()), 'straps_bn_stats_finalize')
        if track and not ctx.defer_nbt:
            bn.num_batches_tracked.add_(1)
    if not apply:                 # the caller fuses the normalisation into its consumer (stem: straps_bn_relu_maxpool_fwd)
        if rec is not None:
            rec.update(raw=raw, stats=ss, out=None)
        return ss
    if ctx.x3 and relu and (planes or (_RELU_BITS and residual is not None and rec is not None)):
        # bf16x3 route: every ReLU output of the residual stages feeds a convolution -- its planes are written here, not by a split pass.
        # keep_fp32 = False: nothing reads the fp32 activation (its only consumers, the next convolution and that layer's weight
        # gradient, run on the planes; the ReLU mask of the backward is re-derived from raw): it is not written -- y is then an
        # empty tensor that only carries the identity the planes are looked up by.
        # planes = False (round 6): every consumer reads the fp32 tensor (1x1 layers on the fp32-operand route, the pooling): no planes are written
        keep_fp32 = keep_fp32 or not planes
        y = torch.empty_like(raw) if keep_fp32 else raw.new_empty(0)
        planes, ps = _new_planes(raw) if planes else (None, 0)
        if _RELU_BITS and residual is not None and rec is not None:
            # the last BatchNorm of a residual unit, with a backward to come: the unit's ReLU decisions also as bits -- the backward reads
            # them (1/32 of the bytes) wherever it would read this fp32 activation for its sign (autograd_ops._bn_bwd, _conv_dgrad)
            bits = torch.empty(rows, C // 32, device=raw.device, dtype=torch.int32)
            hipabi.check(L.straps_bn_apply_bits_x3(hipabi.ptr(raw), hipabi.ptr(ss[0]), hipabi.ptr(ss[1]), hipabi.ptr(residual),
                                                   hipabi.ptr(y if keep_fp32 else None), hipabi.ptr(planes), ps, hipabi.ptr(bits), rows, C,
                                                   hipabi.stream_ptr()), 'straps_bn_apply_bits_x3')
            if planes is not None:
                ctx.planes[id(y)] = (y, planes, ps)
            rec.update(raw=raw, stats=ss, out=y if keep_fp32 else None, bits=bits)
            return y
        hipabi.check(L.straps_bn_apply_x3(hipabi.ptr(raw), hipabi.ptr(ss[0]), hipabi.ptr(ss[1]), hipabi.ptr(residual), int(relu),
                                          hipabi.ptr(y if keep_fp32 else None), hipabi.ptr(planes), ps, rows, C, hipabi.stream_ptr()),
                     'straps_bn_apply_x3')
        ctx.planes[id(y)] = (y, planes, ps)
        if rec is not None:
            rec.update(raw=raw, stats=ss, out=y if keep_fp32 else None)
        return y
    y = torch.empty_like(raw)
    hipabi.check(L.straps_bn_apply(hipabi.ptr(raw), hipabi.ptr(ss[0]), hipabi.ptr(ss[1]), hipabi.ptr(residual), int(relu),
                                   hipabi.ptr(y), rows, C, hipabi.stream_ptr()), 'straps_bn_apply')
    if rec is not None:
        rec.update(raw=raw, stats=ss, out=y)
    return y


def split3(L, t):
    """fp32 activation / gradient tensor [..., C] (NHWC) -> its three bf16 planes [3][ps] (t = p1 + p2 + p3 exactly) in the chunk-major
    order the bf16x3 convolution kernels read (csrc/common.h cm_index: 32-channel chunks outermost); ps = numel rounded up to 8."""
    n = t.numel()
    C = t.shape[-1]
    ps = (n + 7) // 8 * 8
    planes = torch.empty(3, ps, device=t.device, dtype=torch.int16)
    hipabi.check(L.straps_split3_bf16_cm(hipabi.ptr(t), hipabi.ptr(planes), n // C, C, ps, hipabi.stream_ptr()), 'straps_split3_bf16_cm')
    return planes, ps


def weight_planes(L, w, dgrad=False):
    """chunk-major bf16x3 planes [3][ps] of ONE convolution weight (OIHW fp32, cin % 32 == 0; dgrad: the flipped data-gradient layout,
    cout % 32 == 0) through the batched pack with a single descriptor -- what ResNet.prepack does for all layers at once."""
    import numpy as np
    wd = w.detach().float().contiguous()
    O, C = wd.shape[0], wd.shape[1]
    if (O if dgrad else C) % 32:
        raise RuntimeError('weight_planes: the reduction extent (%d) must be a multiple of 32' % (O if dgrad else C))
    n = wd.numel()
    ps = (n + 7) // 8 * 8
    planes = torch.zeros(3, ps, device=wd.device, dtype=torch.int16)
    d = hipabi.PackDesc(wd.data_ptr(), None, None, O, C, wd.shape[2], wd.shape[3], 0)
    table = torch.from_numpy(np.frombuffer(bytes(d), dtype=np.uint8).copy()).to(wd.device)
    hipabi.check(L.straps_pack_conv_weights_batched_x3(hipabi.ptr(table), 1, n, hipabi.ptr(None if dgrad else planes), hipabi.ptr(planes if dgrad else None),
                                                       ps, hipabi.stream_ptr()), 'straps_pack_conv_weights_batched_x3')
    torch.cuda.current_stream().synchronize()          # (the descriptor table and wd die with this frame)
    return planes, ps


def _planes_of(ctx, t):
    hit = ctx.planes.get(id(t))
    if hit is None or hit[0] is not t:
        hit = (t,) + split3(ctx.L, t)
        ctx.planes[id(t)] = hit
    return hit[1], hit[2]


def _conv_launch(ctx, net, x, wpk, conv, ss, residual, relu, y, part, geom, tile_cfg, fmode=False, a_bn=None):
    """the convolution itself on the route net.conv_precision selects: 'fp32' = exact-fp32 MFMA chain (csrc/conv.hip),
    'bf16x3' = three-plane bf16 operands, six products per term, fp32 accumulate (csrc/conv_x3.hip); fmode: the same arithmetic with the A
    operand read from the fp32 tensor (csrc/conv_x3f.hip: 1x1 layers, x3f_mode)."""
    L = ctx.L
    B, H, W, Cin, Cout, k, stride, pad = geom
    s0, s1 = (hipabi.ptr(ss[0]), hipabi.ptr(ss[1])) if ss is not None else (None, None)
    if fmode:
        if not x.numel():
            raise RuntimeError('fp32-operand convolution: the fp32 activation was not materialised')
        w3, wps = net._packed_weight_x3(conv)
        a0, a1 = (hipabi.ptr(a_bn[0]), hipabi.ptr(a_bn[1])) if a_bn is not None else (None, None)
        hipabi.check(L.straps_conv_fwd_x3f(hipabi.ptr(x), a0, a1, int(a_bn is not None), hipabi.ptr(w3), wps, s0, s1, hipabi.ptr(residual), int(relu), hipabi.ptr(y),
                                           hipabi.ptr(part), B, H, W, Cin, Cout, k, k, stride, pad, tile_cfg, hipabi.stream_ptr()), 'straps_conv_fwd_x3f')
        return
    if getattr(net, 'conv_precision', 'fp32') == 'bf16x3':
        x3, xps = _planes_of(ctx, x)
        w3, wps = net._packed_weight_x3(conv)
        hipabi.check(L.straps_conv_fwd_x3(hipabi.ptr(x3), xps, hipabi.ptr(w3), wps, s0, s1, hipabi.ptr(residual), int(relu), hipabi.ptr(y),
                                          hipabi.ptr(part), B, H, W, Cin, Cout, k, k, stride, pad, tile_cfg, hipabi.stream_ptr()),
                     'straps_conv_fwd_x3')
    else:
        hipabi.check(L.straps_conv_fwd(hipabi.ptr(x), hipabi.ptr(wpk), s0, s1, hipabi.ptr(residual), int(relu), hipabi.ptr(y),
                                       hipabi.ptr(part), B, H, W, Cin, Cout, k, k, stride, pad, tile_cfg, hipabi.stream_ptr()),
                     'straps_conv_fwd')


def conv_stat_blocks(L, net, geom, Ho, Wo, tile_cfg, fmode=False):
    B, H, W, Cin, Cout, k, stride, pad = geom
    if fmode:
        return L.straps_conv_x3f_stat_blocks(B, H, W, Cin, Cout, k, k, stride, pad, tile_cfg)
    if getattr(net, 'conv_precision', 'fp32') == 'bf16x3':
        return L.straps_conv_x3_stat_blocks(B, H, W, Cin, Cout, k, k, stride, pad, tile_cfg)
    return L.straps_conv_stat_blocks(B, Ho, Wo, Cout, k * k * Cin, tile_cfg)


def conv_bn(ctx, net, x, B, H, W, conv, bn, relu, residual=None, tile_cfg=0, keep_fp32=True, planes=True, a_bn=None, defer_apply=False):
    """x NHWC [B,H,W,Cin] -> NHWC [B,Ho,Wo,Cout] through conv + BatchNorm (+residual) (+ReLU).
    keep_fp32 / planes: which forms of the OUTPUT activation its consumers read (training mode on the bf16x3 route).
    a_bn = (scale, shift): x is the RAW output of the previous convolution and that layer's BatchNorm + ReLU is applied in this convolution's operand path
    (fp32-operand route only).  defer_apply: return (raw output, (scale, shift, ...)) instead of the activation -- the consumer applies them that way."""
    L = ctx.L
    Cout, Cin, k = conv.weight.shape[0], conv.weight.shape[1], conv.weight.shape[2]
    stride, pad = conv.stride[0], conv.padding[0]
    Ho, Wo = _conv_out(H, k, stride, pad), _conv_out(W, k, stride, pad)
    wpk = None if ctx.x3 else net._packed_weight(conv)          # (the bf16x3 route reads the weights' planes: _conv_launch)
    y = ctx.empty(B, Ho, Wo, Cout)
    rec = None
    if ctx.tape is not None:
        rec = dict(kind='conv', conv=conv, bn=bn, x=x, geom=(B, H, W, Cin, Cout, k, stride, pad, Ho, Wo), relu=relu,
                   residual=residual)
        ctx.tape[id(conv)] = rec
    if not ctx.training and rec is None:
        ss = net._folded_bn(bn)
        if ctx.x3 and relu:
            # bf16x3 route, inference: every ReLU output feeds a convolution, so the epilogue writes its planes as well (no split pass);
            # the fp32 tensor itself only where something reads it (keep_fp32: a unit's output -- the next identity / the pooling)
            x3, xps = _planes_of(ctx, x)
            w3, wps = net._packed_weight_x3(conv)
            if not keep_fp32:
                y = x.new_empty(0)
            yps = (B * Ho * Wo * Cout + 7) // 8 * 8
            yp = torch.empty(3, yps, device=x.device, dtype=torch.int16)
            hipabi.check(L.straps_conv_fwd_x3p(hipabi.ptr(x3), xps, hipabi.ptr(w3), wps, hipabi.ptr(ss[0]), hipabi.ptr(ss[1]), hipabi.ptr(residual), int(relu),
                                               hipabi.ptr(y if keep_fp32 else None), hipabi.ptr(yp), yps, B, H, W, Cin, Cout, k, k, stride, pad, tile_cfg,
                                               hipabi.stream_ptr()), 'straps_conv_fwd_x3p')
            ctx.planes[id(y)] = (y, yp, yps)
            return y, Ho, Wo
        _conv_launch(ctx, net, x, wpk, conv, ss, residual, relu, y, None, (B, H, W, Cin, Cout, k, stride, pad), tile_cfg)
        return y, Ho, Wo
    # training mode, or eval mode with a tape (frozen statistics: no partials needed)
    fmode = rec is not None and x3f_mode(ctx, conv, B, H, W) and x.numel() > 0
    if a_bn is not None and not fmode:
        raise RuntimeError('operand-path BatchNorm needs the fp32-operand route')
    nblk, part = 0, None
    if ctx.training:
        nblk = conv_stat_blocks(L, net, (B, H, W, Cin, Cout, k, stride, pad), Ho, Wo, tile_cfg, fmode)
        part = ctx.empty(nblk, Cout, 2)
    _conv_launch(ctx, net, x, wpk, conv, None, None, False, y, part, (B, H, W, Cin, Cout, k, stride, pad), tile_cfg, fmode, a_bn)
    if rec is not None and ctx.x3:
        rec['x3'] = ctx.planes.get(id(x))          # (x, planes, plane stride): the weight gradient reads the same planes
        rec['fmode'] = fmode                        # the backward takes the same route (autograd_ops: fp32 gradient, no planes of it)
        rec['a_bn'] = a_bn                          # (x is raw: the weight gradient applies the same scale / shift / ReLU in its operand path)
    if defer_apply:
        ss = _bn_train_finish(ctx, bn, y, part, nblk, B * Ho * Wo, None, relu, rec, apply=False)
        return (y, ss), Ho, Wo
    out = _bn_train_finish(ctx, bn, y, part, nblk, B * Ho * Wo, residual, relu, rec, keep_fp32=keep_fp32, planes=planes)
    return out, Ho, Wo


def encoder_forward(net, x, tape=None, nzmask=None):
    """net: resnet.ResNet; x: float32 [B,C,H,W] NCHW on the GPU.  Returns features [B, 512|2048].
    nzmask: optional non-zero map of x already computed by straps_stem_nzmask (the training step's data pipeline makes it next to
    the input, off the critical path)."""
    hipabi.require_gpu_tensor(x, 'encoder input', torch.float32)
    if x.dim() != 4 or x.shape[1] != net.in_channels:
        raise RuntimeError('encoder expects [B,%d,H,W], got %s' % (net.in_channels, tuple(x.shape)))
    hipabi.require_gpu_tensor(net.conv1.weight, 'encoder parameters (call .to(device))')
    x = x.contiguous()
    B, C, H, W = x.shape
    ctx = _Ctx(x.device, net.training, tape)
    ctx.net = net
    ctx.x3 = getattr(net, 'conv_precision', 'fp32') == 'bf16x3'
    if net.training:
        net._bn_epoch = getattr(net, '_bn_epoch', 0) + 1      # running statistics are about to change: folded-BN cache entries expire
    ctx.defer_nbt = getattr(net, '_nbt_flat', None) is not None
    L = ctx.L
    # ---- stem: conv7x7/s2 + BN + ReLU (models/resnet.py:145-148) ----
    Ho, Wo = _conv_out(H, 7, 2, 3), _conv_out(W, 7, 2, 3)
    wfrag = net._packed_weight(net.conv1, stem=True)
    y = ctx.empty(B, Ho, Wo, 64)
    rec = None
    if tape is not None:
        rec = dict(kind='stem', conv=net.conv1, bn=net.bn1, x=x, geom=(B, C, H, W, Ho, Wo), relu=True, residual=None)
        tape["stem"] = rec
    # non-zero map of the input (the proxy representation is ~98 % exact zeros): the stem kernels skip those cells
    nwords = L.straps_stem_nzmask_words(B, C, H, W)
    if getattr(net, 'dense_stem', False):
        nzmask = torch.empty(nwords, device=x.device, dtype=torch.int32)
        nzmask.fill_(-1)          # A/B switch (bench.py --dense-stem): every cell marked non-zero = the plain dense convolution
    elif nzmask is not None:
        if nzmask.dtype != torch.int32 or nzmask.numel() != nwords or nzmask.device != x.device or not nzmask.is_contiguous():
            raise RuntimeError('encoder_forward: nzmask must be a contiguous int32 tensor of %d words on %s' % (nwords, x.device))
    else:
        nzmask = torch.empty(nwords, device=x.device, dtype=torch.int32)
        hipabi.check(L.straps_stem_nzmask(hipabi.ptr(x), hipabi.ptr(nzmask), B, C, H, W, hipabi.stream_ptr()), 'straps_stem_nzmask')
    if rec is not None:
        rec['nzmask'] = nzmask
    if not net.training and tape is None:
        ss = net._folded_bn(net.bn1)
        hipabi.check(L.straps_stem_fwd(hipabi.ptr(x), hipabi.ptr(wfrag), hipabi.ptr(ss[0]), hipabi.ptr(ss[1]), 1, hipabi.ptr(y),
                                       None, hipabi.ptr(nzmask), B, C, H, W, hipabi.stream_ptr()), 'straps_stem_fwd')
    else:
        # training mode -- or eval mode with a tape: the same kernels with the running statistics as frozen constants (_bn_train_finish)
        nblk = L.straps_stem_stat_blocks(B, H, W)
        part = ctx.empty(nblk, 64, 2)
        hipabi.check(L.straps_stem_fwd(hipabi.ptr(x), hipabi.ptr(wfrag), None, None, 0, hipabi.ptr(y), hipabi.ptr(part), hipabi.ptr(nzmask),
                                       B, C, H, W, hipabi.stream_ptr()), 'straps_stem_fwd')
        if tape is not None and not getattr(net, 'unfused_stem_tail', False):      # (attribute: A/B switch for tests / tools)
            # bn1 + relu + maxpool in one pass over the raw conv output: the 268 MB (B=64) activation is never written, the
            # backward gathers the pooled gradient instead of materialising it (straps_bn_bwd_pooled)
            ss = _bn_train_finish(ctx, net.bn1, y, part, nblk, B * Ho * Wo, None, True, rec, apply=False)
            H, W = Ho, Wo
            Hp, Wp = _conv_out(H, 3, 2, 1), _conv_out(W, 3, 2, 1)
            p = ctx.empty(B, Hp, Wp, 64)
            idx = torch.empty(B, Hp, Wp, 64, device=x.device, dtype=torch.uint8)
            u0 = net.layer1[0]
            cons = [u0.conv_bn_pairs()[0][0]] + ([u0.downsample[0]] if u0.downsample is not None else [])
            if ctx.x3 and not all(x3f_mode(ctx, cv, B, Hp, Wp) for cv in cons):      # (planes of the pooled output unless every reader takes the fp32 tensor)
                planes, pstride = _new_planes(p)
                hipabi.check(L.straps_bn_relu_maxpool_fwd_x3(hipabi.ptr(y), hipabi.ptr(ss[0]), hipabi.ptr(ss[1]), hipabi.ptr(p), hipabi.ptr(idx),
                                                             hipabi.ptr(planes), pstride, B, H, W, 64, hipabi.stream_ptr()),
                             'straps_bn_relu_maxpool_fwd_x3')
                ctx.planes[id(p)] = (p, planes, pstride)
            else:
                hipabi.check(L.straps_bn_relu_maxpool_fwd(hipabi.ptr(y), hipabi.ptr(ss[0]), hipabi.ptr(ss[1]), hipabi.ptr(p), hipabi.ptr(idx),
                                                          B, H, W, 64, hipabi.stream_ptr()), 'straps_bn_relu_maxpool_fwd')
            tape['maxpool'] = dict(kind='maxpool_fused', out=p, idx=idx, geom=(B, H, W, 64, Hp, Wp))
            return _residual_stages(ctx, net, p, B, Hp, Wp, tape)
        y = _bn_train_finish(ctx, net.bn1, y, part, nblk, B * Ho * Wo, None, True, rec, planes=False)      # (its consumer is the pooling: no planes)
    # ---- maxpool 3x3/s2/p1 (:149) ----
    H, W = Ho, Wo
    Hp, Wp = _conv_out(H, 3, 2, 1), _conv_out(W, 3, 2, 1)
    p = ctx.empty(B, Hp, Wp, 64)
    if tape is not None:
        idx = torch.empty(B, Hp, Wp, 64, device=x.device, dtype=torch.uint8)      # arg-max tap for the backward
        hipabi.check(L.straps_maxpool_fwd_idx(hipabi.ptr(y), hipabi.ptr(p), hipabi.ptr(idx), B, H, W, 64, hipabi.stream_ptr()),
                     'straps_maxpool_fwd_idx')
        tape['maxpool'] = dict(kind='maxpool', x=y, out=p, idx=idx, geom=(B, H, W, 64, Hp, Wp))
    else:
        hipabi.check(L.straps_maxpool_fwd(hipabi.ptr(y), hipabi.ptr(p), B, H, W, 64, hipabi.stream_ptr()), 'straps_maxpool_fwd')
    return _residual_stages(ctx, net, p, B, Hp, Wp, tape)


def _residual_stages(ctx, net, y, B, H, W, tape):
    L = ctx.L
    train_tape = ctx.x3 and tape is not None
    units = [u for li in range(1, 5) for u in getattr(net, 'layer%d' % li)]

    def needs(t_conv, b, h, w):
        """(fp32, planes) forms a training-mode activation must exist in for the convolution t_conv that reads it (and for that layer's weight gradient)"""
        if x3f_mode(ctx, t_conv, b, h, w):
            return True, False
        k2, s2, p2 = t_conv.weight.shape[2], t_conv.stride[0], t_conv.padding[0]
        return (not L.straps_conv_wgrad_x3_on_planes(b, h, w, t_conv.weight.shape[1], t_conv.weight.shape[0], k2, k2, s2, p2)), True
    # ---- residual stages (:150-156) ----
    for ui, unit in enumerate(units):
        idt = y
        if unit.downsample is not None:
            idt, _, _ = conv_bn(ctx, net, y, B, H, W, unit.downsample[0], unit.downsample[1], relu=False)
        pairs = unit.conv_bn_pairs()
        t, h, w = y, H, W
        a_bn = None
        for ci, (conv, bn) in enumerate(pairs):
            last = ci == len(pairs) - 1
            keep, planes = True, True
            if ctx.x3 and not ctx.training and tape is None:
                keep = last                     # inference: only a unit's output is read as fp32 (identity of the next unit, pooling)
            ho2, wo2 = _conv_out(h, conv.weight.shape[2], conv.stride[0], conv.padding[0]), _conv_out(w, conv.weight.shape[2], conv.stride[0], conv.padding[0])
            defer = False
            if train_tape and not last:
                # the fp32 activation between two convolutions of a unit is dead when the next layer (and its weight gradient) reads planes; its planes
                # are dead when that layer runs on the fp32-operand route -- and then the BatchNorm + ReLU itself moves into that layer's operand path
                keep, planes = needs(pairs[ci + 1][0], B, ho2, wo2)
                defer = X3F_OPERAND_BN and keep and not planes
            elif train_tape and last:
                # a unit's output: fp32 for the identity / the pooling; planes only if a convolution of the next unit reads planes
                planes = False
                if ui + 1 < len(units):
                    nxt = units[ui + 1]
                    cons = [nxt.conv_bn_pairs()[0][0]] + ([nxt.downsample[0]] if nxt.downsample is not None else [])
                    planes = any(needs(cv, B, ho2, wo2)[1] for cv in cons)
            t, h, w = conv_bn(ctx, net, t, B, h, w, conv, bn, relu=True, residual=idt if last else None, keep_fp32=keep, planes=planes, a_bn=a_bn,
                              defer_apply=defer)
            a_bn = None
            if defer:
                t, a_bn = t          # (raw output, its BatchNorm's scale / shift / mean / invstd)
        y, H, W = t, h, w
    # ---- global average pool + flatten (:213-214) ----
    Cf = y.shape[3]
    feat = ctx.empty(B, Cf)
    hipabi.check(L.straps_gap_fwd(hipabi.ptr(y), hipabi.ptr(feat), B, H * W, Cf, hipabi.stream_ptr()), 'straps_gap_fwd')
    if tape is not None:
        tape["gap"] = dict(kind="gap", x=y, geom=(B, H * W, Cf))
    return feat
