"""torch.autograd glue: the drop-in modules stay ordinary autograd citizens (`loss.backward()`,
`optimiser.step()` of the reference train loop :230-233 work unchanged) while every gradient is
computed by the HIP kernels behind the C ABI.  Nothing here does arithmetic in torch.
"""
import ctypes as C

import torch

from . import hipabi
from .encoder_exec import encoder_forward
from .ief_module import EST_LD


def _empty_like(t):
    return torch.empty_like(t, memory_format=torch.contiguous_format)


class GradSink(dict):
    """{param: grad}.  `views` (optional) maps a parameter to a preallocated gradient tensor -- e.g. a slice of the
    flat gradient buffer of train_step.TrainStep -- so kernels write gradients in place (no copies, one all-reduce)."""

    def __init__(self, views=None):
        super().__init__()
        self.views = views

    def buf(self, p, zero=False):
        if self.views is not None and p in self.views:
            return self.views[p]                      # the owner zeroes the flat buffer once per step
        return torch.zeros_like(p) if zero else torch.empty_like(p)


class _SideStream:
    """runs independent work (the weight gradients) on a second HIP stream so MFMA-bound wgrad kernels overlap the
    HBM-bound BatchNorm passes and the tails of the data-gradient kernels of the main stream.  Tensors produced on the
    main stream and consumed on the side stream are kept alive until join()."""

    def __init__(self, stream):
        self.s, self.keep = stream, []

    def run(self, fn, *tensors):
        if self.s is None:
            return fn()
        self.keep.extend(tensors)
        self.s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.s):
            return fn()

    def join(self):
        if self.s is not None:
            torch.cuda.current_stream().wait_stream(self.s)
            self.keep.clear()


# ------------------------------------------------------------------------------------------ encoder
_FUSE_BN_SUMS = True         # (A/B switch: tools set autograd_ops._FUSE_BN_SUMS = False; the product never reads the environment)
_SPARSE_STEM_TAIL = True     # (A/B switch: tools set autograd_ops._SPARSE_STEM_TAIL = False)


def _packed_dgrad_weight(net, conv):
    w = conv.weight

    def make():
        wd = w.detach().contiguous()
        out = torch.empty(wd.numel(), device=wd.device, dtype=torch.float32)
        hipabi.check(hipabi.lib().straps_pack_conv_weight_dgrad(hipabi.ptr(wd), hipabi.ptr(out), wd.shape[0], wd.shape[1], wd.shape[2],
                                                                wd.shape[3], hipabi.stream_ptr()), 'straps_pack_conv_weight_dgrad')
        return out
    return net._cached(('wd', id(conv)), [w], make)


def _bn_bwd(L, rec, dy, masked, want_dz, grads, planes_sink=None, keep_fp32=True, mask_bits=None):
    """BatchNorm(train) + ReLU backward of one tape record; returns (draw, dz|None).
    planes_sink (bf16x3 route): dict that receives id(draw) -> (draw, planes, plane stride), written by the same kernel pass.
    mask_bits: ReLU decisions [rows][C / 32] that mask dy (a residual unit's, for its downsample BatchNorm); a record that carries its own
    bits (encoder_exec: the last BatchNorm of a residual unit) uses those in place of its fp32 activation.  No dz with bits: the consumers
    of the masked gradient apply the bits to dy themselves."""
    bn = rec['bn']
    raw, ss = rec['raw'], rec['stats']
    rows, Cc = raw.numel() // raw.shape[-1], raw.shape[-1]
    ws = torch.empty(L.straps_bn_bwd_workspace_bytes(rows, Cc) // 4, device=raw.device, dtype=torch.float32)
    dgamma, dbeta = grads.buf(bn.weight), grads.buf(bn.bias)
    # keep_fp32 = False (bf16x3 route): both consumers of this gradient -- the data gradient and the weight gradient of the layer --
    # read its planes, so the fp32 tensor is not written; `draw` is then an empty tensor that only carries the identity
    keep_fp32 = keep_fp32 or planes_sink is None
    draw = _empty_like(raw) if keep_fp32 else raw.new_empty(0)
    dz = _empty_like(raw) if want_dz else None
    # ReLU mask: without a residual the activation is relu(raw*scale + shift), so the kernel re-derives it from raw (one
    # tensor read less); with a residual it has to read the stored activation
    from_raw = masked and rec.get('residual') is None
    bits = mask_bits if mask_bits is not None else (rec.get('bits') if masked and not from_raw else None)
    if bits is not None and want_dz:
        raise RuntimeError('BatchNorm backward on ReLU bits writes no masked gradient (dz)')
    planes, ps = None, 0
    if planes_sink is not None:
        ps = (raw.numel() + 7) // 8 * 8
        planes = torch.empty(3, ps, device=raw.device, dtype=torch.int16)
        planes_sink[id(draw)] = (draw, planes, ps)
    flags = 2 if rec.get('frozen') else 0         # bit 1: eval-mode BatchNorm, statistics are constants (encoder_exec._bn_train_finish)
    fused = rec.pop('bwd_partials', None)         # (partials, blocks, dy they belong to): the sums came out of the data gradient's epilogue
    if bits is not None:
        args = (hipabi.ptr(dy), hipabi.ptr(bits), hipabi.ptr(raw), hipabi.ptr(ss[2]), hipabi.ptr(ss[3]), hipabi.ptr(bn.weight), hipabi.ptr(dgamma),
                hipabi.ptr(dbeta), hipabi.ptr(draw if keep_fp32 else None), hipabi.ptr(planes), ps)
        if fused is not None and fused[2] is dy and masked:
            hipabi.check(L.straps_bn_bwd_finish_bits_x3(*args, hipabi.ptr(fused[0]), fused[1], hipabi.ptr(ws), rows, Cc, flags, hipabi.stream_ptr()),
                         'straps_bn_bwd_finish_bits_x3')
        else:
            hipabi.check(L.straps_bn_bwd_bits_x3(*args, hipabi.ptr(ws), rows, Cc, flags, hipabi.stream_ptr()), 'straps_bn_bwd_bits_x3')
    elif fused is not None and fused[2] is dy and masked:
        hipabi.check(L.straps_bn_bwd_finish_x3(hipabi.ptr(dy), hipabi.ptr(rec['out'] if not from_raw else None), hipabi.ptr(raw), hipabi.ptr(ss[2]),
                                               hipabi.ptr(ss[3]), hipabi.ptr(bn.weight), hipabi.ptr(ss[0] if from_raw else None),
                                               hipabi.ptr(ss[1] if from_raw else None), hipabi.ptr(dgamma), hipabi.ptr(dbeta),
                                               hipabi.ptr(draw if keep_fp32 else None), hipabi.ptr(dz), hipabi.ptr(planes), ps, hipabi.ptr(fused[0]),
                                               fused[1], hipabi.ptr(ws), rows, Cc, flags, hipabi.stream_ptr()), 'straps_bn_bwd_finish_x3')
    else:
        hipabi.check(L.straps_bn_bwd_x3(hipabi.ptr(dy), hipabi.ptr(rec['out'] if masked and not from_raw else None), hipabi.ptr(raw), hipabi.ptr(ss[2]),
                                        hipabi.ptr(ss[3]), hipabi.ptr(bn.weight), hipabi.ptr(ss[0] if from_raw else None),
                                        hipabi.ptr(ss[1] if from_raw else None), hipabi.ptr(dgamma), hipabi.ptr(dbeta), hipabi.ptr(draw if keep_fp32 else None),
                                        hipabi.ptr(dz), hipabi.ptr(planes), ps, hipabi.ptr(ws), rows, Cc, flags, hipabi.stream_ptr()), 'straps_bn_bwd')
    grads[bn.weight] = dgamma
    grads[bn.bias] = dbeta
    return draw, dz


def _conv_wgrad(L, rec, draw, grads, planes_sink=None):
    conv = rec['conv']
    B, H, W, Cin, Cout, k, stride, pad, Ho, Wo = rec['geom']
    dw = grads.buf(conv.weight)
    if rec.get('fmode'):
        # the fp32-operand route (csrc/conv_wgrad_x3f.hip): both operands are the fp32 tensors; an input that is a RAW convolution output gets its
        # BatchNorm + ReLU in the operand path (rec['a_bn']: the activation was never materialised)
        a_bn = rec.get('a_bn')
        wsf = torch.empty(max(L.straps_conv_wgrad_x3f_workspace_bytes(B, H, W, Cin, Cout, k, k, stride, pad) // 4, 1), device=draw.device, dtype=torch.float32)
        hipabi.check(L.straps_conv_wgrad_x3f(hipabi.ptr(rec['x']), hipabi.ptr(a_bn[0] if a_bn is not None else None), hipabi.ptr(a_bn[1] if a_bn is not None else None),
                                             int(a_bn is not None), hipabi.ptr(draw), hipabi.ptr(dw), hipabi.ptr(wsf), B, H, W, Cin, Cout, k, k, stride, pad, 0,
                                             hipabi.stream_ptr()), 'straps_conv_wgrad_x3f')
        grads[conv.weight] = dw
        return
    ws = torch.empty(L.straps_conv_wgrad_workspace_bytes(B, H, W, Cin, Cout, k, k, stride, pad) // 4, device=draw.device, dtype=torch.float32)
    fp = lambda t: hipabi.ptr(t if t is not None and t.numel() else None)      # (an empty tensor = "fp32 copy not materialised")
    xh = rec.get('x3')
    gh = planes_sink.get(id(draw)) if planes_sink is not None else None
    if xh is not None and gh is not None and xh[0] is rec['x'] and gh[0] is draw:
        # bf16x3 route: the planes the forward convolution / the data gradient read anyway (3x3 stride-1 layers; other shapes fall
        # through to the fp32 kernels inside the entry point)
        hipabi.check(L.straps_conv_wgrad_x3(fp(rec['x']), fp(draw), hipabi.ptr(xh[1]), xh[2], hipabi.ptr(gh[1]), gh[2], hipabi.ptr(dw),
                                            hipabi.ptr(ws), B, H, W, Cin, Cout, k, k, stride, pad, 0, hipabi.stream_ptr()), 'straps_conv_wgrad_x3')
    else:
        hipabi.check(L.straps_conv_wgrad(hipabi.ptr(rec['x']), hipabi.ptr(draw), hipabi.ptr(dw), hipabi.ptr(ws), B, H, W, Cin, Cout, k, k, stride,
                                         pad, 0, hipabi.stream_ptr()), 'straps_conv_wgrad')
    grads[conv.weight] = dw


def _conv_dgrad(L, net, rec, draw, addend, planes_sink=None, bn_next=None, addend_bits=None):
    """data gradient of one convolution.  bn_next (bf16x3 route): tape record of the BatchNorm (+ ReLU) whose output this convolution
    read -- the returned gradient is that BatchNorm's dy, and the two sums of its backward are accumulated in this launch's epilogue
    (straps_conv_dgrad_x3_bn) and left in bn_next['bwd_partials'] for _bn_bwd.
    addend_bits: the addend is the UNMASKED gradient of the later unit's output and these are that unit's ReLU decisions (the epilogue adds
    bit ? addend : 0); a bn_next that carries bits has its sums masked by them instead of by its fp32 activation."""
    conv = rec['conv']
    B, H, W, Cin, Cout, k, stride, pad, Ho, Wo = rec['geom']
    dx = torch.empty(B, H, W, Cin, device=draw.device, dtype=torch.float32)
    if rec.get('fmode'):
        # the fp32-operand route (csrc/conv_x3f.hip): the gradient is read as the fp32 tensor -- no planes of it were written (encoder_exec.x3f_mode)
        if not draw.numel():
            raise RuntimeError('fp32-operand data gradient: the fp32 gradient was not materialised')
        w3, wps = net._packed_weight_x3(conv, dgrad=True)
        raw = nbits = msc = msh = mean = invstd = part = None
        if bn_next is not None and _FUSE_BN_SUMS:
            ssn = bn_next['stats']
            from_raw = bn_next.get('residual') is None
            nbits = None if from_raw else bn_next.get('bits')
            if from_raw or nbits is not None:          # (a residual unit's BatchNorm without bits -- _RELU_BITS off -- keeps the plane route's fp32-mask form below)
                nblk = L.straps_conv_dgrad_x3f_bn_blocks(B, H, W, Cin, Cout, k, k, stride, pad, 0)
                part = torch.empty(nblk, Cin, 2, device=dx.device, dtype=torch.float64)
                raw, mean, invstd = bn_next['raw'], ssn[2], ssn[3]
                if from_raw:
                    msc, msh = ssn[0], ssn[1]
        if bn_next is None or not _FUSE_BN_SUMS or part is not None:
            hipabi.check(L.straps_conv_dgrad_x3f(hipabi.ptr(draw), hipabi.ptr(w3), wps, hipabi.ptr(addend), hipabi.ptr(addend_bits), hipabi.ptr(dx), B, H, W, Cin, Cout,
                                                 k, k, stride, pad, 0, hipabi.ptr(raw), hipabi.ptr(nbits), hipabi.ptr(msc), hipabi.ptr(msh), hipabi.ptr(mean),
                                                 hipabi.ptr(invstd), hipabi.ptr(part), hipabi.stream_ptr()), 'straps_conv_dgrad_x3f')
            if part is not None:
                bn_next['bwd_partials'] = (part, nblk, dx)
            return dx
    if getattr(net, 'conv_precision', 'fp32') == 'bf16x3':
        if not draw.numel() and (planes_sink is None or planes_sink.get(id(draw)) is None):
            raise RuntimeError('data gradient: neither the fp32 gradient nor its planes exist')
        hit = planes_sink.get(id(draw)) if planes_sink is not None else None      # (kept until the backward ends: the weight gradient may still be reading them on the side stream)
        if hit is not None and hit[0] is draw:
            g3, gps = hit[1], hit[2]
        else:
            from .encoder_exec import split3
            g3, gps = split3(L, draw)
        w3, wps = net._packed_weight_x3(conv, dgrad=True)
        if bn_next is not None and _FUSE_BN_SUMS:
            ssn = bn_next['stats']
            from_raw = bn_next.get('residual') is None
            nblk = L.straps_conv_dgrad_x3_bn_blocks(B, H, W, Cin, Cout, k, k, stride, pad, 0)
            part = torch.empty(nblk, Cin, 2, device=dx.device, dtype=torch.float64)
            nbits = None if from_raw else bn_next.get('bits')
            if addend_bits is not None or nbits is not None:
                hipabi.check(L.straps_conv_dgrad_x3_bn_bits(hipabi.ptr(g3), gps, hipabi.ptr(w3), wps, hipabi.ptr(addend), hipabi.ptr(dx), B, H, W, Cin,
                                                            Cout, k, k, stride, pad, 0, hipabi.ptr(bn_next['raw']),
                                                            hipabi.ptr(None if from_raw or nbits is not None else bn_next['out']),
                                                            hipabi.ptr(ssn[0] if from_raw else None), hipabi.ptr(ssn[1] if from_raw else None),
                                                            hipabi.ptr(ssn[2]), hipabi.ptr(ssn[3]), hipabi.ptr(part), hipabi.ptr(addend_bits),
                                                            hipabi.ptr(nbits), hipabi.stream_ptr()), 'straps_conv_dgrad_x3_bn_bits')
                bn_next['bwd_partials'] = (part, nblk, dx)
                return dx
            hipabi.check(L.straps_conv_dgrad_x3_bn(hipabi.ptr(g3), gps, hipabi.ptr(w3), wps, hipabi.ptr(addend), hipabi.ptr(dx), B, H, W, Cin, Cout,
                                                   k, k, stride, pad, 0, hipabi.ptr(bn_next['raw']), hipabi.ptr(None if from_raw else bn_next['out']),
                                                   hipabi.ptr(ssn[0] if from_raw else None), hipabi.ptr(ssn[1] if from_raw else None),
                                                   hipabi.ptr(ssn[2]), hipabi.ptr(ssn[3]), hipabi.ptr(part), hipabi.stream_ptr()),
                         'straps_conv_dgrad_x3_bn')
            bn_next['bwd_partials'] = (part, nblk, dx)
            return dx
        if addend_bits is not None:
            hipabi.check(L.straps_conv_dgrad_x3_bits(hipabi.ptr(g3), gps, hipabi.ptr(w3), wps, hipabi.ptr(addend), hipabi.ptr(dx), B, H, W, Cin, Cout,
                                                     k, k, stride, pad, 0, hipabi.ptr(addend_bits), hipabi.stream_ptr()), 'straps_conv_dgrad_x3_bits')
            return dx
        hipabi.check(L.straps_conv_dgrad_x3(hipabi.ptr(g3), gps, hipabi.ptr(w3), wps, hipabi.ptr(addend), hipabi.ptr(dx), B, H, W, Cin, Cout,
                                            k, k, stride, pad, 0, hipabi.stream_ptr()), 'straps_conv_dgrad_x3')
        return dx
    if addend_bits is not None:
        raise RuntimeError('ReLU bits exist only on the bf16x3 route')
    hipabi.check(L.straps_conv_dgrad(hipabi.ptr(draw), hipabi.ptr(_packed_dgrad_weight(net, conv)), hipabi.ptr(addend), hipabi.ptr(dx), B, H, W,
                                     Cin, Cout, k, k, stride, pad, 0, hipabi.stream_ptr()), 'straps_conv_dgrad')
    return dx


def encoder_backward(net, tape, dfeat, views=None, side_stream=None, after_layer3=None):
    """tape: dict filled by encoder_forward(net, x, tape) in training mode.  Returns {param: grad}.
    side_stream: optional second torch stream for the weight-gradient kernels (joined before returning).
    after_layer3: optional callback run once the gradients of layer4 and layer3 (78 % / 94 % of resnet18 / 50's parameters)
    are final on the current stream -- the training step starts their all-reduce there (and switches hipGraphs)."""
    L = hipabi.lib()
    grads = GradSink(views)
    side = _SideStream(side_stream)
    sink = {} if getattr(net, 'conv_precision', 'fp32') == 'bf16x3' else None      # planes of the gradients the data-gradient kernels read

    def sink_of(rec):   # planes of this layer's output gradient: not on the fp32-operand route (its data and weight gradient read the fp32 tensor)
        return None if rec.get('fmode') else sink

    def keep(rec):      # does anything read the fp32 gradient of this layer's raw output?  (its weight gradient, unless that runs on planes)
        if sink is None or rec.get('x3') is None:
            return True
        B, H, W, Cin, Cout, k, stride, pad, Ho, Wo = rec['geom']
        return not L.straps_conv_wgrad_x3_on_planes(B, H, W, Cin, Cout, k, k, stride, pad)
    rec = tape['gap']
    B, HW, Cf = rec['geom']
    dy = _empty_like(rec['x'])
    hipabi.check(L.straps_gap_bwd(hipabi.ptr(dfeat.contiguous()), hipabi.ptr(dy), B, HW, Cf, hipabi.stream_ptr()), 'straps_gap_bwd')
    units = [u for li in range(1, 5) for u in getattr(net, 'layer%d' % li)]
    for li in range(4, 0, -1):
        for unit in reversed(list(getattr(net, 'layer%d' % li))):
            pairs = unit.conv_bn_pairs()
            ui = units.index(unit)
            prev_last = tape[id(units[ui - 1].conv_bn_pairs()[-1][0])] if ui > 0 else None      # the BatchNorm this unit's input came out of
            rec = tape[id(pairs[-1][0])]
            ubits = rec.get('bits') if sink is not None else None      # the unit's ReLU decisions as bits (encoder_exec._RELU_BITS)
            dskip_bits = None
            if ubits is not None:
                # no masked copy dz of the incoming gradient: whoever needs relu'(out) * dy reads dy and the bits
                draw, _ = _bn_bwd(L, rec, dy, True, False, grads, sink_of(rec), keep(rec))
                dz = dy
            else:
                draw, dz = _bn_bwd(L, rec, dy, True, True, grads, sink_of(rec), keep(rec))    # ReLU(out) mask; dz feeds the skip connection
            side.run(lambda rec=rec, draw=draw: _conv_wgrad(L, rec, draw, grads, sink), draw)
            if unit.downsample is not None:
                recd = tape[id(unit.downsample[0])]
                drawd, _ = _bn_bwd(L, recd, dz, False, False, grads, sink_of(recd), keep(recd), mask_bits=ubits)
                side.run(lambda recd=recd, drawd=drawd: _conv_wgrad(L, recd, drawd, grads, sink), drawd)
                dskip = _conv_dgrad(L, net, recd, drawd, None, sink)
            else:
                dskip, dskip_bits = dz, ubits
            for ci in range(len(pairs) - 1, 0, -1):
                rec_prev = tape[id(pairs[ci - 1][0])]
                dt = _conv_dgrad(L, net, tape[id(pairs[ci][0])], draw, None, sink, bn_next=rec_prev if sink is not None else None)
                rec = rec_prev
                draw, _ = _bn_bwd(L, rec, dt, True, False, grads, sink_of(rec), keep(rec))
                side.run(lambda rec=rec, draw=draw: _conv_wgrad(L, rec, draw, grads, sink), draw)
            # (+ skip gradient fused in the epilogue; the result is dy of the PREVIOUS unit's last BatchNorm, whose sums ride along)
            dy = _conv_dgrad(L, net, tape[id(pairs[0][0])], draw, dskip, sink, bn_next=prev_last if sink is not None else None,
                             addend_bits=dskip_bits)
        if li == 3 and after_layer3 is not None:
            side.join()
            after_layer3()
    rec = tape['maxpool']
    B, H, W, Cc, Hp, Wp = rec['geom']
    if rec['kind'] == 'maxpool_fused':
        # max-pool backward + BatchNorm/ReLU backward in one: the un-pooled gradient is gathered, never written
        idx = rec['idx']
        rec = tape['stem']
        bn, raw, ss = rec['bn'], rec['raw'], rec['stats']
        ws = torch.empty(L.straps_bn_bwd_workspace_bytes(B * H * W, Cc) // 4, device=raw.device, dtype=torch.float32)
        dgamma, dbeta = grads.buf(bn.weight), grads.buf(bn.bias)
        draw = _empty_like(raw)
        # the stem weight gradient -- draw's only reader -- skips the tiles with no non-zero input under them: they stay unwritten
        tact = None
        if rec.get('nzmask') is not None and _SPARSE_STEM_TAIL:
            _, Cin0, Hin, Win = rec['geom'][:4]
            tact = torch.empty(L.straps_stem_tiles(B, Hin, Win), device=raw.device, dtype=torch.uint8)
            hipabi.check(L.straps_stem_tile_activity(hipabi.ptr(rec['nzmask']), hipabi.ptr(tact), B, Cin0, Hin, Win, hipabi.stream_ptr()),
                         'straps_stem_tile_activity')
        hipabi.check(L.straps_bn_bwd_pooled_sparse(hipabi.ptr(dy), hipabi.ptr(idx), hipabi.ptr(raw), hipabi.ptr(ss[2]), hipabi.ptr(ss[3]),
                                                   hipabi.ptr(bn.weight), hipabi.ptr(ss[0]), hipabi.ptr(ss[1]), hipabi.ptr(dgamma), hipabi.ptr(dbeta),
                                                   hipabi.ptr(draw), hipabi.ptr(ws), B, H, W, Cc, 2 if rec.get('frozen') else 0, hipabi.ptr(tact),
                                                   hipabi.stream_ptr()), 'straps_bn_bwd_pooled_sparse')
        grads[bn.weight] = dgamma
        grads[bn.bias] = dbeta
    else:
        dstem = _empty_like(rec['x'])
        hipabi.check(L.straps_maxpool_bwd(hipabi.ptr(dy), hipabi.ptr(rec['idx']), hipabi.ptr(dstem), B, H, W, Cc, hipabi.stream_ptr()),
                     'straps_maxpool_bwd')
        rec = tape['stem']
        draw, _ = _bn_bwd(L, rec, dstem, True, False, grads)
    B, Cin, H, W, Ho, Wo = rec['geom']
    ws = torch.empty(L.straps_stem_wgrad_workspace_bytes(B, Cin, H, W) // 4, device=draw.device, dtype=torch.float32)
    dw = grads.buf(net.conv1.weight)
    hipabi.check(L.straps_stem_wgrad(hipabi.ptr(rec['x']), hipabi.ptr(draw), hipabi.ptr(dw), hipabi.ptr(ws), hipabi.ptr(rec.get('nzmask')),
                                     B, Cin, H, W, 0, hipabi.stream_ptr()), 'straps_stem_wgrad')
    grads[net.conv1.weight] = dw
    side.join()
    return grads


# ------------------------------------------------------------------------------------------ IEF
def ief_backward(ief, feat, tape, dest, views=None):
    """dest: gradient w.r.t. the final estimate [B,160].  Returns (dfeat, {param: grad}).
    Launches: one copy of dest into its slot, three small GEMMs per iteration (the dependent chain: ReLU masks, the `est_out = est_in + ...`
    addend and the running sum dc1 fused into their epilogues), then ONE launch with every weight / bias gradient and the feature
    gradient -- a weight gradient is a single GEMM over the three iterations' stacked rows (K = 3 B), a bias gradient the same against
    a constant one (straps_gemm_multi)."""
    sink = GradSink(views)
    L, st = hipabi.lib(), hipabi.stream_ptr()
    pk = ief._packed(feat.device)
    B, F = feat.shape
    H1, H2, P = ief.fc1.out_features, ief.fc2.out_features, ief.num_output_params
    dev = feat.device
    T = len(tape)
    e = lambda *s: torch.empty(*s, device=dev, dtype=torch.float32)
    D = hipabi.gemm_desc
    ests, h1s, h2s = tape[0]['stacks']
    assert ests.shape == (T + 1, B, EST_LD) and h1s.shape == (T, B, H1) and h2s.shape == (T, B, H2)
    dW3, db3 = sink.buf(ief.fc3.weight), sink.buf(ief.fc3.bias)
    dW2, db2 = sink.buf(ief.fc2.weight), sink.buf(ief.fc2.bias)
    dW1, db1 = sink.buf(ief.fc1.weight), sink.buf(ief.fc1.bias)
    # dests[it + 1] = gradient w.r.t. the estimate iteration it wrote (slot T = the incoming gradient), dests[0] = w.r.t. the initial one
    dests = e(T + 1, B, EST_LD)
    dsrc = dest if dest.stride(-1) == 1 and dest.dim() == 2 else dest.contiguous()
    hipabi.check(L.straps_masked_copy(hipabi.ptr(dsrc), dsrc.stride(0), None, 0, hipabi.ptr(dests[T]), EST_LD, B, EST_LD, 0, st), 'ief dest slot')
    dh2s, dh1s, dc1 = e(T, B, H2), e(T, B, H1), e(B, H1)
    w3, w2 = ief.fc3.weight, ief.fc2.weight
    for it in reversed(range(T)):
        # est_out = est_in + relu(h2pre) @ W3^T + b3;  h2 = relu(h1 @ W2^T + b2);  h1 = relu(c1 + est_in @ W1e^T)
        hipabi.gemm_multi([D(dests[it + 1], EST_LD, 1, w3, H2, 1, dh2s[it], H2, B, H2, P, mask=h2s[it], ldmask=H2)])            # d pre-activation of fc2
        hipabi.gemm_multi([D(dh2s[it], H2, 1, w2, H1, 1, dh1s[it], H1, B, H1, H2, mask=h1s[it], ldmask=H1,
                             c2=dc1, ldc2=H1, accumulate2=int(it != T - 1))])                                                   # d pre-activation of fc1 (+ its sum over iterations)
        hipabi.gemm_multi([D(dh1s[it], H1, 1, pk['w1e'], EST_LD, 1, dests[it], EST_LD, B, P, H1, addend=dests[it + 1], ldadd=EST_LD)])
    one = pk['one']
    dfeat = e(B, F)
    KB = T * B
    hipabi.gemm_multi([
        D(dh2s, 1, H2, h1s, H1, 1, dW2, H1, H2, H1, KB),                                                  # dW2 = sum_it dh2^T h1
        D(dc1, 1, H1, feat, F, 1, dW1, F + P, H1, F, B),                                                   # dW1[:, :F] = dc1^T feat
        D(dh1s, 1, H1, ests, EST_LD, 1, dW1.data_ptr() + 4 * F, F + P, H1, P, KB),                          # dW1[:, F:] = sum_it dh1^T est_in (slots 0..T-1)
        D(dests[1], 1, EST_LD, h2s, H2, 1, dW3, H2, P, H2, KB),                                            # dW3 = sum_it dest_out^T h2 (slots 1..T)
        D(dc1, H1, 1, pk['w1f'], F, 1, dfeat, F, B, F, H1),                                                # dfeat = dc1 @ W1f
        D(one, 0, 0, dh2s, H2, 1, db2, H2, 1, H2, KB),
        D(one, 0, 0, dests[1], EST_LD, 1, db3, P, 1, P, KB),
        D(one, 0, 0, dc1, H1, 1, db1, H1, 1, H1, B)])
    grads = {ief.fc1.weight: dW1, ief.fc1.bias: db1, ief.fc2.weight: dW2, ief.fc2.bias: db2, ief.fc3.weight: dW3, ief.fc3.bias: db3}
    return dfeat, grads


class _RegressorFn(torch.autograd.Function):
    """input [B,C,H,W] -> estimate buffer [B,160]; parameters are explicit inputs so autograd
    accumulates their .grad like for any other module."""

    @staticmethod
    def forward(ctx, reg, x, *params):
        enc_tape, ief_tape = {}, []
        with torch.no_grad():
            feat = encoder_forward(reg.image_encoder, x, enc_tape)
            est = reg.ief_module.forward_estimate(feat, ief_tape)
        ctx.reg, ctx.feat, ctx.enc_tape, ctx.ief_tape, ctx.params = reg, feat, enc_tape, ief_tape, params
        return est

    @staticmethod
    def backward(ctx, dest):
        reg = ctx.reg
        dfeat, g_ief = ief_backward(reg.ief_module, ctx.feat, ctx.ief_tape, dest)
        g_enc = encoder_backward(reg.image_encoder, ctx.enc_tape, dfeat)
        g_enc.update(g_ief)
        out = [g_enc.get(p) if p.requires_grad else None for p in ctx.params]
        ctx.enc_tape = ctx.ief_tape = None
        return (None, None) + tuple(out)


def regressor_autograd(reg, x):
    hipabi.require_gpu_tensor(x, 'regressor input', torch.float32)
    # (eval mode: BatchNorm back-propagates through its running statistics as constants, like nn.BatchNorm2d -- models/resnet.py:47,147)
    params = list(reg.parameters())
    est = _RegressorFn.apply(reg, x, *params)
    P = reg.ief_module.num_output_params
    return est[:, :3], est[:, 3:3 + 24 * 6], est[:, 3 + 24 * 6:P]


class _IefFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ief, feat, *params):
        tape = []
        with torch.no_grad():
            est = ief.forward_estimate(feat, tape)
        ctx.ief, ctx.feat, ctx.tape, ctx.params = ief, feat.detach().contiguous(), tape, params
        return est

    @staticmethod
    def backward(ctx, dest):
        dfeat, g = ief_backward(ctx.ief, ctx.feat, ctx.tape, dest)
        return (None, dfeat) + tuple(g.get(p) if p.requires_grad else None for p in ctx.params)


def ief_autograd(ief, feat):
    return _IefFn.apply(ief, feat, *list(ief.parameters()))


# ------------------------------------------------------------------------------------------ rot6d / SMPL
class _Rot6dFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        from .rigid_transform_utils import _rot6d_fwd
        ctx.save_for_backward(x)
        return _rot6d_fwd(x.detach())

    @staticmethod
    def backward(ctx, dR):
        (x,) = ctx.saved_tensors
        xd = x.detach()
        if xd.dim() == 2 and xd.stride(1) == 1 and xd.shape[1] % 6 == 0 and xd.stride(0) >= xd.shape[1]:
            rows, per_row, ld = xd.shape[0], xd.shape[1] // 6, xd.stride(0)
            dx = torch.empty(rows, xd.shape[1], device=xd.device, dtype=torch.float32)
        else:
            xd = xd.contiguous().view(-1, 6)
            rows, per_row, ld = xd.shape[0], 1, 6
            dx = torch.empty(rows, 6, device=xd.device, dtype=torch.float32)
        hipabi.check(hipabi.lib().straps_rot6d_bwd(hipabi.ptr(xd), ld, per_row, hipabi.ptr(dR.contiguous()), hipabi.ptr(dx), dx.stride(0), rows,
                                                   hipabi.stream_ptr()), 'straps_rot6d_bwd')
        return dx.view(x.shape)


def rot6d_autograd(x):
    return _Rot6dFn.apply(x)


class _SmplFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, smpl, betas, rotmats):
        b, r = betas.detach().float().contiguous(), rotmats.detach().float().contiguous()
        verts, joints = smpl.forward_arrays(b, r)
        ctx.smpl, ctx.b, ctx.r = smpl, b, r
        return verts, joints

    @staticmethod
    def backward(ctx, dverts, djoints):
        smpl, b, r = ctx.smpl, ctx.b, ctx.r
        L = hipabi.lib()
        B = b.shape[0]
        dbetas = torch.empty_like(b)
        drot = torch.empty_like(r)
        ws = torch.empty(L.straps_smpl_bwd_workspace_bytes(B, 0) // 4, device=b.device, dtype=torch.float32)
        dv = dverts.contiguous() if dverts is not None else None
        dj = djoints.contiguous() if djoints is not None else None
        hipabi.check(L.straps_smpl_bwd(C.byref(smpl._model_struct()), hipabi.ptr(b), hipabi.ptr(r), hipabi.ptr(dv), hipabi.ptr(dj),
                                       hipabi.ptr(dbetas), hipabi.ptr(drot), hipabi.ptr(ws), B, 0, hipabi.stream_ptr()), 'straps_smpl_bwd')
        return None, dbetas, drot


def smpl_forward_autograd(smpl, betas, rotmats):
    return _SmplFn.apply(smpl, betas, rotmats)
