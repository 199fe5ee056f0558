"""ResNet-18/50 proxy-representation encoder -- drop-in for reference models/resnet.py (classes
BasicBlock / Bottleneck / ResNet(block, layers, in_channels, ...) without the FC head, factories resnet18/resnet50).

The nn.Conv2d / nn.BatchNorm2d objects below are PARAMETER CONTAINERS only: they give the module
the reference's state-dict keys (`conv1.weight`, `layer2.0.downsample.1.running_var`, ...), its
parameter iteration order and its seeded initialisation, but their own forward is never called.
`ResNet.forward` runs the HIP pipeline (stem kernel from NCHW, implicit-GEMM convs over NHWC with
BatchNorm/ReLU/residual fused in the epilogue, max-pool, global average pool) through the C ABI.
"""
import torch
import torch.nn as nn

from . import hipabi

BN_EPS = 1e-5


def _check_block_args(name, groups, base_width, dilation, norm_layer):
    """the argument checks of models/resnet.py:44-52 (BasicBlock raises for groups / base_width / dilation itself); this build implements the
    plain variants the regressor uses -- grouped / wide / dilated bottlenecks and other normalisation layers have no kernel."""
    if norm_layer is not None and norm_layer is not nn.BatchNorm2d:
        raise NotImplementedError('%s: only nn.BatchNorm2d is implemented by the HIP encoder (got norm_layer=%r)' % (name, norm_layer))
    if groups != 1 or base_width != 64:
        if name == 'BasicBlock':
            raise ValueError('BasicBlock only supports groups=1 and base_width=64')        # models/resnet.py:49-50
        raise NotImplementedError('Bottleneck: groups / width_per_group other than 1 / 64 are not implemented by the HIP encoder')
    if dilation > 1:
        raise NotImplementedError('Dilation > 1 not supported in %s' % name)              # models/resnet.py:51-52 (BasicBlock); no dilated kernel here


class ResidualUnit(nn.Module):
    """One residual block.  kind 'basic': 3x3(s) - 3x3 (models/resnet.py:39-77); kind 'bottleneck':
    1x1 - 3x3(s) - 1x1 with 4x expansion (:80-121).  A 1x1(s)+BN projection on the skip path when the
    shape changes (:183-187) -- built here (`project`) or handed in by the caller (`downsample`, like the reference's `_make_layer`)."""

    def __init__(self, kind, inplanes, planes, stride, project, downsample=None):
        super().__init__()
        self.kind, self.stride = kind, stride
        out_planes = planes * (4 if kind == 'bottleneck' else 1)
        # the reference builds the projection BEFORE the block's own convs (models/resnet.py:183-190);
        # constructing in that order keeps seeded construction bit-identical.
        proj = downsample
        if proj is not None:
            ok = (isinstance(proj, nn.Sequential) and len(proj) == 2 and isinstance(proj[0], nn.Conv2d) and isinstance(proj[1], nn.BatchNorm2d)
                  and proj[0].kernel_size == (1, 1) and proj[0].bias is None and proj[0].in_channels == inplanes
                  and proj[0].out_channels == out_planes and proj[0].stride == (stride, stride))
            if not ok:
                raise NotImplementedError('downsample must be nn.Sequential(conv1x1(inplanes, %d, stride=%d, bias=False), nn.BatchNorm2d(%d)) -- the '
                                          'projection models/resnet.py:183-187 builds; other skip paths have no kernel' % (out_planes, stride, out_planes))
        elif project:
            proj = nn.Sequential(nn.Conv2d(inplanes, out_planes, 1, stride, bias=False), nn.BatchNorm2d(out_planes))
        if kind == 'basic':
            self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
            self.bn1 = nn.BatchNorm2d(planes)
            self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
            self.bn2 = nn.BatchNorm2d(planes)
        else:
            self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
            self.bn1 = nn.BatchNorm2d(planes)
            self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)
            self.bn2 = nn.BatchNorm2d(planes)
            self.conv3 = nn.Conv2d(planes, out_planes, 1, bias=False)
            self.bn3 = nn.BatchNorm2d(out_planes)
        self.downsample = proj            # registered last, like the reference's attribute order
        self.out_planes = out_planes

    def conv_bn_pairs(self):
        names = ['1', '2'] + (['3'] if self.kind == 'bottleneck' else [])
        return [(getattr(self, 'conv' + n), getattr(self, 'bn' + n)) for n in names]

    def forward(self, x):
        raise RuntimeError('residual blocks are parameter containers: the encoder runs as a whole through ResNet.forward (HIP kernels, no per-module '
                           'forward and no CPU fallback)')


class BasicBlock(ResidualUnit):
    """models/resnet.py:39-77, same constructor."""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, groups=1, base_width=64, dilation=1, norm_layer=None):
        _check_block_args('BasicBlock', groups, base_width, dilation, norm_layer)
        super().__init__('basic', inplanes, planes, stride, False, downsample)


class Bottleneck(ResidualUnit):
    """models/resnet.py:80-121, same constructor."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, groups=1, base_width=64, dilation=1, norm_layer=None):
        _check_block_args('Bottleneck', groups, base_width, dilation, norm_layer)
        super().__init__('bottleneck', inplanes, planes, stride, False, downsample)


class ResNet(nn.Module):
    def __init__(self, block, layers, in_channels, num_classes=1000, zero_init_residual=False, groups=1, width_per_group=64,
                 replace_stride_with_dilation=None, norm_layer=None, conv_precision='bf16x3'):
        """The reference's signature (models/resnet.py:124-129): block = BasicBlock | Bottleneck (the strings 'basic' / 'bottleneck' of earlier
        rounds are still accepted), layers = blocks per stage.  num_classes is accepted and unused like there (the FC head is commented out,
        :158); groups / width_per_group / replace_stride_with_dilation / norm_layer other than their defaults raise: no kernel implements them.
        conv_precision (extension, also an attribute that may be set at any time; NOT part of the state dict): arithmetic of the
        3x3 / 1x1 convolutions -- 'bf16x3' (default): fp32 operands as exact bf16 triples on the bf16 matrix pipe, six products per
        term, fp32 accumulate (the fp32 chain's accuracy class: tests/test_gpu_conv_x3.py), 'fp32': the exact fp32-input MFMA chain."""
        super().__init__()
        if conv_precision not in ('fp32', 'bf16x3'):
            raise ValueError("conv_precision must be 'fp32' or 'bf16x3'")
        if isinstance(block, str):
            block = {'basic': BasicBlock, 'bottleneck': Bottleneck}[block]
        if block not in (BasicBlock, Bottleneck):
            raise NotImplementedError('ResNet: block must be BasicBlock or Bottleneck (got %r)' % (block,))
        if replace_stride_with_dilation is None:
            replace_stride_with_dilation = [False, False, False]
        if len(replace_stride_with_dilation) != 3:
            raise ValueError("replace_stride_with_dilation should be None or a 3-element tuple, got {}".format(replace_stride_with_dilation))   # :139-141
        if any(replace_stride_with_dilation):
            raise NotImplementedError('replace_stride_with_dilation: dilated stages are not implemented by the HIP encoder')
        _check_block_args(block.__name__, groups, width_per_group, 1, norm_layer)
        if len(layers) != 4:
            raise ValueError('ResNet: layers must list the blocks of the four stages')
        self.kind, self.in_channels = ('basic' if block is BasicBlock else 'bottleneck'), in_channels
        self.groups, self.base_width, self.dilation = groups, width_per_group, 1
        self.inplanes = 64
        self.conv1 = nn.Conv2d(in_channels, self.inplanes, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(self.inplanes)
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
        self.layer3 = self._make_layer(block, 256, layers[2], stride=2)
        self.layer4 = self._make_layer(block, 512, layers[3], stride=2)
        self.num_features = self.inplanes
        # models/resnet.py:160-165: kaiming-normal(fan_out, relu) convs, BN gamma=1 beta=0, in modules() order
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        if zero_init_residual:
            for m in self.modules():
                if isinstance(m, ResidualUnit):
                    nn.init.constant_((m.bn3 if m.kind == 'bottleneck' else m.bn2).weight, 0)
        self._cache = {}
        # route of the 3x3 / 1x1 convolutions (forward + data gradient): 'bf16x3' = every fp32 operand as three exact bf16 planes, six
        # products per term, fp32 accumulate on the bf16 matrix pipe (csrc/conv_x3.hip; same error class against float64 as the fp32
        # chain: tests/test_gpu_conv_x3.py), 'fp32' = the exact-fp32 MFMA chain (csrc/conv.hip).  The weight gradients and the stem
        # are fp32 MFMA on both routes.
        self.conv_precision = conv_precision
        self._bn_epoch = 0          # bumped by every training-mode forward (running statistics change behind torch's back)

    def _make_layer(self, block, planes, blocks, stride=1, dilate=False):
        """models/resnet.py:177-199: the projection is built first, then the first block, then the rest (same construction order, same RNG use)"""
        if dilate:
            raise NotImplementedError('dilated stages are not implemented by the HIP encoder')
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(nn.Conv2d(self.inplanes, planes * block.expansion, 1, stride, bias=False), nn.BatchNorm2d(planes * block.expansion))
        units = [block(self.inplanes, planes, stride, downsample, self.groups, self.base_width, 1, None)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            units.append(block(self.inplanes, planes, groups=self.groups, base_width=self.base_width, dilation=self.dilation, norm_layer=None))
        return nn.Sequential(*units)

    # ---- packed-weight / folded-BN caches, refreshed when a parameter's version changes ----
    def _cached(self, key, tensors, make, extra=()):
        sig = tuple((t.data_ptr(), t._version) for t in tensors) + tuple(extra)
        hit = self._cache.get(key)
        if hit is None or hit[0] != sig:
            hit = (sig, make())
            self._cache[key] = hit
        return hit[1]

    def _packed_weight(self, conv, stem=False):
        w = conv.weight
        L = hipabi.lib()

        def make():
            wd = w.detach().contiguous()
            if stem:
                out = torch.empty(L.straps_stem_weight_floats(wd.shape[1]), device=wd.device, dtype=torch.float32)
                hipabi.check(L.straps_pack_stem_weight(hipabi.ptr(wd), hipabi.ptr(out), wd.shape[1], hipabi.stream_ptr()),
                             'straps_pack_stem_weight')
            else:
                out = torch.empty(wd.numel(), device=wd.device, dtype=torch.float32)
                hipabi.check(L.straps_pack_conv_weight(hipabi.ptr(wd), hipabi.ptr(out), wd.shape[0], wd.shape[1], wd.shape[2],
                                                       wd.shape[3], hipabi.stream_ptr()), 'straps_pack_conv_weight')
            return out
        return self._cached(('w', id(conv)), [w], make)

    def _packed_weight_x3(self, conv, dgrad=False):
        """(planes, plane stride) of the packed forward / data-gradient weights for the bf16x3 convolution route: chunk-major planes
        (csrc/elementwise.hip wk_index) written by the batched pack.  A stale or missing entry re-packs EVERY layer in one launch
        (`prepack`), which refreshes all entries of the cache."""
        w = conv.weight
        key = ('wd3' if dgrad else 'w3', id(conv))
        sig = ((w.data_ptr(), w._version),)
        hit = self._cache.get(key)
        if hit is None or hit[0] != sig:
            self.prepack(with_dgrad=dgrad or getattr(self, '_prepack_state', {}).get('crsk3') is not None)
            hit = self._cache[key]
        return hit[1]

    def prepack(self, with_dgrad=True):
        """(re)pack the weights of every non-stem conv for the forward (KRSC) and data-gradient (flipped CRSK) kernels in
        ONE launch and seed the caches `_packed_weight` / autograd_ops._packed_dgrad_weight read.  The training step
        calls this after each optimiser update instead of 2 x (19 | 52) tiny launches.  On the bf16x3 route the same launch
        writes the three bf16 planes of both layouts and nothing else (no fp32 packed copies, no split pass)."""
        import ctypes as C
        import numpy as np
        convs = [m for m in self.modules() if isinstance(m, nn.Conv2d) and m is not self.conv1]
        key = tuple(c.weight.data_ptr() for c in convs) + (with_dgrad, getattr(self, 'conv_precision', 'fp32'))
        st = getattr(self, '_prepack_state', None)
        if st is None or st['key'] != key:
            dev = convs[0].weight.device
            hipabi.require_gpu_tensor(convs[0].weight, 'conv weights (call .to(device) first)')
            total = sum(c.weight.numel() for c in convs)
            x3 = getattr(self, 'conv_precision', 'fp32') == 'bf16x3'
            krsc = torch.empty(total, device=dev, dtype=torch.float32) if not x3 else None
            crsk = torch.empty(total, device=dev, dtype=torch.float32) if with_dgrad and not x3 else None
            descs = (hipabi.PackDesc * len(convs))()
            off = 0
            for d, c in zip(descs, convs):
                w = c.weight
                if not w.is_contiguous():
                    raise RuntimeError('prepack: conv weights must be contiguous')
                d.src = w.data_ptr()
                d.dst_krsc = krsc.data_ptr() + 4 * off if krsc is not None else None
                d.dst_crsk = crsk.data_ptr() + 4 * off if crsk is not None else None
                d.o, d.c, d.r, d.s, d.first = w.shape[0], w.shape[1], w.shape[2], w.shape[3], off
                off += w.numel()
            table = torch.from_numpy(np.frombuffer(bytes(descs), dtype=np.uint8).copy()).to(dev)
            st = dict(key=key, krsc=krsc, crsk=crsk, table=table, total=total, convs=convs)
            if x3:
                ps = (total + 7) // 8 * 8
                st['ps'] = ps
                st['krsc3'] = torch.empty(3, ps, device=dev, dtype=torch.int16)
                st['crsk3'] = torch.empty(3, ps, device=dev, dtype=torch.int16) if with_dgrad else None
            self._prepack_state = st
        x3 = st.get('krsc3') is not None
        if x3:      # every conv's planes are slices of the two plane buffers, with their common plane stride
            hipabi.check(hipabi.lib().straps_pack_conv_weights_batched_x3(hipabi.ptr(st['table']), len(st['convs']), st['total'], hipabi.ptr(st['krsc3']),
                                                                          hipabi.ptr(st['crsk3']), st['ps'], hipabi.stream_ptr()),
                         'straps_pack_conv_weights_batched_x3')
        else:
            hipabi.check(hipabi.lib().straps_pack_conv_weights_batched(hipabi.ptr(st['table']), len(st['convs']), st['total'],
                                                                       hipabi.stream_ptr()), 'straps_pack_conv_weights_batched')
        off = 0
        for c in st['convs']:
            w, n = c.weight, c.weight.numel()
            sig = ((w.data_ptr(), w._version),)
            if not x3:
                self._cache[('w', id(c))] = (sig, st['krsc'][off:off + n])
                if with_dgrad:
                    self._cache[('wd', id(c))] = (sig, st['crsk'][off:off + n])
            else:
                self._cache[('w3', id(c))] = (sig, (st['krsc3'][0, off:], st['ps']))
                if with_dgrad:
                    self._cache[('wd3', id(c))] = (sig, (st['crsk3'][0, off:], st['ps']))
            off += n

    def _folded_bn(self, bn):
        def make():
            C = bn.weight.shape[0]
            ss = torch.empty(2, C, device=bn.weight.device, dtype=torch.float32)
            hipabi.check(hipabi.lib().straps_bn_fold(hipabi.ptr(bn.weight), hipabi.ptr(bn.bias), hipabi.ptr(bn.running_mean),
                                                     hipabi.ptr(bn.running_var), bn.eps, hipabi.ptr(ss[0]), hipabi.ptr(ss[1]), C,
                                                     hipabi.stream_ptr()), 'straps_bn_fold')
            return ss
        # running_mean / running_var are updated by straps_bn_stats_finalize through raw pointers (no _version bump): the
        # training-forward counter is part of the signature, so eval() after train-mode forwards never sees stale folds
        return self._cached(('bn', id(bn)), [bn.weight, bn.bias, bn.running_mean, bn.running_var], make, extra=(self._bn_epoch,))

    def _frozen_bn(self, bn):
        """eval-mode BatchNorm as (scale, shift, mean, invstd) [4][C] -- what the training-mode kernels take -- for an eval-mode forward
        that records a tape (gradients through frozen statistics)."""
        def make():
            C = bn.weight.shape[0]
            ss = torch.empty(4, C, device=bn.weight.device, dtype=torch.float32)
            hipabi.check(hipabi.lib().straps_bn_fold_stats(hipabi.ptr(bn.weight), hipabi.ptr(bn.bias), hipabi.ptr(bn.running_mean), hipabi.ptr(bn.running_var),
                                                           bn.eps, hipabi.ptr(ss[0]), hipabi.ptr(ss[1]), hipabi.ptr(ss[2]), hipabi.ptr(ss[3]), C,
                                                           hipabi.stream_ptr()), 'straps_bn_fold_stats')
            return ss
        return self._cached(('bnf', id(bn)), [bn.weight, bn.bias, bn.running_mean, bn.running_var], make, extra=(self._bn_epoch,))

    @hipabi.on_tensor_device
    def forward(self, x):
        from .encoder_exec import encoder_forward
        return encoder_forward(self, x)


def resnet18(in_channels, pretrained=False, progress=True, **kwargs):
    """models/resnet.py:228-236.  `pretrained` weights cannot be fetched (no torchvision head, no
    network); the reference always passes False on this path (models/regressor.py:30)."""
    if pretrained:
        raise NotImplementedError('pretrained ImageNet weights are not available in this build')
    return ResNet(BasicBlock, [2, 2, 2, 2], in_channels, **kwargs)


def resnet50(in_channels, pretrained=False, progress=True, **kwargs):
    """models/resnet.py:250-258."""
    if pretrained:
        raise NotImplementedError('pretrained ImageNet weights are not available in this build')
    return ResNet(Bottleneck, [3, 4, 6, 3], in_channels, **kwargs)
