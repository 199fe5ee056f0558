"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/straps_hip.h
declares (no compute calls -- there is no GPU here)."""
import ctypes as C
import os
import re

import pytest

import straps_amd
from straps_amd import hipabi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, 'include', 'straps_hip.h')).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    return sorted(set(re.findall(r'\b(straps_[a-z0-9_]+)\s*\(', txt)))


@pytest.fixture(scope='module')
def lib():
    hipabi.build()
    return hipabi.load()


def test_header_symbols_exported_and_bound(lib):
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), 'library does not export %s' % n
        assert n in hipabi.SIGNATURES, 'ctypes binding lacks a prototype for %s' % n
    for n in hipabi.SIGNATURES:
        assert n in names, '%s bound in hipabi.py but not declared in straps_hip.h' % n


def test_version_and_error_channel(lib):
    assert isinstance(lib.straps_last_error(), bytes)
    # argument validation happens before any HIP call, so it is checkable without a GPU
    rc = lib.straps_rot6d_fwd(None, 6, 1, None, 1, None)
    assert rc == 1 and b'null pointer' in lib.straps_last_error()
    rc = lib.straps_linear_fwd(None, 0, None, 0, None, None, None, 0, 0, 0, 0, 0, None)
    assert rc == 1
    assert lib.straps_stem_weight_floats(18) == ((18 * 49 + 7) // 8) * 512
    ms = hipabi.SmplModelStruct()
    ms.n_tiles = 224
    assert lib.straps_smpl_workspace_bytes(C.byref(ms), 64) == 64 * (224 + 288 + 8 * 96) * 4
    header = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'include', 'straps_hip.h')).read()
    declared = int(re.search(r'#define STRAPS_ABI_VERSION (\d+)', header).group(1))
    assert lib.straps_abi_version() == declared == hipabi.ABI_VERSION


def test_tile_choice_of_the_implicit_gemm(lib):
    """straps_conv_stat_blocks = ceil(M / BM) exposes the tile rule (host arithmetic only): 128x128 for layer2, 128x64 for
    layer3 and layer1's 64-channel 3x3 layers, 64x64 for layer4's 4096 pixels and short-K 1x1 layers (resnet18, B=64), and the
    explicit tile_cfg overrides."""
    B = 64
    assert lib.straps_conv_stat_blocks(B, 64, 64, 64, 576, 0) == B * 64 * 64 // 128         # layer1: 128x64 tiles
    assert lib.straps_conv_stat_blocks(2, 64, 64, 64, 576, 0) == 2 * 64 * 64 // 64          # ... only with enough rows to fill the chip
    assert lib.straps_conv_stat_blocks(B, 32, 32, 128, 1152, 0) == B * 32 * 32 // 128       # layer2: 128x128
    assert lib.straps_conv_stat_blocks(B, 16, 16, 256, 2304, 0) == B * 16 * 16 // 128       # layer3: 128x64
    assert lib.straps_conv_stat_blocks(B, 8, 8, 512, 4608, 0) == B * 8 * 8 // 64            # layer4: 64x64
    assert lib.straps_conv_stat_blocks(B, 16, 16, 256, 128, 0) == B * 16 * 16 // 64         # 1x1 down-sample: short K
    for cfg, bm in ((1, 128), (2, 128), (3, 64), (3 | 16, 64)):
        assert lib.straps_conv_stat_blocks(2, 10, 10, 128, 1152, cfg) == -(-200 // bm)
    # BatchNorm-backward reduction grid and the workspace that goes with it
    assert lib.straps_bn_bwd_workspace_bytes(4096, 64) == (lib.straps_bn_bwd_blocks(4096, 64) * 64 * 2 + 2 * 64) * 8 + 64 * 4


def test_tile_choice_and_routes_of_the_bf16x3_kernels(lib):
    """host arithmetic only: straps_conv_x3_stat_blocks exposes which kernel / tile straps_conv_fwd_x3 uses for a geometry (resnet18 at
    B = 64: layer1 128x64 two-stage tiles, layer2 256x128, layer3 the halo-patch kernel (128 rows), layer4 128x64 three-stage; the
    halo-patch kernel only for 3x3 / stride 1 maps whose 128-pixel tiles are whole rows or whole images), and
    straps_conv_wgrad_x3_on_planes which weight gradients read planes only."""
    B = 64
    sb = lib.straps_conv_x3_stat_blocks
    assert sb(B, 64, 64, 64, 64, 3, 3, 1, 1, 0) == B * 64 * 64 // 128
    assert sb(B, 32, 32, 128, 128, 3, 3, 1, 1, 0) == B * 32 * 32 // 256
    assert sb(B, 16, 16, 256, 256, 3, 3, 1, 1, 0) == B * 16 * 16 // 128
    assert sb(B, 8, 8, 512, 512, 3, 3, 1, 1, 0) == B * 8 * 8 // 128
    assert sb(B, 64, 64, 64, 128, 3, 3, 2, 1, 0) == B * 32 * 32 // 256                      # stride 2: im2col kernel, 256x128 tiles
    assert sb(3, 10, 24, 128, 128, 3, 3, 1, 1, 0) == -(-3 * 10 * 24 // 64)                  # a grid that cannot fill the chip with 128-row tiles: 64x64 (round 4)
    assert sb(3, 10, 24, 128, 128, 3, 3, 1, 1, 512) == -(-3 * 10 * 24 // 64)                # (no halo kernel: M % 128 != 0)
    assert sb(32, 8, 8, 512, 512, 3, 3, 1, 1, 0) == 32 * 8 * 8 // 64                        # resnet50's layer4 at 32 bodies: 64x64 tiles, one workgroup per CU
    assert sb(16, 32, 32, 256, 128, 3, 3, 1, 1, 0) == 16 * 32 * 32 // 128                   # 128 tile equivalents: still 128x64
    for cfg, bm in ((1, 128), (2, 128), (3, 64), (4, 256), (5, 128), (7, 128), (11, 128), (12, 256)):
        assert sb(2, 10, 10, 128, 128, 3, 3, 1, 1, cfg) == -(-200 // bm), cfg
    on = lib.straps_conv_wgrad_x3_on_planes
    assert on(B, 64, 64, 64, 64, 3, 3, 1, 1) == 1 and on(B, 8, 8, 512, 512, 3, 3, 1, 1) == 1   # halo-patch kernel
    assert on(B, 64, 64, 64, 128, 3, 3, 2, 1) == 1                                              # 3x3 / stride 2: per-tap kernel on planes
    assert on(B, 64, 64, 64, 128, 1, 1, 2, 0) == 0                                              # 1x1 with a 64-channel side: fp32 kernel
    assert on(B, 28, 28, 128, 512, 1, 1, 1, 0) == 1
    assert on(2, 10, 24, 64, 64, 3, 3, 1, 1) == 1                                               # no halo plan (width 24): per-tap kernel on planes


def test_product_library_reads_no_measurement_switches(lib):
    """ablation instantiations (wrong results by design) and STRAPS_* A/B environment switches live only in the
    -DSTRAPS_TOOLS build (tools/bin/libstraps_hip_tools.so, hipabi.build(tools=True)).  The product library carries no STRAPS_* string at
    of a switch -- it cannot even ask for one -- and no source calls getenv."""
    blob = open(hipabi.LIB_PATH, 'rb').read()
    # (the only STRAPS_* strings a product build may hold are mode / constant names inside error messages)
    found = sorted(set(re.findall(rb'STRAPS_[A-Z0-9_]{3,}', blob)))
    assert all(f.startswith((b'STRAPS_SMPL_SPLIT_', b'STRAPS_SMPL_KERNEL_')) for f in found), found
    for name in (b'STRAPS_SMPL_ABLATE', b'STRAPS_SMPL_PF', b'STRAPS_SMPL_RPC', b'STRAPS_SMPL_WVAR', b'STRAPS_WGRAD3_ABL', b'STRAPS_WGRAD_', b'STRAPS_STEM_WGRAD_'):
        assert name not in blob, name
    for src in hipabi._existing_sources():
        txt = open(src).read()
        assert 'getenv' not in txt, '%s reads the environment' % os.path.basename(src)
