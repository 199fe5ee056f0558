"""ctypes binding of libstraps_hip.so (the C ABI declared in include/straps_hip.h).

No torch types cross this boundary: tensors are handed over as raw device pointers + sizes and
the current HIP stream handle.  There is deliberately NO fallback: if the library is not built or
no GPU is visible, every product entry point raises.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, 'csrc')
LIB_PATH = os.path.join(CSRC, 'libstraps_hip.so')
HEADER = os.path.join(os.path.dirname(_HERE), 'include', 'straps_hip.h')
SOURCES = ['abi.hip', 'augment.hip', 'pose.hip', 'ief.hip', 'elementwise.hip', 'conv.hip', 'conv_x3.hip', 'conv_x3_lean.hip', 'conv_x3f.hip', 'conv_wgrad_x3f.hip', 'stem.hip', 'smpl.hip',
           'smpl_bwd.hip', 'backward.hip', 'train.hip', 'metrics.hip', 'image.hip', 'raster.hip', 'exchange.hip']

_lib = None
LINK_LIBS = ['-ldl']
# per-source compiler flags.  smpl.hip: the SLP vectoriser packs the fold of the matrix-pipe SMPL kernels into v_pk_fma_f32 / v_pk_mul_f32, which
# cost several times a scalar v_fma_f32 beside MFMAs (MI355X guide: "packed f32 VALU: an anti-lever beside MFMAs"; measured on the 64-body
# kernel: 3.00 -> 2.87 ms per 65 536 bodies, profiles/r04_smpl_w_ab.txt)
EXTRA_FLAGS = {'smpl.hip': ['-fno-slp-vectorize']}


def _existing_sources():
    return [os.path.join(CSRC, s) for s in SOURCES if os.path.isfile(os.path.join(CSRC, s))]


TOOLS_LIB_PATH = os.path.join(os.path.dirname(_HERE), 'tools', 'bin', 'libstraps_hip_tools.so')


def build(force=False, verbose=False, tools=False):
    """Compile every HIP source for gfx950 into csrc/libstraps_hip.so (in-tree, so the .so travels
    with the repo snapshot).  hipcc cross-compiles without a GPU.
    tools=True: the same sources with -DSTRAPS_TOOLS into tools/bin/libstraps_hip_tools.so -- the measurement build that carries the
    ablation instantiations (wrong results by design) and reads the STRAPS_* A/B environment switches.  The product library has
    neither; tools select the other library explicitly with `use_library(TOOLS_LIB_PATH)` (tools/with_tools_lib.py)."""
    srcs = _existing_sources()
    lib_path = TOOLS_LIB_PATH if tools else LIB_PATH
    if tools and not force and os.environ.get('STRAPS_TOOLS_NO_BUILD') == '1' and os.path.isfile(lib_path):
        # (GPU-box runs of the tools: use the library that travelled with the snapshot, whatever the copy did to the time stamps -- and say which one)
        import sys
        import time
        print('hipabi.build(tools=True): STRAPS_TOOLS_NO_BUILD=1 -- reusing %s (built %s)' % (lib_path, time.strftime('%Y-%m-%d %H:%M:%S', time.localtime(os.path.getmtime(lib_path)))),
              file=sys.stderr)
        return lib_path
    # every header under csrc/ (common.h, conv_igemm.h, ...) and the public header: a change in any of them rebuilds every object --
    # translation units that share a struct (ConvP) can never be linked from different versions of it
    import glob
    hdrs = sorted(glob.glob(os.path.join(CSRC, '*.h'))) + [HEADER]
    deps = srcs + hdrs
    if not force and os.path.isfile(lib_path) and all(os.path.getmtime(lib_path) > os.path.getmtime(d) for d in deps):
        if not tools:
            _audit(lib_path, 'hipabi.build (library up to date)')
        return lib_path
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    flags = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC'] + (['-DSTRAPS_TOOLS'] if tools else [])
    extra = dict(EXTRA_FLAGS)
    if tools and os.environ.get('STRAPS_TOOLS_SMPL_FLAGS'):      # (build-time A/B of compiler flags for one source: tools library only)
        extra['smpl.hip'] = extra.get('smpl.hip', []) + os.environ['STRAPS_TOOLS_SMPL_FLAGS'].split()
    if tools and os.environ.get('STRAPS_TOOLS_SMPL_BWD_FLAGS'):      # (-DSTRAPS_ALLOW_PACKED_FP32: the kernels WITH packed fp32 instructions, as round 5 found them: csrc/common.h)
        extra['smpl_bwd.hip'] = extra.get('smpl_bwd.hip', []) + os.environ['STRAPS_TOOLS_SMPL_BWD_FLAGS'].split()
    if tools and os.environ.get('STRAPS_TOOLS_RASTER_FLAGS'):
        extra['raster.hip'] = extra.get('raster.hip', []) + os.environ['STRAPS_TOOLS_RASTER_FLAGS'].split()
    # one object per source (csrc/build/*.o, compiled in parallel, rebuilt only when the source or a header is newer), then one link
    objdir = os.path.join(os.path.dirname(TOOLS_LIB_PATH), 'build') if tools else os.path.join(CSRC, 'build')
    os.makedirs(objdir, exist_ok=True)
    jobs = []
    for s in srcs:
        o = os.path.join(objdir, os.path.basename(s)[:-4] + '.o')
        if force or not os.path.isfile(o) or any(os.path.getmtime(o) <= os.path.getmtime(d) for d in [s] + hdrs):
            cmd = [hipcc] + flags + extra.get(os.path.basename(s), []) + ['-c', s, '-o', o]
            if verbose:
                print(' '.join(cmd))
            jobs.append((cmd, subprocess.Popen(cmd, cwd=CSRC)))
    for cmd, p in jobs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, cmd)
    objs = [os.path.join(objdir, os.path.basename(s)[:-4] + '.o') for s in srcs]
    cmd = [hipcc, '--offload-arch=gfx950', '-fPIC', '-shared', '-o', lib_path] + objs + LINK_LIBS
    if verbose:
        print(' '.join(cmd))
    subprocess.run(cmd, check=True, cwd=CSRC)
    if not tools:
        # the packed-operand-select hazard of DESIGN section 1 is a BUILD failure (round 6): the library just linked is disassembled, and one such
        # instruction in any kernel raises (the file is moved aside).  The tools library is exempt: it holds the reproducers' victims on purpose.
        _audit(lib_path, 'hipabi.build', fresh=True)
    return lib_path


def _audit(lib_path, what, fresh=False):
    """ISA audit of a product library (isa_audit.py): trusted when its stamp matches, run otherwise; raises on a finding, warns when the LLVM tools
    that it needs are not installed."""
    from . import isa_audit
    if not fresh and isa_audit.stamp_ok(lib_path):
        return
    if not isa_audit.enforce(lib_path, what):
        import warnings
        warnings.warn('%s: llvm-objdump not found under %s -- the packed-operand-select audit of %s (DESIGN section 1: wrong results in lanes 48..63 on '
                      'MI355X) did NOT run; set STRAPS_LLVM_BIN or audit the library where it was built (tools/audit_packed_fp32.py)'
                      % (what, isa_audit.LLVM_BIN, lib_path), RuntimeWarning, stacklevel=3)


class SmplModelStruct(C.Structure):
    """mirror of straps_smpl_model_t"""
    _fields_ = [('blend_frag', C.c_void_p), ('j_template', C.c_void_p), ('j_shapedirs', C.c_void_p),
                ('parents', C.c_void_p), ('depth', C.c_void_p), ('max_depth', C.c_int32), ('skin_k', C.c_int32),
                ('skin_w', C.c_void_p), ('skin_j', C.c_void_p), ('vj_ptr', C.c_void_p), ('n_tiles', C.c_int32),
                ('reserved0', C.c_int32), ('pick_ids', C.c_void_p), ('blend_frag_t', C.c_void_p), ('children', C.c_void_p),
                ('jrt_ptr', C.c_void_p), ('jrt_code', C.c_void_p), ('jrt_w', C.c_void_p), ('dj_ptr', C.c_void_p), ('dj_code', C.c_void_p),
                ('dj_w', C.c_void_p), ('blend_frag_h', C.c_void_p), ('blend_h_unscale', C.c_float), ('reserved1', C.c_int32), ('skin_frag_p', C.c_void_p)]


class PackDesc(C.Structure):
    """mirror of straps_pack_desc_t"""
    _fields_ = [('src', C.c_void_p), ('dst_krsc', C.c_void_p), ('dst_crsk', C.c_void_p), ('o', C.c_int32), ('c', C.c_int32),
                ('r', C.c_int32), ('s', C.c_int32), ('first', C.c_longlong)]


class GemmDesc(C.Structure):
    """mirror of straps_gemm_desc_t"""
    _fields_ = [('a', C.c_void_p), ('sam', C.c_longlong), ('sak', C.c_longlong), ('b', C.c_void_p), ('sbk', C.c_longlong), ('sbn', C.c_longlong),
                ('c', C.c_void_p), ('ldc', C.c_int32), ('accumulate', C.c_int32), ('addend', C.c_void_p), ('ldadd', C.c_int32), ('reserved0', C.c_int32),
                ('mask', C.c_void_p), ('ldmask', C.c_int32), ('reserved1', C.c_int32), ('c2', C.c_void_p), ('ldc2', C.c_int32), ('accumulate2', C.c_int32),
                ('m', C.c_int32), ('n', C.c_int32), ('k', C.c_int32), ('reserved2', C.c_int32)]


def gemm_desc(a, sam, sak, b, sbk, sbn, c, ldc, m, n, k, accumulate=0, addend=None, ldadd=0, mask=None, ldmask=0, c2=None, ldc2=0, accumulate2=0):
    """one problem of straps_gemm_multi; a / b / c / addend / mask / c2: tensors or raw device addresses (int)."""
    adr = lambda t: None if t is None else (t if isinstance(t, int) else t.data_ptr())
    return GemmDesc(adr(a), sam, sak, adr(b), sbk, sbn, adr(c), ldc, accumulate, adr(addend), ldadd, 0, adr(mask), ldmask, 0, adr(c2), ldc2, accumulate2,
                    m, n, k, 0)


def gemm_multi(descs):
    arr = (GemmDesc * len(descs))(*descs)
    check(lib().straps_gemm_multi(arr, len(descs), stream_ptr()), 'straps_gemm_multi')


_P, _I, _L, _F, _Z, _D = C.c_void_p, C.c_int, C.c_longlong, C.c_float, C.c_size_t, C.c_double

# name -> (restype, argtypes); must list every symbol of include/straps_hip.h (tests/test_abi.py checks)
ABI_VERSION = 9      # == STRAPS_ABI_VERSION of include/straps_hip.h (tests/test_abi.py compares the two); load() refuses a library of another version
SIGNATURES = {
    'straps_abi_version': (_I, []),
    'straps_last_error': (C.c_char_p, []),
    'straps_device_count': (_I, []),
    'straps_wall_clock_khz': (_I, []),
    'straps_set_clock_accumulator': (_I, [_P]),
    'straps_selftest_mfma_peak': (_I, [_P, _P, _I, _I, _P]),
    'straps_selftest_mfma_bf16': (_I, [_P, _P, _I, _I, _P]),
    'straps_selftest_mfma_bf16_dense': (_I, [_P, _P, _I, _I, _I, _P]),
    'straps_pack_conv_weight': (_I, [_P, _P, _I, _I, _I, _I, _P]),
    'straps_pack_conv_weight_dgrad': (_I, [_P, _P, _I, _I, _I, _I, _P]),
    'straps_pack_conv_weights_batched': (_I, [_P, _I, _L, _P]),
    'straps_pack_conv_weights_batched_x3': (_I, [_P, _I, _L, _P, _P, _L, _P]),
    'straps_stem_weight_floats': (_Z, [_I]),
    'straps_pack_stem_weight': (_I, [_P, _P, _I, _P]),
    'straps_bn_fold': (_I, [_P, _P, _P, _P, _F, _P, _P, _I, _P]),
    'straps_bn_fold_stats': (_I, [_P, _P, _P, _P, _F, _P, _P, _P, _P, _I, _P]),
    'straps_stem_stat_blocks': (_I, [_I, _I, _I]),
    'straps_stem_nzmask_words': (_Z, [_I, _I, _I, _I]),
    'straps_stem_nzmask': (_I, [_P, _P, _I, _I, _I, _I, _P]),
    'straps_stem_fwd': (_I, [_P, _P, _P, _P, _I, _P, _P, _P, _I, _I, _I, _I, _P]),
    'straps_conv_trace_buffer': (_I, [_P]),
    'straps_split3_bf16': (_I, [_P, _P, _L, _L, _P]),
    'straps_split3_bf16_cm': (_I, [_P, _P, _L, _I, _L, _P]),
    'straps_conv_fwd_x3': (_I, [_P, _L, _P, _L, _P, _P, _P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    'straps_conv_fwd_x3p': (_I, [_P, _L, _P, _L, _P, _P, _P, _I, _P, _P, _L, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    'straps_conv_dgrad_x3': (_I, [_P, _L, _P, _L, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    'straps_conv_x3_stat_blocks': (_I, [_I] * 10),
    'straps_conv_x3f_supported': (_I, [_I] * 6),
    'straps_conv_fwd_x3f': (_I, [_P, _P, _P, _I, _P, _L, _P, _P, _P, _I, _P, _P] + [_I] * 10 + [_P]),
    'straps_conv_x3f_stat_blocks': (_I, [_I] * 10),
    'straps_conv_dgrad_x3f': (_I, [_P, _P, _L, _P, _P, _P] + [_I] * 10 + [_P] * 8),
    'straps_conv_dgrad_x3f_bn_blocks': (_I, [_I] * 10),
    'straps_conv_wgrad_x3f_workspace_bytes': (_Z, [_I] * 9),
    'straps_conv_wgrad_x3f': (_I, [_P, _P, _P, _I, _P, _P, _P] + [_I] * 10 + [_P]),
    'straps_bn_apply_x3': (_I, [_P, _P, _P, _P, _I, _P, _P, _L, _L, _I, _P]),
    'straps_bn_relu_maxpool_fwd_x3': (_I, [_P, _P, _P, _P, _P, _P, _L, _I, _I, _I, _I, _P]),
    'straps_conv_wgrad_x3_on_planes': (_I, [_I] * 9),
    'straps_conv_dgrad_x3_bn_blocks': (_I, [_I] * 10),
    'straps_conv_dgrad_x3_bn': (_I, [_P, _L, _P, _L, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P]),
    'straps_bn_bwd_finish_x3': (_I, [_P] * 13 + [_L, _P, _I, _P, _L, _I, _I, _P]),
    'straps_conv_wgrad_x3': (_I, [_P, _P, _P, _L, _P, _L, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    'straps_bn_bwd_x3': (_I, [_P] * 13 + [_L, _P, _L, _I, _I, _P]),
    # ReLU decisions as bits (ABI 8)
    'straps_bn_apply_bits_x3': (_I, [_P, _P, _P, _P, _P, _P, _L, _P, _L, _I, _P]),
    'straps_bn_bwd_bits_x3': (_I, [_P] * 10 + [_L, _P, _L, _I, _I, _P]),
    'straps_bn_bwd_finish_bits_x3': (_I, [_P] * 10 + [_L, _P, _I, _P, _L, _I, _I, _P]),
    'straps_conv_dgrad_x3_bits': (_I, [_P, _L, _P, _L, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P]),
    'straps_conv_dgrad_x3_bn_bits': (_I, [_P, _L, _P, _L, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    'straps_conv_stat_blocks': (_I, [_I, _I, _I, _I, _I, _I]),
    'straps_conv_fwd': (_I, [_P, _P, _P, _P, _P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    'straps_maxpool_fwd': (_I, [_P, _P, _I, _I, _I, _I, _P]),
    'straps_gap_fwd': (_I, [_P, _P, _I, _I, _I, _P]),
    'straps_bn_stats_finalize': (_I, [_P, _I, _I, _L, _P, _P, _F, _F, _P, _P, _P, _P, _P, _P, _P]),
    'straps_bn_apply': (_I, [_P, _P, _P, _P, _I, _P, _L, _I, _P]),
    'straps_linear_fwd': (_I, [_P, _I, _P, _I, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    'straps_broadcast_rows': (_I, [_P, _I, _P, _I, _I, _P]),
    'straps_ief_pack': (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    'straps_gemm_multi': (_I, [C.POINTER(GemmDesc), _I, _P]),
    'straps_rot6d_fwd': (_I, [_P, _L, _I, _P, _L, _P]),
    'straps_rodrigues_fwd': (_I, [_P, _P, _L, _P]),
    'straps_orthographic_project': (_I, [_P, _P, _I, _P, _L, _I, _P]),
    'straps_orthographic_project_bwd': (_I, [_P, _P, _I, _P, _P, _P, _L, _I, _P]),
    'straps_perspective_project': (_I, [_P, _P, _P, _P, _I, _P, _L, _I, _P]),
    'straps_smpl_workspace_bytes': (_Z, [C.POINTER(SmplModelStruct), _L]),
    'straps_smpl_fwd': (_I, [C.POINTER(SmplModelStruct), _P, _P, _P, _P, _P, _L, _I, _I, _P]),
    'straps_smpl_bwd_workspace_bytes': (_Z, [_L, _I]),
    'straps_smpl_bwd': (_I, [C.POINTER(SmplModelStruct), _P, _P, _P, _P, _P, _P, _P, _L, _I, _P]),
    'straps_conv_dgrad': (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    'straps_conv_wgrad_workspace_bytes': (_Z, [_I, _I, _I, _I, _I, _I, _I, _I, _I]),
    'straps_conv_wgrad': (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    'straps_stem_wgrad_workspace_bytes': (_Z, [_I, _I, _I, _I]),
    'straps_stem_wgrad': (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    'straps_bn_bwd_blocks': (_I, [_L, _I]),
    'straps_bn_bwd_workspace_bytes': (_Z, [_L, _I]),
    'straps_bn_bwd': (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _P]),
    'straps_bn_relu_maxpool_fwd': (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    'straps_bn_bwd_pooled': (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    'straps_bn_bwd_pooled_sparse': (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P]),
    'straps_stem_tiles': (_Z, [_I, _I, _I]),
    'straps_stem_tile_activity': (_I, [_P, _P, _I, _I, _I, _I, _P]),
    'straps_maxpool_fwd_idx': (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    'straps_maxpool_bwd': (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    'straps_gap_bwd': (_I, [_P, _P, _I, _I, _I, _P]),
    'straps_masked_copy': (_I, [_P, _I, _P, _I, _P, _I, _I, _I, _I, _P]),
    'straps_rot6d_bwd': (_I, [_P, _L, _I, _P, _P, _L, _L, _P]),
    'straps_build_proxy_input': (_I, [_P, _P, _P, _I, _I, _I, _P]),
    'straps_build_proxy_input_std': (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    'straps_build_proxy_input_nz': (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P]),
    'straps_loss_workspace_bytes': (_Z, [_L]),
    'straps_loss_fwd_bwd': (_I, [_P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _P]),
    'straps_loss_fwd_bwd_gm': (_I, [_P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _P, _F, _P]),
    'straps_count_visible': (_I, [_P, _P, _L, _I, _I, _P]),
    'straps_adam_step': (_I, [_P, _P, _P, _P, _L, _I, _F, _F, _F, _F, _F, _P, _P]),
    'straps_mse_fwd': (_I, [_P, _P, _P, _L, _I, _F, _F, _P, _P, _P]),
    'straps_mse_bwd': (_I, [_P, _P, _P, _L, _I, _F, _F, _P, _P, _P]),
    'straps_augment_seg': (_I, [_P, _P, _P, _F, _I, _P, _I, _I, _P]),
    'straps_rasterize_workspace_bytes': (_Z, [_L, _I, _I]),
    'straps_rasterize_parts': (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _I, _I, _F, _F, _P, _D, _D, _P]),
    'straps_philox_fill': (_I, [C.c_ulonglong, _P, _L, C.c_uint, _P, _L, _I, _P]),
    'straps_counter_add': (_I, [_P, _I, _L, _P]),
    'straps_gather_f32': (_I, [_P, _P, _P, _I, _P]),
    'straps_memset_zero': (_I, [_P, _Z, _P]),
    'straps_augment_smpl': (_I, [_P, _L, _P, _P, _P, _P, _I, _P, _D, _D, _P, _P, _P, _L, _P]),
    'straps_augment_cam_t': (_I, [_P, _P, _P, _D, _D, _D, _P, _L, _P]),
    'straps_deviate_verts2d': (_I, [_P, _P, _D, _D, _P, _L, _P]),
    'straps_deviate_joints2d': (_I, [_P, _P, _D, _D, _D, _D, _P, _L, _P]),
    'straps_project_targets': (_I, [_P, _P, _F, _F, _F, _F, _P, _P, _L, _P]),
    'straps_point_metrics': (_I, [_P, _P, _P, _L, _I, _P]),
    'straps_crop_resize': (_I, [_P, _P, _P, _D, _D, _D, _D, _D, _P, _P, _P, _I, _I, _I, _I, _P]),
    'straps_comm_unique_id': (_I, [_P]),
    'straps_comm_init_rank': (_I, [_P, _I, _I, C.POINTER(C.c_void_p)]),
    'straps_comm_destroy': (_I, [_P]),
    'straps_comm_size': (_I, [_P]),
    'straps_comm_library': (C.c_char_p, []),
    'straps_allreduce_grads': (_I, [_P, _L, _P, _P]),
}


def load(path=None):
    """dlopen the library and attach prototypes.  Raises if it has not been built."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.isfile(p):
        raise RuntimeError('libstraps_hip.so is not built (%s missing): run `python -c "import __graft_entry__ as g; '
                           'g.build()"` -- this package has no CPU fallback' % p)
    if os.path.abspath(p) != os.path.abspath(TOOLS_LIB_PATH):
        _audit(p, 'hipabi.load')      # (a library without a matching audit stamp is audited before it is opened; a finding raises)
    lib = C.CDLL(p)
    if hasattr(lib, 'straps_abi_version'):
        lib.straps_abi_version.restype = C.c_int
        got = lib.straps_abi_version()
        if got != ABI_VERSION:
            raise RuntimeError('%s is ABI version %d, this package binds version %d: rebuild it (`python -c "import __graft_entry__ as g; g.build()"`)'
                               % (p, got, ABI_VERSION))
    for name, (res, args) in SIGNATURES.items():
        if not hasattr(lib, name):
            continue                      # symbols of later sources may be absent in partial builds
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if path is None:
        _lib = lib
    return lib


def lib():
    return load()


def use_library(path):
    """make `path` THE library of this process (tools only: the -DSTRAPS_TOOLS build under tools/bin/).  Must run before the first call."""
    global _lib
    _lib = load(path)
    return _lib


_TRACE = os.environ.get('STRAPS_TRACE_CALLS')      # debugging aid: a file that receives the name of every entry point just launched, followed by a
_trace_fh = None                                    # device synchronisation -- after a GPU memory fault its last line names the kernel (tools/README)


def check(rc, what):
    if rc != 0:
        raise RuntimeError('%s failed (code %d): %s' % (what, rc, lib().straps_last_error().decode()))
    if _TRACE:
        import torch
        global _trace_fh
        if _trace_fh is None:
            _trace_fh = open(_TRACE, 'a', buffering=1)
        if not torch.cuda.is_current_stream_capturing():
            _trace_fh.write(what + '\n')
            _trace_fh.flush()
            os.fsync(_trace_fh.fileno())
            torch.cuda.synchronize()


def stream_ptr():
    """the current HIP stream of the CURRENT device: every launch goes there.  `require_gpu_tensor` refuses tensors that
    live on another device, and the module entry points run under `on_tensor_device`, so pointers and stream always belong
    to the same GPU."""
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _first_cuda_tensor(values):
    import torch
    for v in values:
        if isinstance(v, torch.Tensor):
            if v.is_cuda:
                return v
        elif isinstance(v, dict):
            t = _first_cuda_tensor(v.values())
            if t is not None:
                return t
    return None


def on_tensor_device(fn):
    """decorator for module entry points: run under torch.cuda.device(<device of the first GPU tensor argument>), so a
    module living on cuda:N launches on cuda:N's stream whatever the caller's current device is."""
    import functools

    @functools.wraps(fn)
    def wrapper(*a, **k):
        import torch
        t = _first_cuda_tensor(list(a) + list(k.values()))
        if t is None or t.device.index == torch.cuda.current_device():
            return fn(*a, **k)
        with torch.cuda.device(t.device):
            return fn(*a, **k)
    return wrapper


def ptr(t):
    """device pointer of a tensor (None -> NULL)."""
    return C.c_void_p(0 if t is None else t.data_ptr())


def require_gpu_tensor(t, name, dtype=None):
    import torch
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError('%s must be a GPU tensor: the STRAPS hot path runs only through the HIP library '
                           '(no CPU fallback)' % name)
    if dtype is not None and t.dtype != dtype:
        raise RuntimeError('%s must have dtype %s (got %s)' % (name, dtype, t.dtype))
    if t.device.index != torch.cuda.current_device():
        raise RuntimeError('%s lives on %s but the current device is cuda:%d: kernels are launched on the current device\'s '
                           'stream -- call torch.cuda.set_device(%d) or wrap the call in `with torch.cuda.device(...)`'
                           % (name, t.device, torch.cuda.current_device(), t.device.index))
    return t
