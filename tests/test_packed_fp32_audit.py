"""The built library holds no packed fp32 instruction with a low-half operand select (VOP3P `op_sel`): on MI355X those return wrong low results in lanes
48..63 beside a bf16x3 convolution workgroup (round 5; csrc/common.h STRAPS_NO_PACKED_FP32).  Runs without a GPU: the library is
cross-compiled, its code objects are disassembled (tools/audit_packed_fp32.py)."""
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import audit_packed_fp32  # noqa: E402
from straps_amd import hipabi  # noqa: E402

needs_llvm = pytest.mark.skipif(not os.path.isfile(os.path.join(audit_packed_fp32.LLVM_BIN, 'llvm-objdump')), reason='no llvm-objdump')


@needs_llvm
def test_no_kernel_of_the_library_holds_a_packed_fp32_instruction_with_a_low_half_select():
    found, kernels, packed = audit_packed_fp32.audit(hipabi.build())
    assert kernels > 100, 'the auditor saw %d functions: the code objects were not found' % kernels
    assert not found, 'packed fp32 instructions with a low-half operand select (mark the kernel STRAPS_NO_PACKED_FP32):\n' + '\n'.join('%s: %s' % f for f in found[:20])


@needs_llvm
def test_the_auditor_sees_the_instruction_when_there_is_one(tmp_path):
    # (positive control: the compiler forms v_pk_fma_f32 ... op_sel:[0,1,0] for this kernel, and no longer does with the attribute)
    src = tmp_path / 'k.hip'
    src.write_text(textwrap.dedent('''
        #include <hip/hip_runtime.h>
        typedef float f2 __attribute__((ext_vector_type(2)));
        __global__ ATTR void k(const f2* a, const f2* b, f2* c) { int i = threadIdx.x; f2 x = a[i], y = b[i]; c[i] = x * y.y + c[i]; }
    '''))
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    seen = {}
    for name, attr in (('plain', ''), ('marked', '__attribute__((target("no-packed-fp32-ops")))')):
        lib = tmp_path / (name + '.so')
        subprocess.run([hipcc, '--offload-arch=gfx950', '-O3', '-fPIC', '-shared', '-DATTR=' + attr, str(src), '-o', str(lib)], check=True, capture_output=True)
        seen[name] = audit_packed_fp32.audit(str(lib))[0]
    assert any('op_sel:[0,1,0]' in ins for _, ins in seen['plain']), seen['plain']
    assert not seen['marked'], seen['marked']


@needs_llvm
def test_a_library_with_a_planted_instruction_cannot_pass_the_build_audit_or_be_loaded(tmp_path):
    """round 6: the audit is part of hipabi.build() / hipabi.load() (isa_audit.enforce): a library that holds the instruction raises
    and is moved aside; a clean one gets a stamp that load() trusts; a stale stamp (the file changed) is not trusted."""
    from straps_amd import isa_audit
    src = tmp_path / 'k.hip'
    src.write_text(textwrap.dedent('''
        #include <hip/hip_runtime.h>
        typedef float f2 __attribute__((ext_vector_type(2)));
        __global__ ATTR void k(const f2* a, const f2* b, f2* c) { int i = threadIdx.x; f2 x = a[i], y = b[i]; c[i] = x * y.y + c[i]; }
    '''))
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    planted, clean = tmp_path / 'planted.so', tmp_path / 'clean.so'
    subprocess.run([hipcc, '--offload-arch=gfx950', '-O3', '-fPIC', '-shared', '-DATTR=', str(src), '-o', str(planted)], check=True, capture_output=True)
    subprocess.run([hipcc, '--offload-arch=gfx950', '-O3', '-fPIC', '-shared', '-DATTR=__attribute__((target("no-packed-fp32-ops")))', str(src), '-o', str(clean)],
                   check=True, capture_output=True)
    # load(): no stamp -> audited -> refused (before dlopen), and the file is gone from its path
    with pytest.raises(RuntimeError, match='low-half operand select'):
        hipabi.load(str(planted))
    assert not planted.exists() and (tmp_path / 'planted.so.rejected').exists()
    # the build-side call on a clean library: passes, leaves the stamp; the stamp is bound to the file's content
    assert isa_audit.enforce(str(clean), 'test') is True
    assert isa_audit.stamp_ok(str(clean))
    with open(clean, 'ab') as f:
        f.write(b'\\0')
    assert not isa_audit.stamp_ok(str(clean))
