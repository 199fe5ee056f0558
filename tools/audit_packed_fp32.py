#!/usr/bin/env python3
"""tools/audit_packed_fp32.py [library.so] -- does any kernel of the built library hold a packed fp32 instruction whose LOW result reads the HIGH
register of a source (VOP3P `op_sel`, e.g. `v_pk_fma_f32 v[0:1], v[2:3], v[4:5], v[6:7] op_sel:[0,1,0]`)?

Round 5: on MI355X such an instruction returns a wrong low result in lanes 48..63 while a bf16x3 convolution workgroup runs on the
same compute unit (profiles/r05_packed_fp32_victim.txt).  The compiler forms them wherever two fp32 values live in a register pair; the kernels in
which it did are compiled without packed fp32 instructions (csrc/common.h, STRAPS_NO_PACKED_FP32).  This tool is the check that none is left: it takes
the gfx950 code objects out of the library's .hip_fatbin section (clang offload bundles), disassembles them with llvm-objdump and lists every such
instruction with the kernel that holds it.  Exit status 1 if there is one.  tests/test_packed_fp32_audit.py runs it on the product library."""
import os
import re
import struct
import subprocess
import sys
import tempfile

LLVM_BIN = os.environ.get('STRAPS_LLVM_BIN', '/opt/rocm/lib/llvm/bin')
MAGIC = b'__CLANG_OFFLOAD_BUNDLE__'
PACKED = re.compile(r'\bv_pk_(?:(?:fma|mul|add)_f32|mov_b32)\b')      # (v_pk_mov_b32: the 64-bit sibling with the same operand selects -- not measured, not wanted either)
OP_SEL = re.compile(r'\bop_sel:\[([01,]+)\]')


def code_objects(library):
    """the gfx950 code objects inside `library`, as bytes"""
    with tempfile.TemporaryDirectory() as tmp:
        fat = os.path.join(tmp, 'fat.bin')
        subprocess.run([os.path.join(LLVM_BIN, 'llvm-objcopy'), '--dump-section', '.hip_fatbin=' + fat, library], check=True)
        data = open(fat, 'rb').read()
    out = []
    for m in re.finditer(re.escape(MAGIC), data):
        base = m.start()
        n, = struct.unpack_from('<Q', data, base + len(MAGIC))
        q = base + len(MAGIC) + 8
        for _ in range(n):
            off, size, tl = struct.unpack_from('<QQQ', data, q)
            triple = data[q + 24:q + 24 + tl].decode()
            q += 24 + tl
            if 'amdgcn' in triple and size:
                out.append(data[base + off:base + off + size])
    return out


def audit(library):
    """[(kernel symbol, instruction text)] of every packed fp32 instruction with a low-half operand select; also the number of kernels and instructions seen"""
    found, kernels, packed = [], 0, 0
    for co in code_objects(library):
        with tempfile.NamedTemporaryFile(suffix='.co') as f:
            f.write(co)
            f.flush()
            text = subprocess.run([os.path.join(LLVM_BIN, 'llvm-objdump'), '-d', '--mcpu=gfx950', f.name], check=True, capture_output=True, text=True).stdout
        symbol = '?'
        for line in text.splitlines():
            s = re.match(r'^[0-9a-f]+ <(.+)>:$', line)
            if s:
                symbol = s.group(1)
                kernels += 1
                continue
            if PACKED.search(line):
                packed += 1
                sel = OP_SEL.search(line)
                if sel and '1' in sel.group(1):
                    found.append((symbol, line.split('//')[0].strip()))
    return found, kernels, packed


if __name__ == '__main__':
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, 'straps-3dhumanshapepose_amd', 'csrc', 'libstraps_hip.so')
    found, kernels, packed = audit(lib)
    print('%s: %d functions, %d packed fp32 instructions, %d of them with a low-half operand select' % (lib, kernels, packed, len(found)))
    for symbol, ins in found[:40]:
        print('   %s: %s' % (symbol[:100], ins))
    sys.exit(1 if found else 0)
