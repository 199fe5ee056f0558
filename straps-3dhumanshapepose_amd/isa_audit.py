"""Audit of the built library's gfx950 code objects for the instruction form DESIGN section 1 names: a packed (VOP3P) instruction whose LOW result
reads the HIGH register of a source -- `op_sel` with a 1 in it, e.g. `v_pk_fma_f32 v[0:1], v[2:3], v[4:5], v[6:7] op_sel:[0,1,0]`.  On MI355X such an
instruction returns a wrong low result in lanes 48..63 while a bf16x3 convolution workgroup runs on the same compute unit
(profiles/r05_packed_fp32_victim.txt, tools/packed_fp32_hazard_repro.hip; the forms measured are listed in DESIGN section 1).

Round 6: this is a BUILD property, not a convention -- `hipabi.build()` audits the library it has just linked and raises (the file is moved aside), and
`hipabi.load()` refuses a library whose audit stamp is missing or stale and that fails the audit.  Where llvm-objdump is absent (a deployment box without
the ROCm LLVM tools) the check cannot run: build() and load() then say so with a warning instead of passing silently.
`tools/audit_packed_fp32.py` is the command-line front end; `tests/test_packed_fp32_audit.py` keeps the positive control."""
import hashlib
import os
import re
import struct
import subprocess
import tempfile

LLVM_BIN = os.environ.get('STRAPS_LLVM_BIN', '/opt/rocm/lib/llvm/bin')
MAGIC = b'__CLANG_OFFLOAD_BUNDLE__'
# mnemonics audited.  Measured to misexecute with a low-half select on src1 (round 5): v_pk_fma_f32, v_pk_mul_f32, v_pk_add_f32.  The other VOP3P forms
# with operand selects a compiler can emit are audited as well -- v_pk_mov_b32 and the packed 16-bit arithmetic (round 6: measured beside the same
# aggressor by tools/packed_fp32_hazard_repro.hip, result in profiles/r06_packed_forms_repro.txt); none of them occurs in this library with a select,
# so refusing them costs nothing.
PACKED = re.compile(r'\bv_pk_(?:(?:fma|mul|add)_f32|mov_b32|(?:fma|add|mul|min|max)_f16|(?:add|sub|mad|mul_lo|min|max)_[ui]16|(?:lshlrev|lshrrev|ashrrev)_b16)\b')
OP_SEL = re.compile(r'\bop_sel:\[([01,]+)\]')


class AuditUnavailable(RuntimeError):
    """llvm-objcopy / llvm-objdump are not installed: the audit cannot run"""


def available():
    return all(os.path.isfile(os.path.join(LLVM_BIN, t)) for t in ('llvm-objcopy', 'llvm-objdump'))


def code_objects(library):
    """the gfx950 code objects inside `library`, as bytes (uncompressed clang offload bundles of its .hip_fatbin section)"""
    if not available():
        raise AuditUnavailable('llvm-objcopy / llvm-objdump not found under %s (STRAPS_LLVM_BIN)' % LLVM_BIN)
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


def disassemble(code_object):
    with tempfile.NamedTemporaryFile(suffix='.co') as f:
        f.write(code_object)
        f.flush()
        return subprocess.run([os.path.join(LLVM_BIN, 'llvm-objdump'), '-d', '--mcpu=gfx950', f.name], check=True, capture_output=True, text=True).stdout


def audit(library):
    """([(kernel symbol, instruction text)] of every audited packed instruction with a low-half operand select, functions seen, packed instructions seen)"""
    found, kernels, packed = [], 0, 0
    for co in code_objects(library):
        symbol = '?'
        for line in disassemble(co).splitlines():
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


# ---- stamp: build() leaves `<library>.audit` = sha256 of the audited file; load() trusts a matching stamp and audits otherwise ----
def _sha256(path):
    h = hashlib.sha256()
    with open(path, 'rb') as f:
        for blk in iter(lambda: f.read(1 << 20), b''):
            h.update(blk)
    return h.hexdigest()


def stamp_path(library):
    return library + '.audit'


def stamp_ok(library):
    try:
        rec = open(stamp_path(library)).read().split()
    except OSError:
        return False
    return len(rec) >= 2 and rec[0] == 'clean' and rec[1] == _sha256(library)


def write_stamp(library, kernels, packed):
    with open(stamp_path(library), 'w') as f:
        f.write('clean %s functions=%d packed=%d\n' % (_sha256(library), kernels, packed))


def describe(found, limit=20):
    return '\n'.join('   %s: %s' % (sym[:100], ins) for sym, ins in found[:limit]) + ('\n   ... %d more' % (len(found) - limit) if len(found) > limit else '')


def enforce(library, what):
    """audit `library`; raise RuntimeError (after moving the file aside, so that nothing can load it) on a finding; write the stamp when clean.
    Returns True when the audit ran, False when the LLVM tools are missing (the caller warns)."""
    try:
        found, kernels, packed = audit(library)
    except AuditUnavailable:
        return False
    if kernels < 1:
        raise RuntimeError('%s: the ISA audit found no device function in %s (compressed offload bundles?): the check did not run' % (what, library))
    if found:
        rejected = library + '.rejected'
        try:
            os.replace(library, rejected)
        except OSError:
            rejected = library
        try:
            os.remove(stamp_path(library))
        except OSError:
            pass
        raise RuntimeError('%s: %d packed instruction(s) with a low-half operand select in the device code -- on MI355X these return wrong results in lanes '
                           '48..63 beside bf16 / fp16 32x32x16 MFMA work (DESIGN section 1).  Mark the kernel STRAPS_NO_PACKED_FP32 (csrc/common.h) or '
                           'restructure it; the library was moved to %s.\n%s' % (what, len(found), rejected, describe(found)))
    write_stamp(library, kernels, packed)
    return True
