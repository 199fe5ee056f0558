#!/usr/bin/env python3
"""Audit of the libraries whose kernels run BESIDE the training step for the packed-operand-select hazard of DESIGN section 1 ("whether
RCCL's or torch's own kernels beside the step carry the form -- they become co-residents at N > 1").

The victim form is a VOP3P instruction whose LOW result reads the HIGH register of a source (`op_sel` with a 1); beside workgroups that stream 32x32x16
bf16/fp16 MFMAs fed by `ds_read_b128` -- this repo's convolutions -- it returns a wrong low result in lanes 48..63.  At N > 1 the all-reduce kernels of RCCL
run on the exchange stream WHILE the backward convolutions run, so an RCCL reduction kernel holding the form would be a victim.  This tool disassembles the
gfx950 code objects of the given libraries (compressed offload bundles included: they are unbundled with clang-offload-bundler) and counts, per library:
functions, audited packed instructions, the ones with a low-half select (the hazard), and the `op_sel_hi`-only forms (scalar broadcast; measured clean:
form 3 of tools/packed_fp32_hazard_repro.hip).

No GPU.  Usage:  python tools/audit_neighbours.py [--dump rows.tsv] [library ...]   (default: every librccl.so this image holds + libtorch_hip.so)
Result of the round-6 run: profiles/r06_neighbour_audit.txt."""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from straps_amd import isa_audit  # noqa: E402

TARGET = 'hipv4-amdgcn-amd-amdhsa--gfx950'


def gfx950_objects(library, tmp):
    """paths of the gfx950 code objects of `library`: uncompressed bundles through isa_audit.code_objects, compressed ones through the bundler"""
    out = []
    fat = os.path.join(tmp, 'fat.bin')
    r = subprocess.run([os.path.join(isa_audit.LLVM_BIN, 'llvm-objcopy'), '--dump-section', '.hip_fatbin=' + fat, library], capture_output=True, text=True)
    if r.returncode or not os.path.isfile(fat):
        return out, 'no .hip_fatbin section'
    data = open(fat, 'rb').read()
    import struct
    if data.startswith(b'CCOB'):
        # compressed bundles, one per translation unit, each `CCOB` + version(u16) + method(u16) + file size + uncompressed size + hash, padded to a page
        bundler = os.path.join(isa_audit.LLVM_BIN, 'clang-offload-bundler')
        n_b = 0
        for m in re.finditer(b'CCOB', data):
            p0 = m.start()
            ver, method = struct.unpack_from('<HH', data, p0 + 4)
            if p0 % 8 or ver not in (2, 3) or method > 1:
                continue                                                  # the four letters inside compressed data
            size = struct.unpack_from('<I' if ver == 2 else '<Q', data, p0 + 8)[0]
            one = os.path.join(tmp, 'b%d.bin' % n_b)
            with open(one, 'wb') as f:
                f.write(data[p0:p0 + size])
            targets = subprocess.run([bundler, '--list', '--type=o', '--input=' + one], capture_output=True, text=True).stdout.split()
            mine = [t for t in targets if t.startswith(TARGET)]
            if mine:
                co = os.path.join(tmp, 'gfx950_%d.co' % n_b)
                subprocess.run([bundler, '--unbundle', '--type=o', '--input=' + one, '--targets=' + mine[0], '--output=' + co], check=True)
                out.append(co)
            os.unlink(one)
            n_b += 1
        return out, '%d compressed bundles, %d with a gfx950 object' % (n_b, len(out))
    n_obj = 0
    for m in re.finditer(re.escape(isa_audit.MAGIC), data):
        base = m.start()
        n, = struct.unpack_from('<Q', data, base + len(isa_audit.MAGIC))
        q = base + len(isa_audit.MAGIC) + 8
        for _ in range(n):
            off, size, tl = struct.unpack_from('<QQQ', data, q)
            triple = data[q + 24:q + 24 + tl].decode()
            q += 24 + tl
            if 'gfx950' in triple and size:
                p = os.path.join(tmp, 'o%d.co' % n_obj)
                n_obj += 1
                with open(p, 'wb') as f:
                    f.write(data[base + off:base + off + size])
                out.append(p)
    return out, '%d uncompressed gfx950 objects' % n_obj


def audit(library):
    functions = packed = 0
    hazard, hi_only, kinds = [], collections.Counter(), collections.Counter()
    with tempfile.TemporaryDirectory(dir=os.environ.get('TMPDIR', '/tmp')) as tmp:
        objs, how = gfx950_objects(library, tmp)
        for co in objs:
            p = subprocess.Popen([os.path.join(isa_audit.LLVM_BIN, 'llvm-objdump'), '-d', '--mcpu=gfx950', co], stdout=subprocess.PIPE, text=True)
            symbol = '?'
            for line in p.stdout:
                if line[:1] in '0123456789abcdef' and line.rstrip().endswith('>:'):
                    s = re.match(r'^[0-9a-f]+ <(.+)>:$', line.rstrip())
                    if s:
                        symbol = s.group(1)
                        functions += 1
                        continue
                if 'v_pk_' not in line:
                    continue
                m = isa_audit.PACKED.search(line)
                if not m:
                    continue
                packed += 1
                kinds[m.group(0)] += 1
                sel = isa_audit.OP_SEL.search(line)
                if sel and '1' in sel.group(1):
                    hazard.append((symbol, line.split('//')[0].strip()))
                else:
                    h = re.search(r'op_sel_hi:\[[01,]+\]', line)
                    if h:
                        hi_only[m.group(0) + ' ' + h.group(0)] += 1
            p.wait()
    return dict(how=how, functions=functions, packed=packed, kinds=kinds, hazard=hazard, hi_only=hi_only)


def default_libraries():
    libs = []
    try:
        import torch
        tl = os.path.join(os.path.dirname(torch.__file__), 'lib')
        libs += [os.path.join(tl, 'librccl.so'), os.path.join(tl, 'libtorch_hip.so')]
    except Exception:
        pass
    libs.append('/opt/rocm/lib/librccl.so.1')
    return [os.path.realpath(x) for x in libs if os.path.exists(x)]


def main():
    args = sys.argv[1:]
    dump = None
    if '--dump' in args:
        i = args.index('--dump')
        dump = args[i + 1]
        del args[i:i + 2]
    libs = args or default_libraries()
    bad = 0
    for lib in libs:
        r = audit(lib)
        print('%s  (%s)' % (lib, r['how']))
        print('   gfx950 functions %d, audited packed instructions %d  %s' % (r['functions'], r['packed'], dict(r['kinds'].most_common(8))))
        print('   with a LOW-half operand select (the hazard form): %d' % len(r['hazard']))
        if r['hazard']:
            src1 = [(sym, ins) for sym, ins in r['hazard'] if re.match(r'v_pk_(fma|mul|add)_f32', ins) and isa_audit.OP_SEL.search(ins).group(1).split(',')[1] == '1']
            print('      of these, the form MEASURED to misexecute (packed fp32 arithmetic, low-half select on src1): %d instructions in %d functions'
                  % (len(src1), len(set(sym for sym, _ in src1))))
            if dump:
                with open(dump, 'a') as f:
                    for sym, ins in r['hazard']:
                        f.write('%s\t%s\t%s\n' % (os.path.basename(lib), sym, ins))
            forms = collections.Counter((ins.split()[0], (isa_audit.OP_SEL.search(ins) or [''])[0]) for _, ins in r['hazard'])
            print('      by form: %s' % ', '.join('%s %s x %d' % (a, b, n) for (a, b), n in forms.most_common(12)))
            per_fn = collections.Counter(sym for sym, _ in r['hazard'])
            print('      in %d distinct functions; by family (demangled name up to its first template bracket):' % len(per_fn))
            fam = collections.Counter()
            names = list(per_fn)
            import shutil
            filt = shutil.which('c++filt') or shutil.which('llvm-cxxfilt')
            dem = subprocess.run([filt], input='\n'.join(names), capture_output=True, text=True).stdout.splitlines() if filt else names
            for raw, d in zip(names, dem):
                fam[re.sub(r'^void ', '', d).split('<')[0].split('(')[0][:90]] += 1
            for f, n in fam.most_common(25):
                print('         %5d functions  %s' % (n, f))
            for sym, ins in r['hazard'][:6]:
                print('      e.g. %s: %s' % (sym[:100], ins))
        print('   op_sel_hi-only forms (scalar broadcast; measured clean): %s' % dict(r['hi_only'].most_common(6)))
        bad += len(r['hazard'])
    print('TOTAL with a low-half select: %d' % bad)
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
