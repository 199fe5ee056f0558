#!/usr/bin/env python3
"""tools/audit_packed_fp32.py [library.so] -- does any kernel of the built library hold a packed (VOP3P) instruction whose LOW result reads the HIGH
register of a source (`op_sel`, e.g. `v_pk_fma_f32 v[0:1], v[2:3], v[4:5], v[6:7] op_sel:[0,1,0]`)?  Command-line front end of
straps-3dhumanshapepose_amd/isa_audit.py -- the same audit `hipabi.build()` runs on every library it links and `hipabi.load()` on every library without
a matching stamp (round 6: a finding is a build failure).  Exit status 1 if there is one."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from straps_amd import isa_audit  # noqa: E402
from straps_amd.isa_audit import LLVM_BIN, audit, code_objects  # noqa: E402,F401  (re-exported: tests and probes import them from here)

if __name__ == '__main__':
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, 'straps-3dhumanshapepose_amd', 'csrc', 'libstraps_hip.so')
    found, kernels, packed = audit(lib)
    print('%s: %d functions, %d audited packed instructions, %d of them with a low-half operand select' % (lib, kernels, packed, len(found)))
    print(isa_audit.describe(found, 40) if found else '', end='')
    sys.exit(1 if found else 0)
