#!/usr/bin/env python3
"""tools/isa_blocks.py file.s -- per basic block of one kernel's ISA: VALU / convert / MFMA / LDS / SALU counts."""
import re
import sys

lines = open(sys.argv[1]).read().split('\n')
segs = {}
cur = 'entry'
segs[cur] = []
for l in lines:
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m:
        cur = m.group(1)
        segs[cur] = []
    else:
        segs[cur].append(l.strip())
for k, v in segs.items():
    valu = sum(1 for x in v if x.startswith('v_') and not x.startswith('v_mfma'))
    mf = sum(1 for x in v if x.startswith('v_mfma'))
    cvt = sum(1 for x in v if x.startswith('v_cvt_scalef32'))
    ds = sum(1 for x in v if x.startswith('ds_'))
    vm = sum(1 for x in v if x.startswith('buffer_') or x.startswith('global_'))
    sal = sum(1 for x in v if x.startswith('s_'))
    print(f"{k:10s} valu {valu:4d} (fp4 cvt {cvt:4d})  mfma {mf:3d}  lds {ds:3d}  vmem {vm:3d}  salu {sal:4d}")
