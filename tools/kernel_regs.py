#!/usr/bin/env python3
"""tools/kernel_regs.py <file.s> [filter] -- per kernel of a hipcc -save-temps assembly: VGPRs, AGPRs, spills, LDS bytes, scratch (from the .amdhsa metadata)."""
import re
import sys

text = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for blk in re.findall(r"- \.agpr_count:.*?\.wavefront_size: *\d+", text, re.S):
    name = re.search(r"\.name: +(\S+)", blk).group(1)
    if flt not in name:
        continue
    g = lambda k: int(re.search(r"\.%s: +(\d+)" % k, blk).group(1))
    short = re.sub(r"^_ZN9petit_amd\d+", "", name)[:110]
    print(f"vgpr {g('vgpr_count'):4d} agpr {g('agpr_count'):4d} spill {g('vgpr_spill_count'):3d} scratch {g('private_segment_fixed_size'):5d} lds {g('group_segment_fixed_size'):6d} wg {g('max_flat_workgroup_size'):5d}  {short}")
