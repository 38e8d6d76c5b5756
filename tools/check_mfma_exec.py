#!/usr/bin/env python3
"""tools/check_mfma_exec.py -- static check: no MFMA may execute under a lane-divergent EXEC mask.

MFMA reads its A/B operands from ALL 64 lanes; when the compiler sinks one into a divergent region
(hipcc 7.2 does that to the block-scaled builtin, see gemm_native.hpp::pin_acc) the masked lanes'
rows are garbage.  Compiles every GEMM translation unit to gfx950 assembly and scans each kernel
for a v_mfma between an EXEC modification and its restore.  Takes a few minutes (no GPU needed).
"""
import re
import subprocess
import sys
import tempfile
from pathlib import Path

CSRC = Path(__file__).resolve().parent.parent / "petit-kernel_amd" / "csrc"
INC = CSRC.parent.parent / "include"


def scan(asm: str):
    kern, masked, bad, count = None, 0, {}, 0
    for line in asm.splitlines():
        m = re.match(r"^(_ZN\S+):", line)
        if m:
            kern, masked = m.group(1), 0
            count += 1
            continue
        if "s_endpgm" in line:
            kern = None
        if not kern:
            continue
        if "s_and_saveexec_b64" in line or re.search(r"s_(and|andn2|mov)_b64 exec", line):
            masked += 1
        if re.search(r"s_or_b64 exec, exec", line):
            masked = max(0, masked - 1)
        if "v_mfma" in line and masked:
            bad[kern] = bad.get(kern, 0) + 1
    return count, bad


def main() -> int:
    rc = 0
    with tempfile.TemporaryDirectory() as tmp:
        procs = []
        for src in sorted(CSRC.glob("gemm_*.hip")):
            out = Path(tmp) / (src.stem + ".s")
            procs.append((src, out, subprocess.Popen(
                ["hipcc", "-O3", "-std=c++20", "--offload-arch=gfx950", f"-I{INC}", "--cuda-device-only", "-S", str(src),
                 "-o", str(out)], stderr=subprocess.DEVNULL)))
        for src, out, proc in procs:
            if proc.wait() != 0:
                print(f"{src.name}: compile failed")
                rc = 1
                continue
            count, bad = scan(out.read_text())
            print(f"{src.name}: {count} kernels, {len(bad)} with an MFMA under a modified EXEC")
            for k, v in bad.items():
                print(f"    {k}: {v}")
                rc = 1
    return rc


if __name__ == "__main__":
    sys.exit(main())
