#!/usr/bin/env python3
"""tools/split_buckets.py -- one-off (round 5): the open-ended M bucket [257, 1 << 20] of both arch tables becomes 257-512 / 513-1024 / 1025-4096 /
4097+ (csrc/tune.hip tune_bucket, tools/make_tuned_inc.py BUCKET).

A row measured at M = 512 keeps its kernel in the new buckets until a measured row replaces it (tools/build_table.py --ms 1024,2048,8192), but NOT
its cross-workgroup K split where that split no longer makes sense at the bucket's representative M (1024 / 2048 / 8192): the rule is the library's
own (csrc/api.hip guarded_splitk: no split once the unsplit grid has >= 2 workgroups per CU; halve it while the fp32 slabs outgrow the operands),
restated here on the id's fields and checked against the library by tests/test_layout_and_abi.py::test_table_rows_pass_the_split_guard.
Rows of the 257-512 bucket are guarded at M = 512 as well (tune_candidates applies the rule, so a re-tuned row could not name such a split)."""
import re
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
NUM_CUS = 256
REP = {(257, 512): 512, (513, 1024): 1024, (1025, 4096): 2048, (4097, 1 << 20): 8192}


def tile(sol):
    kind = (sol >> 48) & 0xF
    bm = (32 if kind in (12, 13) else 16) * (sol & 0xFF)
    bn = 16 * ((sol >> 8) & 0xFF)
    return bm, bn


def guarded_splitk(sol, m, n, k):
    sk = (sol >> 60) & 0xF
    if sk <= 1:
        return sk
    bm, bn = tile(sol)
    tiles = -(-m // bm) * -(-n // bn)
    if tiles >= 2 * NUM_CUS:
        return 1
    group = 16 if (sol >> 28) & 0xF == 1 else 32
    cap = n * k // 2 + n * k // group + 2 * m * k + 2 * m * n
    while sk > 1 and sk * m * n * 4 > cap:
        sk >>= 1
    return sk


def with_splitk(sol, sk):
    return (sol & ~(0xF << 60)) | (sk << 60)


def main():
    for name in ("tuned_gfx950.inc", "tuned_native_gfx950.inc"):
        path = ROOT / "petit-kernel_amd" / "csrc" / name
        text = path.read_text()
        head = [ln for ln in text.splitlines() if ln.startswith("//")]
        rows = [(int(at), int(bt), int(n), int(k), int(lo), int(hi), int(sol, 16))
                for at, bt, n, k, lo, hi, sol in re.findall(r"\{(\d+), (\d+), (\d+)u, (\d+)u, (\d+)u, (\d+)u, 0x([0-9a-f]+)ull\}", text)]
        out, changed = [], 0
        for at, bt, n, k, lo, hi, sol in rows:
            if hi != 1 << 20 or lo != 257:
                # every other row: guarded at the M it was measured at (the bucket's upper end; the representative M of the prefill buckets)
                g = guarded_splitk(sol, REP.get((lo, hi), hi), n, k)
                changed += g != (sol >> 60) & 0xF
                out.append((at, bt, n, k, lo, hi, with_splitk(sol, g)))
                continue
            for (blo, bhi), rep in REP.items():
                g = guarded_splitk(sol, rep, n, k)
                changed += g != (sol >> 60) & 0xF
                out.append((at, bt, n, k, blo, bhi, with_splitk(sol, g)))
        native = "native" in name
        key = (lambda r: (r[0], r[1], r[2], r[3], r[4], r[5], (r[6] >> 32) & 7)) if native else (lambda r: r[:6])
        out.sort(key=key)
        body = [f"{{{at}, {bt}, {n}u, {k}u, {lo}u, {hi}u, 0x{sol:016x}ull}}," for at, bt, n, k, lo, hi, sol in out]
        path.write_text("\n".join(head + body) + "\n")
        print(f"{name}: {len(rows)} -> {len(out)} rows, {changed} K splits reduced by the guard")


if __name__ == "__main__":
    sys.exit(main())
