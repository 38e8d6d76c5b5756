#!/usr/bin/env python3
"""tools/adopt_rows.py --log NEW.csv[.gz] [--confirm OLD.csv[.gz]] [--ms 32,64] [--klass 8|6|4] --out rows.tune.txt -- which rows of csrc/tuned_gfx950.inc (with --klass: of
csrc/tuned_native_gfx950.inc, the class whose activations are MXFP8 / MXFP6 / MXFP4) a tuning session may replace.

The in-library tuner's per-candidate log ($PETIT_AMD_TUNE_LOG, tools/build_table.py) times the table's current pick and every challenger in ONE session.  A row is
replaced only when (single-session timings sit inside the noise the picks are made on)
  * the problem was timed in two sessions (--confirm): the candidate with the best geometric-mean time beats the table's row by >= 3 % in EACH session;
  * otherwise: the session's winner beats the table's row by >= 5 %.
Writes the rows to adopt in the $PETIT_AMD_TUNE_FILE format for tools/make_tuned_inc.py and prints what it did."""
import argparse
import math
import re
import statistics
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tools"))
from table_from_candidates import read  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log", required=True)
    ap.add_argument("--confirm", default="")
    ap.add_argument("--ms", default="")
    ap.add_argument("--out", required=True)
    ap.add_argument("--table", default="")
    ap.add_argument("--keep-untimed", action="store_true", help="a row whose kernel the session did not time (no longer a candidate at this M) stays, instead of being replaced by the session's winner")
    ap.add_argument("--klass", type=int, default=0, choices=[0, 8, 6, 4], help="0: the exact class; 8 / 6 / 4: the native class with MXFP8 / MXFP6 / MXFP4 activations")
    a = ap.parse_args()
    a.table = a.table or str(ROOT / "petit-kernel_amd" / "csrc" / ("tuned_native_gfx950.inc" if a.klass else "tuned_gfx950.inc"))
    act_code = {0: None, 8: 2, 6: 4, 4: 6}[a.klass]       # bits 32-34 of a native id
    b2, c2 = read(a.log)
    c1 = read(a.confirm)[1] if a.confirm else {}
    ms = {int(x) for x in a.ms.split(",")} if a.ms else None
    rows = {}
    for at, bt, n, k, lo, hi, sol in re.findall(r"\{(\d+), (\d+), (\d+)u, (\d+)u, (\d+)u, (\d+)u, 0x([0-9a-f]+)ull\}", Path(a.table).read_text()):
        if act_code is None or (int(sol, 16) >> 32) & 7 == act_code:
            rows.setdefault((int(at), int(bt), int(n), int(k)), []).append((int(lo), int(hi), int(sol, 16)))
    out, gains, kept, two, one = [], [], 0, 0, 0
    for key in sorted(b2):
        at, bt, klass, m, n, k = key
        if klass != a.klass or (ms and m not in ms):
            continue
        old = [s for lo, hi, s in rows.get((at, bt, n, k), []) if lo <= m <= hi]
        o2 = c2[key].get(old[0]) if old else None
        if o2 is None:          # no row, or the row's kernel was not timed (its output check failed at this shape): take the session's winner
            if old and a.keep_untimed:
                kept += 1
                continue
            out.append((at, bt, n, k, m, b2[key][0]))
            continue
        if key in c1 and c1[key].get(old[0]) is not None:
            o1 = c1[key][old[0]]
            gm, cand = min((math.sqrt(c1[key][s] * c2[key][s]), s) for s in c2[key] if s in c1[key])
            win, us = b2[key]
            if c1[key][cand] < 0.97 * o1 and c2[key][cand] < 0.97 * o2 and not (win not in c1[key] and us < 0.95 * c2[key][cand]):
                out.append((at, bt, n, k, m, cand))
                gains.append(math.sqrt(o1 * o2) / gm)
                two += 1
            elif win not in c1[key] and us < 0.95 * o2:   # a kernel the earlier session did not have: the single-session margin
                out.append((at, bt, n, k, m, win))
                gains.append(o2 / us)
                one += 1
            else:
                kept += 1
        else:
            cand, us = b2[key]
            if us < 0.95 * o2:
                out.append((at, bt, n, k, m, cand))
                gains.append(o2 / us)
                one += 1
            else:
                kept += 1
    with open(a.out, "w") as f:
        f.write(f"# a_type b_type n k m_lo m_hi solution   (tools/adopt_rows.py --log {Path(a.log).name}" + (f" --confirm {Path(a.confirm).name}" if a.confirm else "") + ")\n")
        for at, bt, n, k, m, sid in out:
            f.write(f"{at} {bt} {n} {k} {m} {m} {sid:x}\n")
    print(f"{len(out)} rows to adopt ({two} confirmed in two sessions by >= 3 % each, {one} by >= 5 % in one session, {len(out) - two - one} without a timed incumbent), {kept} kept"
          + (f"; gain of the replaced rows: median {statistics.median(gains):.3f}, max {max(gains):.3f}" if gains else ""))


if __name__ == "__main__":
    main()
