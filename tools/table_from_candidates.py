#!/usr/bin/env python3
"""tools/table_from_candidates.py candidates.csv[.gz] [--out rows.tune.txt] -- the winners of a tools/build_table.py run, recovered from the
per-candidate log the library wrote ($PETIT_AMD_TUNE_LOG: a_type,b_type,klass,m,n,k,solution,us_median,samples): per problem the fastest
candidate that got its full set of samples (a candidate dropped as hopeless after one sample is never a winner).  Held-out shapes
(tools/build_table.py HELDOUT) are left out: they measure the heuristic."""
import csv
import gzip
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent))
from build_table import HELDOUT


def read(path):
    op = gzip.open if str(path).endswith(".gz") else open
    best, cands = {}, {}
    with op(path, "rt") as f:
        for row in csv.reader(f):
            if len(row) != 9:
                continue
            at, bt, klass, m, n, k = (int(x) for x in row[:6])
            sid, us, samples = int(row[6], 16), float(row[7]), int(row[8])
            key = (at, bt, klass, m, n, k)
            cands.setdefault(key, {})[sid] = min(us, cands.get(key, {}).get(sid, 1e30))
            if samples >= 3 and (key not in best or us < best[key][1]):
                best[key] = (sid, us)
    return best, cands


if __name__ == "__main__":
    src = sys.argv[1]
    out = sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else str(Path(src).with_suffix("").with_suffix(".tune.txt"))
    best, _ = read(src)
    rows = [(k, v) for k, v in sorted(best.items()) if k[2] == 0 and (k[4], k[5]) not in HELDOUT]
    with open(out, "w") as f:
        f.write("# a_type b_type n k m_lo m_hi solution   (tools/table_from_candidates.py; $PETIT_AMD_TUNE_FILE format)\n")
        for (at, bt, klass, m, n, k), (sid, us) in rows:
            f.write(f"{at} {bt} {n} {k} {m} {m} {sid:x}\n")
    print(f"{len(rows)} rows -> {out}  ({len(best) - len(rows)} held-out / native problems left out)")
