#!/usr/bin/env python3
"""tools/pack_sweeps.py <tag> <dir> [<dir> ...] -- add the per-sweep CSVs tools/tune.py wrote (gpurun_out/<x>_sweeps/*.csv) to
profiles/<tag>_sweeps.csv.gz (one table, first column = sweep name = the CSV's stem; a sweep of the same name is replaced)."""
import csv
import gzip
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
tag, dirs = sys.argv[1], sys.argv[2:]
dst = ROOT / "profiles" / f"{tag}_sweeps.csv.gz"
rows, header = [], None
if dst.exists():
    with gzip.open(dst, "rt", newline="") as f:
        r = csv.reader(f)
        header = next(r)
        rows = list(r)
for d in dirs:
    for f in sorted(Path(d).glob("*.csv")):
        with open(f, newline="") as fh:
            r = csv.reader(fh)
            h = ["sweep"] + next(r)
            if header is None:
                header = h
            assert h == header, (f, h, header)
            rows = [x for x in rows if x[0] != f.stem]
            rows += [[f.stem] + x for x in r]
with gzip.open(dst, "wt", newline="") as f:
    w = csv.writer(f)
    w.writerow(header)
    w.writerows(rows)
print(f"{len(rows)} rows -> {dst}")
