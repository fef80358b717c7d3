"""Durations of the LAST n dispatches of each kernel whose name contains one of the given substrings, from a rocprofv3 rocpd database -- e.g. the timed loop at the end of
`python -m phc_amd.learning.bench_policy` (the env step in the trained-policy regime) without the training that precedes it.  python last_calls.py <db> <n> <substring> ..."""
import sqlite3
import sys

import numpy as np


def main(path, n, subs):
    db = sqlite3.connect(path)
    rows = db.execute("select name, start, duration, grid_x from kernels order by start").fetchall()
    print(f"# source: {path}; last {n} dispatches of each kernel")
    print(f"# {'kernel':50s} {'calls':>6s} {'avg_us':>8s} {'median':>8s} {'min_us':>8s} {'max_us':>8s} {'grid':>8s}")
    names = sorted({r[0] for r in rows if any(s in r[0] for s in subs)})
    for name in names:
        d = np.array([r[2] for r in rows if r[0] == name][-n:]) / 1e3
        g = max(r[3] for r in rows if r[0] == name)
        print(f"  {name.split('(')[0].replace('void ', '')[:50]:50s} {len(d):6d} {d.mean():8.2f} {np.median(d):8.2f} {d.min():8.2f} {d.max():8.2f} {g:8d}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]), sys.argv[3:])
