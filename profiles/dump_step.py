"""Kernels of ONE PPO optimizer step from a rocprofv3 rocpd database: the dispatches between two consecutive fused-Adam launches
(in the middle of the update), in launch order with duration and grid, then aggregated by kernel.  python dump_step.py <db> [k]"""
import collections
import sqlite3
import sys


def main(path, which=20):
    db = sqlite3.connect(path)
    rows = db.execute("select name, start, duration, grid_x from kernels order by start").fetchall()
    adam = [i for i, r in enumerate(rows) if "adam" in r[0].lower()]
    if len(adam) < which + 2:
        which = len(adam) // 2
    a, b = adam[which], adam[which + 1]
    step = rows[a + 1:b + 1]
    t0 = step[0][1]
    print(f"# optimizer step {which}: {len(step)} dispatches, {sum(r[2] for r in step) / 1e3:.1f} us busy, {(step[-1][1] + step[-1][2] - t0) / 1e3:.1f} us wall")
    for name, start, dur, grid in step:
        print(f"{(start - t0) / 1e3:9.1f} us  {dur / 1e3:7.1f} us  grid {grid:9d}  {name.split('(')[0][:110]}")
    agg = collections.defaultdict(lambda: [0, 0])
    for name, start, dur, grid in step:
        k = (name.split("(")[0][:90], grid)
        agg[k][0] += 1
        agg[k][1] += dur
    print("\n# aggregated")
    for (name, grid), (n, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:60]:
        print(f"{d / 1e3:8.1f} us  x{n:<3d} grid {grid:9d}  {name}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 20)
