"""Kernels of ONE rollout step of the learner from a rocprofv3 rocpd database: the dispatches between two consecutive stepper launches (k_sim_step) in the middle of a
rollout, in launch order with duration and grid, then aggregated by kernel.  python dump_rollout_step.py <db> [k]"""
import collections
import sqlite3
import sys


def main(path, which=-40):
    db = sqlite3.connect(path)
    rows = db.execute("select name, start, duration, grid_x from kernels order by start").fetchall()
    sim = [i for i, r in enumerate(rows) if "k_sim_step" in r[0]]
    # a step inside a rollout: the gap to the next stepper launch is short (no update in between)
    gaps = [(rows[sim[j + 1]][1] - rows[sim[j]][1]) for j in range(len(sim) - 1)]
    inside = [j for j, g in enumerate(gaps) if g < 2e6]
    j = inside[which]
    a, b = sim[j], sim[j + 1]
    step = rows[a:b]
    t0 = step[0][1]
    print(f"# rollout step (stepper launch {j} of {len(sim)}): {len(step)} dispatches, {sum(r[2] for r in step) / 1e3:.1f} us busy, {(rows[b][1] - t0) / 1e3:.1f} us from stepper launch to stepper launch")
    for name, start, dur, grid in step:
        print(f"{(start - t0) / 1e3:9.1f} us  {dur / 1e3:7.1f} us  grid {grid:9d}  {name.split('(')[0][:110]}")
    agg = collections.defaultdict(lambda: [0, 0])
    for name, start, dur, grid in step:
        k = (name.split("(")[0][:90], grid)
        agg[k][0] += 1
        agg[k][1] += dur
    print("\n# aggregated")
    for (name, grid), (n, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
        print(f"{d / 1e3:8.1f} us  x{n:<3d} grid {grid:9d}  {name}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else -40)
