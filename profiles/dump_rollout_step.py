"""Kernels of ONE rollout step of the PPO epoch from a rocprofv3 rocpd database: the dispatches between two consecutive stepper launches
that are followed by a policy step (i.e. inside play_steps), in launch order.  python dump_rollout_step.py <db> [k]"""
import sqlite3
import sys


def main(path, which=10):
    db = sqlite3.connect(path)
    rows = db.execute("select name, start, duration, grid_x from kernels order by start").fetchall()
    adam = [i for i, r in enumerate(rows) if "k_adam" in r[0]]
    sims = [i for i, r in enumerate(rows) if "k_sim_step" in r[0] and i > adam[len(adam) // 2]]   # a rollout after the first update
    a, b = sims[which], sims[which + 1]
    step = rows[a:b]
    t0 = step[0][1]
    print(f"# rollout step {which}: {len(step)} dispatches, {sum(r[2] for r in step) / 1e3:.1f} us busy, {(rows[b][1] - t0) / 1e3:.1f} us wall")
    for name, start, dur, grid in step:
        print(f"{(start - t0) / 1e3:9.1f} us  {dur / 1e3:7.1f} us  grid {grid:9d}  {name.split('(')[0][:110]}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 10)
