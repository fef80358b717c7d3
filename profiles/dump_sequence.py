"""Run-length-encoded, time-ordered list of the kernel dispatches in a rocprofv3 rocpd database (what launches when)."""
import sqlite3
import sys


def main(path, max_lines=120):
    db = sqlite3.connect(path)
    rows = db.execute("select name, start, duration, grid_x, queue_id from kernels order by start").fetchall()
    t0 = rows[0][1]
    out, prev, n, tstart, dur = [], None, 0, 0, 0
    for name, start, d, grid, q in rows:
        key = (name.split("(")[0][:60], grid, q)
        if key != prev:
            if prev is not None:
                out.append(f"{(tstart - t0) / 1e6:10.3f} ms  x{n:<5d} total {dur / 1e3:10.1f} us  grid {prev[1]:9d} q{prev[2]}  {prev[0]}")
            prev, n, tstart, dur = key, 0, start, 0
        n += 1
        dur += d
    out.append(f"{(tstart - t0) / 1e6:10.3f} ms  x{n:<5d} total {dur / 1e3:10.1f} us  grid {prev[1]:9d} q{prev[2]}  {prev[0]}")
    print(f"# {len(rows)} dispatches, {len(out)} runs; first {max_lines} runs:")
    print("\n".join(out[:max_lines]))


if __name__ == "__main__":
    main(sys.argv[1])
