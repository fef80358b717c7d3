"""Turn a rocprofv3 (ROCm 7.x rocpd sqlite) kernel trace into the per-kernel summary table committed under profiles/.

    rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o bench -- python bench.py ...
    python profiles/summarize_rocpd.py gpurun_out/prof/bench_results.db > profiles/rNN_<what>.txt
"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)").fetchall()]
    # an earlier table showed 112 in the `vgpr` column for the stepper, whose code object holds 224 VGPRs (`-Rpass-analysis=kernel-resource-usage`,
    # the number DESIGN.md quotes).  rocpd keeps architectural and accumulation registers in separate columns where it has both: the table prints their sum
    # and, on its first line, the columns this trace's `kernels` view has -- read the column against the compiler's number, not instead of it.
    vg = "max(vgpr_count" + (" + accum_vgpr_count" if "accum_vgpr_count" in cols else "") + ")"
    rows = cur.execute(f"select name, count(*), sum(duration), avg(duration), min(duration), max(duration), {vg}, max(sgpr_count), "
                       "max(lds_size), max(scratch_size), max(grid_x), max(workgroup_x) from kernels group by name order by sum(duration) desc").fetchall()
    print(f"# kernels view columns: {' '.join(cols)}")
    total = sum(r[2] for r in rows)
    print(f"# source: {path}")
    print(f"# {'kernel':58s} {'calls':>6s} {'total_ms':>9s} {'avg_us':>8s} {'min_us':>8s} {'max_us':>8s} {'pct':>6s} {'vgpr':>5s} {'sgpr':>5s} {'lds':>6s} {'scr':>4s} {'grid':>8s} {'wg':>4s}")
    for r in rows:
        name = r[0].split("(")[0].replace("void ", "")[:58]
        print(f"  {name:58s} {r[1]:6d} {r[2] / 1e6:9.3f} {r[3] / 1e3:8.2f} {r[4] / 1e3:8.2f} {r[5] / 1e3:8.2f} {100 * r[2] / total:6.2f} {r[6]:5d} {r[7]:5d} {r[8]:6d} {r[9]:4d} {r[10]:8d} {r[11]:4d}")


if __name__ == "__main__":
    main(sys.argv[1])
