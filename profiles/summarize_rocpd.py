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
    # an earlier table showed 112 in the `vgpr` column for the stepper, whose code object allocates 224 VGPRs (221 used: `-Rpass-analysis=kernel-resource-usage`,
    # the number DESIGN.md quotes).  rocpd derives `vgpr_count` from the kernel descriptor's granulated field with a 4-register granule; wave64 kernels on gfx9 / gfx950 allocate
    # in granules of 8, so the view reports HALF the allocation (check: 112 -> 224 for k_sim_step; 256 -> 512 = the whole unified file for the hipBLASLt MFMA kernels, whose
    # `accum_vgpr_count` column is 0 in this view).  The table prints 2 x (vgpr_count + accum_vgpr_count).
    vg = "2 * max(vgpr_count" + (" + accum_vgpr_count" if "accum_vgpr_count" in cols else "") + ")"
    rows = cur.execute(f"select name, count(*), sum(duration), avg(duration), min(duration), max(duration), {vg}, max(sgpr_count), "
                       "max(lds_size), max(scratch_size), max(grid_x), max(workgroup_x) from kernels group by name order by sum(duration) desc").fetchall()
    print("# vgpr = 2 x rocpd's vgpr_count (+ accum_vgpr_count): wave64 allocation granule 8, see profiles/summarize_rocpd.py")
    total = sum(r[2] for r in rows)
    print(f"# source: {path}")
    print(f"# {'kernel':58s} {'calls':>6s} {'total_ms':>9s} {'avg_us':>8s} {'min_us':>8s} {'max_us':>8s} {'pct':>6s} {'vgpr':>5s} {'sgpr':>5s} {'lds':>6s} {'scr':>4s} {'grid':>8s} {'wg':>4s}")
    for r in rows:
        name = r[0].split("(")[0].replace("void ", "")[:58]
        print(f"  {name:58s} {r[1]:6d} {r[2] / 1e6:9.3f} {r[3] / 1e3:8.2f} {r[4] / 1e3:8.2f} {r[5] / 1e3:8.2f} {100 * r[2] / total:6.2f} {r[6]:5d} {r[7]:5d} {r[8]:6d} {r[9]:4d} {r[10]:8d} {r[11]:4d}")


if __name__ == "__main__":
    main(sys.argv[1])
