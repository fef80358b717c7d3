"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; values are KiB per dispatch).

gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE counts 128-B fabric requests at 64 B -> read bytes = 2 x FETCH_SIZE;
WRITE_SIZE is taken as is.  Both factors are CALIBRATED here on a kernel of this very path whose byte count is known:
k_im_post_physics reads 7 774 B/env (body 1248 + dof 552 + dof_force 276 + 4 reference frames x 1248 + AMP history shift 7056 on one step
in ten -- the window-in-a-strip layout of round 2; 14 124 with plain shifted buffers) and writes 6 522 B/env (obs 3736 + new AMP frame 784 +
shift 7056 / 10 + ref_* side buffers 1236 + flags 60; 12 872 before); the table prints measured/expected."""
import collections
import csv
import json
import sys


def load(path, name):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == name:
            agg[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    return agg


def main(fetch_csv, write_csv, n_envs=4096):
    f, w = load(fetch_csv, "FETCH_SIZE"), load(write_csv, "WRITE_SIZE")
    out = {}
    print(f"# {'kernel':28s} {'calls':>5s} {'FETCH_KiB':>10s} {'WRITE_KiB':>10s} {'read_MB(2x)':>12s} {'write_MB':>9s} {'total_MB':>9s} {'B/env':>8s}")
    def pick(prefix):
        c = [k for k in f if k.startswith(prefix)]
        return max(c, key=lambda k: len(f[k])) if c else None

    for k in filter(None, (pick("k_sim_step<true"), pick("k_im_post_physics"), pick("k_im_reset"))):
        # steady state: drop the first dispatches (full reset of all envs) by taking the median
        fv, wv = sorted(f[k])[len(f[k]) // 2], sorted(w[k])[len(w[k]) // 2]
        rd, wr = 2 * fv * 1024, wv * 1024
        out[k] = {"fetch_kib": fv, "write_kib": wv, "read_bytes": rd, "write_bytes": wr, "traffic_bytes": rd + wr}
        print(f"  {k:28s} {len(f[k]):5d} {fv:10.1f} {wv:10.1f} {rd / 1e6:12.2f} {wr / 1e6:9.2f} {(rd + wr) / 1e6:9.2f} {(rd + wr) / n_envs:8.0f}")
    post = pick("k_im_post_physics")
    if post in out:
        o = out[post]
        print(f"# calibration on k_im_post_physics: read {o['read_bytes'] / n_envs:.0f} B/env measured vs 7774 expected "
              f"({o['read_bytes'] / n_envs / 7774:.2f}x), write {o['write_bytes'] / n_envs:.0f} vs 6522 ({o['write_bytes'] / n_envs / 6522:.2f}x)")
    print("JSON " + json.dumps(out))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
