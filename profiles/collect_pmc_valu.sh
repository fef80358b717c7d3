#!/bin/bash
# Instruction-issue evidence for the stepper (why it is latency- and not HBM-bound): SQ counters per kernel, one counter
# group per rocprofv3 pass (--pmc only together with --kernel-trace).  Run on the GPU box from the repo root.
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_THREAD_CYCLES_VALU" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/pmcv_$tag -o pmc -- python bench.py --steps 20 --warmup 5 --ppo-epochs 0 --no-cpu-baseline --no-pmc --no-other-workloads > /dev/null 2>/tmp/pmcv_$tag.err || tail -3 /tmp/pmcv_$tag.err
done
python - <<'PY'
import collections, csv, glob
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/tmp/pmcv_*/pmc_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if k.startswith(("k_sim_step", "k_im_")):
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted({c for k in agg for c in agg[k]})
print("# median per dispatch; kernel, then counters:", " ".join(names))
for k in sorted(agg):
    med = {c: sorted(v)[len(v) // 2] for c, v in agg[k].items()}
    print(k, " ".join(f"{c}={med.get(c, float('nan')):.4g}" for c in names))
    if "SQ_INSTS_VALU" in med and "SQ_WAVES" in med and med["SQ_WAVES"]:
        print(f"   VALU instructions per wavefront: {med['SQ_INSTS_VALU'] / med['SQ_WAVES']:.0f}; LDS per wavefront: {med.get('SQ_INSTS_LDS', 0) / med['SQ_WAVES']:.0f}")
    if "SQ_THREAD_CYCLES_VALU" in med and med.get("SQ_ACTIVE_INST_VALU"):
        print(f"   active-lane share of the VALU work (SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU)): {med['SQ_THREAD_CYCLES_VALU'] / (64 * med['SQ_ACTIVE_INST_VALU']):.3f}")
    if "SQ_WAVE_CYCLES" in med and "SQ_ACTIVE_INST_VALU" in med and med["SQ_WAVE_CYCLES"]:
        print(f"   VALU-active share of wavefront cycles: {med['SQ_ACTIVE_INST_VALU'] / med['SQ_WAVE_CYCLES']:.3f}; waiting-on-anything share: {med.get('SQ_WAIT_INST_ANY', 0) / med['SQ_WAVE_CYCLES']:.3f}")
PY
