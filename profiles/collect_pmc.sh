#!/bin/bash
# HBM traffic of the env-step kernels from PMC counters, as /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes:
# FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 passes (TCC slot limit), --pmc only with --kernel-trace.
# Run on the GPU box from the repo root:  bash profiles/collect_pmc.sh > profiles/rNN_pmc_traffic.txt
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -o pmc -- python bench.py --steps 20 --warmup 5 --ppo-epochs 0 --no-cpu-baseline --no-pmc --no-other-workloads > /dev/null 2>&1
done
python profiles/pmc_summary.py /tmp/pmc_FETCH_SIZE/pmc_counter_collection.csv /tmp/pmc_WRITE_SIZE/pmc_counter_collection.csv
