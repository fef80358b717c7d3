#!/bin/bash
# stepper check after a kernel change: shape tests, kernel-trace summary (scratch / VGPR columns), PMC traffic
O=gpurun_out/${1:-stepchk}
mkdir -p $O
python -m pytest tests/test_dynamics.py tests/test_env_gpu.py -m gpu -q -k "per_env_body or shape_variation or two_slot or step_matches" 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d /tmp/prof -o bench -- python bench.py --steps 200 --warmup 20 --ppo-epochs 0 --no-cpu-baseline --no-pmc > $O/prof_bench.json 2> $O/prof.err
python profiles/summarize_rocpd.py $(find /tmp/prof -name '*.db' | head -1) > $O/env_step_kernel_stats.txt
head -6 $O/env_step_kernel_stats.txt | cut -c1-150
bash profiles/collect_pmc.sh > $O/pmc_traffic.txt 2>> $O/prof.err
head -5 $O/pmc_traffic.txt | cut -c1-150
