#!/bin/bash
# Round-2 quick measurement: gpu tests subset, phase profile of the stepper, kernel trace of the env step, (optional) PMC
O=gpurun_out/${1:-r2d}
mkdir -p $O
python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; tail -15 $O/pytest_gpu.log | cut -c1-300
python scripts/sim_phase_profile.py run 4096 > $O/phase_profile.txt 2>&1; tail -12 $O/phase_profile.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d /tmp/prof -o bench -- python bench.py --steps 200 --warmup 20 --ppo-epochs 0 --no-cpu-baseline --no-pmc > $O/prof_bench.json 2> $O/prof.err
python profiles/summarize_rocpd.py $(find /tmp/prof -name '*.db' | head -1) > $O/kernel_stats.txt; head -8 $O/kernel_stats.txt | cut -c1-200
