#!/bin/bash
# round 3: the dispatches of one rollout step and of one optimizer step (end-of-round state)
O=gpurun_out/r03_32; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace -d /tmp/prof2 -o ppo -- python bench.py --steps 20 --warmup 5 --ppo-epochs 1 --no-cpu-baseline --no-pmc --no-other-workloads > $O/prof_ppo.json 2> $O/prof.err
DB=$(find /tmp/prof2 -name '*.db' | head -1)
python profiles/dump_rollout_step.py $DB 40 > $O/rollout_step_kernels.txt 2>&1; head -50 $O/rollout_step_kernels.txt | cut -c1-150
python profiles/dump_step.py $DB > $O/optimizer_step_kernels.txt 2>&1; head -3 $O/optimizer_step_kernels.txt
