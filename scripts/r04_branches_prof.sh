#!/bin/bash
OUT=gpurun_out/r04e; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/prof2
rocprofv3 --kernel-trace --stats -d /tmp/prof2 -o ppo -- python bench.py --steps 20 --warmup 5 --ppo-epochs 1 --no-cpu-baseline --no-pmc --no-other-workloads ${BENCH_ARGS} > $OUT/prof_ppo.json 2> $OUT/prof.err
python profiles/dump_step.py $(find /tmp/prof2 -name '*.db' | head -1) > $OUT/ppo_optimizer_step_kernels.txt 2>> $OUT/prof.err || true
head -3 $OUT/ppo_optimizer_step_kernels.txt
