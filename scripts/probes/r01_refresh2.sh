#!/bin/bash
# Late round-1 refresh after the learner kernels: default bench line, PPO kernel trace (graph and eager), env-step trace.
mkdir -p gpurun_out/refresh2
O=$GRAFT_REPO_ROOT/gpurun_out/refresh2
python bench.py > $O/bench_random.json 2> $O/bench_random.err
python bench.py --actions tracking --no-cpu-baseline > $O/bench_tracking.json 2> $O/bench_tracking.err
python bench.py --no-cpu-baseline --no-update-graph --steps 50 --warmup 10 > $O/bench_eager_update.json 2>> $O/bench_tracking.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof2 -o ppo -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --ppo-epochs 3 --no-cpu-baseline > $O/prof_ppo.json 2> $O/prof.err
cd $GRAFT_REPO_ROOT
python profiles/summarize_rocpd.py $(find /tmp/prof2 -name '*.db' | head -1) > $O/kernel_stats_ppo.txt
python profiles/dump_step.py $(find /tmp/prof2 -name '*.db' | head -1) 100 > $O/step_kernels.txt 2>&1
cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof1 -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 20 --ppo-epochs 0 --no-cpu-baseline > $O/prof_bench.json 2>> $O/prof.err
cd $GRAFT_REPO_ROOT
python profiles/summarize_rocpd.py $(find /tmp/prof1 -name '*.db' | head -1) > $O/kernel_stats.txt
cut -c1-2200 $O/bench_random.json; head -12 $O/kernel_stats_ppo.txt; head -6 $O/kernel_stats.txt
