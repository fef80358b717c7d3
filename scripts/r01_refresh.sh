#!/bin/bash
# Round-1 measurement refresh on the GPU box: gpu tests, bench lines, env-count sweep, kernel trace summaries, PMC traffic.
set -x
mkdir -p gpurun_out/refresh
O=gpurun_out/refresh
python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
python bench.py > $O/bench_random.json 2> $O/bench_random.err
python bench.py --actions tracking --no-cpu-baseline > $O/bench_tracking.json 2> $O/bench_tracking.err
for n in 8192 16384 32768; do
  python bench.py --envs $n --steps 100 --warmup 10 --ppo-epochs 0 --no-cpu-baseline > $O/bench_envs_$n.json 2>> $O/sweep.err
done
( time python bench.py --envs 8192 --motion-clips 2048 --steps 100 --warmup 10 --ppo-epochs 0 --no-cpu-baseline > $O/bench_cfg3_shape.json ) 2> $O/bench_cfg3_shape.err
python bench.py --robot h1 --ppo-epochs 0 --no-cpu-baseline > $O/bench_h1_random.json 2>> $O/sweep.err
python bench.py --robot h1 --actions tracking --ppo-epochs 0 --no-cpu-baseline > $O/bench_h1_tracking.json 2>> $O/sweep.err
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d /tmp/prof -o bench -- python bench.py --steps 200 --warmup 20 --ppo-epochs 0 --no-cpu-baseline > $O/prof_bench.json 2> $O/prof.err
python profiles/summarize_rocpd.py $(find /tmp/prof -name '*.db' | head -1) > $O/kernel_stats.txt
rocprofv3 --kernel-trace --stats -d /tmp/prof2 -o ppo -- python bench.py --steps 20 --warmup 5 --ppo-epochs 1 --no-cpu-baseline > $O/prof_ppo.json 2>> $O/prof.err
python profiles/summarize_rocpd.py $(find /tmp/prof2 -name '*.db' | head -1) > $O/kernel_stats_ppo.txt
bash profiles/collect_pmc.sh > $O/pmc_traffic.txt 2>> $O/prof.err
cat $O/bench_random.json | cut -c1-1500; head -8 $O/kernel_stats.txt; cat $O/pmc_traffic.txt
