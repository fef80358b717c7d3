#!/bin/bash
# G1 (38 bodies, 64-lane kernels): GPU tests, both stepper mappings timed, and the SMPL / H1 bench lines re-timed after the
# lanes-per-env templating (must be unchanged).
mkdir -p gpurun_out/g1
O=gpurun_out/g1
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
for m in 1 2; do
python bench.py --robot g1 --envs 4096 --lane-mapping $m --actions tracking --steps 100 --warmup 10 --ppo-epochs 0 --no-cpu-baseline > $O/bench_g1_map$m.json 2> $O/bench_g1.err || tail -5 $O/bench_g1.err
python -c "
import json; d=json.load(open('$O/bench_g1_map$m.json')); print('g1 map$m: env-steps/s %.2fM  ms/step %.4f  k_sim_step %.1f us' % (d['value']/1e6, d['ms_per_step'], d['roofline']['kernel_ms']*1e3))"
done
for r in smpl h1; do
python bench.py --robot $r --steps 200 --warmup 20 --ppo-epochs 0 --no-cpu-baseline > $O/bench_$r.json 2> $O/bench_$r.err || tail -5 $O/bench_$r.err
python -c "
import json; d=json.load(open('$O/bench_$r.json')); print('$r: env-steps/s %.2fM  ms/step %.4f  k_sim_step %.1f us' % (d['value']/1e6, d['ms_per_step'], d['roofline']['kernel_ms']*1e3))"
done
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_g1 -o g1 -- python $GRAFT_REPO_ROOT/bench.py --robot g1 --envs 4096 --actions tracking --steps 50 --warmup 10 --ppo-epochs 0 --no-cpu-baseline > /dev/null 2>&1
cd $GRAFT_REPO_ROOT && python profiles/summarize_rocpd.py $(ls /tmp/prof_g1/*/*.db /tmp/prof_g1/*.db 2>/dev/null | head -1) > $O/g1_kernel_stats.txt 2>&1; head -12 $O/g1_kernel_stats.txt
