#!/bin/bash
# round 3, GPU call 1: microbenchmark (VALU issue), device tests after the kinematics / two-slot changes, stepper time at 1 / 2 / 4 waves per SIMD, VALU counters
O=gpurun_out/r03_1; mkdir -p $O
( cd profiles/microbench && ./valu_issue ../../$O/valu_issue.json > ../../$O/valu_issue.txt 2>&1 ); tail -3 $O/valu_issue.txt
timeout 900 python -m pytest tests/test_dynamics.py tests/test_env_gpu.py tests/test_h1.py -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -5 $O/pytest_gpu.log
for n in 1024 2048 4096 8192; do
python bench.py --envs $n --steps 200 --warmup 20 --ppo-epochs 0 --no-cpu-baseline --no-pmc --no-other-workloads > $O/bench_$n.json 2> $O/bench_$n.err || tail -3 $O/bench_$n.err
python -c "
import json; d=json.load(open('$O/bench_$n.json')); print('envs $n: env-steps/s %.2fM  ms/step %.4f  k_sim_step %.1f us' % (d['value']/1e6, d['ms_per_step'], d['roofline']['kernel_ms']*1e3))"
done
bash profiles/collect_pmc_valu.sh > $O/pmc_valu.txt 2>&1; tail -12 $O/pmc_valu.txt
