#!/bin/bash
# round 3: Unitree G1 line and the one-rank RCCL self-test on the final kernels
O=gpurun_out/r03_34; mkdir -p $O
timeout 300 python bench.py --robot g1 --ppo-epochs 0 --no-cpu-baseline --no-pmc --no-other-workloads > $O/bench_g1.json 2> $O/bench_g1.err; python -c "
import json; d=json.load(open('$O/bench_g1.json')); print('g1', round(d['value']/1e6,2), 'M', round(d['ms_per_step']*1e3,1), 'us/step; stepper', round(d['roofline']['kernel_ms']*1e3,1))" || tail -3 $O/bench_g1.err
timeout 400 python bench.py --force-rccl --steps 50 --warmup 10 --ppo-epochs 3 --no-cpu-baseline --no-pmc --no-other-workloads > $O/bench_force_rccl.json 2> $O/bench_force_rccl.err; python -c "
import json; d=json.load(open('$O/bench_force_rccl.json')); print('force-rccl', round(d['ppo_samples_per_s']), d.get('ppo_comm'))" || tail -3 $O/bench_force_rccl.err
