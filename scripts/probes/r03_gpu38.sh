#!/bin/bash
O=gpurun_out/r03_38; mkdir -p $O
for rep in 1 2; do python bench.py --steps 20 --warmup 5 --ppo-epochs 0 --no-cpu-baseline --no-pmc --no-other-workloads 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('K=20:', round(d['value']/1e6,2), 'M', round(d['ms_per_step']*1e3,1), 'us/step; stepper', round(d['roofline']['kernel_ms']*1e3,1), 'resets<5', round(d['envs_within_5_steps_of_a_reset'],3))"; done
