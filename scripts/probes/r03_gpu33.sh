#!/bin/bash
# round 3: BASELINE configs[2] shape (8192 envs, thousands of clips) on the final kernels -- the AMP table at that size
O=gpurun_out/r03_33; mkdir -p $O
timeout 500 python bench.py --config 3 --ppo-epochs 0 --no-cpu-baseline --no-pmc --no-other-workloads > $O/bench_config3.json 2> $O/bench_config3.err; tail -c 300 $O/bench_config3.err
python -c "
import json; d=json.load(open('$O/bench_config3.json')); print(round(d['value']/1e6,2), 'M env-steps/s', round(d['ms_per_step']*1e3,1), 'us/step; stepper', round(d['roofline']['kernel_ms']*1e3,1), 'us'); print(d['config']); print({k: d[k] for k in d if 'cfg3' in k or 'config3' in k})"
