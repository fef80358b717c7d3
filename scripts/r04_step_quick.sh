#!/bin/bash
# quick A/B of the stepper on one box: dynamics parity on hip + env-step timing (K = 300)
OUT=gpurun_out/r04c; mkdir -p $OUT
python -m pytest tests/test_dynamics.py -m gpu -q -x 2>&1 | tail -3 > $OUT/dyn.txt
python bench.py --steps 300 --warmup 30 --ppo-epochs 0 --no-cpu-baseline --no-pmc --no-other-workloads > $OUT/bench_env.json 2> $OUT/bench_env.err
tail -2 $OUT/dyn.txt
python -c "
import json; d=json.loads(open('$OUT/bench_env.json').read().strip().splitlines()[-1]); print('M env-steps/s', d['value']/1e6, 'us/step', d['ms_per_step']*1e3, 'stepper us', d['roofline']['kernel_ms']*1e3)"
