#!/bin/bash
mkdir -p gpurun_out/check
O=gpurun_out/check
for a in random tracking; do
python bench.py --robot h1 --actions $a --steps 200 --warmup 20 --ppo-epochs 0 --no-cpu-baseline > $O/bench_h1_$a.json 2> $O/bench_h1.err; tail -2 $O/bench_h1.err
python -c "
import json; d=json.load(open('$O/bench_h1_$a.json')); print('$a: env-steps/s %.2fM  ms/step %.4f  k_sim_step %.1f us resets %.2f' % (d['value']/1e6, d['ms_per_step'], d['roofline']['kernel_ms']*1e3, d['envs_within_5_steps_of_a_reset']))"
done
