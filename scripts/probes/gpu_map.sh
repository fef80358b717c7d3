#!/bin/bash
for rep in 1 2; do
for m in 1 2; do
python bench.py --lane-mapping $m --steps 200 --warmup 20 --ppo-epochs 0 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('smpl map $m: %.2fM  ms/step %.4f  k_sim_step %.1f us' % (d['value']/1e6, d['ms_per_step'], d['roofline']['kernel_ms']*1e3))"
python bench.py --robot h1 --lane-mapping $m --steps 200 --warmup 20 --ppo-epochs 0 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('h1   map $m: %.2fM  ms/step %.4f  k_sim_step %.1f us' % (d['value']/1e6, d['ms_per_step'], d['roofline']['kernel_ms']*1e3))"
done
done
