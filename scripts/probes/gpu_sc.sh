#!/bin/bash
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for sc in 0 1; do
python bench.py --self-collision $sc --steps 200 --warmup 20 --ppo-epochs 0 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('smpl self_collision=$sc: %.2fM  ms/step %.4f  k_sim_step %.1f us' % (d['value']/1e6, d['ms_per_step'], d['roofline']['kernel_ms']*1e3))"
python bench.py --robot h1 --self-collision $sc --steps 200 --warmup 20 --ppo-epochs 0 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('h1   self_collision=$sc: %.2fM  ms/step %.4f  k_sim_step %.1f us' % (d['value']/1e6, d['ms_per_step'], d['roofline']['kernel_ms']*1e3))"
done
