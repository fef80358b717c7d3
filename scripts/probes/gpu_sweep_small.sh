#!/bin/bash
mkdir -p gpurun_out/check
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for n in 4096 8192 32768; do
python bench.py --envs $n --steps 100 --warmup 10 --ppo-epochs 0 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$n envs: %.2fM  ms/step %.4f  k_sim_step %.1f us' % (d['value']/1e6, d['ms_per_step'], d['roofline']['kernel_ms']*1e3))"
done
python bench.py --robot h1 --steps 100 --warmup 10 --ppo-epochs 0 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('h1 4096 envs: %.2fM  ms/step %.4f  k_sim_step %.1f us' % (d['value']/1e6, d['ms_per_step'], d['roofline']['kernel_ms']*1e3))"
python __graft_entry__.py --smoke 2>&1 | tail -3
