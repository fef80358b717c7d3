#!/bin/bash
mkdir -p gpurun_out/final
O=gpurun_out/final
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
python __graft_entry__.py --smoke 2>&1 | tail -2
for i in 1 2; do
python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>$O/bench.err | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('env %.2fM  k_sim %.1f us | ppo update %.1f ms  play %.1f ms  samples/s %.0f graph=%s' % (d['value']/1e6, d['roofline']['kernel_ms']*1e3, d['ppo_update_ms'], d['ppo_play_ms'], d['ppo_samples_per_s'], d['ppo_config']['update_graph']))" || tail -5 $O/bench.err
done
