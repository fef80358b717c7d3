#!/bin/bash
mkdir -p gpurun_out/learn
O=gpurun_out/learn
timeout 900 python -m pytest tests/test_env_gpu.py tests/test_learn_gpu.py -q -x > $O/pytest_graph.log 2>&1; tail -15 $O/pytest_graph.log
for i in 1 2; do
for v in 0 1; do
if [ $v = 1 ]; then G=--no-update-graph; else G=; fi
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline $G 2>$O/bench_graph.err | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('PHC_NO_GRAPH=$v', 'update %.1f ms  play %.1f ms  samples/s %.0f' % (d['ppo_update_ms'], d['ppo_play_ms'], d['ppo_samples_per_s']))" || tail -5 $O/bench_graph.err
done; done
