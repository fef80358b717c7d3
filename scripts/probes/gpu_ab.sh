#!/bin/bash
# A/B on one box: PPO epoch time with / without an env toggle ($1), three runs each, interleaved
for i in 1 2 3; do
for v in 0 1; do
if [ $v = 1 ]; then export $1=1; else unset $1; fi
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1=$v', 'update %.1f ms  play %.1f ms  samples/s %.0f' % (d['ppo_update_ms'], d['ppo_play_ms'], d['ppo_samples_per_s']))"
done; done
