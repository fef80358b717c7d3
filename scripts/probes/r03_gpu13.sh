#!/bin/bash
# round 3: whole-step rollout graphs: task-level equivalence test, device suite for env + learner, PPO bench A/B
O=gpurun_out/r03_13; mkdir -p $O
timeout 900 python -m pytest tests/test_env_gpu.py tests/test_learn_gpu.py tests/test_learner_parity.py -m gpu -x -q > $O/pytest.log 2>&1; tail -6 $O/pytest.log
for v in 0 1; do
  if [ $v = 1 ]; then export PHC_NO_STEP_GRAPH=1; else unset PHC_NO_STEP_GRAPH; fi
  python bench.py --steps 20 --warmup 5 --ppo-epochs 6 --no-cpu-baseline --no-pmc --no-other-workloads 2>$O/bench_$v.err | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('PHC_NO_STEP_GRAPH=$v', 'update %.1f ms  play %.1f ms  samples/s %.0f' % (d['ppo_update_ms'], d['ppo_play_ms'], d['ppo_samples_per_s']))" || tail -5 $O/bench_$v.err
done
