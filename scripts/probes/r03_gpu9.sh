#!/bin/bash
# round 3: stepper (two-phase body-body contact, touching-points-first ground contact) -- device parity, same-box A/B against HEAD, timeline
O=gpurun_out/r03_9; mkdir -p $O
timeout 900 python -m pytest tests/test_dynamics.py tests/test_h1.py -m gpu -x -q > $O/pytest_dyn.log 2>&1; tail -3 $O/pytest_dyn.log
for i in 1 2; do
for v in old new; do
  if [ $v = old ]; then d=_ab_old; else d=.; fi
  ( cd $d && python bench.py --steps 200 --warmup 20 --ppo-epochs 0 --no-cpu-baseline --no-pmc --no-other-workloads 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v: env-steps/s %.2fM  ms/step %.4f  k_sim_step %.1f us' % (d['value']/1e6, d['ms_per_step'], d['roofline']['kernel_ms']*1e3))" )
done; done 2>&1 | tee $O/ab.txt
python scripts/probes/sim_timeline.py 2048 300 2>&1 | grep -v amdgpu.ids > $O/timeline_2048.txt; head -12 $O/timeline_2048.txt
python scripts/probes/sim_ablation.py 4096 2>&1 | grep -v amdgpu.ids > $O/ablation_4096.txt; cat $O/ablation_4096.txt
python bench.py --robot h1 --steps 100 --warmup 10 --ppo-epochs 0 --no-cpu-baseline --no-pmc --no-other-workloads 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('h1: env-steps/s %.2fM  k_sim_step %.1f us' % (d['value']/1e6, d['roofline']['kernel_ms']*1e3))"
( cd _ab_old && python bench.py --robot h1 --steps 100 --warmup 10 --ppo-epochs 0 --no-cpu-baseline --no-pmc --no-other-workloads 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('h1 old: env-steps/s %.2fM  k_sim_step %.1f us' % (d['value']/1e6, d['roofline']['kernel_ms']*1e3))" )
timeout 900 python scripts/learning_curve.py 3000 4096 $O/learning_curve_squat.json env.motion_file=squat:10 > $O/learning_curve_squat.log 2>&1; tail -3 $O/learning_curve_squat.log | cut -c1-420
