#!/bin/bash
# round 3: stepper with the set-bit loop for body-body contact: parity, same-box A/B, timeline; squat clip under plain PD tracking for the whole clip
O=gpurun_out/r03_10; mkdir -p $O
timeout 900 python -m pytest tests/test_dynamics.py tests/test_h1.py -m gpu -x -q > $O/pytest_dyn.log 2>&1; tail -3 $O/pytest_dyn.log
for i in 1 2; do
for v in old new; do
  if [ $v = old ]; then d=_ab_old; else d=.; fi
  ( cd $d && python bench.py --steps 200 --warmup 20 --ppo-epochs 0 --no-cpu-baseline --no-pmc --no-other-workloads 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v: env-steps/s %.2fM  ms/step %.4f  k_sim_step %.1f us' % (d['value']/1e6, d['ms_per_step'], d['roofline']['kernel_ms']*1e3))" )
done; done 2>&1 | tee $O/ab.txt
python scripts/probes/sim_timeline.py 2048 300 2>&1 | grep -v amdgpu.ids > $O/timeline_2048.txt; head -8 $O/timeline_2048.txt
for r in h1 g1; do
python bench.py --robot $r --steps 100 --warmup 10 --ppo-epochs 0 --no-cpu-baseline --no-pmc --no-other-workloads 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$r new: env-steps/s %.2fM  k_sim_step %.1f us' % (d['value']/1e6, d['roofline']['kernel_ms']*1e3))"
( cd _ab_old && python bench.py --robot $r --steps 100 --warmup 10 --ppo-epochs 0 --no-cpu-baseline --no-pmc --no-other-workloads 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$r old: env-steps/s %.2fM  k_sim_step %.1f us' % (d['value']/1e6, d['roofline']['kernel_ms']*1e3))" )
done 2>&1 | tee -a $O/ab.txt
python scripts/probes/track_probe.py squat:10 298 2>&1 | grep -v amdgpu.ids > $O/squat_pd_tracking.txt; awk 'NR%8==1' $O/squat_pd_tracking.txt | cut -c1-200 | tail -16
