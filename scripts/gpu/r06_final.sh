#!/bin/bash
# Round 6, last GPU call: the refresh (suite, smoke, bench lines, kernel traces, PMC) and the squat clip alone with half the power coefficient on the two seeds that fail as shipped.
bash scripts/r06_refresh.sh r6f
O=gpurun_out/r6f
for seed in 0 2; do
  PHC_QUIET=1 timeout 400 python scripts/learning_curve.py 2500 2048 $O/squat_halfpower_s$seed.json env.motion_file=squat +env.power_coefficient=0.00025 --seed=$seed > $O/squat_halfpower_s$seed.log 2>&1
  echo "== squat, half power coefficient, seed $seed: $(grep -E 'epoch 2500|success' $O/squat_halfpower_s$seed.log | tail -2 | tr '\n' ' ' | cut -c1-400)"
done
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python profiles/dump_rollout_step.py $(find /tmp/prof2 -name '*.db' | head -1) -60 > $O/rollout_step_kernels.txt 2>> $O/prof.err || true
