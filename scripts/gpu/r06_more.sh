#!/bin/bash
# Round 6, second batch of policy-level runs: (i) the rigid contact model on a failing, a flipping and a passing seed, (ii) the shipped yaml at 4096 and 8192 envs
# (48 / 96 optimizer steps per rollout: round 5's "fail" rows, one seed each then), (iii) the squat clip ALONE under four seeds (rounds 3-5: seed 0 only, never learned).
O=gpurun_out/${1:-r06_more}; mkdir -p $O
bash scripts/gpu/r06_cliff_ab.sh ${1:-r06_more} 120 "tgs" "0 2 4"
ENVS=4096 bash scripts/gpu/r06_cliff_ab.sh ${1:-r06_more}/envs4096 130 "shipped" "1 3"
ENVS=8192 bash scripts/gpu/r06_cliff_ab.sh ${1:-r06_more}/envs8192 150 "shipped" "1 3"
for seed in 0 1 2 3; do
  PHC_QUIET=1 timeout 400 python scripts/learning_curve.py 2500 2048 $O/squat_s$seed.json env.motion_file=squat --seed=$seed > $O/squat_s$seed.log 2>&1
  echo "== squat seed $seed: $(grep -E 'epoch 2500|success' $O/squat_s$seed.log | tail -2 | tr '\n' ' ')"
done
