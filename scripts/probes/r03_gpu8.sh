#!/bin/bash
# round 3: locomotion-class physics acceptance (squat / step in place / walk) + the switch test again
O=gpurun_out/r03_8; mkdir -p $O
timeout 300 python -m pytest tests/test_env_gpu.py -m gpu -x -q -k "switches" > $O/pytest_switches.log 2>&1; tail -3 $O/pytest_switches.log
for spec in "squat 1200" "stepinplace 2000" "walk 3000"; do
  set -- $spec
  timeout 900 python scripts/learning_curve.py $2 4096 $O/learning_curve_$1.json env.motion_file=$1:10 > $O/learning_curve_$1.log 2>&1
  tail -4 $O/learning_curve_$1.log | cut -c1-400
done
