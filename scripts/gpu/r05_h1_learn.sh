#!/bin/bash
# Round 5: the rollout-graph pool fix (resample on a robot) + the H1 stand / arm-swing learning runs (BASELINE configs[4] at policy level).
O=gpurun_out/h1_learn; mkdir -p $O
timeout 600 python -m pytest tests/test_env_gpu.py -m gpu -q -x -k "resample" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
H1="robot=unitree_h1 env=env_im_h1_phc sim=robot_sim control=robot_control"
timeout 240 python scripts/learning_curve.py 400 2048 $O/h1_stand_2048envs.json $H1 env.motion_file=stand:10 > $O/stand.log 2>&1; tail -4 $O/stand.log
timeout 240 python scripts/learning_curve.py 400 2048 $O/h1_armswing_2048envs.json $H1 env.motion_file=armswing:10 > $O/armswing.log 2>&1; tail -4 $O/armswing.log
