#!/bin/bash
# Round 5: Unitree G1 (38 bodies, the 64-lane stepper kernels) stand / arm-swing learning runs.
O=gpurun_out/g1_learn; mkdir -p $O
G1="robot=unitree_g1 env=env_im_g1_phc sim=robot_sim control=robot_control"
timeout 300 python scripts/learning_curve.py 400 2048 $O/g1_stand_2048envs.json $G1 env.motion_file=stand:10 > $O/stand.log 2>&1; tail -4 $O/stand.log
timeout 300 python scripts/learning_curve.py 400 2048 $O/g1_armswing_2048envs.json $G1 env.motion_file=armswing:10 > $O/armswing.log 2>&1; tail -4 $O/armswing.log
